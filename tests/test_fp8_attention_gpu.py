"""fp8 attention (BASELINE config 5): e4m3 Q/K/V/P, e5m2 dO/dS, against (a) a torch restatement with the SAME quantisation points
(tight: only accumulation order and bf16 output rounding differ) and (b) exact fp32 attention (the tolerance ladder of the format).
Reference op: nn.MultiheadAttention of the fusion encoder (allenact_dino_transformer.py:545-552,702-708)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from safevla_amd import ops as o
    return o


def bf(t):
    return t.bfloat16().float()


def _q(t, dtype, target):
    """per-[S,64]-slice quantisation: t [rows, H, S, 64] -> (dequantised values, scale [rows, H, 1, 1])"""
    amax = t.abs().amax(dim=(-1, -2), keepdim=True)
    s = amax / target
    return (t / s.clamp_min(1e-30)).to(dtype).float(), s


def _heads(x, rows, S, H):
    return x.view(rows, S, H, 64).transpose(1, 2).contiguous()


def _case(rows, S, H, seed):
    g = torch.Generator().manual_seed(seed)
    qkv = bf(torch.randn(rows * S, 3 * H * 64, generator=g) * 0.7)
    do = bf(torch.randn(rows * S, H * 64, generator=g) * 0.02)
    return qkv, do


def _exact(qkv, do, rows, S, H, scale):
    q, k, v = [_heads(qkv[:, i * H * 64:(i + 1) * H * 64], rows, S, H).requires_grad_(True) for i in range(3)]
    o = torch.softmax((q @ k.transpose(-1, -2)) * scale, -1) @ v
    o.backward(_heads(do, rows, S, H))
    return o, q.grad, k.grad, v.grad


def _flat(t, rows, S, H):
    return t.transpose(1, 2).reshape(rows * S, H * 64)


@pytest.mark.parametrize("S", [1, 5, 40, 64, 65, 100, 128, 129, 181, 192, 193, 233, 256])
def test_fp8_forward_matches_quantisation_aware_reference(ops, S):
    rows, H, scale = (3 if S > 1 else 1), 8, 0.125
    qkv, _ = _case(rows, S, H, S)
    q, k, v = [_heads(qkv[:, i * H * 64:(i + 1) * H * 64], rows, S, H) for i in range(3)]
    (q8, sq), (k8, sk), (v8, sv) = [_q(t, torch.float8_e4m3fn, 448.0) for t in (q, k, v)]
    x = (q8 @ k8.transpose(-1, -2)) * (sq * sk * scale)
    mx = x.amax(-1, keepdim=True)
    pu = torch.exp(x - mx)
    p8 = (pu * 256.0).to(torch.float8_e4m3fn).float()
    want = (p8 @ v8) * sv / (256.0 * pu.sum(-1, keepdim=True))
    want_lse = (mx + torch.log(pu.sum(-1, keepdim=True))).squeeze(-1)
    d = qkv.to(DEV).bfloat16()
    f8 = ops.attn_fp8_quant(d, 3 * H * 64, rows, S, H)
    out, lse = ops.attn_fp8_fwd(f8, scale)
    torch.cuda.synchronize()
    assert torch.allclose(f8.scales.cpu().view(rows, H, 3), torch.stack([sq, sk, sv], -1).view(rows, H, 3), rtol=1e-6)
    got = _heads(out.float().cpu(), rows, S, H)
    dl = (lse.cpu() - want_lse).abs()       # isolated elements round to the other e4m3 neighbour (x * (448/amax) here, x / (amax/448) there)
    assert dl.max().item() < 1e-2 and dl.mean().item() < 1e-4, (dl.max().item(), dl.mean().item())
    err = (got - want).abs()
    assert err.max().item() <= 6e-2 * want.abs().max().item() and err.mean().item() <= 2e-3 * want.abs().max().item(), (err.max().item(), err.mean().item())


@pytest.mark.parametrize("S,p", [(17, 0.0), (64, 0.0), (65, 0.1), (181, 0.0), (193, 0.0), (233, 0.0), (233, 0.1), (256, 0.1)])
def test_fp8_forward_backward_tolerance_ladder_vs_fp32(ops, S, p):
    """Against exact fp32 attention, measured on N(0, 0.7) inputs: O 4.1-4.2 % relative Frobenius error, dV 6.0-6.2 % (cosine 0.998),
    dQ / dK 8.6 % (cosine 0.9963) at every S -- the price of e4m3 operands / probabilities (3 mantissa bits) and e5m2 dO / dS
    (2 mantissa bits); the bf16 kernels sit at 0.23 % / 0.24 % on the same inputs.  The gates leave ~1.4x margin."""
    from oracle.ref_model import hash_dropout
    rows, H, scale = 4, 8, 0.125
    qkv, do = _case(rows, S, H, 100 + S)
    drop = ops.Dropout(seed=0xBEEF, stream=3, p=p) if p > 0 else None
    q, k, v = [_heads(qkv[:, i * H * 64:(i + 1) * H * 64], rows, S, H).requires_grad_(True) for i in range(3)]
    pr = torch.softmax((q @ k.transpose(-1, -2)) * scale, -1)
    if p > 0:
        pr = hash_dropout(pr, 0xBEEF, 3, p, attn_S=S)
    o = pr @ v
    o.backward(_heads(do, rows, S, H))
    d = qkv.to(DEV).bfloat16()
    f8 = ops.attn_fp8_quant(d, 3 * H * 64, rows, S, H)
    out, lse = ops.attn_fp8_fwd(f8, scale, drop=drop)
    dqkv = torch.zeros_like(d)
    ops.attn_fp8_bwd(f8, out, lse, do.to(DEV).bfloat16(), dqkv, dqkv[:, H * 64:], dqkv[:, 2 * H * 64:], 3 * H * 64, scale, drop=drop)
    torch.cuda.synchronize()
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    cos = lambda a, b: torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
    got_o = _heads(out.float().cpu(), rows, S, H)
    assert torch.isfinite(dqkv.float()).all()
    assert rel(got_o, o.detach()) < 0.06, rel(got_o, o.detach())
    for i, (n, t) in enumerate((("dQ", q), ("dK", k), ("dV", v))):
        got = _heads(dqkv[:, i * H * 64:(i + 1) * H * 64].float().cpu(), rows, S, H)
        assert rel(got, t.grad) < 0.12, (n, rel(got, t.grad))
        assert cos(got, t.grad) > 0.994, (n, cos(got, t.grad))


def test_fp8_backward_is_zero_for_zero_upstream_gradient(ops):
    rows, S, H = 2, 181, 8
    qkv, _ = _case(rows, S, H, 5)
    d = qkv.to(DEV).bfloat16()
    f8 = ops.attn_fp8_quant(d, 3 * H * 64, rows, S, H)
    out, lse = ops.attn_fp8_fwd(f8, 0.125)
    dqkv = torch.full_like(d, 7.0)
    ops.attn_fp8_bwd(f8, out, lse, torch.zeros_like(out), dqkv, dqkv[:, H * 64:], dqkv[:, 2 * H * 64:], 3 * H * 64, 0.125)
    assert (dqkv == 0).all()


@pytest.mark.parametrize("S,p", [(40, 0.0), (65, 0.1), (181, 0.0), (181, 0.1), (233, 0.1), (256, 0.0)])
def test_fp8_backward_matches_quantisation_aware_reference(ops, S, p):
    """svla_attn_fp8_bwd against oracle/ref_fp8_attn.bwd: the same e4m3 / e5m2 casts at the same points (P x 256, dS x 2^-13, per-slice scales),
    fed with the kernel's own forward outputs -- what is left is accumulation order, exp2 vs exp, and the borderline values those two push to the OTHER
    fp8 neighbour (one e4m3 step is 6 % of the value, one e5m2 step 12-25 %: a per-cent of flipped roundings is a per-cent of Frobenius error).
    Measured on the MI355X: O 0.6-1.0 %, dV / dK 0.5-1.0 %, dQ 1.0-1.5 % relative Frobenius, mean |error| <= 1.5e-3 of the largest element.  Gates: 3 %
    Frobenius and 3e-3 mean -- four times tighter than the fp32 ladder's 12 % and a third of the formats' own 8.6 % error; a wrong e5m2 scale or a
    mis-permuted reduction slot is a > 30 % error here.  The forward with dropout is checked against the same restatement on the way."""
    from oracle import ref_fp8_attn as R
    from oracle.ref_model import hash_dropout
    rows, H, scale = 3, 8, 0.125
    qkv, do = _case(rows, S, H, 300 + S)
    q, k, v = [_heads(qkv[:, i * H * 64:(i + 1) * H * 64], rows, S, H) for i in range(3)]
    dropc = ops.Dropout(seed=0xC0FFEE, stream=4, p=p) if p > 0 else None
    keep = (hash_dropout(torch.ones(rows, H, S, S), 0xC0FFEE, 4, p, attn_S=S) > 0) if p > 0 else None
    ds = 1.0 / (1.0 - p)
    d = qkv.to(DEV).bfloat16()
    f8 = ops.attn_fp8_quant(d, 3 * H * 64, rows, S, H)
    out, lse = ops.attn_fp8_fwd(f8, scale, drop=dropc)
    dqkv = torch.zeros_like(d)
    ops.attn_fp8_bwd(f8, out, lse, do.to(DEV).bfloat16(), dqkv, dqkv[:, H * 64:], dqkv[:, 2 * H * 64:], 3 * H * 64, scale, drop=dropc)
    torch.cuda.synchronize()
    o_k, lse_k = _heads(out.float().cpu(), rows, S, H), lse.cpu().view(rows, H, S)
    o_ref, lse_ref = R.fwd(q, k, v, scale, keep=keep, drop_scale=ds)
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    assert rel(o_k, o_ref) < 2e-2 and (lse_k - lse_ref).abs().mean().item() < 5e-4, (rel(o_k, o_ref), (lse_k - lse_ref).abs().mean().item())
    dq, dk, dv = R.bwd(q, k, v, o_k, lse_k, _heads(do, rows, S, H), scale, keep=keep, drop_scale=ds)
    for i, (n, want) in enumerate((("dQ", dq), ("dK", dk), ("dV", dv))):
        got = _heads(dqkv[:, i * H * 64:(i + 1) * H * 64].float().cpu(), rows, S, H)
        mean_err = ((got - want).abs().mean() / want.abs().max()).item()
        assert rel(got, want) < 3e-2 and mean_err < 3e-3, (n, S, p, rel(got, want), mean_err)
