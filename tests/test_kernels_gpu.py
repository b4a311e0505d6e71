"""Per-kernel parity: every C-ABI entry point (called through the ctypes binding, i.e. through the C ABI) against a
plain fp32/fp64 torch restatement of the same op on identical seeded inputs.  bf16 kernels are fed bf16-exact inputs,
so the only differences are accumulation order and the final bf16 rounding (2^-9 relative)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from safevla_amd import ops as o

    return o


DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def bf(x):  # bf16-exact fp32 values
    return x.to(torch.bfloat16).float()


def close(got, want, rtol, atol, name=""):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    if bad.any() or not torch.isfinite(got).all():
        i = int(torch.argmax(err - tol))
        idx = np.unravel_index(i, got.shape) if got.dim() else ()
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{got.numel()} off; worst at {idx}: got {got.flatten()[i]:.6g} want {want.flatten()[i]:.6g}"
            f" | max abs err {err.max():.3g}, want absmax {want.abs().max():.3g}, finite={bool(torch.isfinite(got).all())}"
        )


# ------------------------------------------------------------------------------------------------ GAE
@pytest.mark.parametrize("T,B", [(37, 5), (256, 32), (128, 200), (1, 1)])
def test_gae_bit_exact(ops, T, B):
    from oracle.ref_rollout import gae_scan

    r, c = rnd(T, B, 1, seed=1), rnd(T, B, 1, seed=2).abs().round()
    v, cv = rnd(T, B, 1, seed=3), rnd(T, B, 1, seed=4)
    m = (torch.rand(T + 1, B, 1, generator=torch.Generator().manual_seed(5)) > 0.1).float()
    nv, ncv = rnd(B, 1, seed=6), rnd(B, 1, seed=7)
    ret, adv = gae_scan(r, v, m, nv)
    cret, cadv = gae_scan(c, cv, m, ncv)
    d = lambda t: t.reshape(t.shape[0], B).contiguous().to(DEV)
    got = ops.gae_scan(d(r), d(c), d(v), d(cv), d(m), nv.reshape(B).to(DEV), ncv.reshape(B).to(DEV))
    for g_, w_, n in zip(got, (ret, adv, cret, cadv), ("ret", "adv", "c_ret", "c_adv")):
        assert torch.equal(g_.cpu(), w_.reshape(T, B)), n  # bit-exact


def test_gae_full_size_properties(ops):
    """C4-size (T=256, B=256): episode boundaries cut the recursion; linear in (rewards)."""
    T, B = 256, 256
    r, v = rnd(T, B, seed=1).to(DEV), rnd(T, B, seed=3).to(DEV)
    z = torch.zeros(T, B, device=DEV)
    m = (torch.rand(T + 1, B, generator=torch.Generator().manual_seed(5)) > 0.02).float().to(DEV)
    nv = rnd(B, seed=6).to(DEV)
    a = ops.gae_scan(r, z, v, z, m, nv, torch.zeros(B, device=DEV))
    b = ops.gae_scan(2 * r, r, 2 * v, v, m, 2 * nv, nv)
    close(b[1], 2 * a[1], 1e-5, 1e-5, "linearity")
    close(b[3], a[1], 1e-5, 1e-5, "cost twin == reward twin on same data")
    r2 = r.clone()
    r2[100:] += 50.0
    m2 = m.clone()
    m2[100] = 0
    a1 = ops.gae_scan(r, z, v, z, m2, nv, torch.zeros(B, device=DEV))
    a2 = ops.gae_scan(r2, z, v, z, m2, nv, torch.zeros(B, device=DEV))
    assert torch.equal(a1[1][:99], a2[1][:99])


# ------------------------------------------------------------------------------------------------ losses
@pytest.mark.parametrize("clipped", [False, True])
@pytest.mark.parametrize("lam", [0.0, 0.37, 5.0])
def test_ppo_lag_loss_vs_reference_golden(ops, lam, clipped):
    g = dict(np.load(os.path.join(G, "g6_losses.npz")))
    T, B, A = g["raw_logits"].shape
    R = T * B
    t = lambda k: torch.from_numpy(g[k]).reshape(R, *g[k].shape[2:]).contiguous().to(DEV)
    sums, dl, dv = ops.ppo_lag_loss_fwd_bwd(
        t("raw_logits"), t("values_pred").reshape(R), t("batch:actions"), t("batch:old_action_log_probs"),
        t("batch:adv_targ").reshape(R), t("batch:c_adv_targ").reshape(R), t("batch:returns").reshape(R),
        t("batch:values").reshape(R), lam, 0.1, 0.5, 1.0, 0.01, clipped, 1.0 / R)
    key = f"safe:{int(clipped)}:{lam}"
    s = sums.cpu().numpy() / R
    value, action, ent = (0.5 * s[0], s[1], s[2])
    total = 0.5 * value + action + 0.01 * ent
    np.testing.assert_allclose([total, value, action, ent], g[key + ":scalars"], rtol=2e-5)
    close(dl.reshape(T, B, A), torch.from_numpy(g[key + ":dlogits"]), 1e-4, 1e-8, "dlogits")
    close(dv.reshape(T, B, 1), torch.from_numpy(g[key + ":dvalues"]), 1e-4, 1e-8, "dvalues")


def test_value_mse(ops):
    v, r = rnd(1000, seed=1), rnd(1000, seed=2)
    sums, dv = ops.value_mse_fwd_bwd(v.to(DEV), r.to(DEV), 1.0, 1.0 / 1000)
    vv = v.clone().requires_grad_(True)
    l = 0.5 * (r - vv).pow(2).mean()
    l.backward()
    np.testing.assert_allclose(0.5 * sums.item() / 1000, l.item(), rtol=1e-5)
    close(dv, vv.grad, 1e-5, 1e-9, "dv")


@pytest.mark.parametrize("W_", [512, 768])
@pytest.mark.parametrize("N,T,B", [(20, 0, 0), (1, 0, 0), (20, 7, 5), (1, 7, 5)])
def test_small_linear(ops, N, T, B, W_):
    rows = T * B if T else 37
    x, W, b = rnd(rows, W_, seed=1), rnd(N, W_, seed=2, scale=0.05), rnd(N, seed=3)
    out = ops.small_linear_fwd(x.to(DEV), W.to(DEV), b.to(DEV), T, B)
    xs = x.view(B, T, W_).permute(1, 0, 2).reshape(rows, W_) if T else x  # (b*T+t) storage -> (t*B+b) rows
    want = xs @ W.t() + b
    close(out, want, 1e-4, 1e-5, "fwd")
    dout = rnd(rows, N, seed=4)
    dx = torch.zeros(rows, W_, device=DEV)
    dW = torch.ones(N, W_, device=DEV)
    db = torch.ones(N, device=DEV)
    ops.small_linear_bwd(x.to(DEV), W.to(DEV), dout.to(DEV), dx, dW, db, T, B)
    dxs = dout @ W
    if T:
        dxs = dxs.view(T, B, W_).permute(1, 0, 2).reshape(rows, W_)
    close(dx, dxs, 1e-4, 1e-5, "dx")
    close(dW, 1 + dout.t() @ xs, 1e-4, 1e-4, "dW")
    close(db, 1 + dout.sum(0), 1e-4, 1e-4, "db")


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("W_", [512, 768])
@pytest.mark.parametrize("rms", [False, True])
def test_norm_fwd_bwd_plain(ops, rms, W_):
    rows = 333
    x = bf(rnd(rows, W_, seed=1) * 2 + 0.3)
    gma, bta = 1 + 0.1 * rnd(W_, seed=2), 0.1 * rnd(W_, seed=3)
    eps = 1e-5
    xr = x.clone().requires_grad_(True)
    gr, br = gma.clone().requires_grad_(True), bta.clone().requires_grad_(True)
    if rms:
        want = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + eps) * gr
    else:
        want = F.layer_norm(xr, (W_,), gr, br, eps)
    y, mean, rstd = ops.norm_fwd(x.to(DEV).bfloat16(), gma.to(DEV), bta.to(DEV), eps, rows, rms=rms, D=W_)
    close(y.float(), want, 8e-3, 8e-3, "y")
    dy = bf(rnd(rows, W_, seed=4))
    want.backward(dy)
    dg, db = torch.zeros(W_, device=DEV), torch.zeros(W_, device=DEV)
    dx = ops.norm_bwd(dy.to(DEV).bfloat16(), x.to(DEV).bfloat16(), gma.to(DEV), bta.to(DEV), mean, rstd, rows, dg, db, rms=rms, D=W_)
    close(dx.float(), xr.grad, 1e-2, 1e-2, "dx")
    close(dg, gr.grad, 2e-3, 2e-3 * gr.grad.abs().max().item(), "dgamma")
    if not rms:
        close(db, br.grad, 2e-3, 2e-3 * br.grad.abs().max().item(), "dbeta")


@pytest.mark.parametrize("W_", [512, 768])
def test_norm_adapter_rowmap_relu_tok(ops, W_):
    """Linear->LN->ReLU(+camera token) writing into the [R,S,D] fusion input: rows m=(r*2+cam)*84+p -> r*S+1+m%168."""
    R, S, Gp = 5, 181, 168
    rows = R * Gp
    x = bf(rnd(rows, W_, seed=1))
    gma, bta, tok = 1 + 0.1 * rnd(W_, seed=2), 0.1 * rnd(W_, seed=3), rnd(2, W_, seed=4)
    xr, gr, br, tr = [t.clone().requires_grad_(True) for t in (x, gma, bta, tok)]
    z = F.relu(F.layer_norm(xr, (W_,), gr, br, 1e-5)).view(R, 2, 84, W_) + tr.view(1, 2, 1, W_)
    x0 = torch.full((R, S, W_), 7.0, device=DEV, dtype=torch.bfloat16)
    _, mean, rstd = ops.norm_fwd(x.to(DEV).bfloat16(), gma.to(DEV), bta.to(DEV), 1e-5, rows, relu=True, tok=tok.to(DEV),
                                 tok_group=84, y=x0, ymap=(Gp, S, 1), D=W_)
    close(x0[:, 1:169].float(), z.reshape(R, Gp, W_), 8e-3, 8e-3, "y slice")
    assert (x0[:, 0] == 7).all() and (x0[:, 169:] == 7).all()
    dx0 = bf(rnd(R, S, W_, seed=5))
    z.backward(dx0[:, 1:169].reshape(R, 2, 84, W_))
    dg, db, dt = torch.zeros(W_, device=DEV), torch.zeros(W_, device=DEV), torch.zeros(2, W_, device=DEV)
    dx = ops.norm_bwd(dx0.to(DEV).bfloat16(), x.to(DEV).bfloat16(), gma.to(DEV), bta.to(DEV), mean, rstd, rows, dg, db, relu=True,
                      dtok=dt, tok_group=84, dymap=(Gp, S, 1), D=W_)
    close(dx.float(), xr.grad, 1e-2, 1e-2, "dx")
    close(dg, gr.grad, 2e-3, 2e-3 * gr.grad.abs().max().item(), "dgamma")
    close(db, br.grad, 2e-3, 2e-3 * br.grad.abs().max().item(), "dbeta")
    close(dt, tr.grad, 2e-3, 2e-3 * tr.grad.abs().max().item(), "dtok")


def test_norm_fwd_384(ops):
    x = bf(rnd(100, 384, seed=1))
    gma, bta = 1 + 0.1 * rnd(384, seed=2), 0.1 * rnd(384, seed=3)
    y, _, _ = ops.norm_fwd(x.to(DEV).bfloat16(), gma.to(DEV), bta.to(DEV), 1e-6, 100, D=384)
    close(y.float(), F.layer_norm(x, (384,), gma, bta, 1e-6), 8e-3, 8e-3, "y384")


# ------------------------------------------------------------------------------------------------ GEMMs
@pytest.mark.parametrize("small", [False, True])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 128, 64), (1000, 512, 512), (777, 1536, 384), (2500, 512, 2048),
                                   (33000, 512, 512), (11111, 1536, 96), (66000, 256, 128)])
def test_gemm_nt_plain(ops, M, N, K, small):
    """both tile variants: 128x128 (forced) and the 256x256 kernel that big row-streaming shapes dispatch to"""
    A, B = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2) / math.sqrt(K))
    ops.gemm_force_small_tile(small)
    try:
        out = ops.gemm_nt(A.to(DEV).bfloat16(), B.to(DEV).bfloat16(), M, N, K)
    finally:
        ops.gemm_force_small_tile(False)
    close(out.float(), A @ B.t(), 6e-3, 6e-3, "C")


@pytest.mark.parametrize("M,N", [(650, 256), (17000, 1024)])
def test_gemm_nt_epilogues(ops, M, N):
    K = 192
    A, B = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2) / math.sqrt(K))
    bias, res, msk = rnd(N, seed=3), bf(rnd(M, N, seed=4)), bf(rnd(M, N, seed=5))
    d = lambda t: t.to(DEV).bfloat16()
    base = A @ B.t()
    close(ops.gemm_nt(d(A), d(B), M, N, K, bias=bias.to(DEV), act=ops.ACT_RELU).float(), F.relu(base + bias), 6e-3, 6e-3, "bias+relu")
    close(ops.gemm_nt(d(A), d(B), M, N, K, bias=bias.to(DEV), residual=d(res)).float(), base + bias + res, 8e-3, 4e-2, "bias+res")  # cancellation: abs tol ~ ulp of the addends
    close(ops.gemm_nt(d(A), d(B), M, N, K, relu_mask=d(msk), residual=d(res)).float(), base * (msk > 0) + res, 8e-3, 4e-2, "mask+res")
    close(ops.gemm_nt(d(A), d(B), M, N, K, bias=bias.to(DEV), act=ops.ACT_GELU).float(), F.gelu(base + bias), 6e-3, 6e-3, "gelu")
    o32 = ops.gemm_nt(d(A), d(B), M, N, K, bias=bias.to(DEV), out_f32=True, alpha=0.5)
    assert o32.dtype == torch.float32
    close(o32, 0.5 * base + bias, 1e-4, 1e-4, "f32 out")
    # strided views: A is a column slice of a wider matrix, C a column slice of a wider output
    wide = bf(rnd(M, 3 * K, seed=6))
    outw = torch.zeros(M, 2 * N, device=DEV, dtype=torch.bfloat16)
    wd = d(wide)
    ops.gemm_nt(wd[:, K : 2 * K], d(B), M, N, K, out=outw[:, N:], lda=3 * K, ldc=2 * N)
    close(outw[:, N:].float(), wide[:, K : 2 * K] @ B.t(), 6e-3, 6e-3, "strided")
    assert (outw[:, :N] == 0).all()


@pytest.mark.parametrize("small", [False, True])
@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (1000, 128, 128), (5000, 512, 384), (3001, 1536, 512), (20000, 512, 2048),
                                   (65536, 256, 256), (100000 // 32 * 32, 512, 512), (70016, 1536, 256), (256 * 233, 512, 512), (256 * 233, 2048, 512)])
def test_gemm_tn(ops, M, N, K, small):
    """128x128 register-staged kernel (forced / small shapes) and the 256x256 LDS-DMA kernel, with the fused bias gradient"""
    dY, X = bf(rnd(M, N, seed=1)), bf(rnd(M, K, seed=2))
    dW = torch.ones(N, K, device=DEV)
    db = torch.ones(N, device=DEV)
    ops.gemm_force_small_tile(small)
    try:
        ops.gemm_tn_acc(dY.to(DEV).bfloat16(), X.to(DEV).bfloat16(), dW, M, N, K, db=db)
    finally:
        ops.gemm_force_small_tile(False)
    want = 1 + dY.double().t() @ X.double()
    close(dW, want, 2e-4, 2e-4 * math.sqrt(M), "dW")
    close(db, 1 + dY.double().sum(0), 2e-4, 2e-4 * math.sqrt(M), "db")


def test_colsum(ops):
    for M, N, rs in ((1000, 512, 1), (333, 2048, 1), (50, 512, 181), (100, 1536, 1)):
        dY = bf(rnd(M * rs, N, seed=1))
        db = torch.ones(N, device=DEV)
        ops.colsum_acc(dY.to(DEV).bfloat16(), db, M, N, row_stride=rs)
        close(db, 1 + dY[::rs].sum(0), 1e-4, 1e-3, f"colsum {M} {N} {rs}")


# ------------------------------------------------------------------------------------------------ attention
def attn_ref(q, k, v, scale, mask=None, bias=None):
    s = (q @ k.transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    return torch.softmax(s, -1) @ v


def _attn_case(ops, rows, S, H, mask_mode=0, traj=None, bias=None, kvalid=None, scale=0.125, bwd=True, seed=0):
    qkv = bf(rnd(rows * S, 3 * H * 64, seed=seed + 1))
    d_qkv = qkv.to(DEV).bfloat16()
    ld = 3 * H * 64
    q, k, v = [qkv[:, i * H * 64 : (i + 1) * H * 64].view(rows, S, H, 64).transpose(1, 2).clone().requires_grad_(True) for i in range(3)]
    mask = None
    if mask_mode == 1:
        mask = torch.tril(traj[:, :, None] == traj[:, None, :])[:, None]
    if kvalid is not None:
        km = kvalid.bool()[:, None, None, :]
        mask = km if mask is None else (mask & km)
    want = attn_ref(q, k, v, scale, mask, bias[None] if bias is not None else None)
    out, lse = ops.attn_fwd(d_qkv, d_qkv[:, H * 64 :], d_qkv[:, 2 * H * 64 :], ld, rows, S, H, scale, mask_mode=mask_mode,
                            traj=None if traj is None else traj.int().to(DEV), bias=None if bias is None else bias.to(DEV),
                            kvalid=None if kvalid is None else kvalid.to(torch.uint8).to(DEV))
    close(out.float().view(rows, S, H, 64), want.transpose(1, 2), 1e-2, 1e-2, f"O S={S}")
    if not bwd:
        return
    do = bf(rnd(rows * S, H * 64, seed=seed + 2))
    want.backward(do.view(rows, S, H, 64).transpose(1, 2))
    dqkv = torch.zeros_like(d_qkv)
    ops.attn_bwd(d_qkv, d_qkv[:, H * 64 :], d_qkv[:, 2 * H * 64 :], ld, out, H * 64, lse, do.to(DEV).bfloat16(), H * 64,
                 dqkv, dqkv[:, H * 64 :], dqkv[:, 2 * H * 64 :], ld, rows, S, H, scale, mask_mode=mask_mode,
                 traj=None if traj is None else traj.int().to(DEV), bias=None if bias is None else bias.to(DEV),
                 kvalid=None if kvalid is None else kvalid.to(torch.uint8).to(DEV))
    for i, (n, t) in enumerate((("dQ", q), ("dK", k), ("dV", v))):
        got = dqkv[:, i * H * 64 : (i + 1) * H * 64].float().view(rows, S, H, 64)
        w = t.grad.transpose(1, 2)
        close(got, w, 2e-2, 2e-2 * w.abs().max().item() + 1e-3, f"{n} S={S}")


@pytest.mark.parametrize("S", [5, 16, 64, 100, 181, 192, 233, 256])
def test_attn_nomask(ops, S):
    _attn_case(ops, 3, S, 8)


@pytest.mark.parametrize("Sq", [1, 5, 20])
def test_attn_query_subset(ops, Sq):
    """Sq > 0: only the first Sq queries of every row (last fusion layer: only sequence position 0 is consumed)."""
    rows, S, H = 4, 181, 8
    kv = bf(rnd(rows * S, 2 * H * 64, seed=1))
    qs = bf(rnd(rows * Sq, H * 64, seed=2))
    d_kv, d_q = kv.to(DEV).bfloat16(), qs.to(DEV).bfloat16()
    k, v = [kv[:, i * H * 64 : (i + 1) * H * 64].view(rows, S, H, 64).transpose(1, 2).clone().requires_grad_(True) for i in range(2)]
    q = qs.view(rows, Sq, H, 64).transpose(1, 2).clone().requires_grad_(True)
    want = attn_ref(q, k, v, 0.125)
    out, lse = ops.attn_fwd(d_q, d_kv, d_kv[:, H * 64 :], 2 * H * 64, rows, S, H, 0.125, Sq=Sq, ldq=H * 64)
    assert out.shape == (rows * Sq, H * 64) and lse.shape == (rows, H, Sq)
    close(out.float().view(rows, Sq, H, 64), want.transpose(1, 2), 1e-2, 1e-2, "O subset")
    do = bf(rnd(rows * Sq, H * 64, seed=3))
    want.backward(do.view(rows, Sq, H, 64).transpose(1, 2))
    dq = torch.zeros_like(d_q)
    dkv = torch.zeros_like(d_kv)
    ops.attn_bwd(d_q, d_kv, d_kv[:, H * 64 :], 2 * H * 64, out, H * 64, lse, do.to(DEV).bfloat16(), H * 64, dq, dkv, dkv[:, H * 64 :],
                 2 * H * 64, rows, S, H, 0.125, Sq=Sq, ldq=H * 64, lddq=H * 64)
    close(dq.float().view(rows, Sq, H, 64), q.grad.transpose(1, 2), 2e-2, 2e-2 * q.grad.abs().max().item() + 1e-3, "dQ subset")
    for i, (n, t) in enumerate((("dK", k), ("dV", v))):
        w = t.grad.transpose(1, 2)
        close(dkv[:, i * H * 64 : (i + 1) * H * 64].float().view(rows, S, H, 64), w, 2e-2, 2e-2 * w.abs().max().item() + 1e-3, n + " subset")


@pytest.mark.parametrize("W_", [512, 768])
def test_rows_add(ops, W_):
    R, S = 37, 11
    dst, src = bf(rnd(R, S, W_, seed=1)), bf(rnd(R, W_, seed=2))
    d = dst.to(DEV).bfloat16()
    ops.rows_add(d, S * W_, src.to(DEV).bfloat16(), W_, R, W_)
    want = dst.clone()
    want[:, 0] = bf(dst[:, 0] + src)
    assert torch.equal(d.float().cpu(), want)


def test_attn_block_causal(ops):
    for S, rows in ((128, 4), (256, 2), (32, 3)):
        g = torch.Generator().manual_seed(S)
        traj = torch.cumsum((torch.rand(rows, S, generator=g) < 0.05).long(), dim=1) + 3
        _attn_case(ops, rows, S, 8, mask_mode=1, traj=traj)


def test_attn_t5_bias_and_padding(ops):
    rows, S, H = 5, 11, 8
    bias = rnd(H, S, S, seed=9)
    kvalid = torch.ones(rows, S)
    for i, n in enumerate([11, 4, 7, 1, 9]):
        kvalid[i, n:] = 0
    _attn_case(ops, rows, S, H, bias=bias, kvalid=kvalid, scale=1.0, bwd=False)


@pytest.mark.parametrize("M,N,K", [(64, 1536, 512), (768, 2048, 512), (130, 512, 512), (64, 3072, 512)])
def test_gemm_nt_rmsnorm_fused(ops, M, N, K):
    """svla_gemm_nt_rmsa_bf16: RMSNorm(A) @ W^T with gamma folded into W and the row statistics taken from the GEMM's own A fragments (the pre-norm linears of
    the frozen T5 block and of the llama decoder block in an acting step), against fp32 torch and against the two-launch path (norm kernel, then GEMM)."""
    A = bf(rnd(M, K, seed=81, scale=1.7)).to(DEV).bfloat16(); W = bf(rnd(N, K, seed=82, scale=0.05)).to(DEV).bfloat16()
    gamma = (1.0 + 0.2 * rnd(K, seed=83)).to(DEV)
    res = bf(rnd(M, N, seed=84)).to(DEV).bfloat16()
    Wg = (W.float() * gamma[None, :]).bfloat16()
    got = ops.gemm_nt_rmsa(A, Wg, M, N, K, 1e-5, residual=res, act=ops.ACT_RELU)
    a32 = A.float()
    want = torch.relu((a32 * torch.rsqrt((a32 * a32).mean(-1, keepdim=True) + 1e-5)) @ Wg.float().t()) + res.float()
    close(got.float(), want, 1e-2, 2e-2, "norm-fused GEMM vs fp32 torch")
    n1, _, _ = ops.norm_fwd(A, gamma, None, 1e-5, M, rms=True, save_stats=False)
    two = ops.gemm_nt(n1, W, M, N, K, residual=res, act=ops.ACT_RELU)
    close(got.float(), two.float(), 2e-2, 3e-2, "norm-fused GEMM vs norm kernel + GEMM")


@pytest.mark.parametrize("S,kv_rows,window", [(500, 500, (37, 121)), (500, 500, (0, 499)), (45, 500, (3, 44)), (181, 181, None), (64, 64, (10, 10)), (500, 500, "none")])
def test_attn_single_query_decode_kernel(ops, S, kv_rows, window):
    """The single-query forward (Sq == 1: the KV-cached acting step of the llama decoder, allenact_dino_transformer.py:388-397; the pruned last fusion layer in
    eval mode) on the decode kernel that reads only the valid keys, against the tile kernels (hook 4) and fp32 torch: cache windows per row (kvalid), a cache
    larger than the attended length (kv_rows), all keys valid, a single valid key, NO valid key (output 0), and the log-sum-exp."""
    from safevla_amd._lib import lib
    rows, H, D = 5, 8, 512
    kv = bf(rnd(rows * kv_rows, 2 * D, seed=71)).to(DEV).bfloat16()
    q = bf(rnd(rows, 3 * D, seed=72)).to(DEV).bfloat16()
    kvalid = None
    if window is not None:
        kvalid = torch.zeros(rows, S, dtype=torch.uint8, device=DEV)
        if window != "none":
            for r in range(rows):
                lo, hi = min(window[0] + r, window[1]), window[1]
                kvalid[r, lo:hi + 1] = 1
    outs = {}
    try:
        for hook in (4, 0):
            lib().call("svla_attn_bwd_two_pass", hook)
            o, lse = ops.attn_fwd(q, kv, kv[:, D:], 2 * D, rows, S, H, 0.125, kvalid=kvalid, save_lse=True, Sq=1, ldq=3 * D, kv_rows=kv_rows)
            torch.cuda.synchronize()
            outs[hook] = (o.float(), lse.clone())
    finally:
        lib().call("svla_attn_bwd_two_pass", 0)
    K = kv[:, :D].float().view(rows, kv_rows, H, 64)[:, :S]; V = kv[:, D:].float().view(rows, kv_rows, H, 64)[:, :S]
    sc = torch.einsum("rhd,rshd->rhs", q[:, :D].float().view(rows, H, 64), K) * 0.125
    if kvalid is not None:
        sc = sc.masked_fill(~kvalid.bool()[:, None, :], float("-inf"))
    pr = torch.softmax(sc, -1).nan_to_num(0.0)
    want = torch.einsum("rhs,rshd->rhd", pr, V).reshape(rows, D)
    close(outs[0][0], want, 1e-2, 1e-2, "decode kernel vs fp32 torch")
    close(outs[0][0], outs[4][0], 1e-2, 2e-3, "decode kernel vs tile kernel")
    if window != "none":
        ok = torch.isfinite(outs[4][1])
        assert torch.allclose(outs[0][1][ok], outs[4][1][ok], rtol=1e-4, atol=1e-4)
        assert torch.allclose(outs[0][1].view(rows, H), torch.logsumexp(sc, -1), rtol=1e-3, atol=1e-3)
    else:
        assert (outs[0][0] == 0).all()


@pytest.mark.parametrize("S,H", [(433, 6), (257, 12), (300, 8), (448, 6), (449, 6), (512, 2), (417, 6), (272, 6), (273, 6), (288, 4), (289, 4)])      # 417 / 273: exact-tile kernels with an all-padding last tile
def test_attn_vit_length_fwd(ops, S, H):
    """S > 256 (ViT-S/14: 433 tokens, SigLIP-B/16: 257): the eight-wave long-sequence forward, both tile counts (28 / 32), ragged and full last tiles."""
    _attn_case(ops, 2, S, H, bwd=False)


# ------------------------------------------------------------------------------------------------ glue
def test_feat_to_tokens(ops):
    R = 7
    f0, f1 = rnd(R, 384, 7, 12, seed=1), rnd(R, 384, 7, 12, seed=2)
    out = torch.zeros(R, 2, 84, 384, device=DEV, dtype=torch.bfloat16)
    ops.feat_to_tokens(f0.to(DEV), out, 0)
    ops.feat_to_tokens(f1.to(DEV), out, 1)
    want = torch.stack([f0.flatten(2).permute(0, 2, 1), f1.flatten(2).permute(0, 2, 1)], 1)
    assert torch.equal(out.float().cpu(), bf(want))


@pytest.mark.parametrize("W_", [512, 768])
def test_fusion_fill_and_text_bwd(ops, W_):
    T, B, L, S = 9, 4, 6, 181
    R, U = T * B, 5
    ft, text = rnd(W_, seed=1), bf(rnd(U, L, W_, seed=2))
    g = torch.Generator().manual_seed(3)
    gid = torch.zeros(T, B, dtype=torch.int32)
    cur = torch.randint(0, U, (B,), generator=g)
    for t in range(T):
        flip = torch.rand(B, generator=g) < 0.3
        cur = torch.where(flip, torch.randint(0, U, (B,), generator=g), cur)
        gid[t] = cur
    x0 = torch.full((R, S, W_), 3.0, device=DEV, dtype=torch.bfloat16)
    ops.fusion_fill(ft.to(DEV), text.to(DEV).bfloat16(), gid.reshape(R).to(DEV), x0, R, S, L, 169)
    assert torch.equal(x0[:, 0].float().cpu(), bf(ft).expand(R, W_))
    assert torch.equal(x0[:, 169 : 169 + L].float().cpu(), text[gid.reshape(R).long()])
    assert (x0[:, 1:169] == 3).all() and (x0[:, 169 + L :] == 3).all()
    dx0 = bf(rnd(R, S, W_, seed=4))
    dtext = torch.zeros(U, L, W_, device=DEV)
    ops.fusion_text_bwd(dx0.to(DEV).bfloat16(), gid.reshape(R).to(DEV), T, B, S, L, 169, dtext)
    want = torch.zeros(U, L, W_).index_add_(0, gid.reshape(R).long(), dx0[:, 169 : 169 + L])
    close(dtext, want, 1e-5, 1e-5, "dtext")


@pytest.mark.parametrize("W_", [512, 768])
def test_decoder_embed(ops, W_):
    T, B, S = 6, 5, 181
    R = T * B
    g = torch.Generator().manual_seed(0)
    xf = bf(rnd(R, S, W_, seed=1))
    act, hand_t = rnd(22, W_, seed=2, scale=0.3), rnd(3, W_, seed=3, scale=0.3)
    div = torch.exp(torch.arange(0, W_, 2) * (-math.log(10000.0) / W_))
    pa = torch.randint(0, 20, (T, B), generator=g)
    masks = (torch.rand(T, B, generator=g) > 0.3).float()
    hand = torch.randint(0, 2, (T, B), generator=g)
    ts = torch.randint(0, 500, (T, B), generator=g)
    out = torch.empty(B * T, W_, device=DEV, dtype=torch.bfloat16)
    ops.decoder_embed_fwd(xf.to(DEV).bfloat16(), S * W_, act.to(DEV), hand_t.to(DEV), div.to(DEV), pa.to(DEV), masks.to(DEV),
                          hand.to(DEV), ts.to(DEV), T, B, out)
    pe = torch.zeros(T, B, W_)
    pe[..., 0::2] = torch.sin(ts.unsqueeze(-1) * div)
    pe[..., 1::2] = torch.cos(ts.unsqueeze(-1) * div)
    idx = torch.where(masks != 0, pa, torch.full_like(pa, 20))
    want = pe + xf[:, 0].view(T, B, W_) + act[idx] + hand_t[hand]
    close(out.float().view(B, T, W_).permute(1, 0, 2), want, 8e-3, 8e-3, "joint")
    dout = bf(rnd(B * T, W_, seed=5))
    dxf = torch.zeros(R, S, W_, device=DEV, dtype=torch.bfloat16)
    da, dh = torch.zeros(22, W_, device=DEV), torch.zeros(3, W_, device=DEV)
    ops.decoder_embed_bwd(dout.to(DEV).bfloat16(), pa.to(DEV), masks.to(DEV), hand.to(DEV), T, B, dxf, S * W_, da, dh)
    d_tb = dout.view(B, T, W_).permute(1, 0, 2).reshape(R, W_)
    assert torch.equal(dxf[:, 0].float().cpu(), d_tb) and (dxf[:, 1:] == 0).all()
    close(da, torch.zeros(22, W_).index_add_(0, idx.reshape(R), d_tb), 1e-5, 1e-4, "d act")
    close(dh, torch.zeros(3, W_).index_add_(0, hand.reshape(R), d_tb), 1e-5, 1e-4, "d hand")


def test_swiglu(ops):
    M, Hd = 333, 1536
    ab = bf(rnd(M, 2 * Hd, seed=1))
    abr = ab.clone().requires_grad_(True)
    want = F.silu(abr[:, :Hd]) * abr[:, Hd:]
    g = ops.swiglu_fwd(ab.to(DEV).bfloat16(), M, Hd)
    close(g.float(), want, 8e-3, 8e-3, "g")
    dg = bf(rnd(M, Hd, seed=2))
    want.backward(dg)
    dab = ops.swiglu_bwd(ab.to(DEV).bfloat16(), dg.to(DEV).bfloat16(), M, Hd)
    close(dab.float(), abr.grad, 1e-2, 1e-2, "dab")


def test_embed_gather_and_casts(ops):
    tab = rnd(1000, 512, seed=1)
    ids = torch.randint(0, 1000, (77,), generator=torch.Generator().manual_seed(2))
    out = ops.embed_gather(tab.to(DEV), ids.to(DEV))
    assert torch.equal(out.float().cpu(), bf(tab[ids]))
    w = rnd(1536, 512, seed=3)
    d1 = torch.empty(1536, 512, device=DEV, dtype=torch.bfloat16)
    d2 = torch.empty(512, 1536, device=DEV, dtype=torch.bfloat16)
    ops.cast_bf16(w.to(DEV), d1)
    ops.transpose_cast_bf16(w.to(DEV), d2)
    assert torch.equal(d1.float().cpu(), bf(w)) and torch.equal(d2.float().cpu(), bf(w).t())


def test_adam_clip_matches_torch(ops):
    n = 100_003
    p0, grads = rnd(n, seed=1), [rnd(n, seed=10 + i, scale=0.01 * (i + 1)) for i in range(3)]
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pt], lr=2e-5)
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pb = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    for i, g in enumerate(grads):
        pt.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([pt], 0.5)
        opt.step()
        gd = g.to(DEV)
        ss = torch.zeros(1, device=DEV, dtype=torch.float64)
        ops.sumsq(gd, ss)
        np.testing.assert_allclose(ss.item(), float(g.double().pow(2).sum()), rtol=1e-6)
        ops.adam_step(p, gd, m, v, pb, 2e-5, i + 1, gnorm_sq=ss, max_norm=0.5)
    close(p, pt.data, 1e-6, 1e-7, "adam p")
    assert torch.equal(pb.float().cpu(), bf(p.cpu()))


@pytest.mark.parametrize("M,force_small", [(700, True), (256 * 300 + 77, False)])
def test_gemm_nt_relu_bit_mask_roundtrip(ops, M, force_small):
    """relu_bits_out of a ReLU GEMM == (output > 0) packed LSB-first, and relu_bits masks exactly like relu_mask=<that output>."""
    torch.manual_seed(3)
    N, K = 512, 256
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16); B = (torch.randn(N, K, device=DEV) * 0.1).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV) * 0.1
    ops.gemm_force_small_tile(force_small)
    try:
        bits = torch.zeros(ops.relu_bits_bytes(M, N), device=DEV, dtype=torch.uint8)
        y = ops.gemm_nt(A, B, M, N, K, bias=bias, act=ops.ACT_RELU, relu_bits_out=bits)
        want = (y.float() > 0).view(M, N // 8, 8).to(torch.int32)
        packed = (want << torch.arange(8, device=DEV, dtype=torch.int32)).sum(-1).to(torch.uint8)       # [M, N/8] row-major
        MP = (M + 31) // 32 * 32                                                                       # -> blocked [M/32][N/64][32][8]
        pad = torch.zeros(MP, N // 8, device=DEV, dtype=torch.uint8); pad[:M] = packed
        blocked = pad.view(MP // 32, 32, N // 64, 8).permute(0, 2, 1, 3).reshape(-1)
        valid = torch.zeros(MP, N // 8, device=DEV, dtype=torch.bool); valid[:M] = True
        vb = valid.view(MP // 32, 32, N // 64, 8).permute(0, 2, 1, 3).reshape(-1)
        assert torch.equal(bits[vb], blocked[vb])
        assert 0.2 < want.float().mean().item() < 0.8
        dY = torch.randn(M, K, device=DEV).to(torch.bfloat16); W = (torch.randn(N, K, device=DEV) * 0.1).to(torch.bfloat16)
        a = ops.gemm_nt(dY, W, M, N, K, relu_mask=y)
        b = ops.gemm_nt(dY, W, M, N, K, relu_bits=bits)
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    finally:
        ops.gemm_force_small_tile(False)


def test_ce_loss_fwd_bwd_vs_torch(ops):
    """svla_ce_loss_fwd_bwd_f32 == F.cross_entropy(ignore_index=-1) and its gradient (early_fusion_tsfm_models.py:93,115-117)."""
    torch.manual_seed(5)
    R, A = 777, 20
    logits = (torch.randn(R, A, device=DEV) * 3).requires_grad_(True)
    tgt = torch.randint(0, A, (R,), device=DEV)
    tgt[torch.rand(R, device=DEV) < 0.3] = -1
    want = F.cross_entropy(logits, tgt, ignore_index=-1)
    want.backward()
    n_valid = (tgt != -1).sum().float().reshape(1)
    dl = torch.empty(R, A, device=DEV)
    sums = torch.zeros(1, device=DEV, dtype=torch.float64)
    ops.ce_loss_fwd_bwd(logits.detach(), tgt, n_valid, dl, sums)
    assert abs(sums.item() / n_valid.item() - want.item()) < 2e-6 * max(1.0, abs(want.item()))
    assert torch.allclose(dl, logits.grad, rtol=1e-5, atol=1e-8)
    assert (dl[tgt == -1] == 0).all()


# ------------------------------------------------------------------------------------------------ train-mode dropout
def _keep_np(seed, stream, p, idx):
    from oracle.ref_model import hash_keep
    return torch.from_numpy(hash_keep(seed, stream, p, idx))


@pytest.mark.parametrize("S,mask_mode", [(100, 0), (181, 0), (40, 1), (233, 0), (250, 0)])
def test_attn_dropout_fwd_bwd_vs_hash_reference(ops, S, mask_mode):
    """Dropout on the attention probabilities: generic kernels (S = 100, block-causal S = 40) and the persistent-forward /
    exact-tile backward kernels (S = 181) against torch with the SAME counter-based masks (include/svla.h: svla_dropout)."""
    from oracle.ref_model import hash_dropout
    rows, H, scale, p = 3, 8, 0.125, 0.1
    drop = ops.Dropout(seed=0xC0FFEE, stream=8, p=p)
    qkv = bf(rnd(rows * S, 3 * H * 64, seed=11))
    d_qkv = qkv.to(DEV).bfloat16()
    ld = 3 * H * 64
    q, k, v = [qkv[:, i * H * 64:(i + 1) * H * 64].view(rows, S, H, 64).transpose(1, 2).clone().requires_grad_(True) for i in range(3)]
    traj = None
    s = (q @ k.transpose(-1, -2)) * scale
    if mask_mode == 1:
        g = torch.Generator().manual_seed(3)
        traj = torch.sort(torch.randint(0, 3, (rows, S), generator=g), dim=1).values
        s = s.masked_fill(~torch.tril(traj[:, :, None] == traj[:, None, :])[:, None], float("-inf"))
    pr = hash_dropout(torch.softmax(s, -1), 0xC0FFEE, 8, p, attn_S=S)
    want = pr @ v
    kw = dict(mask_mode=mask_mode, traj=None if traj is None else traj.int().to(DEV), drop=drop)
    out, lse = ops.attn_fwd(d_qkv, d_qkv[:, H * 64:], d_qkv[:, 2 * H * 64:], ld, rows, S, H, scale, **kw)
    close(out.float().view(rows, S, H, 64), want.transpose(1, 2), 1e-2, 1e-2, f"O drop S={S}")
    nodrop, _ = ops.attn_fwd(d_qkv, d_qkv[:, H * 64:], d_qkv[:, 2 * H * 64:], ld, rows, S, H, scale, mask_mode=mask_mode, traj=kw["traj"])
    assert (nodrop.float() - out.float()).abs().max() > 0.05         # the mask really is applied
    do = bf(rnd(rows * S, H * 64, seed=12))
    want.backward(do.view(rows, S, H, 64).transpose(1, 2))
    dqkv = torch.zeros_like(d_qkv)
    ops.attn_bwd(d_qkv, d_qkv[:, H * 64:], d_qkv[:, 2 * H * 64:], ld, out, H * 64, lse, do.to(DEV).bfloat16(), H * 64,
                 dqkv, dqkv[:, H * 64:], dqkv[:, 2 * H * 64:], ld, rows, S, H, scale, **kw)
    for i, (n, t) in enumerate((("dQ", q), ("dK", k), ("dV", v))):
        w = t.grad.transpose(1, 2)
        close(dqkv[:, i * H * 64:(i + 1) * H * 64].float().view(rows, S, H, 64), w, 2e-2, 2e-2 * w.abs().max().item() + 1e-3, f"{n} drop S={S}")


@pytest.mark.parametrize("S,p", [(50, 0.0), (181, 0.0), (181, 0.1), (233, 0.1)])
def test_attn_bwd_two_pass_pair_still_matches_reference(ops, S, p):
    """The dQ + dK/dV kernel pair (kept behind svla_attn_bwd_two_pass for A/B) against torch, and against the default single-pass
    kernel on the same inputs: identical dropout masks, gradients equal up to bf16 rounding of differently ordered fp32 sums."""
    ops.attn_bwd_two_pass(True)
    try:
        if p > 0:
            test_attn_dropout_fwd_bwd_vs_hash_reference(ops, S, 0)
        else:
            _attn_case(ops, 3, S, 8)
    finally:
        ops.attn_bwd_two_pass(False)
    rows, H = 5, 8
    qkv = bf(rnd(rows * S, 3 * H * 64, seed=31)).to(DEV).bfloat16()
    kw = dict(drop=ops.Dropout(seed=5, stream=2, p=p)) if p > 0 else {}
    out, lse = ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, rows, S, H, 0.125, **kw)
    do = bf(rnd(rows * S, H * 64, seed=32)).to(DEV).bfloat16()
    res = []
    for two in (True, False):
        ops.attn_bwd_two_pass(two)
        try:
            d = torch.zeros_like(qkv)
            ops.attn_bwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, out, 512, lse, do, 512, d, d[:, 512:], d[:, 1024:], 1536, rows, S, H, 0.125, **kw)
            res.append(d.float())
        finally:
            ops.attn_bwd_two_pass(False)
    assert (res[0] - res[1]).abs().max().item() <= 2 ** -7 * res[0].abs().max().item()
    assert torch.nn.functional.cosine_similarity(res[0].flatten(), res[1].flatten(), dim=0).item() > 0.99999


@pytest.mark.parametrize("M,force_small,row_mult", [(700, True, 1), (256 * 300 + 77, False, 1), (256 * 300, False, 3)])
def test_gemm_nt_epilogue_dropout(ops, M, force_small, row_mult):
    """Dropout in the NT-GEMM epilogue (after the activation, before the residual add), 128- and 256-tile kernels, row_mult > 1
    (only every row_mult-th row of the logical tensor is materialised), and its interplay with the ReLU sign bits."""
    N, K, p = 512, 256, 0.1
    A = bf(rnd(M, K, seed=21)); B = bf(rnd(N, K, seed=22, scale=0.1)); bias = rnd(N, seed=23, scale=0.1); res = bf(rnd(M, N, seed=24))
    drop = ops.Dropout(seed=77, stream=5, p=p, row_mult=row_mult)
    idx = (np.arange(M, dtype=np.uint64)[:, None] * np.uint64(row_mult * N)) + np.arange(N, dtype=np.uint64)[None, :]
    keep = _keep_np(77, 5, p, idx).float()
    assert abs(keep.mean().item() - 0.9) < 5e-3
    d = lambda t: t.to(DEV).bfloat16()
    ops.gemm_force_small_tile(force_small)
    try:
        y = ops.gemm_nt(d(A), d(B), M, N, K, bias=bias.to(DEV), residual=d(res), drop=drop)
        want = bf((A @ B.t() + bias) * keep / (1 - p) + res)
        close(y.float(), want, 1e-2, 2e-2, "sub-layer output dropout")
        bits = torch.zeros(ops.relu_bits_bytes(M, N), device=DEV, dtype=torch.uint8)
        h = ops.gemm_nt(d(A), d(B), M, N, K, bias=bias.to(DEV), act=ops.ACT_RELU, relu_bits_out=bits, drop=drop)
        want_h = torch.relu(A @ B.t() + bias) * keep / (1 - p)
        close(h.float(), want_h, 1e-2, 2e-2, "activation dropout")
        assert ((h.float() == 0) | (keep.to(DEV) > 0)).all()            # every dropped element is exactly zero
        dY = d(bf(rnd(M, K, seed=25))); W = d(bf(rnd(N, K, seed=26, scale=0.1)))
        a = ops.gemm_nt(dY, W, M, N, K, relu_mask=h, alpha=1 / (1 - p))
        b = ops.gemm_nt(dY, W, M, N, K, relu_bits=bits, alpha=1 / (1 - p))
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))   # bits encode (pre-activation > 0 AND kept)
    finally:
        ops.gemm_force_small_tile(False)


def test_norm_bwd_dropout_output(ops):
    """norm_bwd's second output = dx * keep / (1-p): the gradient of the dropped-out sub-layer output feeding the residual stream."""
    M, D, p, rm = 1000, 512, 0.1, 181
    x, dy = bf(rnd(M, D, seed=31)), bf(rnd(M, D, seed=32))
    gamma, beta = 1 + 0.1 * rnd(D, seed=33), 0.1 * rnd(D, seed=34)
    d = lambda t: t.to(DEV).bfloat16()
    y, mean, rstd = ops.norm_fwd(d(x), gamma.to(DEV), beta.to(DEV), 1e-5, M)
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    drop = ops.Dropout(seed=5, stream=1, p=p, row_mult=rm)
    dxd = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    dx = ops.norm_bwd(d(dy), d(x), gamma.to(DEV), beta.to(DEV), mean, rstd, M, dg, db, dx_drop=dxd, drop=drop)
    idx = (np.arange(M, dtype=np.uint64)[:, None] * np.uint64(rm * D)) + np.arange(D, dtype=np.uint64)[None, :]
    keep = _keep_np(5, 1, p, idx).to(DEV)
    assert ((dxd.float() == 0) | keep).all()
    close(dxd.float()[keep], (dx.float() / (1 - p))[keep], 1e-2, 1e-3, "dx_drop kept elements")


# ------------------------------------------------------------------------------------------------ error behaviour / edge cases
def test_invalid_arguments_are_refused_not_computed(ops):
    """Every entry point validates its arguments and returns SVLA_EINVAL (-1) -> the binding raises; nothing is launched."""
    from safevla_amd._lib import SvlaError
    bf = torch.bfloat16
    A = torch.zeros(64, 64, device=DEV, dtype=bf); W = torch.zeros(128, 64, device=DEV, dtype=bf)
    with pytest.raises(SvlaError):
        ops.gemm_nt(A, W[:100], 64, 100, 64)                       # N not a multiple of 128
    with pytest.raises(SvlaError):
        ops.gemm_nt(A[:, :40], W[:, :40], 64, 128, 40)             # K not a multiple of 32
    with pytest.raises(SvlaError):
        ops.gemm_nt(A, W, 64, 128, 64, act=7)                      # unknown activation
    with pytest.raises(SvlaError):                                 # a bit mask cannot be combined with a residual
        ops.gemm_nt(A, W, 64, 128, 64, residual=torch.zeros(64, 128, device=DEV, dtype=bf),
                    relu_bits=torch.zeros(ops.relu_bits_bytes(64, 128), device=DEV, dtype=torch.uint8))
    qkv = torch.zeros(2 * 600, 3 * 64, device=DEV, dtype=bf)
    with pytest.raises(SvlaError):
        ops.attn_fwd(qkv, qkv[:, 64:], qkv[:, 128:], 192, 2, 600, 1, 0.125)       # S > 512
    with pytest.raises(SvlaError):
        ops.attn_fwd(qkv, qkv[:, 64:], qkv[:, 128:], 192, 2, 16, 1, 0.125, mask_mode=ops.MASK_BLOCK_CAUSAL)   # block-causal without traj ids
    with pytest.raises(SvlaError):
        ops.norm_fwd(torch.zeros(4, 256, device=DEV, dtype=bf), torch.ones(256, device=DEV), None, 1e-5, 4, D=256)  # only D = 384 / 512 are built
    with pytest.raises((ValueError, TypeError)):
        ops.gemm_nt(A.float(), W, 64, 128, 64)                     # wrong dtype is caught before the C call


def test_degenerate_sizes(ops):
    """Smallest legal problems: one row, one env, one step, one key."""
    # GAE with T = 1, B = 1
    one = lambda v: torch.tensor([[v]], device=DEV, dtype=torch.float32)
    ret, adv, cret, cadv = ops.gae_scan(one(1.0), one(2.0), one(0.5), one(0.25), torch.ones(2, 1, device=DEV), torch.tensor([0.7], device=DEV),
                                        torch.tensor([0.1], device=DEV), 0.99, 0.95)
    assert abs(adv.item() - (1.0 + 0.99 * 0.7 - 0.5)) < 1e-6 and abs(ret.item() - (adv.item() + 0.5)) < 1e-6
    assert abs(cadv.item() - (2.0 + 0.99 * 0.1 - 0.25)) < 1e-6
    # GEMM with a single row (128-tile kernel, ragged everywhere)
    A = bf(rnd(1, 64, seed=1)); W = bf(rnd(128, 64, seed=2))
    y = ops.gemm_nt(A.to(DEV).bfloat16(), W.to(DEV).bfloat16(), 1, 128, 64)
    close(y.float(), A @ W.t(), 1e-2, 1e-2, "1-row gemm")
    # attention with one key / one query
    qkv = bf(rnd(1, 3 * 64, seed=3)).to(DEV).bfloat16()
    o, lse = ops.attn_fwd(qkv, qkv[:, 64:], qkv[:, 128:], 192, 1, 1, 1, 0.125)
    close(o.float(), qkv[:, 128:].float(), 1e-2, 1e-2, "S = 1 attention returns V")


# ------------------------------------------------------------------------------------------------ generated assembly GEMMs (asmgen/)
def _asm_off(ops, off):
    from safevla_amd._lib import lib
    lib().call("svla_gemm_force_small_tile", 10 + (8192 if off else 0))


@pytest.mark.parametrize("N", [512, 1536, 384])      # 384: N % 256 == 128, the last 1-KiB bias chunk is clamped by the descriptor
@pytest.mark.parametrize("flavour", ["bias", "relu_bits", "relu_drop_bits", "bits_in"])
def test_gemm_nt_assembly_kernels(ops, N, flavour):
    """The A-stationary assembly kernels (svla_nt_as_*, asmgen/nt_as_gen.py; K = 512, >= 512 row panels) behind svla_gemm_nt_bf16: against the
    fp32 torch restatement of the epilogue (dropout mask from the counter definition), against the HIP kernels they replace (flag 8192 = assembly
    off: at most one bf16 rounding apart -- the bias is the accumulator's initial value instead of an addition at the end), sign bits equal,
    identical from run to run, ragged M (the tail rows run on the 128-tile kernel with the global row index)."""
    M, K, p = 256 * 520 + 77, 512, 0.1
    A = bf(rnd(M, K, seed=41)).to(DEV).bfloat16(); B = bf(rnd(N, K, seed=42, scale=0.05)).to(DEV).bfloat16(); bias = rnd(N, seed=43, scale=0.5).to(DEV)
    kw, want = {}, None
    acc = A.float() @ B.float().t()
    if flavour == "bias":
        kw = dict(bias=bias)
        want = acc + bias
    elif flavour in ("relu_bits", "relu_drop_bits"):
        kw = dict(bias=bias, act=ops.ACT_RELU)
        want = torch.relu(acc + bias)
        if flavour == "relu_drop_bits":
            kw["drop"] = ops.Dropout(seed=99, stream=3, p=p)
            idx = (np.arange(M, dtype=np.uint64)[:, None] * np.uint64(N)) + np.arange(N, dtype=np.uint64)[None, :]
            want = want * _keep_np(99, 3, p, idx).float().to(DEV) / (1 - p)
    else:
        mask = torch.rand(M, N, device=DEV) < 0.6
        h = torch.where(mask, 1.0, -1.0).bfloat16()
        bits = torch.zeros(ops.relu_bits_bytes(M, N), device=DEV, dtype=torch.uint8)
        _asm_off(ops, True)
        ops.gemm_nt(h, torch.eye(N, device=DEV).bfloat16(), M, N, N, act=ops.ACT_RELU, relu_bits_out=bits)      # sign bits of h through the HIP kernel
        kw = dict(relu_bits=bits, alpha=1 / (1 - p))
        want = torch.where(mask, acc / (1 - p), torch.zeros((), device=DEV))
    outs = {}
    try:
        for off in (True, False, False):
            _asm_off(ops, off)
            k2 = dict(kw)
            if "relu" in flavour:
                k2["relu_bits_out"] = torch.zeros(ops.relu_bits_bytes(M, N), device=DEV, dtype=torch.uint8)
            y = ops.gemm_nt(A, B, M, N, K, **k2)
            torch.cuda.synchronize()
            # the dispatch itself is asserted: assembly on -> the A-stationary flavour of this epilogue, off -> a HIP kernel
            name, mnk = ops.gemm_last_kernel()
            want_name = {"bias": "svla_nt_as_f0", "relu_bits": "svla_nt_as_f1", "relu_drop_bits": "svla_nt_as_f1d", "bits_in": "svla_nt_as_f3"}[flavour]
            assert (name == want_name and mnk == (M // 256 * 256, N, K)) if not off else name.startswith("gemm_nt"), (off, name, mnk)
            outs.setdefault(off, []).append((y, k2.get("relu_bits_out")))
    finally:
        _asm_off(ops, False)
    (hip, hip_bits), (a1, b1), (a2, b2) = outs[True][0], outs[False][0], outs[False][1]
    assert torch.equal(a1.view(torch.int16), a2.view(torch.int16)), "assembly kernel differs from run to run"
    close(a1.float(), want, 1e-2, 2e-2, f"asm {flavour} vs fp32 torch")
    d = (a1.float() - hip.float()).abs()
    assert (d <= hip.float().abs() * 2.0 ** -7 + 1e-6).all(), f"asm vs HIP kernel: max {d.max().item()}"
    if b1 is not None:
        assert torch.equal(b1, b2)
        assert (b1 != hip_bits).float().mean().item() < 1e-4          # a value that rounds to exactly 0 in one of the two may flip its bit
        pos = (a1.float() > 0)
        chk = torch.zeros(ops.relu_bits_bytes(M, N), device=DEV, dtype=torch.uint8)
        _asm_off(ops, True)
        try:
            ops.gemm_nt(torch.where(pos, 1.0, -1.0).bfloat16(), torch.eye(N, device=DEV).bfloat16(), M, N, N, act=ops.ACT_RELU, relu_bits_out=chk)
        finally:
            _asm_off(ops, False)
        assert torch.equal(chk, b1), "sign bits are not (output > 0)"


@pytest.mark.parametrize("M,N,K,flavour", [(256 * 45 + 64, 1536, 512, "bias"), (256 * 233, 2048, 512, "relu_drop_bits"), (256 * 60 + 5, 512, 512, "bits_in"),
                                             (256 * 30, 1024, 512, "relu_bits"), (55424, 1152, 384, "bias"), (55424, 384, 384, "bias"), (55424, 1536, 384, "gelu"), (256 * 600 + 9, 512, 384, "relu_bits")])
def test_gemm_nt_assembly_mid_m_launches(ops, M, N, K, flavour):
    """Mid-M launches of the A-stationary kernels (round 5: an acting step's 45 row panels, the 233 of the batch-256 probe, the ViT-S/14's 216 at K = 384):
    grid = panel slots x n-ranges (workgroup_id_y sweeps its own N / nsplit columns), no phases.  Forced on (hook 2) so that every flavour is exercised
    at sizes the cost model would leave to the tile kernels; against the HIP kernels (at most one bf16 rounding apart), fp32 torch, run to run."""
    from safevla_amd._lib import lib
    p = 0.1
    A = bf(rnd(M, K, seed=61)).to(DEV).bfloat16(); B = bf(rnd(N, K, seed=62, scale=0.05)).to(DEV).bfloat16(); bias = rnd(N, seed=63, scale=0.5).to(DEV)
    acc = A.float() @ B.float().t()
    if flavour == "bias":
        kw, want = dict(bias=bias), acc + bias
    elif flavour == "gelu":      # the frozen ViT's fc1: exact erf-GELU (torch) against the kernels' shared degree-9 polynomial form (asmgen/gelu_poly.py)
        kw, want = dict(bias=bias, act=ops.ACT_GELU), F.gelu(acc + bias)
    elif flavour in ("relu_bits", "relu_drop_bits"):
        kw, want = dict(bias=bias, act=ops.ACT_RELU), torch.relu(acc + bias)
        if flavour == "relu_drop_bits":
            kw["drop"] = ops.Dropout(seed=99, stream=3, p=p)
            idx = (np.arange(M, dtype=np.uint64)[:, None] * np.uint64(N)) + np.arange(N, dtype=np.uint64)[None, :]
            want = want * _keep_np(99, 3, p, idx).float().to(DEV) / (1 - p)
    else:
        mask = torch.rand(M, N, device=DEV) < 0.6
        bits = torch.zeros(ops.relu_bits_bytes(M, N), device=DEV, dtype=torch.uint8)
        _asm_off(ops, True)
        ops.gemm_nt(torch.where(mask, 1.0, -1.0).bfloat16(), torch.eye(N, device=DEV).bfloat16(), M, N, N, act=ops.ACT_RELU, relu_bits_out=bits)
        kw, want = dict(relu_bits=bits, alpha=1 / (1 - p)), torch.where(mask, acc / (1 - p), torch.zeros((), device=DEV))
    want_name = {"bias": "svla_nt_as_f0" if K == 512 else "svla_nt_as_k384_f0", "relu_bits": "svla_nt_as_f1" if K == 512 else "svla_nt_as_k384_f1",
                 "relu_drop_bits": "svla_nt_as_f1d", "bits_in": "svla_nt_as_f3", "gelu": "svla_nt_as_k384_f2"}[flavour]
    outs = []
    try:
        for mode in ("hip", "asm", "asm"):
            if mode == "hip":
                _asm_off(ops, True)
            else:
                lib().call("svla_gemm_force_small_tile", 2)
            k2 = dict(kw)
            if "relu" in flavour:
                k2["relu_bits_out"] = torch.zeros(ops.relu_bits_bytes(M, N), device=DEV, dtype=torch.uint8)
            y = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
            ops.gemm_nt(A, B, M, N, K, out=y, **k2)
            torch.cuda.synchronize()
            name, mnk = ops.gemm_last_kernel()
            assert (name == want_name and mnk == (M // 256 * 256, N, K)) if mode == "asm" else name.startswith("gemm_nt"), (mode, name, mnk)
            outs.append((y, k2.get("relu_bits_out")))
    finally:
        _asm_off(ops, False)
    (hip, hb), (a1, b1), (a2, b2) = outs
    assert torch.equal(a1.view(torch.int16), a2.view(torch.int16)), "assembly kernel differs from run to run"
    close(a1.float(), want, 1e-2, 2e-2, f"mid-M asm {flavour} vs fp32 torch")
    d = (a1.float() - hip.float()).abs()
    # (GELU: the two kernels form z = t^2 * zscale - 1 with different roundings: a few 1e-7 before the bf16 rounding, on values near zero too)
    assert (d <= hip.float().abs() * 2.0 ** -7 + (1e-5 if flavour == "gelu" else 1e-6)).all(), f"asm vs HIP kernel: max {d.max().item()}"
    if b1 is not None:
        assert torch.equal(b1, b2) and (b1 != hb).float().mean().item() < 1e-4


def test_gemm_nt_mid_m_cost_model_dispatch(ops):
    """Normal dispatch (no hook): the batch-256 probe's 233 panels x N = 1536 and the ViT's qkv projection go to the assembly kernels, an acting step's
    45 panels x N = 512 stay on the tile kernel (profiles/r05_midm_sweep.txt)."""
    for (M, N, K, want) in [(59648, 1536, 512, "svla_nt_as_f0"), (55424, 1152, 384, "svla_nt_as_k384_f0"), (11584, 512, 512, "gemm_nt_bf16_kernel")]:
        A = torch.zeros(M, K, device=DEV, dtype=torch.bfloat16); B = torch.zeros(N, K, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(A, B, M, N, K, bias=torch.zeros(N, device=DEV))
        torch.cuda.synchronize()
        assert ops.gemm_last_kernel()[0] == want, (M, N, K, ops.gemm_last_kernel())


@pytest.mark.parametrize("N,K", [(512, 2048), (512, 1536), (1024, 384), (512, 512)])
@pytest.mark.parametrize("flavour", ["plain", "bias", "res", "bias_res"])
def test_gemm_nt_output_stationary_assembly_kernels(ops, N, K, flavour):
    """The output-stationary assembly kernels (svla_nt_os_*, asmgen/nt_os_gen.py; K >= 384, no dropout) behind svla_gemm_nt_bf16 -- the input-gradient GEMMs
    through linear1 / in_proj of the fusion encoder (allenact_dino_transformer.py:545-552): against fp32 torch, against the 8-phase HIP kernel
    they replace (flag 8192 = assembly off; bit for bit without a residual), identical from run to run, ragged M (tail rows on the 128-tile kernel)."""
    if K == 512 and "res" not in flavour:
        pytest.skip("K = 512 without a residual runs on the A-stationary kernels")
    M = 256 * (520 if N == 512 else 260) + 77
    A = bf(rnd(M, K, seed=51)).to(DEV).bfloat16(); B = bf(rnd(N, K, seed=52, scale=0.05)).to(DEV).bfloat16()
    kw = {}
    want = A.float() @ B.float().t()
    if "bias" in flavour:
        kw["bias"] = rnd(N, seed=53, scale=0.5).to(DEV)
        want = want + kw["bias"]
    if "res" in flavour:
        kw["residual"] = bf(rnd(M, N, seed=54)).to(DEV).bfloat16()
        want = want + kw["residual"].float()
    outs = {}
    try:
        for off in (True, False, False):
            _asm_off(ops, off)
            y = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
            ops.gemm_nt(A, B, M, N, K, out=y, **kw)
            torch.cuda.synchronize()
            name, mnk = ops.gemm_last_kernel()
            want_name = "svla_nt_os_" + {"plain": "p", "bias": "b", "res": "r", "bias_res": "br"}[flavour]
            assert (name == want_name and mnk == (M // 256 * 256, N, K)) if not off else name.startswith("gemm_nt"), (off, name, mnk)
            outs.setdefault(off, []).append(y)
    finally:
        _asm_off(ops, False)
    hip, a1, a2 = outs[True][0], outs[False][0], outs[False][1]
    assert torch.equal(a1.view(torch.int16), a2.view(torch.int16)), "assembly kernel differs from run to run"
    close(a1.float(), want, 1e-2, 2e-2, f"asm {flavour} vs fp32 torch")
    if "res" in flavour:
        # the residual is added by v_dot2c_f32_bf16 (one instruction per element instead of unpack + add): the DOT unit's fp32 sum is not always the
        # correctly rounded one, so ~1e-5 of the elements land on the other side of a bf16 rounding boundary (measured: 99.9996 % identical)
        d = (a1.float() - hip.float()).abs()
        assert (d <= hip.float().abs() * 2.0 ** -7 + 1e-6).all(), f"asm vs HIP kernel: max {d.max().item()}"
        assert (a1.view(torch.int16) != hip.view(torch.int16)).float().mean().item() < 1e-4
    else:
        assert torch.equal(a1.view(torch.int16), hip.view(torch.int16)), f"asm vs HIP kernel: max {(a1.float() - hip.float()).abs().max().item()}"


@pytest.mark.parametrize("N,K", [(512, 512), (1536, 512), (512, 2048)])
def test_gemm_tn_assembly_kernel(ops, N, K):
    """The output-stationary assembly weight-gradient kernel (svla_tn_os, asmgen/tn_os_gen.py) behind svla_gemm_tn_f32acc: accumulation into a
    non-zero dW, fused bias gradient, against fp32 torch and the HIP 8-phase kernel."""
    M = 64 * 700
    dY = bf(rnd(M, N, seed=51)).to(DEV).bfloat16(); X = bf(rnd(M, K, seed=52)).to(DEV).bfloat16()
    w0, b0 = rnd(N, K, seed=53).to(DEV), rnd(N, seed=54).to(DEV)
    want_w = w0 + dY.float().t() @ X.float(); want_b = b0 + dY.float().sum(0)
    res = {}
    try:
        for off in (False, True):
            _asm_off(ops, off)
            dW, db = w0.clone(), b0.clone()
            ops.gemm_tn_acc(dY, X, dW, M, N, K, db=db)
            torch.cuda.synchronize()
            assert ops.gemm_last_kernel()[0] == ("gemm_tn8p_bf16_kernel" if off else "svla_tn_os"), (off, ops.gemm_last_kernel())
            res[off] = (dW, db)
    finally:
        _asm_off(ops, False)
    for off in (False, True):
        assert ((res[off][0] - want_w).abs().max() / want_w.abs().max()).item() < 1e-4
        assert ((res[off][1] - want_b).abs().max() / want_b.abs().max()).item() < 1e-4
