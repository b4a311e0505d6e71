"""Oracle pieces whose arithmetic is NOT in /root/reference (parity unpinned): property tests + the T5 pin."""
import numpy as np
import torch

from oracle import ref_rollout
from oracle.detfill import fill_state_dict
from oracle.ref_t5 import RefT5Encoder


def test_t5_restatement_matches_hf():
    from transformers import T5Config, T5EncoderModel

    cfg = T5Config(vocab_size=32128, d_model=512, d_kv=64, d_ff=2048, num_layers=6, num_heads=8, feed_forward_proj="relu")
    hf = T5EncoderModel(cfg).eval()
    fill_state_dict(hf, seed=3)
    mine = RefT5Encoder().eval()
    missing = mine.load_state_dict(hf.state_dict(), strict=True)
    rs = np.random.RandomState(0)
    ids = torch.from_numpy(rs.randint(3, 32000, size=(5, 11)))
    am = torch.ones(5, 11, dtype=torch.int64)
    for i, n in enumerate([11, 4, 7, 1, 9]):
        ids[i, n:] = 0
        am[i, n:] = 0
    with torch.no_grad():
        want = hf(input_ids=ids, attention_mask=am).last_hidden_state
    got = mine(ids, am)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=1e-5)


def test_gae_scan_equals_definition():
    rs = np.random.RandomState(0)
    T, B = 37, 5
    r = torch.from_numpy(rs.standard_normal((T, B, 1)).astype(np.float32))
    v = torch.from_numpy(rs.standard_normal((T, B, 1)).astype(np.float32))
    m = torch.from_numpy((rs.rand(T + 1, B, 1) > 0.15).astype(np.float32))
    nv = torch.from_numpy(rs.standard_normal((B, 1)).astype(np.float32))
    ret, adv = ref_rollout.gae_scan(r, v, m, nv)
    ret2, adv2 = ref_rollout.gae_definition(r, v, m, nv)
    np.testing.assert_allclose(ret.numpy(), ret2.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(adv.numpy(), adv2.numpy(), rtol=1e-5, atol=1e-5)
    # episode boundary: nothing after a done leaks backwards
    m[10] = 0.0
    r2 = r.clone()
    r2[10:] += 100.0
    _, a = ref_rollout.gae_scan(r, v, m, nv)
    _, b = ref_rollout.gae_scan(r2, v, m, nv)
    np.testing.assert_allclose(a[:10].numpy(), b[:10].numpy(), rtol=0, atol=0)


def test_lagrange_monotone_and_clamped():
    lag = ref_rollout.RefLagrange(cost_limit=2.0, init=0.001, lr=0.035)
    up = [lag.update(5.0) for _ in range(5)]
    assert all(b > a for a, b in zip(up, up[1:]))
    down = [lag.update(0.0) for _ in range(400)]
    assert down[-1] == 0.0 and min(down) >= 0.0


def test_dropout_hash_statistics():
    """The counter-based dropout mask (include/svla.h: svla_dropout; oracle.ref_model.hash_keep == csrc/common.h: drop_bits): keep rate and
    independence of neighbouring elements / sites on 2e6 indices beyond 2^32 (both index words in play)."""
    import numpy as np
    from oracle.ref_model import hash_keep

    n = 2_000_000
    idx = np.arange(n, dtype=np.uint64) + np.uint64(6_000_000_000)
    sig = (0.9 * 0.1 / n) ** 0.5
    for seed, stream in [(0x5AFE, 3), (987654321, 40)]:
        k = hash_keep(seed, stream, 0.1, idx).astype(np.float64)
        assert abs(k.mean() - (1 - 6554 / 65536)) < 5 * sig
        for lag in (1, 2, 3, 4, 512, 2048):
            assert abs(np.corrcoef(k[:-lag], k[lag:])[0, 1]) < 5 / n ** 0.5
        k2 = hash_keep(seed, stream + 1, 0.1, idx).astype(np.float64)
        assert abs(np.corrcoef(k, k2)[0, 1]) < 5 / n ** 0.5


def test_branch_free_gelu_of_the_gemm_epilogue_is_the_erf_gelu():
    """csrc/gemm.hip gelu_f (the ViT MLP epilogue) evaluates nn.GELU's exact erf form through Abramowitz-Stegun 7.1.26 without a branch;
    this is its arithmetic restated in float32 numpy: absolute error <= 3e-7 (+ one float32 ulp of the value) everywhere, relative error below one bf16 ulp (the epilogue's
    output precision) wherever the value is representable at all."""
    import numpy as np
    import torch

    x = np.linspace(-12.0, 12.0, 400001, dtype=np.float32)
    az = np.abs(x) * np.float32(0.70710678118654752)
    t = np.float32(1.0) / (np.float32(0.3275911) * az + np.float32(1.0))
    q = t * np.float32(1.061405429) + np.float32(-1.453152027)
    q = q * t + np.float32(1.421413741)
    q = q * t + np.float32(-0.284496736)
    q = q * t + np.float32(0.254829592)
    q = q * t * np.exp2(x * x * np.float32(-0.72134752044448170))
    got = np.float32(0.5) * x * np.where(x >= 0, np.float32(2.0) - q, q)
    want = torch.nn.functional.gelu(torch.from_numpy(x).double()).numpy()
    assert (np.abs(got - want) <= 3e-7 + 1.2e-7 * np.abs(want)).all()          # 2e-7 of the formula + float32 rounding of the result
    big = np.abs(want) > 1e-6
    assert (np.abs(got - want)[big] / np.abs(want)[big]).max() < 2.0 ** -8


def test_fp8_attention_restatement_sits_on_the_format_ladder():
    """oracle/ref_fp8_attn.py (the quantisation-aware reference the fp8 kernels are gated against at 3 % relative Frobenius / 3e-3 mean error: tests/test_fp8_attention_gpu.py) against exact fp32 attention: e4m3 operands /
    probabilities and e5m2 gradients cost 4 % on O, 6 % on dV, 9 % on dQ / dK on N(0, 0.7) inputs -- the same figures the kernels measure
    (tests/test_fp8_attention_gpu.py), so restatement and kernels share their quantisation points and nothing else needs explaining."""
    import torch

    from oracle import ref_fp8_attn as R

    torch.manual_seed(0)
    rows, H, S = 2, 4, 100
    q, k, v = [(torch.randn(rows, H, S, 64) * 0.7).bfloat16().float().requires_grad_(True) for _ in range(3)]
    do = (torch.randn(rows, H, S, 64) * 0.02).bfloat16().float()
    o = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v
    o.backward(do)
    o8, lse8 = R.fwd(q.detach(), k.detach(), v.detach(), 0.125)
    dq, dk, dv = R.bwd(q.detach(), k.detach(), v.detach(), o8.bfloat16().float(), lse8, do, 0.125)
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    assert 0.02 < rel(o8, o.detach()) < 0.06
    assert 0.03 < rel(dv, v.grad) < 0.09 and 0.04 < rel(dq, q.grad) < 0.12 and 0.04 < rel(dk, k.grad) < 0.12
    lse = torch.logsumexp(q.detach() @ k.detach().transpose(-1, -2) * 0.125, -1)
    assert (lse8 - lse).abs().max().item() < 0.1
