"""Size-independent properties at the FULL benchmark sizes (BASELINE configs: T=256 x 32 envs per GPU => 8192 rows x 181
tokens = 1,482,752 GEMM rows), where the CPU oracle is too slow to run: linearity, two independent routes to the same
gradient, softmax invariants, norm statistics, chunking exactness."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
R, S = 8192, 181
M = R * S


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from safevla_amd import ops as o

    return o


def rb(*s, seed=0, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*s, device=DEV, generator=g) * scale).to(torch.bfloat16)


def relerr(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


def test_gemm_nt_linearity_and_row_independence(ops):
    """C(A, B1 + B2) == C(A, B1) + C(A, B2); every 4097th output row equals a direct dot product."""
    N, K = 512, 512
    A, B1, B2 = rb(M, K, seed=1), rb(N, K, seed=2, scale=0.05), rb(N, K, seed=3, scale=0.05)
    c1, c2 = ops.gemm_nt(A, B1, M, N, K, out_f32=False), ops.gemm_nt(A, B2, M, N, K)
    c12 = ops.gemm_nt(A, (B1.float() + B2.float()).to(torch.bfloat16), M, N, K)
    assert relerr(c12, c1.float() + c2.float()) < 8e-3
    rows = torch.arange(0, M, 4097, device=DEV)
    want = A[rows].float() @ B1.float().t()
    assert relerr(c1[rows], want) < 4e-3
    assert torch.isfinite(c1.float()).all()


def test_weight_grad_two_routes(ops):
    """dW from the TN kernel == sum over row blocks of small TN calls == (A^T-route) NT GEMM on a subsample."""
    N, K = 512, 512
    dY, X = rb(M, N, seed=4, scale=0.1), rb(M, K, seed=5)
    dW = torch.zeros(N, K, device=DEV)
    db = torch.zeros(N, device=DEV)
    ops.gemm_tn_acc(dY, X, dW, M, N, K, db=db)
    dW2 = torch.zeros(N, K, device=DEV)
    ops.gemm_force_small_tile(True)
    try:
        for c0 in range(0, M, M // 8):
            c1 = min(M, c0 + M // 8)
            ops.gemm_tn_acc(dY[c0:c1], X[c0:c1], dW2, c1 - c0, N, K)
    finally:
        ops.gemm_force_small_tile(False)
    assert relerr(dW, dW2) < 2e-4
    assert relerr(db, dY.float().sum(0)) < 2e-4
    # linear functional probe: <dW, G> == sum_m <dY_m, (X_m G^T)> evaluated through the NT kernel
    G = rb(N, K, seed=6, scale=0.05)
    XG = ops.gemm_nt(X, G, M, N, K, out_f32=False)          # [M, N] = X @ G^T
    lhs = (dW.double() * G.double()).sum().item()
    rhs = (dY.double() * XG.double()).sum().item()
    assert abs(lhs - rhs) < 2e-2 * (abs(rhs) + math.sqrt(M))


def test_attention_invariants_full_size(ops):
    """constant V => output == V; scaling all keys' values is linear; LSE shift under a constant score offset."""
    qkv = rb(M, 1536, seed=7)
    qkv[:, 1024:] = 0.75                                   # V = const
    out, lse = ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125)
    assert (out.float() - 0.75).abs().max().item() < 8e-3
    assert torch.isfinite(lse).all() and lse.shape == (R, 8, S)
    # dO with constant V: dQ and dK must vanish (P rows sum to 1 => dP constant per row => dS = 0)
    do = rb(M, 512, seed=8)
    dqkv = torch.empty_like(qkv)
    ops.attn_bwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, out, 512, lse, do, 512, dqkv, dqkv[:, 512:], dqkv[:, 1024:], 1536, R, S, 8, 0.125)
    assert dqkv[:, :1024].float().abs().max().item() < 2e-2
    # sum over keys of dV == sum over queries of dO (columns of P sum over keys to 1 per query)
    dv = dqkv[:, 1024:].float().view(R, S, 512).sum(1)
    dq = do.float().view(R, S, 512).sum(1)
    assert relerr(dv, dq) < 1e-2


def test_layernorm_statistics_full_size(ops):
    x = rb(M, 512, seed=9, scale=3.0)
    g, b = torch.ones(512, device=DEV), torch.zeros(512, device=DEV)
    y, mean, rstd = ops.norm_fwd(x, g, b, 1e-5, M)
    yf = y.float()
    assert yf.mean(-1).abs().max().item() < 2e-2 and (yf.var(-1, unbiased=False) - 1).abs().max().item() < 5e-2
    # backward of sum(y) is identically zero for LayerNorm (dy = 1 lies in the null space)
    dg, db_ = torch.zeros(512, device=DEV), torch.zeros(512, device=DEV)
    dx = ops.norm_bwd(torch.ones_like(x), x, g, b, mean, rstd, M, dg, db_)
    assert dx.float().abs().max().item() < 2e-2
    assert abs(db_.sum().item() - M * 512) < 1e-3 * M * 512


def test_gae_full_size_vs_fp64_recurrence(ops):
    T, B = 256, 256
    g = torch.Generator().manual_seed(0)
    r, v = torch.randn(T, B, generator=g), torch.randn(T, B, generator=g)
    m = (torch.rand(T + 1, B, generator=g) > 0.02).float()
    nv = torch.randn(B, generator=g)
    z = torch.zeros(T, B)
    ret, adv, _, _ = ops.gae_scan(r.to(DEV), z.to(DEV), v.to(DEV), z.to(DEV), m.to(DEV), nv.to(DEV), torch.zeros(B, device=DEV))
    gg, vn, want = torch.zeros(B, dtype=torch.float64), nv.double(), torch.zeros(T, B, dtype=torch.float64)
    for t in reversed(range(T)):
        d = r[t].double() + 0.99 * vn * m[t + 1].double() - v[t].double()
        gg = d + 0.99 * 0.95 * m[t + 1].double() * gg
        want[t] = gg
        vn = v[t].double()
    assert (adv.cpu().double() - want).abs().max().item() < 1e-4
