#!/usr/bin/env python3
"""Per-kernel microbenchmarks at the bench workload's shapes (HIP-event timed, random bf16 data)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops

dev = "cuda"
def t_ms(fn, n=5, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

R, S = 8192, 181
M = R * S
M2 = R * 168
which = sys.argv[1:] or ["nt", "tn", "attn", "norm"]
rb = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
if "nt" in which:
    for (m, n, k, tag) in [(M, 1536, 512, "qkv"), (M, 512, 512, "out"), (M, 2048, 512, "ffn1"), (M, 512, 2048, "ffn2"), (M, 512, 1536, "dx_in"), (M2, 512, 384, "c1"), (M2, 512, 512, "c2"), (8192, 512, 512, "dec")]:
        A, B = rb(m, k), rb(n, k)
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        ms = t_ms(lambda: ops.gemm_nt(A, B, m, n, k, out=out))
        byt = (m * k + n * k + m * n) * 2
        print(f"gemm_nt {tag:6s} M={m} N={n} K={k}: {ms:8.3f} ms  {2*m*n*k/ms/1e9:8.1f} TF  {byt/ms/1e6:7.1f} GB/s(min)")
        res = rb(m, n); bias = torch.randn(n, device=dev)
        ms = t_ms(lambda: ops.gemm_nt(A, B, m, n, k, out=out, bias=bias, residual=res))
        print(f"   +bias+residual: {ms:8.3f} ms  {2*m*n*k/ms/1e9:8.1f} TF")
        del A, B, out, res
if "tn" in which:
    for (m, n, k, tag) in [(M, 1536, 512, "qkv"), (M, 512, 512, "out"), (M, 2048, 512, "ffn1"), (M, 512, 2048, "ffn2"), (M2, 512, 384, "c1")]:
        dY, X = rb(m, n), rb(m, k)
        dW = torch.zeros(n, k, device=dev)
        ms = t_ms(lambda: ops.gemm_tn_acc(dY, X, dW, m, n, k))
        byt = (m * k + m * n) * 2
        print(f"gemm_tn {tag:6s} M={m} N={n} K={k}: {ms:8.3f} ms  {2*m*n*k/ms/1e9:8.1f} TF  {byt/ms/1e6:7.1f} GB/s(min)")
        db = torch.zeros(n, device=dev)
        ms = t_ms(lambda: ops.gemm_tn_acc(dY, X, dW, m, n, k, db=db))
        print(f"   +fused bias grad: {ms:8.3f} ms  {2*m*n*k/ms/1e9:8.1f} TF")
        del dY, X
if "attn" in which:
    qkv = rb(M, 1536)
    out, lse = ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125)
    ms = t_ms(lambda: ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125, out=out))
    fl = 4 * S * S * 64 * R * 8
    print(f"attn_fwd R={R} S={S}: {ms:8.3f} ms {fl/ms/1e9:8.1f} TF")
    do = rb(M, 512); dqkv = torch.empty_like(qkv)
    ms = t_ms(lambda: ops.attn_bwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, out, 512, lse, do, 512, dqkv, dqkv[:, 512:], dqkv[:, 1024:], 1536, R, S, 8, 0.125), n=3, w=1)
    print(f"attn_bwd R={R} S={S}: {ms:8.3f} ms {2.5*fl/ms/1e9:8.1f} TF (5 GEMM-units algorithmic)")
    del qkv, out, do, dqkv
if "norm" in which:
    x = rb(M, 512); g = torch.ones(512, device=dev); b = torch.zeros(512, device=dev)
    y, mean, rstd = ops.norm_fwd(x, g, b, 1e-5, M)
    ms = t_ms(lambda: ops.norm_fwd(x, g, b, 1e-5, M, y=y))
    print(f"norm_fwd M={M}: {ms:8.3f} ms {M*512*4/ms/1e6:8.1f} GB/s")
    dg, db = torch.zeros(512, device=dev), torch.zeros(512, device=dev)
    dx = torch.empty_like(x)
    ms = t_ms(lambda: ops.norm_bwd(y, x, g, b, mean, rstd, M, dg, db, dx=dx))
    print(f"norm_bwd M={M}: {ms:8.3f} ms {M*512*6/ms/1e6:8.1f} GB/s")
    ms = t_ms(lambda: ops.colsum_acc(x, db, M, 512))
    print(f"colsum   M={M} N=512: {ms:8.3f} ms {M*512*2/ms/1e6:8.1f} GB/s")
