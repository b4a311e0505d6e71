#!/usr/bin/env python3
"""Does the row stride of C (L2 channel interleave) matter for the NT epilogue stores?  Same GEMM, padded leading dimension."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
M = 8192 * 181
def t_ms(f, n=10, w=2):
    for _ in range(w): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (n, k) in [(2048, 512), (512, 512), (1536, 512)]:
    A = torch.randn(M, k, device="cuda").to(torch.bfloat16); B = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
    for pad in (0, 64, 128, 192, 256):
        buf = torch.empty(M, n + pad, device="cuda", dtype=torch.bfloat16)
        out = buf[:, :n]
        ms = t_ms(lambda: ops.gemm_nt(A, B, M, n, k, out=out))
        print(f"N={n} K={k} ldc={n+pad}: {ms:.3f} ms ({2*M*n*k/ms/1e9:.0f} TF)", flush=True)
        del buf, out
    # A's leading dimension too (reads)
    for pad in (64,):
        abuf = torch.empty(M, k + pad, device="cuda", dtype=torch.bfloat16); abuf[:, :k] = A
        out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
        ms = t_ms(lambda: ops.gemm_nt(abuf[:, :k], B, M, n, k, out=out))
        print(f"N={n} K={k} lda={k+pad}: {ms:.3f} ms ({2*M*n*k/ms/1e9:.0f} TF)", flush=True)
        del abuf, out
