#!/usr/bin/env python3
"""ViT fc2 (M = 55 424 + pad, N = 384, K = 1536, bias + residual): one launch on the 8-phase HIP kernel (one and a half 256-tiles per row panel) against
the split N = 256 (output-stationary assembly kernel, forced) + N = 128 (tile kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
from safevla_amd._lib import lib
M, N, K = 55552, 384, 1536
A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").to(torch.bfloat16)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out1 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
one = lambda: ops.gemm_nt(A, W, M, N, K, bias=bias, residual=res, out=out1)
lib().call("svla_gemm_force_small_tile", 0)
print("one launch:", f"{t(one):.1f} us", ops.gemm_last_kernel())
out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
def split():
    lib().call("svla_gemm_force_small_tile", 2)
    ops.gemm_nt(A, W[:256], M, 256, K, bias=bias[:256], residual=res, ldr=N, out=out2, ldc=N)
    lib().call("svla_gemm_force_small_tile", 0)
    ops.gemm_nt(A, W[256:], M, 128, K, bias=bias[256:], residual=res[:, 256:], ldr=N, out=out2[:, 256:], ldc=N)
print("split 256 + 128:", f"{t(split):.1f} us", ops.gemm_last_kernel())
one(); split(); torch.cuda.synchronize()
d = (out1.float() - out2.float()).abs().max().item()
print("max diff", d)
lib().call("svla_gemm_force_small_tile", 2)
ops.gemm_nt(A, W[:256], M, 256, K, bias=bias[:256], residual=res, ldr=N, out=out2, ldc=N); print(ops.gemm_last_kernel())
p256 = lambda: ops.gemm_nt(A, W[:256], M, 256, K, bias=bias[:256], residual=res, ldr=N, out=out2, ldc=N)
print("N=256 part alone (forced asm):", f"{t(p256):.1f} us")
lib().call("svla_gemm_force_small_tile", 0)
p128 = lambda: ops.gemm_nt(A, W[256:], M, 128, K, bias=bias[256:], residual=res[:, 256:], ldr=N, out=out2[:, 256:], ldc=N)
print("N=128 part alone:", f"{t(p128):.1f} us", ops.gemm_last_kernel())
# proj: K = 384
K2 = 384
A2 = torch.randn(M, K2, device="cuda").to(torch.bfloat16); W2 = (torch.randn(N, K2, device="cuda") * 0.05).to(torch.bfloat16)
onep = lambda: ops.gemm_nt(A2, W2, M, N, K2, bias=bias, residual=res, out=out1)
print("proj one launch:", f"{t(onep):.1f} us", ops.gemm_last_kernel())
