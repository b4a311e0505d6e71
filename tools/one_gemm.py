#!/usr/bin/env python3
"""One GEMM shape launched N times (for rocprofv3 --kernel-trace --stats: GPU-side duration of a single launch, without the host's launch rate).
  python tools/one_gemm.py M N K [asm|hip] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
from safevla_amd._lib import lib
M, N, K = (int(x) for x in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "asm"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 50
lib().call("svla_gemm_force_small_tile", 10 + (8192 if mode == "hip" else 0))
A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(reps):
    ops.gemm_nt(A, B, M, N, K, out=out, bias=bias)
torch.cuda.synchronize()
print(ops.gemm_last_kernel())
