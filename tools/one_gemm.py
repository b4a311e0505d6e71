#!/usr/bin/env python3
"""One GEMM shape launched N times (for rocprofv3 --kernel-trace [--pmc ...]: GPU-side duration / counters of a single launch, without the host's launch rate).
  python tools/one_gemm.py M N K [asm|hip] [reps] [flavour]      flavour: bias (default) | relu_drop_bits | bits_in | res_drop
bench.py runs it under two rocprofv3 --pmc passes for the live `roofline.traffic` of the update's largest-share GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
from safevla_amd._lib import lib
M, N, K = (int(x) for x in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "asm"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 50
flav = sys.argv[6] if len(sys.argv) > 6 else "bias"
lib().call("svla_gemm_force_small_tile", 10 + (8192 if mode == "hip" else 0))
torch.manual_seed(0)
A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
kw = dict(bias=bias)
if flav == "relu_drop_bits":
    kw.update(act=ops.ACT_RELU, relu_bits_out=torch.zeros(ops.relu_bits_bytes(M, N), device="cuda", dtype=torch.uint8), drop=ops.Dropout(77, 3, 0.1))
elif flav == "bits_in":
    kw = dict(relu_bits=torch.randint(0, 255, (ops.relu_bits_bytes(M, N),), device="cuda", dtype=torch.uint8), alpha=1.0 / 0.9)
elif flav == "res_drop":
    kw.update(residual=torch.randn(M, N, device="cuda").to(torch.bfloat16), drop=ops.Dropout(1234, 5, 0.1))
for _ in range(reps):
    ops.gemm_nt(A, B, M, N, K, out=out, **kw)
torch.cuda.synchronize()
print(ops.gemm_last_kernel())
