#!/usr/bin/env python3
"""One shape of the 256-tile NT GEMM, a few launches (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
M = 8192 * 181
n, k = int(sys.argv[1]), int(sys.argv[2])
A = torch.randn(M, k, device="cuda").to(torch.bfloat16); B = torch.randn(n, k, device="cuda").to(torch.bfloat16)
out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
for _ in range(3): ops.gemm_nt(A, B, M, n, k, out=out)
torch.cuda.synchronize()
