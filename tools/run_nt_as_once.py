#!/usr/bin/env python3
"""a few launches of one K = 512 NT GEMM shape (for rocprofv3 --pmc / --kernel-trace runs): N from argv[1], optional variant from the environment"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
M = int(os.environ.get("AB_ROWS", 16384)) * 181
A = torch.randn(M, 512, device="cuda").to(torch.bfloat16); B = (torch.randn(n, 512, device="cuda") * 0.05).to(torch.bfloat16)
bias = torch.randn(n, device="cuda"); out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
for _ in range(int(os.environ.get("REPS", 4))): ops.gemm_nt(A, B, M, n, 512, bias=bias, out=out)
torch.cuda.synchronize()
