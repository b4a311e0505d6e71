#!/usr/bin/env python3
"""One shape of the 256-tile TN GEMM, a few launches (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
M = 8192 * 181
n, k = int(sys.argv[1]), int(sys.argv[2])
dY = torch.randn(M, n, device="cuda").to(torch.bfloat16); X = torch.randn(M, k, device="cuda").to(torch.bfloat16)
dW = torch.zeros(n, k, device="cuda"); db = torch.zeros(n, device="cuda")
for _ in range(3): ops.gemm_tn_acc(dY, X, dW, M, n, k, db=db)
torch.cuda.synchronize()
