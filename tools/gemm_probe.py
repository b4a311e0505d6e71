#!/usr/bin/env python3
"""Launch a few GEMMs of the bench shapes (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
dev = "cuda"
M = 8192 * 181
rb = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
for (n, k) in [(512, 512), (1536, 512), (512, 2048)]:
    A, B = rb(M, k), rb(n, k)
    out = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm_nt(A, B, M, n, k, out=out)
    dW = torch.zeros(n, k, device=dev)
    dY = rb(M, n)
    for _ in range(3):
        ops.gemm_tn_acc(dY, A, dW, M, n, k)
    torch.cuda.synchronize()
    del A, B, out, dY
