#!/usr/bin/env python3
"""A few launches of the fusion-shape attention forward/backward (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
R, S = 8192, 181
qkv = (torch.randn(R * S, 1536, device="cuda") * 0.5).to(torch.bfloat16)
out, lse = ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125)
for _ in range(2): ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125, out=out)
if len(sys.argv) > 1 and sys.argv[1] == "bwd":
    do = torch.randn_like(out); dqkv = torch.empty_like(qkv)
    for _ in range(2): ops.attn_bwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, out, 512, lse, do, 512, dqkv, dqkv[:, 512:], dqkv[:, 1024:], 1536, R, S, 8, 0.125)
torch.cuda.synchronize()
