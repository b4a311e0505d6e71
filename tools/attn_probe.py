#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
R, S = 2048, 181
M = R * S
qkv = torch.randn(M, 1536, device="cuda").to(torch.bfloat16)
out, lse = ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125)
for _ in range(2):
    ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125, out=out)
do = torch.randn(M, 512, device="cuda").to(torch.bfloat16); dqkv = torch.empty_like(qkv)
for _ in range(2):
    ops.attn_bwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, out, 512, lse, do, 512, dqkv, dqkv[:, 512:], dqkv[:, 1024:], 1536, R, S, 8, 0.125)
torch.cuda.synchronize()
