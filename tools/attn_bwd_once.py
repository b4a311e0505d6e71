#!/usr/bin/env python3
"""wall clock of the fusion-encoder attention forward / backward at the update's size (16 384 rows x 8 heads, S = 181, dropout 0.1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
R, S = int(os.environ.get("AB_ROWS", 16384)), int(os.environ.get("AB_S", 181))
drop = ops.Dropout(77, 3, 0.1)
qkv = (torch.randn(R * S, 1536, device="cuda") * 0.5).to(torch.bfloat16)
out, lse = ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125, drop=drop)
do = torch.randn_like(out); dqkv = torch.zeros_like(qkv)
f = lambda: ops.attn_bwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, out, 512, lse, do, 512, dqkv, dqkv[:, 512:], dqkv[:, 1024:], 1536, R, S, 8, 0.125, drop=drop)
g = lambda: ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125, drop=drop, out=out)
from safevla_amd._lib import lib
for name, fn in (("bwd", f), ("bwd row-major items", f), ("bwd", f), ("bwd row-major items", f), ("fwd", g), ("fwd row-major items", g), ("fwd", g), ("fwd row-major items", g)):
    lib().call("svla_attn_bwd_two_pass", 2 if "row-major" in name else 0)
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, f"{e0.elapsed_time(e1)/5:.3f} ms", "checksum", float(dqkv.float().abs().sum()) if name.startswith("bwd") else float(out.float().abs().sum()))
