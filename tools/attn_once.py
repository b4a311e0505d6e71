import os, sys
sys.path.insert(0, "/root/repo")
import torch
from safevla_amd import ops
R, S = 16384, 181
drop = ops.Dropout(77, 3, 0.1)
qkv = (torch.randn(R * S, 1536, device="cuda") * 0.5).to(torch.bfloat16)
out, lse = ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125, drop=drop)
do = torch.randn_like(out); dqkv = torch.zeros_like(qkv)
f = lambda: ops.attn_bwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, out, 512, lse, do, 512, dqkv, dqkv[:, 512:], dqkv[:, 1024:], 1536, R, S, 8, 0.125, drop=drop)
g = lambda: ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125, drop=drop, out=out)
for name, fn in (("bwd", f), ("fwd", g)):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, f"{e0.elapsed_time(e1)/5:.3f} ms")
