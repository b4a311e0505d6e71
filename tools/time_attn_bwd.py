#!/usr/bin/env python3
"""Phase timing of the single-pass attention backward (build with -DATT_TIMING: SVLA_EXTRA_FLAGS=-DATT_TIMING python safevla_amd/build.py --force):
cycles per (row, head) item and wave in [stage Q / dO / D / lse] [dK, dV] [fragments -> registers, K / V -> LDS] [dQ + stores]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
R, S = int(os.environ.get("AB_ROWS", 16384)), int(os.environ.get("AB_S", 181))
dbg = torch.zeros(R * 8 * 4 * 4, device="cuda", dtype=torch.int32)
os.environ["SVLA_ATTN_DBGBUF"] = hex(dbg.data_ptr())
drop = ops.Dropout(77, 3, 0.1)
qkv = (torch.randn(R * S, 1536, device="cuda") * 0.5).to(torch.bfloat16)
out, lse = ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125, drop=drop)
do = torch.randn_like(out); dqkv = torch.zeros_like(qkv)
f = lambda: ops.attn_bwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, out, 512, lse, do, 512, dqkv, dqkv[:, 512:], dqkv[:, 1024:], 1536, R, S, 8, 0.125, drop=drop)
f(); torch.cuda.synchronize(); dbg.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
d = dbg.view(R * 8, 4, 4).double()
n = R * 8
print(f"R={R} S={S}: {ms:.3f} ms; items {n}")
for w in range(4):
    c = d[:, w, :].mean(0)
    print(f"  wave {w}: stage {c[0]:.0f}  dK/dV {c[1]:.0f}  switch {c[2]:.0f}  dQ + stores {c[3]:.0f}  total {c.sum():.0f} cycles/item")
tot = d[:, 0, :].sum(-1).mean().item()
print(f"  item time x items / (768 workgroup slots) = {tot * n / 768 / 1e6:.2f} Mcycles -> {tot * n / 768 / ms / 1e6:.2f} GHz-equivalent")
