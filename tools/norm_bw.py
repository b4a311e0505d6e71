#!/usr/bin/env python3
"""HBM rate of the LayerNorm kernels at the update's size (2.97 M rows x 512), per grid cap (SVLA_NORM_GRID, read once per process: run one cap per invocation)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
M = 16384 * 181
x = torch.randn(M, 512, device="cuda").to(torch.bfloat16); dy = torch.randn(M, 512, device="cuda").to(torch.bfloat16)
g = torch.ones(512, device="cuda"); b = torch.zeros(512, device="cuda"); dg = torch.zeros(512, device="cuda"); db = torch.zeros(512, device="cuda")
y, mean, rstd = ops.norm_fwd(x, g, b, 1e-5, M)
dx = torch.empty_like(x)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
f = t(lambda: ops.norm_fwd(x, g, b, 1e-5, M, y=y))
bw = t(lambda: ops.norm_bwd(dy, x, g, b, mean, rstd, M, dg, db, dx=dx))
nb = M * 512 * 2
print(f"grid cap {os.environ.get('SVLA_NORM_GRID', 'default')}: norm_fwd {f:.3f} ms = {2 * nb / f / 1e9:.2f} TB/s   norm_bwd {bw:.3f} ms = {3 * nb / bw / 1e9:.2f} TB/s")
