#!/usr/bin/env python3
"""Timing-only ablations of the 256-tile GEMM main loop (results are wrong by construction for k > 0)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
from safevla_amd._lib import lib
M = 8192 * 181
names = {0: "full", 1: "no MFMA", 2: "no vmcnt/barrier", 3: "no ds_read", 4: "no DMA"}
for (n, k) in [(512, 512), (512, 2048)]:
    A = torch.randn(M, k, device="cuda").to(torch.bfloat16); B = torch.randn(n, k, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
    for abl in (0, 1, 2, 3, 4):
        lib().call("svla_gemm_force_small_tile", 10 + abl)
        for _ in range(2): ops.gemm_nt(A, B, M, n, k, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.gemm_nt(A, B, M, n, k, out=out)
        e1.record(); torch.cuda.synchronize()
        print(f"N={n} K={k} {names[abl]:18s}: {e0.elapsed_time(e1)/5:7.3f} ms")
    lib().call("svla_gemm_force_small_tile", 0)
