#!/usr/bin/env python3
"""Localise mismatches of the rotating NT kernel (flag 512) against the 128-tile kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import Counter
from safevla_amd import ops
from safevla_amd._lib import lib
torch.manual_seed(0)
M, n, k = [int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (85348, 1024, 1024))]
A = torch.randn(M, k, device="cuda").to(torch.bfloat16); B = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
lib().call("svla_gemm_force_small_tile", 1); ref = ops.gemm_nt(A, B, M, n, k).float(); torch.cuda.synchronize()
for rep in range(12):
    lib().call("svla_gemm_force_small_tile", 10 + 512)
    out = torch.full((M, n), float("nan"), device="cuda", dtype=torch.bfloat16)
    ops.gemm_nt(A, B, M, n, k, out=out); torch.cuda.synchronize()
    d = (out.float() - ref).abs(); bad = ~(d <= 8e-3 * ref.abs() + 2e-2)
    nb = bad.sum().item()
    if not nb: print(f"rep {rep}: ok"); continue
    idx = bad.nonzero()
    rows, cols = idx[:, 0], idx[:, 1]
    c = Counter(zip((rows // 256).tolist(), ((rows % 256) // 32).tolist(), (cols // 64).tolist()))
    print(f"rep {rep}: {nb} bad elements in {len(c)} (m-block, row block of 32, 64-col group) slabs; nan {(out != out).sum().item()}")
    for (mb, rb_, cg), cnt in sorted(c.items())[:24]:
        sub = bad[mb * 256 + rb_ * 32: mb * 256 + rb_ * 32 + 32, cg * 64: cg * 64 + 64]
        print(f"   m-block {mb} (xcd {mb % 8}, t {mb // 8}) rows {rb_*32}-{rb_*32+31} (group {rb_ // 4} block {rb_ % 4}) cols {cg*64}-{cg*64+63} (n-tile {cg // 4} wn {cg % 4}): {cnt} bad; bad rows in slab {sorted(set(sub.nonzero()[:,0].tolist()))[:12]} bad cols {sorted(set(sub.nonzero()[:,1].tolist()))[:16]}")
lib().call("svla_gemm_force_small_tile", 0)
