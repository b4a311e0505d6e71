#!/usr/bin/env python3
"""Time the NT GEMM on arbitrary shapes: shape_gemm.py variants M,N,K [M,N,K ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
from safevla_amd._lib import lib
variants = [int(v) for v in sys.argv[1].split(",")]
for spec in sys.argv[2:]:
    M, n, k = [int(x) for x in spec.split(",")]
    A = torch.randn(M, k, device="cuda").to(torch.bfloat16); B = torch.randn(n, k, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
    res = {}
    for rep in range(3):
        for v in variants:
            lib().call("svla_gemm_force_small_tile", 10 + v)
            for _ in range(2): ops.gemm_nt(A, B, M, n, k, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.gemm_nt(A, B, M, n, k, out=out)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 10)
    lib().call("svla_gemm_force_small_tile", 0)
    print(f"M={M} N={n} K={k}: " + "  ".join(f"v{a}: {min(t):.3f} ms ({2*M*n*k/min(t)/1e9:.0f} TF)" for a, t in res.items()), flush=True)
    del A, B, out
