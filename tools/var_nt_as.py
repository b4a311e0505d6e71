#!/usr/bin/env python3
"""wall-clock A/B of generator variants of the assembly NT kernel (SVLA_ASM_DEBUG_VARIANTS=1 builds): interleaved rounds, min and median"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
M = int(os.environ.get("AB_ROWS", 16384)) * 181
variants = sys.argv[1].split(",")
for n in [int(x) for x in os.environ.get("AB_N", "512,1536,2048").split(",")]:
    A = torch.randn(M, 512, device="cuda").to(torch.bfloat16); B = (torch.randn(n, 512, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda"); out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
    res = {v: [] for v in variants}
    for rnd in range(5):
        for var in variants:
            if var == "base": os.environ.pop("SVLA_NT_AS_VARIANT", None)
            else: os.environ["SVLA_NT_AS_VARIANT"] = var
            ops.gemm_nt(A, B, M, n, 512, bias=bias, out=out); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops.gemm_nt(A, B, M, n, 512, bias=bias, out=out)
            e1.record(); torch.cuda.synchronize()
            res[var].append(e0.elapsed_time(e1) / 5)
    print(f"N={n}: " + "  ".join(f"{v}: {min(t):.3f}/{statistics.median(t):.3f} ms ({2*M*n*512/min(t)/1e9:.0f} TF)" for v, t in res.items()), flush=True)
    del A, B, out
