#!/usr/bin/env python3
"""wall-clock A/B of generator variants of the output-stationary assembly NT kernel (SVLA_ASM_DEBUG_VARIANTS=1 builds; residual flavour)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
M = int(os.environ.get("AB_ROWS", 16384)) * 181
variants = sys.argv[1].split(",")
for (n, K) in [(512, 2048), (512, 1024)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = (torch.randn(n, K, device="cuda") * 0.05).to(torch.bfloat16)
    R = torch.randn(M, n, device="cuda").to(torch.bfloat16) if os.environ.get("AB_FLAV", "r") == "r" else None; out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
    res = {v: [] for v in variants}
    for rnd in range(4):
        for var in variants:
            if var == "base": os.environ.pop("SVLA_NT_OS_VARIANT", None)
            else: os.environ["SVLA_NT_OS_VARIANT"] = var
            kw = dict(ldr=0) if (R is not None and os.environ.get("AB_LDR0")) else {}      # AB_LDR0: every residual row is row 0 (L2-resident): latency probe
            ops.gemm_nt(A, B, M, n, K, residual=R, out=out, **kw); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): ops.gemm_nt(A, B, M, n, K, residual=R, out=out, **kw)
            e1.record(); torch.cuda.synchronize()
            res[var].append(e0.elapsed_time(e1) / 4)
    print(f"N={n} K={K}: " + "  ".join(f"{v}: {min(t):.3f} ms ({2*M*n*K/min(t)/1e9:.0f} TF)" for v, t in res.items()), flush=True)
    del A, B, out, R
