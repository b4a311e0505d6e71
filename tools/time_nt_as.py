#!/usr/bin/env python3
"""Step timing of the assembly NT kernel: cycles between consecutive workgroup barriers by kind of n-step (first / mid / last of a panel), for the
timing-only ablation builds (SVLA_ASM_DEBUG_VARIANTS=1 python safevla_amd/build.py).  4096 cycles = MFMA-bound."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
from safevla_amd._lib import lib
M = int(os.environ.get("AB_ROWS", 16384)) * 181
variants = (sys.argv[1] if len(sys.argv) > 1 else "time,time_nostore,time_nodma,time_noepi,time_nox,time_nobarwait,time_noepi_nodma_nox").split(",")
dbg = torch.zeros(256 * 4 * 8, device="cuda", dtype=torch.int32)
os.environ["SVLA_NT_AS_DBGBUF"] = hex(dbg.data_ptr())
for n in (512, 2048):
    A = torch.randn(M, 512, device="cuda").to(torch.bfloat16); B = (torch.randn(n, 512, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda"); out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
    for var in variants:
        os.environ["SVLA_NT_AS_VARIANT"] = var.replace("v1:", "")
        lib().call("svla_gemm_force_small_tile", 10 + (16384 if var.startswith("v1:") else 0))
        ts = []
        for rep in range(3):
            dbg.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm_nt(A, B, M, n, 512, bias=bias, out=out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        d = dbg.view(256, 4, 8).float()
        cyc = [(d[:, :, i].sum() / d[:, :, 3 + i].sum().clamp(min=1)).item() for i in range(3)]
        tot = d[:, :, :3].sum(-1).max().item()
        print(f"N={n} {var}: {min(ts):.3f} ms ({2*M*n*512/min(ts)/1e9:.0f} TF); cycles per step first {cyc[0]:.0f} mid {cyc[1]:.0f} last {cyc[2]:.0f}; "
              f"slowest wave {tot/1e6:.2f} Mcycles -> {tot/min(ts)/1e6:.2f} GHz-equivalent", flush=True)
    del A, B, out
