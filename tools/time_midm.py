#!/usr/bin/env python3
"""Step timing of the mid-M / K = 384 launches of the A-stationary kernels (cycles between consecutive workgroup barriers by kind of n-step), timing builds:
  SVLA_ASM_DEBUG_VARIANTS=1 SVLA_EXTRA_FLAGS=-DSVLA_ASM_DEBUG python safevla_amd/build.py --force     (rebuild without them afterwards)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
from safevla_amd._lib import lib
dbg = torch.zeros(256 * 4 * 8 * 8, device="cuda", dtype=torch.int32)
os.environ["SVLA_NT_AS_DBGBUF"] = hex(dbg.data_ptr())
CASES = [(55424, 1536, 384, 0, "time,time_nostore,time_nodma,time_noepi,time_noepi_nodma"), (55424, 1152, 384, 0, "time"),
         (59648, 1536, 512, 0, "time,time_nostore,time_nodma,time_noepi"), (16384 * 181, 1536, 512, 0, "time")]
for (M, n, K, act, variants) in CASES:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = (torch.randn(n, K, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda"); out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
    for var in variants.split(","):
        os.environ["SVLA_NT_AS_VARIANT"] = var
        lib().call("svla_gemm_force_small_tile", 2 if M < 10**6 else 0)
        ts = []
        for rep in range(3):
            dbg.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm_nt(A, B, M, n, K, bias=bias, out=out, act=act); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        name = ops.gemm_last_kernel()[0]
        d = dbg.view(-1, 4, 8).float()
        cyc = [(d[:, :, i].sum() / d[:, :, 3 + i].sum().clamp(min=1)).item() for i in range(3)]
        tot = d[:, :, :3].sum(-1).max().item()
        print(f"M={M} N={n} K={K} act={act} {name}: {min(ts)*1e3:.1f} us ({2*M*n*K/min(ts)/1e9:.0f} TF); cycles per step first {cyc[0]:.0f} mid {cyc[1]:.0f} last {cyc[2]:.0f}; "
              f"slowest wave {tot/1e3:.1f} kcycles in steps -> {tot/min(ts)/1e6:.2f} GHz-equivalent if the steps were the whole launch", flush=True)
    del A, B, out
