#!/usr/bin/env python3
"""Weight-gradient GEMM: the output-stationary assembly kernel (asmgen/tn_os_gen.py, variant 0) vs the 8-phase HIP kernel (gemm_tn8p, flag 8192) vs the
2-buffer kernel (gemm_tn256, flag 128): agreement with fp32 torch, then timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
from safevla_amd._lib import lib
sel = lambda v: lib().call("svla_gemm_force_small_tile", 10 + v)
torch.manual_seed(0)
rb = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
for (M, n, k) in [(64 * 700, 512, 512), (64 * 1111, 256, 768), (64 * 600, 1536, 512), (64 * 513, 512, 2048)]:
    dY, X = rb(M, n), rb(M, k)
    ref = dY.float().t() @ X.float(); refb = dY.float().sum(0)
    outs = {}
    for v in (0, 8192):
        for rep in range(3):
            sel(v); dW = torch.zeros(n, k, device="cuda"); db = torch.zeros(n, device="cuda")
            ops.gemm_tn_acc(dY, X, dW, M, n, k, db=db); torch.cuda.synchronize()
            e = ((dW - ref).abs().max() / ref.abs().max()).item(); eb = ((db - refb).abs().max() / refb.abs().max()).item()
            if e > 2e-3 or eb > 2e-3: print(f"MISMATCH v{v} M={M} N={n} K={k}: rel err dW {e:.2e} db {eb:.2e} (rep {rep})", flush=True)
        outs[v] = (e, eb)
    print(f"checked M={M} N={n} K={k}: rel err vs fp32 torch  asm {outs[0][0]:.1e}/{outs[0][1]:.1e}  hip 8-phase {outs[8192][0]:.1e}/{outs[8192][1]:.1e}", flush=True)
    del dY, X
sel(0)
M = int(os.environ.get("AB_ROWS", 16384)) * 181
VAR = [int(v) for v in os.environ.get('AB_VARIANTS', '0,8192').split(',')]
for (n, k, bias) in [(512, 512, True), (1536, 512, True), (2048, 512, True), (512, 2048, True), (1024, 512, True), (512, 512, False)]:
    dY, X = rb(M, n), rb(M, k)
    dW = torch.zeros(n, k, device="cuda"); db = torch.zeros(n, device="cuda") if bias else None
    res = {}
    for rep in range(3):
        for v in VAR:
            sel(v)
            for _ in range(2): ops.gemm_tn_acc(dY, X, dW, M, n, k, db=db)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops.gemm_tn_acc(dY, X, dW, M, n, k, db=db)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 5)
    cyc = {}
    for v in VAR:      # shader cycles per phase (flag 4: the kernel writes them instead of the gradient)
        sel(v | 4); dW.zero_(); ops.gemm_tn_acc(dY, X, dW, M, n, k, db=db); torch.cuda.synchronize()
        c = dW.view(-1)[:256]; cyc[v] = (c[c > 0].mean().item(), c.max().item())
    sel(0)
    print("   cycles/phase (mean, max over workgroups; 512 = MFMA-bound): " + "  ".join(f"v{a}: {m:.0f}/{x:.0f}" for a, (m, x) in cyc.items()), flush=True)
    print(f"TN N={n} K={k} bias={bias}: " + "  ".join(f"v{a}: {min(t):.3f} ms ({2*M*n*k/min(t)/1e9:.0f} TF, {(M*(n+k)*2)/min(t)/1e9:.2f} TB/s)" for a, t in res.items()), flush=True)
    del dY, X
