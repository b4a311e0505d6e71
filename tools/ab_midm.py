#!/usr/bin/env python3
"""Mid-M launches of the A-stationary assembly kernels (grid = panel slots x n-ranges, no phases; K = 512 and the K = 384 ViT-S flavour) against the HIP tile
kernels they replace (flag 8192 = assembly off): correctness on the shapes of an acting step (M = 11 584), the batch-256 probe (59 648) and the ViT
(55 424), then timing.  SVLA_NT_AS_MIN_PANELS sweeps the threshold."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
from safevla_amd._lib import lib
sel = lambda v: lib().call("svla_gemm_force_small_tile", 10 + v)

def kwargs(flav, M, n):
    bias = torch.randn(n, device="cuda")
    if flav == "bias": return dict(bias=bias)
    if flav == "relu+drop": return dict(bias=bias, act=1, relu_bits_out=torch.zeros(ops.relu_bits_bytes(M, n), device="cuda", dtype=torch.uint8), drop=ops.Dropout(77, 3, 0.1))
    if flav == "relu": return dict(bias=bias, act=1, relu_bits_out=torch.zeros(ops.relu_bits_bytes(M, n), device="cuda", dtype=torch.uint8))
    if flav == "bits_in": return dict(relu_bits=torch.randint(0, 255, (ops.relu_bits_bytes(M, n),), device="cuda", dtype=torch.uint8), alpha=1.0 / 0.9)
    if flav == "gelu": return dict(bias=bias, act=2)
    if flav == "res": return dict(bias=bias, residual=torch.randn(M, n, device="cuda").to(torch.bfloat16))
    raise ValueError(flav)

torch.manual_seed(0)
nbad = 0
CASES = [(11584, 1536, 512, "bias"), (11584, 512, 512, "bias"), (11584, 2048, 512, "relu+drop"), (11584, 2048, 512, "bits_in"), (11584, 1024, 512, "relu"),
         (59648, 1536, 512, "bias"), (59648, 2048, 512, "relu+drop"), (59648, 512, 512, "bias"), (4096 + 13, 512, 512, "bias"), (100 * 256, 1536, 512, "bias"), (300 * 256, 512, 512, "bias"),
         (55424, 1152, 384, "bias"), (55424, 384, 384, "bias"), (55424, 1536, 384, "bias"), (10752, 512, 384, "bias")]
EXTRA = [c for c in [(55424, 1536, 384, "gelu"), (55424, 384, 384, "res")] if os.environ.get("AB_EXTRA")]
for (M, n, K, flav) in CASES + EXTRA:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = (torch.randn(n, K, device="cuda") * 0.05).to(torch.bfloat16)
    kw = kwargs(flav, M, n)
    kw_ref = dict(kw)
    if "relu_bits_out" in kw: kw_ref["relu_bits_out"] = torch.zeros_like(kw["relu_bits_out"])
    sel(8192); ref = ops.gemm_nt(A, B, M, n, K, **kw_ref); torch.cuda.synchronize()
    hipname = ops.gemm_last_kernel()[0]
    prev = None
    for rep in range(2):
        sel(0); out = torch.full_like(ref, float("nan")); ops.gemm_nt(A, B, M, n, K, out=out, **kw); torch.cuda.synchronize()
        name = ops.gemm_last_kernel()[0]
        d = (out.float() - ref.float()).abs()
        tol = ref.float().abs() * 2.0 ** -6 + 1e-2
        bad = int((~(d <= tol)).sum().item())
        same = prev is None or bool((out.view(torch.int16) == prev.view(torch.int16)).all().item())
        if "relu_bits_out" in kw:
            bb = int((kw["relu_bits_out"] != kw_ref["relu_bits_out"]).sum().item())
            if bb > 0.001 * kw["relu_bits_out"].numel(): bad += bb
        if bad or not same:
            nbad += 1
            print(f"MISMATCH M={M} N={n} K={K} {flav} rep {rep} ({name}): {bad} off (max {d.max().item():.4f}), repeatable {same}", flush=True)
            rows = torch.nonzero(~(d <= tol))[:, 0]
            if rows.numel(): print("   bad rows:", rows.min().item(), rows.max().item(), rows.unique().numel(), " cols:", torch.nonzero(~(d <= tol))[:, 1].unique()[:16].tolist(), flush=True)
            break
        prev = out
    res = {}
    outb = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
    for rep in range(3):
        for v in (0, 8192):
            sel(v)
            for _ in range(3): ops.gemm_nt(A, B, M, n, K, out=outb, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.gemm_nt(A, B, M, n, K, out=outb, **kw)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 20)
    sel(0)
    fl = 2.0 * M * n * K
    print(f"M={M:6d} N={n:5d} K={K:4d} {flav:10s} {name:22s} {min(res[0])*1e3:7.1f} us ({fl/min(res[0])/1e9:5.0f} TF)   {hipname:26s} {min(res[8192])*1e3:7.1f} us ({fl/min(res[8192])/1e9:5.0f} TF)", flush=True)
    del A, B, out, outb, kw, ref
print(f"{nbad} FAILURES" if nbad else "all correct", flush=True)
if os.environ.get("AB_SWEEP"):      # threshold study: panels x N, bias flavour, K = 512 (and the K = 384 flavour at the ViT's N)
    for K, Ns in ((512, (512, 1024, 1536, 2048)), (384, (384, 1152, 1536))):
        for n in Ns:
            for P in (24, 32, 45, 64, 96, 128, 160, 200, 233, 256, 300, 400, 500):
                M = 256 * P
                A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = (torch.randn(n, K, device="cuda") * 0.05).to(torch.bfloat16)
                bias = torch.randn(n, device="cuda"); outb = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
                res = {}
                for rep in range(3):
                    for v in (0, 8192):
                        sel(v)
                        for _ in range(3): ops.gemm_nt(A, B, M, n, K, out=outb, bias=bias)
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(20): ops.gemm_nt(A, B, M, n, K, out=outb, bias=bias)
                        e1.record(); torch.cuda.synchronize()
                        res.setdefault(v, []).append(e0.elapsed_time(e1) / 20)
                        if v == 0: nm = ops.gemm_last_kernel()[0]
                sel(0)
                print(f"SWEEP K={K} N={n:5d} panels={P:4d} asm({nm[:18]}) {min(res[0])*1e3:7.1f} us  hip {min(res[8192])*1e3:7.1f} us  ratio {min(res[8192])/min(res[0]):.2f}", flush=True)
                del A, B, outb
sys.exit(1 if nbad else 0)
