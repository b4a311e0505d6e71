#!/usr/bin/env python3
"""Output-stationary assembly NT kernels (asmgen/nt_os_gen.py) vs the 8-phase HIP kernel: correctness on ragged M (against the HIP kernel, bit for bit
where the epilogue arithmetic is the same, and repeated: races show up as run-to-run differences), then timing A/B (flag 16384 of
svla_gemm_force_small_tile(10 + f) = these kernels off)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
from safevla_amd._lib import lib
sel = lambda v: lib().call("svla_gemm_force_small_tile", 10 + v)

def kwargs(flav, M, n):
    kw = {}
    if "b" in flav: kw["bias"] = torch.randn(n, device="cuda")
    if "r" in flav: kw["residual"] = torch.randn(M, n, device="cuda").to(torch.bfloat16)
    return kw

torch.manual_seed(0)
nbad = 0
for (M, n, K) in [(256 * 600 + 77, 512, 2048), (256 * 520, 512, 1536), (256 * 300 + 255, 1024, 1024), (256 * 513 + 1, 512, 384), (256 * 512, 512, 512), (256 * 260, 1024, 256)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = (torch.randn(n, K, device="cuda") * 0.05).to(torch.bfloat16)
    for flav in ("p", "b", "r", "br"):
        if K == 512 and "r" not in flav: continue        # those run on the A-stationary kernels
        kw = kwargs(flav, M, n)
        sel(16384); ref = ops.gemm_nt(A, B, M, n, K, **kw); torch.cuda.synchronize()
        prev = None
        for rep in range(3):
            sel(0); out = torch.full_like(ref, float("nan")); ops.gemm_nt(A, B, M, n, K, out=out, **kw); torch.cuda.synchronize()
            d = (out.float() - ref.float()).abs()
            tol = ref.float().abs() * 2.0 ** -6 + 1e-2
            bad = int((~(d <= tol)).sum().item())
            same = prev is None or bool((out.view(torch.int16) == prev.view(torch.int16)).all().item())
            if bad or not same:
                nbad += 1
                print(f"MISMATCH M={M} N={n} K={K} {flav} rep {rep}: {bad} elements off (max {d.max().item():.4f}), repeatable {same}", flush=True)
                rows = torch.nonzero(~(d <= tol))[:, 0]
                if rows.numel(): print("   bad rows (first/last/count of distinct):", rows.min().item(), rows.max().item(), rows.unique().numel(), " cols:", torch.nonzero(~(d <= tol))[:, 1].unique()[:16].tolist(), flush=True)
                break
            prev = out
        exact = float((out.view(torch.int16) == ref.view(torch.int16)).float().mean().item())
        print(f"checked M={M} N={n} K={K} {flav}: identical to the HIP kernel in {100 * exact:.3f} % of the elements", flush=True)
    del A, B
if nbad: print(f"{nbad} FAILURES", flush=True)
if os.environ.get("AB_NOTIME"): sys.exit(1 if nbad else 0)
M = int(os.environ.get("AB_ROWS", 16384)) * 181
for (n, K, flav) in [(512, 2048, "r"), (512, 1536, "r"), (512, 1024, "r"), (512, 384, "r"), (512, 2048, "br"), (512, 512, "br")]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); B = (torch.randn(n, K, device="cuda") * 0.05).to(torch.bfloat16)
    out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
    kw = kwargs(flav, M, n)
    res = {}
    for rep in range(3):
        for v in (0, 16384):
            sel(v)
            for _ in range(2): ops.gemm_nt(A, B, M, n, K, out=out, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops.gemm_nt(A, B, M, n, K, out=out, **kw)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 5)
    f = 2.0 * M * n * K
    print(f"N={n} K={K} {flav}: " + "  ".join(f"{'asm' if v == 0 else 'hip'}: {min(t):.3f} ms ({f / min(t) / 1e9:.0f} TF)" for v, t in res.items()), flush=True)
    del A, B, out, kw
sel(0)
