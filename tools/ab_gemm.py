#!/usr/bin/env python3
"""256-tile NT GEMM: bit-compare with the 128-tile kernel over all epilogue flavours, then time (optionally with the
timing-only ablation flags f of svla_gemm_force_small_tile(10 + f): 1 = no C stores, 2 = no epilogue)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
from safevla_amd._lib import lib
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
sel = lambda v: lib().call("svla_gemm_force_small_tile", 10 + v)

# ---- correctness: every epilogue flavour, ragged M, several K
torch.manual_seed(0)
for (M, n, k) in [(256 * 300 + 77, 512, 512), (256 * 270, 1536, 128), (256 * 700 + 1, 256, 384), (256 * 256 + 255, 2048, 2048), (256 * 333 + 100, 1024, 1024)]:
    A = torch.randn(M, k, device="cuda").to(torch.bfloat16); B = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda"); res = torch.randn(M, n, device="cuda").to(torch.bfloat16)
    mask = torch.randn(M, n, device="cuda").to(torch.bfloat16)
    for name, kw in [("plain", {}), ("bias+relu", dict(bias=bias, act=1)), ("bias+gelu", dict(bias=bias, act=2)), ("bias+res", dict(bias=bias, residual=res)),
                     ("mask", dict(relu_mask=mask)), ("mask+res", dict(relu_mask=mask, residual=res, alpha=0.5)), ("relu+res", dict(bias=bias, act=1, residual=res)), ("gelu+res", dict(bias=bias, act=2, residual=res))]:
        lib().call("svla_gemm_force_small_tile", 1); ref = ops.gemm_nt(A, B, M, n, k, **kw); torch.cuda.synchronize()
        for v in [0, 128]:
            for rep in range(3):
                sel(v); out = torch.full_like(ref, float("nan")); ops.gemm_nt(A, B, M, n, k, out=out, **kw); torch.cuda.synchronize()
                bad = (out.view(torch.int16) != ref.view(torch.int16)).sum().item()
                if bad: print(f"MISMATCH v{v} M={M} N={n} K={k} {name}: {bad} elements differ (rep {rep})", flush=True); break
    del A, B, res, mask
    print(f"checked M={M} N={n} K={k}", flush=True)
sel(0)

M = int(os.environ.get("AB_ROWS", 16384)) * 181
for (n, k, flav) in [(512, 512, "plain"), (512, 512, "res"), (512, 512, "res+drop"), (1536, 512, "bias"), (2048, 512, "relu"), (2048, 512, "relu+drop"), (512, 2048, "res"), (512, 2048, "res+drop"), (512, 1536, "res"), (1024, 512, "bias"), (2048, 512, "bits_in")]:
    A = torch.randn(M, k, device="cuda").to(torch.bfloat16); B = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
    if os.environ.get("AB_CONST"):  # operands with few toggling bits: separates the schedule from the power-limited clock
        A.fill_(1.0); B.fill_(0.5)
    out = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(n, device="cuda")
    kw = {}
    if flav.startswith("res"): kw = dict(bias=bias, residual=torch.randn(M, n, device="cuda").to(torch.bfloat16))
    elif flav == "bias": kw = dict(bias=bias)
    elif flav.startswith("relu"): kw = dict(bias=bias, act=1, relu_bits_out=torch.empty(ops.relu_bits_bytes(M, n), device="cuda", dtype=torch.uint8))
    elif flav == "bits_in": kw = dict(relu_bits=torch.randint(0, 255, (ops.relu_bits_bytes(M, n),), device="cuda", dtype=torch.uint8), alpha=1.0 / 0.9)
    if flav.endswith("+drop"): kw["drop"] = ops.Dropout(1234, 5, 0.1)
    res = {}
    for rep in range(3):
        for abl in variants:
            sel(abl)
            for _ in range(2): ops.gemm_nt(A, B, M, n, k, out=out, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.gemm_nt(A, B, M, n, k, out=out, **kw)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(abl, []).append(e0.elapsed_time(e1) / 10)
    if os.environ.get("AB_CYCLES"):
        for extra, tag in ((0, "full"), (2, "no epilogue"), (2 + 8, "no epilogue, operands from L2"), (2 + 32, "no epilogue, no DMA")):
            sel(256 + 16 + extra); ops.gemm_nt(A, B, M, n, k, out=out, **kw); torch.cuda.synchronize()
            c = out.view(-1)[:1024].view(torch.float32)[:512].view(256, 2)
            print(f"   8-phase kernel ({tag}): {c[:, 0].mean().item():.0f} shader cycles per K-tile in the main loop (2048 = MFMA-bound), {c[:, 1].mean().item():.0f} per tile in the epilogue", flush=True)
    if os.environ.get("AB_KT"):
        for extra, tag in ((0, "full"), (2, "no epilogue"), (1, "no C stores")):
            sel(256 + 4096 + extra); ops.gemm_nt(A, B, M, n, k, out=out, **kw); torch.cuda.synchronize()
            c = out.view(-1)[16384:16384 + 4096].view(torch.float32)[:2048].view(256, 8).mean(0)
            print(f"   cycles by K-tile position after a tile boundary ({tag}): " + " ".join(f"{v:.0f}" for v in c.tolist()), flush=True)
    sel(0)
    print(f"N={n} K={k} {flav}: " + "  ".join(f"v{a}: {min(t):.3f} ms ({2*M*n*k/min(t)/1e9:.0f} TF)" for a, t in res.items()), flush=True)
    del A, B, out, kw
