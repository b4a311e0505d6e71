#!/usr/bin/env python3
"""CPU-only: random launch geometries of the generated kernels (A-stationary NT; with --family os / tn the output-stationary NT and the weight-gradient kernel) through the lane-accurate emulator -- flavour x row panels x N x workgroups x n-range split (mid-M
launches) x wave issue order x dropout row multiplier -- against the numpy restatement of the epilogues (the harness of tests/test_asm_emulator_cpu.py).  The emulator
checks what hardware runs cannot show: a read of LDS-DMA data that has not landed, a counted s_waitcnt that is one too high, the DOT-result hazard, in EVERY wave order.

    python tools/fuzz_asm_emulator.py [--seed 0] [--cases 24]
"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_asm_emulator_cpu import check, run_kernel, run_nt_os, run_tn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=24)
    ap.add_argument("--family", default="as", choices=["as", "os", "tn"])
    args = ap.parse_args()
    rng = random.Random(args.seed)
    bad = 0
    for i in range(args.cases if args.family != "as" else 0):
        t0 = time.time()
        try:
            if args.family == "os":      # svla_nt_os_*: K > 512-style shapes (K a multiple of 128, >= 384), tiles of 256 x 256, any grid, both wave orders
                fl, mt, nt, Kc = rng.choice(["p", "b", "r", "br"]), rng.choice([1, 1, 2, 3]), rng.choice([1, 2, 3, 4]), rng.choice([384, 512, 640, 768, 1024])
                grid, order = rng.randint(1, mt * nt), rng.choice([None, [3, 2, 1, 0], [2, 0, 3, 1]])
                tag = f"nt_os_{fl} M={256 * mt} N={256 * nt} K={Kc} grid={grid} order={order}"
                out, ref = run_nt_os(fl, 256 * mt, 256 * nt, Kc, grid, order=order, seed=rng.randint(0, 999))
                check(out, ref)
            else:                        # svla_tn_os: row chunks of 32-row slots over the workgroups, 256 x 256 tiles of dW, fused bias gradient
                N, Kc = 256 * rng.choice([1, 1, 2]), 256 * rng.choice([1, 2])
                chunk = 32 * rng.randint(1, 6)
                chunks = rng.randint(1, 3)
                grid = (N // 256) * (Kc // 256) * chunks          # one workgroup per (256 x 256 tile of dW, row chunk): csrc/gemm.hip svla_gemm_tn_f32acc
                M = chunk * (chunks - 1) + 32 * rng.randint(1, chunk // 32)
                wb = rng.random() < 0.7
                tag = f"tn_os M={M} N={N} K={Kc} chunk_rows={chunk} chunks={chunks} grid={grid} bias_grad={int(wb)}"
                dW, rW, db, rb = run_tn(M, N, Kc, chunk, grid, with_bias=wb, seed=rng.randint(0, 999))
                assert np.abs(dW - rW).max() < 2e-3 * np.abs(rW).max(), np.abs(dW - rW).max()
                assert np.abs(db - rb).max() < 2e-3 * max(1.0, np.abs(rb).max()), np.abs(db - rb).max()
            print(f"ok   {tag} ({time.time() - t0:.1f} s)", flush=True)
        except Exception as e:
            bad += 1
            print(f"FAIL {tag}: {e!r}"[:400], flush=True)
    for i in range(args.cases if args.family == "as" else 0):
        fl = rng.choice(["f0", "f1", "f1d", "f3", "k384_f0", "k384_f1", "k384_f2"])
        K = 384 if fl.startswith("k384") else 512
        panels = rng.choice([1, 1, 2, 3])
        N = rng.choice([256, 384, 512, 640, 768, 1024] if not fl.startswith("k384") else [256, 384, 512, 768, 1152])
        if fl in ("f1", "f1d", "f3", "k384_f1") and N % 256:
            N = N // 256 * 256 or 256                   # the sign-bit flavours are dispatched at N % 256 == 0 (csrc/gemm.hip nt_as_try)
        mid = rng.random() < 0.45
        nsplit = rng.choice([d for d in (1, 2, 3, 4) if (N // 128) % d == 0 and (N // d) % 128 == 0]) if mid else 1
        grid = rng.randint(1, panels)
        order = rng.choice([None, [3, 2, 1, 0], [2, 0, 3, 1], [1, 3, 0, 2]])
        kw = dict(order=order, seed=rng.randint(0, 999), K=K)
        if mid:
            kw.update(nsplit=nsplit, flags=1)
        if fl == "f1d":
            kw.update(row_mult=rng.choice([1, 3, 181]), p_drop=rng.choice([0.1, 0.25]))
        if fl == "f3":
            kw.update(alpha=rng.choice([1.0, 1.0 / 0.9]))
        tag = f"{fl} M={256 * panels} N={N} K={K} grid={grid} nsplit={nsplit} mid={int(mid)} order={order}"
        t0 = time.time()
        try:
            out, ref, bits, g = run_kernel(fl, 256 * panels, N, grid, **kw)
            check(out, ref)
            print(f"ok   {tag} ({time.time() - t0:.1f} s)", flush=True)
        except Exception as e:
            bad += 1
            print(f"FAIL {tag}: {e!r}"[:400], flush=True)
    print(f"{bad} failing launch geometr(y/ies) of {args.cases} (seed {args.seed})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
