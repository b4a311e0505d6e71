#!/usr/bin/env python3
"""CU-partition probe: do an MFMA-bound GEMM and an HBM / latency-bound kernel of the update finish sooner SIDE BY SIDE on two halves of the chip
(two streams created with hipExtStreamCreateWithCUMask) than one after the other on all 256 CUs?

Motivation: the assembly GEMMs hold the MFMA pipe 57-75 % busy and the chip answers with 1.55-1.65 GHz (power), while the attention / LayerNorm kernels
run at 2.0-2.5 GHz with the MFMA pipe 12-20 % busy and HBM at 40-65 %.  The weight-gradient GEMMs of the backward (svla_tn_os, 18.5 % of the update) do not depend on the
dX chain that runs next to them (attention backward, LayerNorm backward), so they could run on a second stream -- but a persistent assembly GEMM takes every register of a CU
(512 per lane x 4 waves), so the two kinds of kernels cannot share a CU: the overlap needs a CU partition.  This tool measures whether the partition pays before anything is
restructured around it.

Part 1 finds out how the mask bits map to XCDs (a probe kernel records XCC_ID / HW_ID per workgroup); part 2 times the pairs."""
import collections
import ctypes
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = "/tmp/whereami.so"
if not os.path.exists(SO):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "probes", "whereami_probe.hip"), "-o", SO], check=True)
lib = ctypes.CDLL(SO)
NCU = torch.cuda.get_device_properties(0).multi_processor_count
NW = (NCU + 31) // 32


def masked_stream(bits):
    words = [0] * NW
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (ctypes.c_uint32 * NW)(*words)
    s = ctypes.c_void_p()
    rc = lib.make_masked_stream(ctypes.byref(s), NW, arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(s.value)


def where(stream, grid=8192, spin=60000):
    out = torch.zeros(2 * grid, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rc = lib.whereami_launch(ctypes.c_void_p(out.data_ptr()), grid, ctypes.c_longlong(spin), ctypes.c_void_p(stream.cuda_stream))
    assert rc == 0, rc
    torch.cuda.synchronize()
    o = out.cpu().view(grid, 2)
    xcc = (o[:, 0] & 0xF).tolist()
    hw = o[:, 1].tolist()
    ids = {(x, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 0xF) for x, h in zip(xcc, hw)}
    return collections.Counter(xcc), len(ids)


def wall_ms(jobs, reps=3):
    """jobs: [(stream, fn, n)] issued back to back from this thread; wall time of the whole set, best of ``reps``"""
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for st, fn, n in jobs:
            with torch.cuda.stream(st):
                for _ in range(n):
                    fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


def main():
    print(f"{NCU} CUs, {NW} mask words")
    main_s = torch.cuda.current_stream()
    masks = {"low half (bits 0..N/2-1)": [b for b in range(NCU) if b < NCU // 2], "even bits": [b for b in range(NCU) if b % 2 == 0],
             "bit % 8 < 4": [b for b in range(NCU) if b % 8 < 4], "(bit // 8) % 2 == 0": [b for b in range(NCU) if (b // 8) % 2 == 0],
             "(bit // 32) % 2 == 0": [b for b in range(NCU) if (b // 32) % 2 == 0]}
    print("== part 1: where workgroups of a masked stream run (XCC histogram of 8192 workgroups, distinct (xcc, se, sh, cu))")
    c, n = where(main_s)
    print(f"  unmasked: {dict(sorted(c.items()))}, {n} distinct CUs")
    best = None
    for name, bits in masks.items():
        try:
            st = masked_stream(bits)
        except RuntimeError as e:
            print(f"  {name}: {e}")
            continue
        c, n = where(st)
        print(f"  {name} ({len(bits)} bits): XCCs {dict(sorted(c.items()))}, {n} distinct CUs")
        if best is None or (len(c), abs(n - NCU // 2)) < best[0]:
            best = ((len(c), abs(n - NCU // 2)), name, bits)
    name, bits = best[1], best[2]
    comp = [b for b in range(NCU) if b not in set(bits)]
    sa, sb = masked_stream(bits), masked_stream(comp)
    ca, na = where(sa)
    cb, nb = where(sb)
    print(f"  partition used: A = '{name}' ({na} CUs on XCCs {sorted(ca)}), B = complement ({nb} CUs on XCCs {sorted(cb)})")

    print("== part 2: update-sized kernels, one after the other on all CUs vs side by side on the two halves")
    R, S = int(os.environ.get("CS_ROWS", 16384)), 181
    M = R * S
    dev = "cuda"
    bf = torch.bfloat16
    X = (torch.randn(M, 512, device=dev) * 0.5).to(bf)
    dY = (torch.randn(M, 2048, device=dev) * 0.5).to(bf)
    dW = torch.zeros(2048, 512, device=dev)
    W1 = (torch.randn(2048, 512, device=dev) * 0.05).to(bf)
    Y1 = torch.empty(M, 2048, device=dev, dtype=bf)
    bias = torch.zeros(2048, device=dev)
    g_tn = lambda: ops.gemm_tn_acc(dY, X, dW, M, 2048, 512)
    g_nt = lambda: ops.gemm_nt(X, W1, M, 2048, 512, bias=bias, out=Y1)
    qkv = (torch.randn(M, 1536, device=dev) * 0.5).to(bf)
    drop = ops.Dropout(77, 3, 0.1)
    out, lse = ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125, drop=drop)
    do = torch.randn_like(out)
    dqkv = torch.zeros_like(qkv)
    a_bwd = lambda: ops.attn_bwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, out, 512, lse, do, 512, dqkv, dqkv[:, 512:], dqkv[:, 1024:], 1536, R, S, 8, 0.125, drop=drop)
    a_fwd = lambda: ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125, drop=drop, out=out)
    gam, bet = torch.ones(512, device=dev), torch.zeros(512, device=dev)
    y, mean, rstd = ops.norm_fwd(X, gam, bet, 1e-5, M)
    dgam, dbet = torch.zeros(512, device=dev), torch.zeros(512, device=dev)
    dxn = torch.empty_like(X)
    n_bwd = lambda: ops.norm_bwd(do.view(M, 512) if do.shape[0] == M else y, X, gam, bet, mean, rstd, M, dgam, dbet, dx=dxn)
    n_fwd = lambda: ops.norm_fwd(X, gam, bet, 1e-5, M, y=y)
    kernels = {"tn_os 2048x512 (dW of linear1)": g_tn, "nt_as N=2048 K=512 (linear1 fwd)": g_nt, "attn_bwd S=181 dropout": a_bwd, "attn_fwd S=181 dropout": a_fwd,
               "norm_bwd": n_bwd, "norm_fwd": n_fwd}
    for fn in kernels.values():
        fn()
    torch.cuda.synchronize()
    full, half = {}, {}
    for k, fn in kernels.items():
        full[k] = wall_ms([(main_s, fn, 4)]) / 4
        half[k] = wall_ms([(sa, fn, 4)]) / 4
        print(f"  {k}: all CUs {full[k]:.3f} ms, half A alone {half[k]:.3f} ms ({half[k] / full[k]:.2f}x)")
    pairs = [("tn_os 2048x512 (dW of linear1)", "attn_bwd S=181 dropout"), ("tn_os 2048x512 (dW of linear1)", "norm_bwd"), ("tn_os 2048x512 (dW of linear1)", "attn_fwd S=181 dropout"),
             ("nt_as N=2048 K=512 (linear1 fwd)", "attn_bwd S=181 dropout"), ("nt_as N=2048 K=512 (linear1 fwd)", "norm_fwd"), ("attn_bwd S=181 dropout", "norm_bwd")]
    for ka, kb in pairs:
        # balance the two sides on their half-chip times
        na_, nb_ = 1, 1
        if half[ka] > half[kb]:
            nb_ = max(1, round(half[ka] / half[kb]))
        else:
            na_ = max(1, round(half[kb] / half[ka]))
        na_, nb_ = 2 * na_, 2 * nb_
        seq = na_ * full[ka] + nb_ * full[kb]
        conc = wall_ms([(sa, kernels[ka], na_), (sb, kernels[kb], nb_)])
        unm = wall_ms([(main_s, kernels[ka], na_), (torch.cuda.Stream(), kernels[kb], nb_)]) if os.environ.get("CS_UNMASKED", "1") == "1" else float("nan")
        print(f"  {na_} x [{ka}] || {nb_} x [{kb}]: one after the other on all CUs {seq:.2f} ms, side by side on the halves {conc:.2f} ms ({seq / conc:.3f}x), "
              f"two unmasked streams {unm:.2f} ms ({seq / unm:.3f}x)")


if __name__ == "__main__":
    main()
