#!/usr/bin/env python3
"""Run ON the GPU box after rocprofv3: reduce result DBs (written under /tmp) to small text/JSON summaries in gpurun_out/.
  prof_summarize.py stats <db> <out.txt> <title...>     kernel-trace --stats table
  prof_summarize.py pmc <fetch_db> <write_db> <out.json> HBM traffic per launch from FETCH_SIZE / WRITE_SIZE passes
  prof_summarize.py pmc_shapes <fetch_db> <write_db> <launch_log> <out.json>   the same keyed by (kernel, M, N, K)"""
import json, sqlite3, sys

def stats(db, out, title):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 45").fetchall()
    tot = sum(r[2] for r in cur.execute("select name,total_calls,total_duration from top_kernels"))
    o = [f"# {title}", "# columns: kernel | calls | total_ms | avg_ms | pct", f"# total kernel time {tot/1e3:.1f} ms"]
    import re
    for n, c, t, a, p in rows:
        # kernels that exist only in the tower-grouped form (csrc/launch.h): _Z12svla_groupedITnDaXadL_Z<len><body name>I<template args>... -> "svla_grouped<body name ...>"
        n = re.sub(r"^_Z12svla_groupedITnDaXadL_Z\d+", "svla_grouped<", n)
        o.append(f"{n[:90]:90s} | {c:6d} | {t/1e3:10.2f} | {a/1e3:9.4f} | {p:6.2f}")
    open(out, "w").write("\n".join(o) + "\n")
    print("\n".join(o[:14]))

def timeline(db, out, n_last=140):
    """the last ``n_last`` kernel dispatches in start order: short name, workgroups, duration, idle gap before it (one acting step is ~100 dispatches)"""
    con = sqlite3.connect(db)
    cur = con.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    cand = [n for n in names if n == "kernels"] or [n for n in names if "kernel_dispatch" in n]
    if not cand:
        open(out, "w").write("no kernel table: " + ", ".join(names) + "\n"); return
    t = cand[0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({t})")]
    pick = lambda *c: next((x for x in c if x in cols), None)
    nm, st, en = pick("name", "kernel_name"), pick("start", "start_timestamp"), pick("end", "end_timestamp")
    gx, wx = pick("grid_x", "grid_size_x", "grid_size"), pick("workgroup_x", "workgroup_size_x", "workgroup_size")
    gz = pick("grid_z", "grid_size_z")
    if not (nm and st and en):
        open(out, "w").write(f"table {t}: columns {cols}\n"); return
    sel = ", ".join(x for x in (nm, st, en, gx, wx, gz) if x)
    rows = cur.execute(f"select {sel} from {t} order by {st} desc limit {int(n_last)}").fetchall()[::-1]
    o = [f"# last {len(rows)} dispatches of {db} ({t}); columns: kernel | workgroups (x) | z | us | gap_us"]
    prev = None
    for r in rows:
        name, s0, e0 = r[0], r[1], r[2]
        wg = (r[3] // r[4]) if (gx and wx and r[4]) else -1
        z = r[5] if gz else 1
        short = name.replace("_Z12svla_groupedITnDaXadL_Z", "G:").split("(")[0][:70]
        o.append(f"{short:70s} | {wg:6d} | {z} | {(e0 - s0) / 1e3:8.1f} | {((s0 - prev) / 1e3 if prev else 0):7.1f}")
        prev = e0
    open(out, "w").write("\n".join(o) + "\n")

def _sources_sha():
    """Same hash as bench.py: kernel_sources_sha() -- bench.py refuses a traffic file measured on other kernel sources."""
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.kernel_sources_sha()

def pmc_shapes(fdb, wdb, log, out):
    """per (kernel, M, N, K): mean FETCH / WRITE bytes per launch, joined with the launch log of SVLA_GEMM_LOG (i-th dispatch of a kernel name
    <-> i-th log line of that name), next to the algorithmic bytes of the shape"""
    import collections
    def rows(path, counter):
        cur = sqlite3.connect(path).cursor()
        d = collections.defaultdict(list)
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        order = "dispatch_id" if "dispatch_id" in cols else ("start" if "start" in cols else "rowid")
        for n, v, dur in cur.execute(f"select kernel_name, value, duration from counters_collection where counter_name=? order by {order}", (counter,)):
            d[n.split("(")[0].replace("void ", "").split("<")[0]].append((v, dur))
        return d
    f, w = rows(fdb, "FETCH_SIZE"), rows(wdb, "WRITE_SIZE")
    logs = collections.defaultdict(list)
    for line in open(log):
        k, M, N, K, *x = line.split()
        logs[k].append((int(M), int(N), int(K), int(x[0]) if x else 0))      # 5th column: bytes per row beyond A and C (residual, sign bits)
    res = {"_method": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2), --kernel-trace only, on `python bench.py --steps 1 --warmup 0 "
                      "--no-cpu-baseline --no-roofline --no-secondary` with SVLA_GEMM_LOG: the i-th dispatch of a kernel name is the i-th logged launch of that "
                      "name; counters are KiB; gfx950: FETCH_SIZE reports half of a wide (16 B/lane) coalesced stream (MI355X_MICROARCH.md, HBM) => fetch bytes = "
                      "2*FETCH_SIZE*1024, WRITE_SIZE as is.  algorithmic_bytes: NT 2*(M*K + M*N) + M*extra_bytes_per_row (residual / ReLU-mask rows 2 N, sign bits N/8: counted since round 5), TN 2*M*(N + K).",
           "kernel_sources_sha256": _sources_sha(), "shapes": []}
    for k, shp in logs.items():
        fk, wk = f.get(k, []), w.get(k, [])
        if len(fk) != len(shp) or len(wk) != len(shp):
            res["shapes"].append({"kernel": k, "error": f"{len(shp)} logged launches vs {len(fk)} / {len(wk)} counter rows"})
            continue
        agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
        for (M, N, K, X), (fv, fd), (wv, wd) in zip(shp, fk, wk):
            a = agg[(M, N, K, X)]
            a[0] += 1; a[1] += 2 * fv * 1024; a[2] += wv * 1024; a[3] += fd
        for (M, N, K, X), (n, fb, wb, dur) in sorted(agg.items(), key=lambda kv: -kv[1][3]):
            alg = 2 * M * (N + K) if "tn" in k else 2 * (M * K + M * N) + M * X
            res["shapes"].append({"kernel": k, "M": M, "N": N, "K": K, "extra_bytes_per_row": X, "launches": n, "fetch_bytes_per_launch": fb / n, "write_bytes_per_launch": wb / n,
                                  "hbm_bytes_per_launch": (fb + wb) / n, "algorithmic_bytes_per_launch": alg, "ratio_to_algorithmic": round((fb + wb) / n / alg, 3),
                                  "avg_duration_us_profiled": dur / n / 1e3, "total_ms_profiled": dur / 1e6})
    res["shapes"].sort(key=lambda r: -r.get("total_ms_profiled", 0))
    json.dump(res, open(out, "w"), indent=1)
    for r in res["shapes"][:14]:
        if "error" in r: print(r); continue
        print(f'{r["kernel"][:28]:28s} M={r["M"]:8d} N={r["N"]:5d} K={r["K"]:5d} n={r["launches"]:4d} hbm={r["hbm_bytes_per_launch"]/1e9:7.3f} GB alg={r["algorithmic_bytes_per_launch"]/1e9:7.3f} GB x{r["ratio_to_algorithmic"]:.2f} {r["avg_duration_us_profiled"]:9.1f} us')


def pmc(fdb, wdb, out):
    def q(path, counter):
        cur = sqlite3.connect(path).cursor()
        return {n: (v, c, d) for n, v, c, d in cur.execute(
            "select kernel_name, avg(value), count(*), avg(duration) from counters_collection where counter_name=? group by kernel_name", (counter,))}
    f, w = q(fdb, "FETCH_SIZE"), q(wdb, "WRITE_SIZE")
    res = {"_method": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2), each with --kernel-trace only, on "
                      "`python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline`; means per launch; counters are KiB; "
                      "gfx950: FETCH_SIZE reports half of a wide (16 B/lane) coalesced stream (MI355X_MICROARCH.md, HBM) => fetch bytes = "
                      "2*FETCH_SIZE*1024; WRITE_SIZE as is", "kernel_sources_sha256": _sources_sha(), "kernels": {}}
    for n in sorted(set(f) | set(w)):
        if any(k in n for k in ("gemm", "attn", "norm", "colsum", "adam", "svla_", "patchify", "vit_tokens", "adaptive_pool")):
            fv, fc, fd = f.get(n, (0, 0, 0)); wv, wc, wd = w.get(n, (0, 0, 0))
            res["kernels"][n.split("(")[0]] = {"launches": fc, "fetch_bytes_per_launch": 2 * fv * 1024, "write_bytes_per_launch": wv * 1024,
                                               "hbm_bytes_per_launch": 2 * fv * 1024 + wv * 1024, "avg_duration_us_profiled": fd / 1e3,
                                               "hbm_TBps_profiled": round((2 * fv * 1024 + wv * 1024) / max(fd, 1) / 1e3, 4)}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print(f'{k[:42]:42s} n={v["launches"]:5d} hbm={v["hbm_bytes_per_launch"]/1e9:8.3f} GB/launch  {v["avg_duration_us_profiled"]:9.1f} us')

if sys.argv[1] == "stats":
    stats(sys.argv[2], sys.argv[3], " ".join(sys.argv[4:]))
elif sys.argv[1] == "timeline":
    timeline(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 140)
elif sys.argv[1] == "pmc_shapes":
    pmc_shapes(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
else:
    pmc(sys.argv[2], sys.argv[3], sys.argv[4])
