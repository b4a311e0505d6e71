#!/usr/bin/env python3
"""Run ON the GPU box after rocprofv3: reduce result DBs (written under /tmp) to small text/JSON summaries in gpurun_out/.
  prof_summarize.py stats <db> <out.txt> <title...>     kernel-trace --stats table
  prof_summarize.py pmc <fetch_db> <write_db> <out.json> HBM traffic per launch from FETCH_SIZE / WRITE_SIZE passes"""
import json, sqlite3, sys

def stats(db, out, title):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 45").fetchall()
    tot = sum(r[2] for r in cur.execute("select name,total_calls,total_duration from top_kernels"))
    o = [f"# {title}", "# columns: kernel | calls | total_ms | avg_ms | pct", f"# total kernel time {tot/1e3:.1f} ms"]
    for n, c, t, a, p in rows:
        o.append(f"{n[:90]:90s} | {c:6d} | {t/1e3:10.2f} | {a/1e3:9.4f} | {p:6.2f}")
    open(out, "w").write("\n".join(o) + "\n")
    print("\n".join(o[:14]))

def _sources_sha():
    """Same hash as bench.py: kernel_sources_sha() -- bench.py refuses a traffic file measured on other kernel sources."""
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.kernel_sources_sha()

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
        if any(k in n for k in ("gemm", "attn", "norm", "colsum", "adam")):
            fv, fc, fd = f.get(n, (0, 0, 0)); wv, wc, wd = w.get(n, (0, 0, 0))
            res["kernels"][n.split("(")[0]] = {"launches": fc, "fetch_bytes_per_launch": 2 * fv * 1024, "write_bytes_per_launch": wv * 1024,
                                               "hbm_bytes_per_launch": 2 * fv * 1024 + wv * 1024, "avg_duration_us_profiled": fd / 1e3}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print(f'{k[:42]:42s} n={v["launches"]:5d} hbm={v["hbm_bytes_per_launch"]/1e9:8.3f} GB/launch  {v["avg_duration_us_profiled"]:9.1f} us')

if sys.argv[1] == "stats":
    stats(sys.argv[2], sys.argv[3], " ".join(sys.argv[4:]))
else:
    pmc(sys.argv[2], sys.argv[3], sys.argv[4])
