#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc result DBs (one counter set per run) to per-kernel means and derived utilisations.
   pmc_util_summary.py out.json db1 [db2 ...]
Derived per kernel (means per launch): clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / duration; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES /
(1024 SIMDs x cycles); lds_active = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles); lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import json, sqlite3, sys
out, dbs = sys.argv[1], sys.argv[2:]
acc = {}
for path in dbs:
    cur = sqlite3.connect(path).cursor()
    for name, ctr, val, cnt, dur in cur.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection "
                                                 "group by kernel_name, counter_name"):
        if not any(k in name for k in ("gemm", "attn", "norm", "svla_")):
            continue
        k = name.split("(")[0]
        d = acc.setdefault(k, {"launches": cnt, "avg_duration_us_profiled": dur / 1e3})
        d[ctr] = val
res = {"_method": "rocprofv3 --pmc <set> --kernel-trace, one counter set per run, on `python bench.py --steps 1 --warmup 0 --no-cpu-baseline "
                  "--no-roofline` (train mode); means per launch", "kernels": {}}
for k, d in sorted(acc.items()):
    if d["avg_duration_us_profiled"] < 50:
        continue
    cyc = d.get("GRBM_GUI_ACTIVE", 0) / 8.0
    e = dict(d)
    if cyc:
        e["clock_ghz"] = round(cyc / (d["avg_duration_us_profiled"] * 1e3), 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d: e["mfma_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4)
        if "SQ_LDS_IDX_ACTIVE" in d: e["lds_active"] = round(d["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), 4)
    if d.get("SQ_LDS_IDX_ACTIVE"): e["lds_conflict_frac"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0) / d["SQ_LDS_IDX_ACTIVE"], 4)
    res["kernels"][k] = e
    print(f'{k[:44]:44s} n={d["launches"]:4d} {d["avg_duration_us_profiled"]:8.1f} us  clock {e.get("clock_ghz", 0):5.2f} GHz  mfma_busy {e.get("mfma_busy", 0):6.3f}  '
          f'lds_active {e.get("lds_active", 0):6.3f}  lds_conflict {e.get("lds_conflict_frac", 0):6.3f}')
json.dump(res, open(out, "w"), indent=1)
