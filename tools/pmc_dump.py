#!/usr/bin/env python3
"""Print per-kernel means of every counter in rocprofv3 --pmc result DBs:  pmc_dump.py <name filter> db1 [db2 ...]"""
import sqlite3, sys
flt, dbs = sys.argv[1], sys.argv[2:]
acc = {}
for path in dbs:
    cur = sqlite3.connect(path).cursor()
    for name, ctr, val, cnt, dur in cur.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection "
                                                 "group by kernel_name, counter_name"):
        if flt not in name: continue
        d = acc.setdefault(name.split("(")[0], {})
        d[ctr] = val; d["_n"] = cnt; d["_us"] = dur / 1e3
for k, d in sorted(acc.items()):
    print(f"{k}  launches {d['_n']}  avg {d['_us']:.1f} us")
    for c, v in sorted(d.items()):
        if not c.startswith("_"): print(f"    {c:34s} {v:18.0f}")
