#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc results DB: per (kernel, grid) mean of each counter and mean duration."""
import sqlite3, sys
for path in sys.argv[1:]:
    db = sqlite3.connect(path); cur = db.cursor()
    rows = cur.execute("""select kernel_name, grid_size, counter_name, avg(value), count(*), avg(duration)
                          from counters_collection group by kernel_name, grid_size, counter_name order by kernel_name, grid_size""").fetchall()
    for n, g, c, v, cnt, d in rows:
        if "gemm" in n or "attn" in n:
            print(f"{n[:40]:40s} grid={g:>10} {c:28s} mean={v:16.1f} n={cnt} dur_us={d/1e3:9.1f}")
