#!/usr/bin/env python3
"""ON the GPU box, after `rocprofv3 --kernel-trace -d DIR -o kt -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary`:
how much of the timed update is kernel time and how much is idle between kernels?   python tools/update_gaps.py <results.db>
(the timed update = the dispatches between the last two adam_kernel bursts)"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
pick = lambda *c: next(x for x in c if x in cols)
nm, st, en = pick("name", "kernel_name"), pick("start", "start_timestamp"), pick("end", "end_timestamp")
rows = cur.execute(f"select {nm}, {st}, {en} from kernels order by {st}").fetchall()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
# Adam launches come in bursts of 3 (one per tower) per epoch: 4 epochs per update -> the last 12 belong to the timed update, the 12 before to the warm-up update
assert len(adam) >= 24, len(adam)
lo, hi = adam[-13] + 1, adam[-1]
seg = rows[lo:hi + 1]
busy = sum(e - s for _, s, e in seg)
span = seg[-1][2] - seg[0][1]
gaps = sorted(((seg[i + 1][1] - seg[i][2], seg[i][0][:60], seg[i + 1][0][:60]) for i in range(len(seg) - 1)), reverse=True)
idle = sum(g for g, _, _ in gaps if g > 0)
print(f"timed update: {len(seg)} dispatches, span {span / 1e6:.1f} ms, kernel time {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %), idle between kernels {idle / 1e6:.1f} ms ({100 * idle / span:.1f} %)")
print("largest gaps (us | after kernel | before kernel):")
for g, a, b in gaps[:15]:
    print(f"  {g / 1e3:9.1f} | {a} | {b}")
big = sum(g for g, _, _ in gaps if g > 20e3)
print(f"gaps > 20 us: {sum(1 for g, _, _ in gaps if g > 20e3)} totalling {big / 1e6:.1f} ms")
