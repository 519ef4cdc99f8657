#!/usr/bin/env python3
"""Steady-state timeline of a rocprofv3 --kernel-trace CSV: for each kernel (name, grid) the mean duration and the mean
gap to the end of the kernel before it on the device, over the middle half of the trace."""
import csv
import glob
import sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:],
                     r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
rows.sort()
lo, hi = len(rows) // 4, 3 * len(rows) // 4
acc = defaultdict(lambda: [0, 0.0, 0.0])
for i in range(max(lo, 1), hi):
    s, e, k, g = rows[i]
    a = acc[(k, g)]
    a[0] += 1
    a[1] += (e - s) / 1e3
    a[2] += (s - rows[i - 1][1]) / 1e3
tot = (rows[hi - 1][1] - rows[lo][0]) / 1e3
print(f"{hi - lo} launches in {tot:.0f} us")
for (k, g), (n, d, gap) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:46s} grid {g:>9s} n={n:5d} dur {d / n:7.1f} us  gap-before {gap / n:6.1f} us  share {(d + gap) / tot:5.1%}")
