#!/usr/bin/env python3
"""Mean duration of every kernel of a rocprofv3 --kernel-trace CSV, split by grid size (the same kernel launched on
different grids -- e.g. the marching CG step with and without its sending blocks -- is listed per grid)."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-48:], r.get("Grid_Size", r.get("Grid_Size_X", "?")))
        acc[k][0] += 1
        acc[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (k, g), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if t / n > 3.0 and n >= 20:
        print(f"{k:50s} grid {g:>9s}  n={n:5d}  avg {t / n:8.1f} us")
