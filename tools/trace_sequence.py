#!/usr/bin/env python3
"""The device's timeline between two launches of a marker kernel (default: init_residual_kernel = one solve of the
fused CG loop) from a rocprofv3 --kernel-trace CSV: every launch with its start offset, duration and the idle gap in
front of it; runs of the same kernel are folded."""
import csv
import glob
import sys

marker = sys.argv[2] if len(sys.argv) > 2 else "init_residual_kernel"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
i0, i1 = marks[which], marks[which + 1]
# back up to the first launch of this solve (whatever precedes the marker after the previous solve's last kernel)
t0 = rows[i0][0]
prev_end = rows[i0 - 1][1]
print(f"{i1 - i0} launches, {(rows[i1][0] - t0) / 1e3:.1f} us from marker to marker; idle before the marker {(t0 - prev_end) / 1e3:.1f} us")
busy = gap_total = 0.0
i = i0
while i < i1:
    j = i
    dur = gaps = 0.0
    while j < i1 and rows[j][2] == rows[i][2]:
        dur += (rows[j][1] - rows[j][0]) / 1e3
        gaps += (rows[j][0] - rows[j - 1][1]) / 1e3
        j += 1
    print(f"  +{(rows[i][0] - t0) / 1e3:8.1f} us  {rows[i][2]:42s} x{j - i:3d}  busy {dur:8.1f}  idle-in-front {gaps:7.1f}")
    busy += dur
    gap_total += gaps
    i = j
print(f"busy {busy:.1f} us, idle {gap_total:.1f} us")
