#!/usr/bin/env python3
"""One steady-state iteration of a solver loop from a rocprofv3 --kernel-trace CSV, BOTH streams: every launch between
two launches of a marker kernel (default bicg_update_kernel<true> = the end of a BiCGStab iteration) with its queue /
stream, start offset, duration and the idle time of ITS queue in front of it; below it the iteration's critical path
by queue (busy, idle).  `python tools/trace_iteration_streams.py <dir> [marker] [which]`."""
import csv
import glob
import sys

marker = sys.argv[2] if len(sys.argv) > 2 else "bicg_update_kernel<true>"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -3
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        q = r.get("Stream_Id") or r.get("Queue_Id") or "?"
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("storm::", "").split("(")[0][-52:], q))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
i0, i1 = marks[which], marks[which + 1]
t0 = rows[i0][1]  # the marker's END: the iteration starts behind it
print(f"{i1 - i0} launches, {(rows[i1][1] - t0) / 1e3:.1f} us from the end of one {marker} to the end of the next")
last_end = {}
busy = {}
for s, e, k, q in rows[i0 + 1 : i1 + 1]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    print(f"  q{q:>3s} +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  idle-before {gap:6.1f}  {k}")
    last_end[q] = e
    busy[q] = busy.get(q, 0.0) + (e - s) / 1e3
for q, b in busy.items():
    print(f"queue {q}: busy {b:.1f} us")
