#!/usr/bin/env python3
"""Print the kernel timeline of ONE solver iteration from a rocprofv3 --kernel-trace CSV (start offset, duration,
stream/queue, name), to see where the time between kernels goes:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 tools/comm_path_overhead.py 256 ipc_only
    python tools/trace_iteration.py /tmp/tr cg_xp_kernel"""
import csv
import glob
import os
import re
import sys

d, anchor = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"),
                         re.sub(r"\(.*", "", r["Kernel_Name"])[:70]))
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor in r[3]]
if len(idx) < 12:
    raise SystemExit(f"{len(idx)} launches of {anchor}")
a, b = idx[-6], idx[-5]  # one steady-state iteration: from one anchor launch to the next
t0 = rows[a][1]
print(f"iteration length {(rows[b][1] - rows[a][1]) / 1e3:.1f} us")
for r in rows[a + 1:b + 1]:
    print(f"  +{(r[0] - t0) / 1e3:7.1f} us  {(r[1] - r[0]) / 1e3:7.1f} us  q{r[2]:>3}  {r[3]}")
