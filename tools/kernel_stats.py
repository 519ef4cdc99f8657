#!/usr/bin/env python3
"""Print (kernel, calls, average us, total ms) from a rocprofv3 --stats output directory."""
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.2f} us {float(r['TotalDurationNs']) / 1e6:9.2f} ms")
