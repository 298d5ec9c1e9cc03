#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs for asdr_update_kernel: mean per dispatch of every counter."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
acc = defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        per_dispatch = defaultdict(float)
        for row in csv.DictReader(fh):
            if "asdr_update_kernel" not in row.get("Kernel_Name", ""):
                continue
            per_dispatch[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
        for (d, name), v in per_dispatch.items():
            acc[name].append(v)
for name in sorted(acc):
    v = sorted(acc[name])
    print("%-24s n=%3d  median %.6g" % (name, len(v), v[len(v) // 2]))
