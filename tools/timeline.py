#!/usr/bin/env python3
"""Timeline of the lt:: kernels of the last run in a rocprofv3 --kernel-trace CSV (start offset, duration, name)."""
import csv
import glob
import sys

import os
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)  # (the newest, if old runs were merged into the directory)
rows = [r for r in csv.DictReader(open(f)) if "lt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "lt_plan_kernel" in r["Kernel_Name"]]
last = rows[idx[-1]:] if idx else rows
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    name = r["Kernel_Name"].split("(")[0].replace("lt::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {name}")
print(f"total {(int(last[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
