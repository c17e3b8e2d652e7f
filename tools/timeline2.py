#!/usr/bin/env python3
"""Timeline of the lt:: kernels of one step in a rocprofv3 --kernel-trace CSV, with the queue each ran on
(overlap mode: two queues).  usage: timeline2.py <dir> [step index from the end, default 3]"""
import csv
import glob
import sys

import os
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)  # (the newest, if old runs were merged into the directory)
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = [r for r in csv.DictReader(open(f)) if "lt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
begins = [i for i, r in enumerate(rows) if "lt_plan_kernel" in r["Kernel_Name"]]
i0 = begins[-back - 1]
i1 = begins[-back + 1] if back > 1 else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    name = r["Kernel_Name"].split("(")[0].replace("lt::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>3}  {name}")
