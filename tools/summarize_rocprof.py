#!/usr/bin/env python3
"""Trim a rocprofv3 *_kernel_stats.csv to short kernel names so it can be committed under profiles/."""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
with open(src) as f, open(dst, "w", newline="") as g:
    r = csv.reader(f)
    w = csv.writer(g)
    for row in r:
        if row:
            name = row[0]
            row[0] = name if len(name) <= 96 else name[:93] + "..."
        w.writerow(row)
