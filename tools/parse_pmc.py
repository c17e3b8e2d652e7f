#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_traffic.sh: per-launch FETCH_SIZE / WRITE_SIZE
of lt_chain_kernel and of the decimator (calibration), and profiles/traffic.json for bench.py."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]


def counter(run, kernel_substr, name):
    vals = []
    for f in glob.glob(os.path.join(root, run, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel_substr in row.get("Kernel_Name", "") and row.get("Counter_Name") == name:
                    vals.append(float(row["Counter_Value"]))
    return vals


out = {}
for run, kern in (("mix", "lt_chain_kernel"), ("decim", "lt_decim_kernel")):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = counter(f"{run}_{c}", kern, c)
        out[f"{run}_{c}"] = v
        print(run, c, "per launch (counter units):", v)
# units: FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B? print raw and derive with
# the decimator, whose true read volume is known (64 ch * 1.92e6 * 8 B)
true_read = 64 * 1_920_000 * 8
true_write = true_read // 8
if out["decim_FETCH_SIZE"] and out["mix_FETCH_SIZE"]:
    fd = sorted(out["decim_FETCH_SIZE"])[len(out["decim_FETCH_SIZE"]) // 2]
    wd = sorted(out["decim_WRITE_SIZE"])[len(out["decim_WRITE_SIZE"]) // 2] if out["decim_WRITE_SIZE"] else None
    fm = sorted(out["mix_FETCH_SIZE"])[len(out["mix_FETCH_SIZE"]) // 2]
    wm = sorted(out["mix_WRITE_SIZE"])[len(out["mix_WRITE_SIZE"]) // 2] if out["mix_WRITE_SIZE"] else None
    bytes_per_fetch_unit = true_read / fd
    bytes_per_write_unit = (true_write / wd) if wd else None
    print(f"calibration: decimator reads {true_read} B = {fd} FETCH units -> {bytes_per_fetch_unit:.1f} B/unit")
    if wd:
        print(f"calibration: decimator writes {true_write} B = {wd} WRITE units -> {bytes_per_write_unit:.1f} B/unit")
    hbm_read = fm * bytes_per_fetch_unit
    hbm_write = wm * bytes_per_write_unit if (wm is not None and bytes_per_write_unit) else 0.0
    print(f"lt_chain_kernel per launch: read {hbm_read / 1e6:.1f} MB, write {hbm_write / 1e6:.1f} MB "
          f"(algorithmic 983.04 MB)")
    res = {"hbm_bytes_per_launch": hbm_read + hbm_write, "read_bytes": hbm_read, "write_bytes": hbm_write,
           "fetch_units": fm, "write_units": wm, "bytes_per_fetch_unit": bytes_per_fetch_unit,
           "bytes_per_write_unit": bytes_per_write_unit,
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on tools/probe.py --mode mix; "
                     "units calibrated on lt_decim_kernel (known 983.04 MB read with the same 8 B/lane loads)"}
    with open(os.path.join(root, "traffic.json"), "w") as f:
        json.dump(res, f, indent=1)
