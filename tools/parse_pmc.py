#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_traffic.sh: FETCH_SIZE / WRITE_SIZE per run of the batched
path, kernel by kernel (a run launches the scan and resolve kernels once per round: their launches are summed),
the decimator as calibration, and profiles/traffic.json for bench.py."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha16  # noqa: E402  (the hash bench.py checks before it quotes these numbers)


RUNS = int(sys.argv[2]) if len(sys.argv) > 2 else 3  # lt_batch_run calls per profiled process (probe.py --reps)


def counter(run, kernel_substr, name, per_run=False):
    """Per-launch values; per_run: the sum over all launches divided by the number of runs."""
    vals = []
    for f in glob.glob(os.path.join(root, run, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel_substr in row.get("Kernel_Name", "") and row.get("Counter_Name") == name:
                    vals.append(float(row["Counter_Value"]))
    if per_run:
        return [sum(vals) / RUNS] if vals else []
    return vals


KERNELS = ("lt_plan_kernel", "lt_scan_kernel", "lt_resolve_kernel", "lt_chain_kernel", "lt_cpsss_kernel", "lt_cp_scan_kernel")
POST = ()


def median(v):
    return sorted(v)[len(v) // 2] if v else None


out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    out[f"decim_{c}"] = counter(f"decim_{c}", "lt_decim_kernel", c)
    print("decim", c, "per launch (counter units):", out[f"decim_{c}"])
    for k in KERNELS:
        out[f"{k}_{c}"] = counter(f"mix_{c}", k, c, per_run=True)
        print(k, c, "per run (counter units):", out[f"{k}_{c}"])
# The counters come in units that depend on the access width on gfx950 (MI355X_MICROARCH.md: FETCH_SIZE
# reads 1/2 of a 16 B/lane stream): calibrate them on the decimator, whose true volume is known
# (64 ch * 1.92e6 * 8 B read with the same 8 B/lane coalesced loads, 1/8 of it written).
true_read = 64 * 1_920_000 * 8
true_write = true_read // 8
fd, wd = median(out["decim_FETCH_SIZE"]), median(out["decim_WRITE_SIZE"])
if fd:
    bytes_per_fetch_unit = true_read / fd
    bytes_per_write_unit = (true_write / wd) if wd else None
    print(f"calibration: decimator reads {true_read} B = {fd} FETCH units -> {bytes_per_fetch_unit:.1f} B/unit")
    if wd:
        print(f"calibration: decimator writes {true_write} B = {wd} WRITE units -> {bytes_per_write_unit:.1f} B/unit")
    res = {"kernels": {}, "bytes_per_fetch_unit": bytes_per_fetch_unit, "bytes_per_write_unit": bytes_per_write_unit,
           "taken": __import__("datetime").date.today().isoformat(),
           "kernel_source_sha16": kernel_source_sha16(),
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on tools/probe.py --mode mix (the bench "
                     "workload), summed over the launches of a run; units calibrated on lt_decim_kernel (known 983.04 MB "
                     "read with 8 B/lane loads, 122.88 MB written)"}
    total = 0.0
    for k in KERNELS:
        fm, wm = median(out[f"{k}_FETCH_SIZE"]), median(out[f"{k}_WRITE_SIZE"])
        if fm is None:
            continue
        rd = fm * bytes_per_fetch_unit
        wr = wm * bytes_per_write_unit if (wm is not None and bytes_per_write_unit) else 0.0
        print(f"{k} per run: read {rd / 1e6:.1f} MB, write {wr / 1e6:.1f} MB")
        res["kernels"][k] = {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
                             "fetch_units": fm, "write_units": wm}
        total += rd + wr
    post = [res["kernels"][k] for k in POST if k in res["kernels"]]
    if post:  # bench.py reports the three per-record kernels as one phase
        res["kernels"]["post_pass"] = {f: sum(p[f] for p in post) for f in ("hbm_bytes_per_launch", "read_bytes", "write_bytes")}
    res["hbm_bytes_per_launch"] = total
    with open(os.path.join(root, "traffic.json"), "w") as f:
        json.dump(res, f, indent=1)
