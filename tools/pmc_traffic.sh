#!/bin/bash
# HBM-traffic counters of the batched path's kernels (run on the GPU box through gpurun).  Separate passes
# for FETCH_SIZE and WRITE_SIZE (they do not fit one pass), plus the decimator as the calibration
# of FETCH_SIZE for 8-byte-per-lane coalesced loads (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of a
# 16 B/lane stream on gfx950; other widths must be calibrated on a known byte count).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc
rm -rf $OUT && mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/mix_$c -- python3 tools/probe.py --mode mix --reps 3 > $OUT/mix_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/decim_$c -- python3 tools/probe.py --mode decim --reps 3 > $OUT/decim_$c.log 2>&1
done
python3 tools/parse_pmc.py $OUT 3 > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
