#!/bin/bash
# round 5, call 6: scan_bench compile-time variants (poll interval, load cache policy), then the whole GPU suite
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B=gr-ltetrigger_amd/build
{
for rep in 1 2 3; do
for v in scan_bench scan_bench_POLL2 scan_bench_POLL4 scan_bench_POLL8 scan_bench_AUX1 scan_bench_AUX2; do
  for a in "64 64 7 1 3 40" "64 64 7 2 3 40"; do
    echo "$v $a: $(timeout -k 10 120 $B/$v $a | grep -E '^scan' | sort -k7 -n | head -1 | awk '{print $7, $8}') $(timeout -k 10 60 $B/$v $a | grep checksum | awk '{print $(NF-4)}')"
  done
done
done
} 2>&1 | tee gpurun_out/r05_scan_bench_variants.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_6.txt 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/r05_gputests_6.txt
tail -12 gpurun_out/r05_gputests_6.txt
