#!/bin/bash
# round 5, call 3: balanced wave map + mixed workgroups -- scan_bench variants, GPU suite, bench A/B by env
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B=gr-ltetrigger_amd/build
{
for rep in 1 2; do
for a in "64 64 7 1 3 40 0 0 0" "64 64 7 1 3 40 0 1 0" "64 64 7 1 3 40 0 1 1" "64 64 7 1 3 40 1 0 0" "64 64 7 1 3 40 1 1 0" "64 64 7 1 3 40 1 0 1" "64 64 7 1 3 40 1 1 1" "64 64 3 1 3 40 0 0 0" "64 64 3 1 3 40 0 1 0" "64 64 7 2 3 40 0 0 0" "64 64 7 2 3 40 0 1 0" "48 64 7 1 3 40 1 0 0" "48 64 7 1 3 40 1 1 1" "100 64 7 1 3 40 1 0 0" "100 64 7 1 3 40 1 1 1"; do
  echo "== scan_bench $a"
  timeout -k 10 120 $B/scan_bench $a | grep -E "^scan|checksum" | tail -2 || exit 1
done
done
} > gpurun_out/r05_scan_bench_balance.txt 2>&1
grep -E "^==|^scan" gpurun_out/r05_scan_bench_balance.txt | paste - - | awk '{print $3,$4,$5,$6,$7,$8,$9,$10,$11, $17}'
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_3.txt 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/r05_gputests_3.txt
tail -15 gpurun_out/r05_gputests_3.txt
if [ $rc -ne 0 ]; then exit $rc; fi
for rep in 1 2; do
for cfg in "0 0" "1 0" "1 1"; do
  set -- $cfg
  echo "balance $1 mix $2: $(LT_SCAN_BALANCE=$1 LT_SCAN_MIX=$2 LIBS=cur LEVELS=4 SNRS="20" REPS=1 bash tools/ab.sh 2>&1 | tail -1)"
done
done | tee gpurun_out/r05_ab_balance.txt
