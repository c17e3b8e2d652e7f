#!/bin/bash
# round 5, call 2: deferred lobe walk -- GPU suite, scan_bench old vs new, library A/B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_2.txt 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/r05_gputests_2.txt
tail -15 gpurun_out/r05_gputests_2.txt
if [ $rc -ge 124 ]; then exit $rc; fi
B=gr-ltetrigger_amd/build
{
for rep in 1 2; do
for a in "64 64 7 1 3 40" "64 64 3 1 3 40" "64 64 7 2 3 40" "64 64 7 3 3 40"; do
  for v in scan_bench_old scan_bench; do
    echo "== $v $a"
    timeout -k 10 120 $B/$v $a | grep -E "^scan|checksum" | tail -3 || exit 1
  done
done
done
echo "== scan_bench_prof 64 64 7 1"
timeout -k 10 120 $B/scan_bench_prof 64 64 7 1 3 8 | grep -vE "^head|^part" | tail -20
} > gpurun_out/r05_scan_bench_deferred.txt 2>&1
cat gpurun_out/r05_scan_bench_deferred.txt | grep -E "^==|^scan" | paste - - - - | awk '{print $2, $3,$4,$5,$6, $(NF-11)}' | head -40
LIBS="old cur" LEVELS="4" SNRS="20 0" REPS=2 bash tools/ab.sh > gpurun_out/r05_ab_deferred.txt 2>&1 || { tail -5 gpurun_out/r05_ab_deferred.txt; exit 1; }
cat gpurun_out/r05_ab_deferred.txt
