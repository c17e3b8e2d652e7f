#!/bin/bash
# Collects the round's measurement artefacts on the GPU box (one gpurun call):
#   bash tools/collect_artifacts.sh r05        -> gpurun_out/final/r05_*
# Every step is checked: a failed step stops the script (no ALLDONE, no stale copies).
# Before the call, here: make -C gr-ltetrigger_amd all scan_bench   (the box runs the binaries it is sent)
set -euo pipefail
TAG="${1:-r05}"
: "${GRAFT_REPO_ROOT:?must run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT="$GRAFT_REPO_ROOT/gpurun_out/final"
rm -rf "$OUT" && mkdir -p "$OUT"
set -x
# HBM traffic first: bench.py quotes profiles/traffic.json only while its hash matches csrc/ -- the box's copy of the file is
# replaced by this run's before the bench lines are taken (copy "$OUT/traffic.json" to profiles/ afterwards)
bash tools/pmc_traffic.sh > "$OUT/pmc_traffic.txt" 2>&1
cp gpurun_out/pmc/summary.txt "$OUT/${TAG}_pmc_traffic_summary.txt"
cp gpurun_out/pmc/traffic.json "$OUT/traffic.json"
cp gpurun_out/pmc/traffic.json profiles/traffic.json
sleep 20
# the driver's command (5 + 20 steps on a process that finds the GPU idle: the figure the docs lead with), then the defaults (50 + 100)
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${TAG}_bench_driver_cmd.json.log" 2> "$OUT/bench_driver.err"
python bench.py > "$OUT/${TAG}_bench.json.log" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_full" -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/${TAG}_bench_under_rocprof.json.log" 2> "$OUT/bench_prof.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_path" -- python3 bench.py --steps 10 --warmup 3 --no-end-to-end --no-config2 --no-config5 --no-cpu-baseline --no-overlap-leg --no-streaming --no-oracle-check > "$OUT/bench_path.json" 2> "$OUT/bench_path.err"
python tools/timeline.py "$OUT/prof_path" > "$OUT/${TAG}_timeline.txt"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_overlap" -- python3 bench.py --overlap --steps 10 --warmup 3 --no-end-to-end --no-config2 --no-config5 --no-cpu-baseline --no-streaming --no-oracle-check > "$OUT/bench_overlap.json" 2> "$OUT/bench_overlap.err"
python tools/timeline2.py "$OUT/prof_overlap" 3 > "$OUT/${TAG}_timeline_overlap.txt"
cp "$OUT"/prof_full/*/*kernel_stats.csv "$OUT/${TAG}_kernel_stats.csv"
cp "$OUT"/prof_path/*/*kernel_stats.csv "$OUT/${TAG}_kernel_stats_path.csv"
bash tools/pmc_scan.sh gr-ltetrigger_amd/build/scan_bench lt_scan_kernel 64 64 7 1 > "$OUT/${TAG}_pmc_scan_kernel.txt" 2>&1
bash tools/pmc_scan.sh python3 lt_decim tools/decim_probe.py 16 50 > "$OUT/${TAG}_pmc_decim_kernel.txt" 2>&1
for a in "4 16" "8 64" "16 50" "16 1" "8 1"; do python tools/decim_probe.py $a; done > "$OUT/${TAG}_decim_probe.txt" 2>&1
# BASELINE configs[1] (one 15.36 Msps channel): the kernels of one pass
rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_config2" -- python3 tools/config2_timeline.py > "$OUT/config2.out" 2> "$OUT/config2.err"
{ cat "$OUT/config2.out"; python tools/timeline.py "$OUT/prof_config2"; python tools/config2_timeline.py; } > "$OUT/${TAG}_timeline_config2.txt"
# marginal cells: a step's kernels at -4 and 0 dB
for level in 3 4; do for snr in -4 0; do
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof_snr${snr}_l$level" -- python3 bench.py --snr-db $snr --window-cache $level --steps 6 --warmup 3 --no-end-to-end --no-config2 --no-config5 --no-cpu-baseline --no-overlap-leg --no-steady-leg --no-streaming --no-oracle-check > "$OUT/bench_snr${snr}_l$level.json" 2> /dev/null
  { echo "== bench.py --snr-db $snr --window-cache $level (64 channels x 1 s): the kernels of one step"; python tools/timeline.py "$OUT/prof_snr${snr}_l$level"; } >> "$OUT/${TAG}_timeline_marginal.txt"
done; done
for a in "64 64 7 1" "64 64 3 1" "64 64 1 1" "64 64 7 2" "64 64 7 3"; do
  echo "== scan_bench $a"
  { gr-ltetrigger_amd/build/scan_bench $a 2>&1 | grep "^scan\|checksum" | tail -2; } 2>/dev/null
done > "$OUT/${TAG}_scan_bench.txt" 2>&1
gr-ltetrigger_amd/build/scan_bench_prof 64 64 7 1 > "$OUT/${TAG}_scan_phases.txt" 2>&1
python tools/percall_latency.py > "$OUT/${TAG}_percall_latency.txt" 2>&1
# parity at volume: the library's default (window cache level 4: deterministic since round 5), level 3, and level 4 runs
# repeated (--runs 3: every run's records must be the first's, byte for byte -- stress_parity checks the last against the oracle)
timeout -k 10 500 python tools/stress_parity.py --scenes 300 --seed0 9000 --out "$OUT/stress" > "$OUT/stress.json" 2> "$OUT/stress.err"
cp "$OUT/stress/stress_parity.json" "$OUT/${TAG}_stress_parity.json"
timeout -k 10 500 python tools/stress_parity.py --scenes 300 --seed0 9000 --window-cache 3 --out "$OUT/stress_l3" > "$OUT/stress_l3.json" 2> "$OUT/stress_l3.err"
cp "$OUT/stress_l3/stress_parity.json" "$OUT/${TAG}_stress_parity_level3.json"
timeout -k 10 500 python tools/stress_parity.py --scenes 200 --seed0 12000 --variants 3 --out "$OUT/stress_var" > "$OUT/stress_var.json" 2> "$OUT/stress_var.err"
cp "$OUT/stress_var/stress_parity.json" "$OUT/${TAG}_stress_parity_variants.json"
# (scenes of 3 .. 11 channels scan from the window cache since the limit is 12: batches of ~ 50 channels for the rounds through the transforms)
timeout -k 10 500 python tools/stress_parity.py --scenes 30 --seed0 15000 --stack 7 --out "$OUT/stress_stack" > "$OUT/stress_stack.json" 2> "$OUT/stress_stack.err"
cp "$OUT/stress_stack/stress_parity.json" "$OUT/${TAG}_stress_parity_stack7.json"
echo ALLDONE
