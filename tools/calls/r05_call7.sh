#!/bin/bash
# round 5, call 7: reset fused into the first launch, 32-lag rows, resolver's head from registers -- suite, timeline, bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_8.txt 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/r05_gputests_8.txt
tail -12 gpurun_out/r05_gputests_8.txt
if [ $rc -ne 0 ]; then exit $rc; fi
B=gr-ltetrigger_amd/build
for a in "64 64 7 1 3 40" "64 64 7 2 3 40"; do echo "scan_bench $a: $(timeout -k 10 120 $B/scan_bench $a | grep -E '^scan' | sort -k7 -n | head -1 | awk '{print $7, $8}')"; done | tee gpurun_out/r05_scan_bench_8.txt
rm -rf gpurun_out/pp8
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp8 -- python3 bench.py --steps 10 --warmup 3 --no-end-to-end --no-config2 --no-config5 --no-cpu-baseline --no-overlap-leg --no-streaming --no-oracle-check > gpurun_out/bench_path8.json 2> gpurun_out/bench_path8.err || { tail -5 gpurun_out/bench_path8.err; exit 1; }
python tools/timeline.py gpurun_out/pp8 | tee gpurun_out/r05_timeline_8.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_cmd_8.json 2> gpurun_out/r05_bench_driver_cmd_8.err || { tail -5 gpurun_out/r05_bench_driver_cmd_8.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_driver_cmd_8.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','cell_id_match_rate','match_vs_oracle')}, d['roofline']['frac'], d.get('steady_state',{}).get('ms_per_step'), d['cpu_baseline']['records_match_gpu'], d['config5']['records_equal'], d['streaming']['chunks_10ms']['us_per_chunk'], d['streaming']['chunks_100ms']['us_per_chunk'])
PY
