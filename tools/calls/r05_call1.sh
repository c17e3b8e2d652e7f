#!/bin/bash
# round 5, call 1: the GPU suite (new tests included), level 3 vs 4 at 20 dB, the driver's bench command
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_1.txt 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/r05_gputests_1.txt
tail -5 gpurun_out/r05_gputests_1.txt
if [ $rc -ge 124 ]; then exit $rc; fi
LIBS=cur LEVELS="3 4" SNRS="20" REPS=3 bash tools/ab.sh > gpurun_out/r05_ab_level_20db.txt 2>&1 || { tail -5 gpurun_out/r05_ab_level_20db.txt; exit 1; }
cat gpurun_out/r05_ab_level_20db.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_cmd_1.json 2> gpurun_out/r05_bench_driver_cmd_1.err || { tail -5 gpurun_out/r05_bench_driver_cmd_1.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_driver_cmd_1.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','cell_id_match_rate','match_vs_oracle')}, d['roofline']['frac'], d.get('steady_state',{}).get('ms_per_step'), d['cpu_baseline']['records_match_gpu'], d['config5']['records_equal'], d['config5']['arg_max_tie_gaps'])
PY
