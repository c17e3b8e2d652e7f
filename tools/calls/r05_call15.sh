#!/bin/bash
# round 5, call 15: the tree as it stands -- suite, smoke, the driver's command
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_15.txt 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/r05_gputests_15.txt
tail -5 gpurun_out/r05_gputests_15.txt
if [ $rc -ne 0 ]; then exit $rc; fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" &&
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_cmd_15.json 2> gpurun_out/r05_bench_driver_cmd_15.err || { tail -5 gpurun_out/r05_bench_driver_cmd_15.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_driver_cmd_15.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','cell_id_match_rate','match_vs_oracle')}, d['roofline']['frac'], d['roofline'].get('traffic'), d.get('steady_state',{}).get('ms_per_step'), d['cpu_baseline']['records_match_gpu'], d['config5']['records_equal'], d['config2']['device_ms'], d['config2']['launches'], d['streaming']['chunks_10ms']['us_per_chunk_p50'])
PY
