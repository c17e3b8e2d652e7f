#!/bin/bash
# round 5, call 9: resolver with helper waves -- suite, timeline, a stress leg, the driver's bench command
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_final.txt 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/r05_gputests_final.txt
tail -12 gpurun_out/r05_gputests_final.txt
if [ $rc -ne 0 ]; then exit $rc; fi
rm -rf gpurun_out/ppf
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ppf -- python3 bench.py --steps 10 --warmup 3 --no-end-to-end --no-config2 --no-config5 --no-cpu-baseline --no-overlap-leg --no-streaming --no-oracle-check > gpurun_out/bench_pathf.json 2> gpurun_out/bench_pathf.err || { tail -5 gpurun_out/bench_pathf.err; exit 1; }
python tools/timeline.py gpurun_out/ppf | tee gpurun_out/r05_timeline_final.txt
timeout -k 10 400 python tools/stress_parity.py --scenes 300 --seed0 43000 --variants 2 --out gpurun_out/stressf > gpurun_out/stressf.log 2> gpurun_out/stressf.err || { tail -3 gpurun_out/stressf.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/stressf/stress_parity.json"))
print("stress: scenes", d["scenes"], "records", d["records_compared"], "variant runs", d.get("variant_runs"), "differing", len(d["differing"]), "ties", [t.get("gap") for t in d.get("arg_max_between_lags_equal_to_1e-5", [])])
PY
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_cmd_final.json 2> gpurun_out/r05_bench_driver_cmd_final.err || { tail -5 gpurun_out/r05_bench_driver_cmd_final.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_driver_cmd_final.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','cell_id_match_rate','match_vs_oracle')}, d['roofline']['frac'], d.get('steady_state',{}).get('ms_per_step'), d['cpu_baseline']['records_match_gpu'], d['config5']['records_equal'], d['streaming']['chunks_10ms']['us_per_chunk'], d['streaming']['chunks_100ms']['us_per_chunk'])
PY
python -c "import __graft_entry__ as g; g.smoke()"
