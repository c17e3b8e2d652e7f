#!/bin/bash
# round 5, call 16: cached scan rounds up to 12 channels -- suite, stress at the batch sizes concerned, crossover once more
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/stress7
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_16.txt 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/r05_gputests_16.txt
tail -5 gpurun_out/r05_gputests_16.txt
if [ $rc -ne 0 ]; then exit $rc; fi
run() { # name, args...
  n=$1; shift
  timeout -k 10 560 python tools/stress_parity.py "$@" --out gpurun_out/stress7/$n > gpurun_out/stress7/$n.log 2> gpurun_out/stress7/$n.err || { echo "$n FAILED"; tail -3 gpurun_out/stress7/$n.err; return 1; }
  cp gpurun_out/stress7/$n/stress_parity.json gpurun_out/stress7/r05_stress3_$n.json
  python - "$n" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/stress7/r05_stress3_{sys.argv[1]}.json"))
print(sys.argv[1], "scenes", d["scenes"], "records", d["records_compared"], "variant runs", d.get("variant_runs"), "differing", len(d["differing"]), "ties", [t.get("gap") for t in d.get("arg_max_between_lags_equal_to_1e-5", [])], "s", d["seconds"])
PY
}
run plain --scenes 500 --seed0 61000 &&
run chunks --scenes 250 --seed0 62000 --chunks &&
run short --scenes 600 --seed0 63000 --windows 1,9 &&
run variants --scenes 150 --seed0 64000 --variants 3 &&
timeout -k 10 300 python tools/cached_crossover.py 4 8 12 16 &&
echo CALL16DONE
