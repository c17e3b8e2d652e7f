#!/bin/bash
# round 5: parity at volume on the kernels as they are after the cached-scan segments, the early counter reads and the decimator's wait
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/stress6
run() { # name, args...
  n=$1; shift
  timeout -k 10 560 python tools/stress_parity.py "$@" --out gpurun_out/stress6/$n > gpurun_out/stress6/$n.log 2> gpurun_out/stress6/$n.err || { echo "$n FAILED"; tail -3 gpurun_out/stress6/$n.err; return 1; }
  cp gpurun_out/stress6/$n/stress_parity.json gpurun_out/stress6/r05_stress2_$n.json
  python - "$n" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/stress6/r05_stress2_{sys.argv[1]}.json"))
print(sys.argv[1], "scenes", d["scenes"], "records", d["records_compared"], "variant runs", d.get("variant_runs"), "differing", len(d["differing"]), "ties", [t.get("gap") for t in d.get("arg_max_between_lags_equal_to_1e-5", [])], "s", d["seconds"])
PY
}
run plain --scenes 600 --seed0 51000 &&
run chunks --scenes 300 --seed0 52000 --chunks &&
run short --scenes 800 --seed0 53000 --windows 1,9 &&
run variants --scenes 120 --seed0 54000 --variants 3 &&
echo STRESSDONE
