#!/bin/bash
# round 5: parity at volume on the final kernels (window cache level 4 = the default)
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/stress5
run() { # name, args...
  n=$1; shift
  timeout -k 10 560 python tools/stress_parity.py "$@" --out gpurun_out/stress5/$n > gpurun_out/stress5/$n.log 2> gpurun_out/stress5/$n.err || { echo "$n FAILED"; tail -3 gpurun_out/stress5/$n.err; return 1; }
  cp gpurun_out/stress5/$n/stress_parity.json gpurun_out/stress5/r05_stress_$n.json
  python - "$n" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/stress5/r05_stress_{sys.argv[1]}.json"))
print(sys.argv[1], "scenes", d["scenes"], "records", d["records_compared"], "variant runs", d.get("variant_runs"), "differing", len(d["differing"]), "ties", [t.get("gap") for t in d.get("arg_max_between_lags_equal_to_1e-5", [])], "s", d["seconds"])
PY
}
run variants --scenes 250 --seed0 38000 --variants 3 &&
run short --scenes 1500 --seed0 39000 --windows 1,9 &&
echo STRESSDONE
