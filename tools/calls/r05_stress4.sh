#!/bin/bash
# round 5: one more stress campaign on the final tree -- the legs that take the scan rounds through the transforms (window cache
# level 1: nothing scans from the cache; three strips per workgroup forced; batches of ~ 50 channels), emitted half-frames, overlap
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/stress9
run() { # name, args...
  n=$1; shift
  timeout -k 10 560 python tools/stress_parity.py "$@" --out gpurun_out/stress9/$n > gpurun_out/stress9/$n.log 2> gpurun_out/stress9/$n.err || { echo "$n FAILED"; tail -3 gpurun_out/stress9/$n.err; return 1; }
  cp gpurun_out/stress9/$n/stress_parity.json gpurun_out/stress9/r05_stress5_$n.json
  python - "$n" <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/stress9/r05_stress5_{sys.argv[1]}.json"))
print(sys.argv[1], "scenes", d["scenes"], "records", d["records_compared"], "variant runs", d.get("variant_runs"), "differing", len(d["differing"]), "ties", [t.get("gap") for t in d.get("arg_max_between_lags_equal_to_1e-5", [])], "s", d["seconds"])
PY
}
run level1 --scenes 500 --seed0 81000 --window-cache 1 &&
run level1_three_strips --scenes 300 --seed0 82000 --window-cache 1 --scan-groups 3 &&
run stack7 --scenes 50 --seed0 83000 --stack 7 &&
run halfframes --scenes 250 --seed0 84000 --halfframes &&
run chunks_level1 --scenes 250 --seed0 85000 --chunks --window-cache 1 &&
echo STRESSDONE
