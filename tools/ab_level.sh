#!/bin/bash
# A/B on one box: a 64-channel step at marginal SNR with the window cache at level 3 (windows) and level 4 (streams)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 3 4 3 4; do
  for snr in ${SNRS:--4 0 20}; do
    rm -rf gpurun_out/pp
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pp -- python3 bench.py --snr-db $snr --window-cache $v --steps 6 --warmup 3 --no-end-to-end --no-config2 --no-config5 --no-cpu-baseline --no-overlap-leg --no-steady-leg > gpurun_out/ab_level_${v}_${snr}.json 2>/dev/null
    echo "level $v SNR $snr: $(python tools/timeline.py gpurun_out/pp | grep -E 'lt_chain|lt_metric' | tr '\n' ' ') | $(python tools/timeline.py gpurun_out/pp | tail -1)"
  done
done
