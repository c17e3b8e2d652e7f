#!/bin/bash
# A/B on one box: builds of the library (gr-ltetrigger_amd/build/lib_<name>.so, names in $LIBS) x window-cache levels ($LEVELS)
# x SNRs ($SNRS): the launches of a 64-channel step under rocprofv3 (tools/timeline.py).  The product library is never touched:
# a variant is selected through LTETRIGGER_HIP_LIB (gr-ltetrigger_amd/_binding.py).  Every line carries the variant's sha256.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LIBS=${LIBS:-cur}
for v in $LIBS; do
  f=gr-ltetrigger_amd/build/lib_$v.so
  [ "$v" = cur ] && f=gr-ltetrigger_amd/libltetrigger_hip.so
  [ -f "$f" ] || { echo "missing $f" >&2; exit 1; }
done
mkdir -p gpurun_out
for rep in $(seq 1 "${REPS:-2}"); do
for v in $LIBS; do
  f=$GRAFT_REPO_ROOT/gr-ltetrigger_amd/build/lib_$v.so
  [ "$v" = cur ] && f=$GRAFT_REPO_ROOT/gr-ltetrigger_amd/libltetrigger_hip.so
  sha=$(sha256sum "$f" | cut -c1-12)
  for level in ${LEVELS:-4}; do for snr in ${SNRS:-20}; do
    rm -rf gpurun_out/pp
    LTETRIGGER_HIP_LIB=$f rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pp -- python3 bench.py --snr-db "$snr" --window-cache "$level" \
      --steps "${STEPS:-6}" --warmup "${WARMUP:-3}" --no-end-to-end --no-config2 --no-config5 --no-cpu-baseline --no-overlap-leg --no-steady-leg --no-oracle-check --no-streaming \
      > "gpurun_out/ab_${v}_${level}_${snr}.json" 2> gpurun_out/ab_stderr.txt || { tail -20 gpurun_out/ab_stderr.txt >&2; exit 1; }
    echo "$v ($sha) level $level SNR $snr: $(python tools/timeline.py gpurun_out/pp | tail -1)"
  done; done
done
done
