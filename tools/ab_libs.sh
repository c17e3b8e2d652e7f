#!/bin/bash
# A/B on one box: builds of the library under gr-ltetrigger_amd/build/lib_<name>.so (names in $LIBS), window cache level $LEVEL,
# the launches of a 64-channel step at the SNRs in $SNRS
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp gr-ltetrigger_amd/libltetrigger_hip.so gr-ltetrigger_amd/build/lib_cur.so
for rep in 1 2; do
for v in ${LIBS:-old new}; do
  cp gr-ltetrigger_amd/build/lib_$v.so gr-ltetrigger_amd/libltetrigger_hip.so
  for snr in ${SNRS:--4 0}; do
    rm -rf gpurun_out/pp
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pp -- python3 bench.py --snr-db $snr --window-cache ${LEVEL:-3} --steps 6 --warmup 3 --no-end-to-end --no-config2 --no-config5 --no-cpu-baseline --no-overlap-leg --no-steady-leg > /dev/null 2>&1
    echo "$v SNR $snr: $(python tools/timeline.py gpurun_out/pp | grep -E 'lt_chain' | sort -k3 -n -r | head -1) $(python tools/timeline.py gpurun_out/pp | tail -1)"
  done
done
done
cp gr-ltetrigger_amd/build/lib_cur.so gr-ltetrigger_amd/libltetrigger_hip.so
