#!/bin/bash
# as tools/ab_libs.sh at several window-cache levels ($LEVELS) and with the default 20 dB step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp gr-ltetrigger_amd/libltetrigger_hip.so gr-ltetrigger_amd/build/lib_cur.so
for rep in 1 2; do
for v in ${LIBS:-old new}; do
  cp gr-ltetrigger_amd/build/lib_$v.so gr-ltetrigger_amd/libltetrigger_hip.so
  for level in ${LEVELS:-3 4}; do for snr in ${SNRS:--4 0}; do
    rm -rf gpurun_out/pp
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pp -- python3 bench.py --snr-db $snr --window-cache $level --steps 6 --warmup 3 --no-end-to-end --no-config2 --no-config5 --no-cpu-baseline --no-overlap-leg --no-steady-leg > /dev/null 2>&1
    echo "$v level $level SNR $snr: $(python tools/timeline.py gpurun_out/pp | tail -1)"
  done; done
done
done
cp gr-ltetrigger_amd/build/lib_cur.so gr-ltetrigger_amd/libltetrigger_hip.so
