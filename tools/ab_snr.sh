#!/bin/bash
# A/B on one box: the per-chain kernel's time per step at marginal SNR for two builds of the library (build/lib_old.so, build/lib_new.so)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp gr-ltetrigger_amd/libltetrigger_hip.so gr-ltetrigger_amd/build/lib_cur.so
for v in old new old new; do
  cp gr-ltetrigger_amd/build/lib_$v.so gr-ltetrigger_amd/libltetrigger_hip.so
  for snr in -4 0; do
    rm -rf gpurun_out/pp
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pp -- python3 bench.py --snr-db $snr --steps 6 --warmup 2 --no-end-to-end --no-config2 --no-config5 --no-cpu-baseline --no-overlap-leg --no-steady-leg > /dev/null 2>&1
    echo "$v SNR $snr: $(python tools/timeline.py gpurun_out/pp | grep lt_chain_kernel | sort -k3 -n -r | head -1) $(python tools/timeline.py gpurun_out/pp | tail -1)"
  done
done
cp gr-ltetrigger_amd/build/lib_cur.so gr-ltetrigger_amd/libltetrigger_hip.so
