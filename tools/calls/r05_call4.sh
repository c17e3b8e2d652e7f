#!/bin/bash
# round 5, call 4: decimator with paired LDS reads -- parity tests, then old / new timing on one box
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
LT_SKIP_REHEARSAL=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "decim or config" > gpurun_out/r05_gputests_4.txt 2>&1
rc=$?
tail -5 gpurun_out/r05_gputests_4.txt
if [ $rc -ne 0 ]; then exit $rc; fi
{
for rep in 1 2 3; do
for v in decim0 cur; do
  f=$GRAFT_REPO_ROOT/gr-ltetrigger_amd/build/lib_$v.so
  [ "$v" = cur ] && f=$GRAFT_REPO_ROOT/gr-ltetrigger_amd/libltetrigger_hip.so
  for a in "16 50" "8 64" "4 64" "16 1" "8 1"; do
    echo "$v: $(LTETRIGGER_HIP_LIB=$f timeout -k 10 120 python tools/decim_probe.py $a | tail -1)"
  done
done
done
} 2>&1 | tee gpurun_out/r05_decim_probe.txt
