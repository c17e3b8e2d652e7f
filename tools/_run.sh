cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "random_scenes" > gpurun_out/t2.log 2>&1; rc=$?; tail -15 gpurun_out/t2.log
if grep -q "HSA_STATUS_ERROR" gpurun_out/t2.log; then echo FAULT; exit 9; fi
exit $rc
