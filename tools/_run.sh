cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_synth.py -x -q > gpurun_out/t2.log 2>&1; rc=$?; tail -3 gpurun_out/t2.log
if grep -q "HSA_STATUS_ERROR" gpurun_out/t2.log; then echo FAULT; exit 9; fi
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mix -- python3 tools/probe.py --mode mix --reps 3 > gpurun_out/pmix.log 2>&1
grep -h "lt::" gpurun_out/prof_mix/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-110 | head -7
