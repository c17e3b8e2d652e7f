cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -15
for m in noise mix; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$m -- python3 tools/probe.py --mode $m --reps 3 > gpurun_out/p$m.log 2>&1
tail -1 gpurun_out/p$m.log | cut -c1-200
grep -h "lt::" gpurun_out/prof_$m/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
