cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -8
for m in mix; do
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$m -- python3 tools/probe.py --mode $m --reps 3 > gpurun_out/p$m.log 2>&1
grep -h "lt::" gpurun_out/prof_$m/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120 | head -5
timeout -k 10 120 python3 tools/probe.py --mode $m --reps 3 --ablate 256 2>&1 | grep -v amdgpu.ids | cut -c1-250
done
