cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 10 --warmup 2 > gpurun_out/bench_prof.log 2>&1
tail -1 gpurun_out/bench_prof.log | cut -c1-300
cat gpurun_out/prof_bench/*/*kernel_stats.csv | cut -c1-160 | head -8
timeout -k 10 600 bash tools/pmc_traffic.sh > gpurun_out/pmc.log 2>&1; tail -12 gpurun_out/pmc.log
timeout -k 10 300 python3 bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-200
