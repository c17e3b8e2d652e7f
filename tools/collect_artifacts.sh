set -x
rm -rf gpurun_out/final && mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/final/r02_bench.json.log 2> gpurun_out/final/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_full -- python3 bench.py > gpurun_out/final/r02_bench_under_rocprof.json.log 2> gpurun_out/final/bench_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_path -- python3 bench.py --steps 10 --warmup 3 --no-end-to-end --no-config2 --no-cpu-baseline --no-overlap-leg > gpurun_out/final/bench_path.json 2> gpurun_out/final/bench_path.err
python tools/timeline.py gpurun_out/final/prof_path > gpurun_out/final/r02_timeline.txt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/final/prof_overlap -- python3 bench.py --overlap --steps 10 --warmup 3 --no-end-to-end --no-config2 --no-cpu-baseline > gpurun_out/final/bench_overlap.json 2> gpurun_out/final/bench_overlap.err
python tools/timeline2.py gpurun_out/final/prof_overlap 3 > gpurun_out/final/r02_timeline_overlap.txt
cp gpurun_out/final/prof_full/*/*kernel_stats.csv gpurun_out/final/r02_kernel_stats.csv
cp gpurun_out/final/prof_path/*/*kernel_stats.csv gpurun_out/final/r02_kernel_stats_path.csv
bash tools/pmc_traffic.sh > gpurun_out/final/pmc_traffic.txt 2>&1
cp gpurun_out/pmc/summary.txt gpurun_out/final/r02_pmc_traffic_summary.txt; cp gpurun_out/pmc/traffic.json gpurun_out/final/traffic.json
bash tools/pmc_scan.sh gr-ltetrigger_amd/build/scan_bench lt_scan_kernel 64 64 7 1 > gpurun_out/final/r02_pmc_scan_kernel.txt 2>&1
gr-ltetrigger_amd/build/ubench_ports > gpurun_out/final/r02_ubench_ports.txt 2>&1
for a in "64 64 7 1" "64 64 3 1" "64 64 1 1" "64 64 7 2" "64 64 7 3"; do echo "== scan_bench $a"; gr-ltetrigger_amd/build/scan_bench $a 2>&1 | grep "^scan\|checksum" | tail -2; done > gpurun_out/final/r02_scan_bench.txt 2>&1
timeout -k 10 500 python tools/stress_parity.py --scenes 400 --seed0 9000 --out gpurun_out/final/stress > gpurun_out/final/stress.json 2> gpurun_out/final/stress.err
cp gpurun_out/final/stress/stress_parity.json gpurun_out/final/r02_stress_parity.json
echo ALLDONE
