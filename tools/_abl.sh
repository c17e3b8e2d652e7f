cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in $ABLS; do
LT_METRIC_ABL=$a rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_abl$a -- python3 tools/probe.py --mode mix --reps 3 > gpurun_out/pabl$a.log 2>&1
echo "abl $a: $(grep metric_kernel gpurun_out/prof_abl$a/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120)"
done
