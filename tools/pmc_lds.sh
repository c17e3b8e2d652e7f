#!/bin/bash
# LDS / VALU counters of the lag-metric and replay kernels (run on the GPU box through gpurun); one
# rocprofv3 --pmc pass per counter group, kernel trace only (see MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_lds
rm -rf $OUT && mkdir -p $OUT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 tools/probe.py --mode mix --reps 2 > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_lds/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        for name in ('lt_metric_kernel', 'lt_chain_kernel', 'lt_edge_kernel'):
            if name in k:
                acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v)
        print(f"  {c:24s} {v[len(v)//2]:.4g}  (per launch, median of {len(v)})")
PY
