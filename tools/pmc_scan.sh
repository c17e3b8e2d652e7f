#!/bin/bash
# SQ / LDS counters of a command (run on the GPU box through gpurun): one rocprofv3 --pmc pass
# per counter group, kernel trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots).  usage: pmc_scan.sh <binary> <kernel substring> [args]
BIN=$1; KERN=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_scan
rm -rf $OUT && mkdir -p $OUT
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- $BIN "$@" > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
python3 - "$KERN" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_scan/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in sorted(acc.items()):
    v = sorted(v)
    print(f"  {c:24s} {v[len(v)//2]:.4g}  (per launch, median of {len(v)})")
PY
