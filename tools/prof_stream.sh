#!/bin/bash
# Where a search out of the stream spends its cycles (window cache level 4): a -DLT_PROF build of the library
# (gr-ltetrigger_amd/build/lib_prof.so, selected through LTETRIGGER_HIP_LIB -- the product library is not touched) under
# bench.py at the SNR given, the slowest chains' lines
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
f=$GRAFT_REPO_ROOT/gr-ltetrigger_amd/build/lib_prof.so
[ -f "$f" ] || { echo "missing $f" >&2; exit 1; }
mkdir -p gpurun_out
LTETRIGGER_HIP_LIB=$f timeout -k 10 200 python3 bench.py --snr-db "${1:--4}" --window-cache 4 --steps 2 --warmup 3 --no-end-to-end --no-config2 --no-config5 --no-cpu-baseline --no-overlap-leg --no-steady-leg --no-oracle-check --no-streaming > gpurun_out/prof_stdout.txt 2> gpurun_out/prof_stderr.txt
grep PROF gpurun_out/prof_stdout.txt | tail -56 | sort -k4 -n | tail -8
