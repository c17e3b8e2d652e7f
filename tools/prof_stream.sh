#!/bin/bash
# Where a search out of the stream spends its cycles (window cache level 4): a -DLT_PROF build of the library
# (gr-ltetrigger_amd/build/lib_prof.so) under bench.py at the SNR given, the slowest chains' lines
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp gr-ltetrigger_amd/libltetrigger_hip.so gr-ltetrigger_amd/build/lib_cur.so
cp gr-ltetrigger_amd/build/lib_prof.so gr-ltetrigger_amd/libltetrigger_hip.so
timeout -k 10 200 python3 bench.py --snr-db ${1:--4} --window-cache 4 --steps 2 --warmup 3 --no-end-to-end --no-config2 --no-config5 --no-cpu-baseline --no-overlap-leg --no-steady-leg > gpurun_out/prof_stdout.txt 2> gpurun_out/prof_stderr.txt
cp gr-ltetrigger_amd/build/lib_cur.so gr-ltetrigger_amd/libltetrigger_hip.so
grep PROF gpurun_out/prof_stdout.txt | tail -56 | sort -k4 -n | tail -8
