#!/bin/bash
# round 5, call 10: early counter reads in the product library -- suite; config 2's timeline
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_10.txt 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/r05_gputests_10.txt
tail -5 gpurun_out/r05_gputests_10.txt
if [ $rc -ne 0 ]; then exit $rc; fi
rm -rf gpurun_out/pp2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp2 -- python3 tools/config2_timeline.py > gpurun_out/config2_timeline.out 2> gpurun_out/config2_timeline.err || { tail -5 gpurun_out/config2_timeline.err; exit 1; }
cat gpurun_out/config2_timeline.out
python tools/timeline.py gpurun_out/pp2 | tee gpurun_out/r05_timeline_config2.txt
python tools/config2_timeline.py && python tools/config2_timeline.py 3
