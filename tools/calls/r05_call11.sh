#!/bin/bash
# round 5, call 11: cached jobs in segments (lt_scan_cached_kernel) -- suite, config 2 against segment count and light cost
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_11.txt 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/r05_gputests_11.txt
tail -5 gpurun_out/r05_gputests_11.txt
if [ $rc -ne 0 ]; then exit $rc; fi
for seg in 0 4 8 16 32; do for light in 0.08 0.15 0.25; do
  echo "seg $seg light $light: $(LT_CACHED_SEG=$seg LT_CACHED_LIGHT=$light timeout -k 10 120 python tools/config2_timeline.py)" | tee -a gpurun_out/r05_config2_segments.txt
  if [ $seg -eq 0 ]; then break; fi
done; done
rm -rf gpurun_out/pp2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp2 -- python3 tools/config2_timeline.py > gpurun_out/config2_timeline.out 2> gpurun_out/config2_timeline.err || { tail -5 gpurun_out/config2_timeline.err; exit 1; }
python tools/timeline.py gpurun_out/pp2 | tee gpurun_out/r05_timeline_config2_seg.txt
