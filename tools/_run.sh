cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4
for m in noise mix; do
timeout -k 10 120 python3 tools/probe.py --mode $m --reps 3 --ablate 256 2>&1 | grep -v amdgpu.ids | cut -c1-250
done
timeout -k 10 120 python3 tools/probe.py --mode mix --reps 3 --fused 2>&1 | grep -v amdgpu.ids | cut -c1-250
