"""N > 1 path of the bench on CPU: two gloo ranks shard four channels, generate their own inputs
from the global channel index, process them (with the oracle standing in for the device) and
aggregate exactly like bench.py does; the result must equal the single-rank run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

PER_RANK = 2
N = 30 * 19200


def _frames():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lte_synth
    return [(lte_synth.make_frame(cid, seed=cid), cid) for cid in (10, 200, 333, 461)]


def _work(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle as O
    import ltetrigger_amd as lt
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    mine = lt.sharding.channel_range(rank, world, PER_RANK)
    iq, expect = bench.make_channels(torch, _frames(), PER_RANK, N, mine[0], 20.0, dev)
    recs = O.batch_run(iq.numpy(), nthreads=2)
    found = bench.cells_found(lt, recs, PER_RANK)
    ok = sum(1 for c in range(PER_RANK) if found[c] == (set() if expect[c] < 0 else {int(expect[c])}))
    elapsed = lt.sharding.max_over_ranks(dist, torch, 1.0 + rank, dev)   # slowest rank wins
    ok_sum, ch_sum = lt.sharding.sum_over_ranks(dist, torch, [ok, PER_RANK], dev)
    sig = [float(np.abs(iq[c].numpy()).sum()) for c in range(PER_RANK)]
    out.put((rank, list(mine), sig, ok_sum, ch_sum, elapsed, [sorted(f) for f in found]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_work, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def test_channel_range_partition():
    import ltetrigger_amd as lt
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            seen += list(lt.sharding.channel_range(r, world, 64))
        assert seen == list(range(64 * world))
        assert all(lt.sharding.owner_of(c, 64) == c // 64 for c in seen)
    assert lt.sharding.aggregate_msamp_per_s(8, 64, 1_920_000, 10, 0.05) == pytest.approx(8 * 64 * 1.92e6 * 10 / 0.05 / 1e6)
    with pytest.raises(ValueError):
        lt.sharding.channel_range(2, 2, 64)


def test_two_gloo_ranks_equal_one_rank():
    two = _run(2)
    # ranks own disjoint consecutive channel blocks
    assert two[0][1] == [0, 1] and two[1][1] == [2, 3]
    # aggregated over ranks: every channel counted once, slowest rank's time
    for r in two:
        assert r[4] == 4.0 and r[5] == 2.0
        assert r[3] == 4.0                       # all four cells found
    assert two[0][6] == [[10], [200]] and two[1][6] == [[333], [461]]
    # the same global channels come out of a single rank that owns all four
    os.environ.pop("MASTER_ADDR", None)
    sys.path.insert(0, ROOT)
    import bench
    iq, _ = bench.make_channels(torch, _frames(), 4, N, 0, 20.0, torch.device("cpu"))
    one = [float(np.abs(iq[c].numpy()).sum()) for c in range(4)]
    assert one == two[0][2] + two[1][2]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset) starts two ranks itself and
    prints exactly one JSON line for the job -- here in its dry-run form (gloo, no GPU, no hot-path work)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["dry_run"] is True
    assert out["config"]["channels_total"] == 128      # 64 per rank, every channel owned once
    assert out["slowest_rank_s"] == 1.5                # max over ranks
    # under a launcher the same file must not start ranks again
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run"], env=env2,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1
