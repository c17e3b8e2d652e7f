"""BASELINE.json configs[2] and configs[3] at full size on ONE GPU, where the driver's `pytest -m gpu` sees them.

configs[2]: 64 channels x 1 s @1.92 Msps -- bench.py's own generator and seeds, rank 0's channels -- every record of the
            192 chains against the CPU oracle, the found cells against construction and against the oracle.
configs[3]: the 512 channels of eight ranks, 64 per GPU.  The channels are independent
            (python/downlink_trigger_c.py:43-45: even the three roots of a channel are independent chains), a rank's
            result does not depend on which GPU ran it: ranks 1..7's channels (global 64..511) go through lt.Batch in seven
            64-channel runs on this GPU, every record against the oracle.  Two of those channels (106 and 346) carry
            rotations that fall into the reference algorithm's window-edge blind spot: no cell, on the device
            and in the oracle alike -- asserted as such.
            What one GPU cannot show is the 8-GPU hardware curve itself; the launcher, the rendezvous and two ranks
            sharing this GPU are rehearsed by test_two_ranks_share_one_gpu (children started in conftest.py before this
            process touches the GPU).
"""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, REHEARSAL

pytestmark = pytest.mark.gpu

N = 1_920_000
BLIND_SPOT = {106, 346}  # bench.make_channels seeds whose rotation puts the PSS across the edge of the first window


def _rank_run(lt, oracle, torch, bench, frames, rank):
    dev = torch.device("cuda", 0)
    iq, expect = bench.make_channels(torch, frames, 64, N, 64 * rank, 20.0, dev)
    b = lt.Batch(64, N, psr_threshold=4.0)
    b.run_device(torch.view_as_real(iq).data_ptr(), N, torch.cuda.current_stream().cuda_stream)
    recs = b.collect()
    b.close()
    x = iq.cpu().numpy()
    del iq
    want = oracle.batch_run(x, nthreads=min(len(os.sched_getaffinity(0)), 16), psr_threshold=4.0)
    ties = []
    for c in range(192):
        ok, tie = oracle.compare_chain(recs[c], want[c], x[c // 3], c % 3)
        if tie is not None:
            ties.append(dict(tie, global_channel=64 * rank + c // 3, root=c % 3))
        assert ok, (f"rank {rank} channel {64 * rank + c // 3} root {c % 3}: {oracle.first_difference(recs[c], want[c])}"
                    f"{' tie ' + json.dumps(tie) if tie else ''}")
    found_g, found_o = bench.cells_found(lt, recs, 64), bench.cells_found(lt, want, 64)
    assert found_g == found_o, f"rank {rank}: found cells differ from the oracle's"
    wrong = {64 * rank + c for c in range(64) if found_g[c] != (set() if expect[c] < 0 else {int(expect[c])})}
    return wrong, found_g, ties


def test_config3_band_scan_full_size_matches_oracle(lt, oracle):
    """BASELINE configs[2]: the bench's 64 x 1 s batch (rank 0), record for record."""
    import torch
    import bench
    frames = bench.base_frames(lt)
    wrong, found, ties = _rank_run(lt, oracle, torch, bench, frames, 0)
    assert wrong == set(), f"channels whose found cells are not the one put there: {sorted(wrong)}"
    assert sum(1 for f in found if f) == 56  # every 8th channel is noise only
    assert all(t["gap"] is not None and t["gap"] <= oracle.ARGMAX_TIE_RTOL for t in ties), ties


def test_config4_other_ranks_channels_match_oracle(lt, oracle):
    """BASELINE configs[3]: global channels 64..511 (ranks 1..7 of the 8-GPU job), one rank's worth at a time."""
    import torch
    import bench
    frames = bench.base_frames(lt)
    wrong_all, ties_all = set(), []
    for rank in range(1, 8):
        wrong, found, ties = _rank_run(lt, oracle, torch, bench, frames, rank)
        for g in wrong:
            assert found[g - 64 * rank] == set(), f"channel {g}: a wrong cell {found[g - 64 * rank]}"
        wrong_all |= wrong
        ties_all += ties
    # (found cells == the oracle's on every channel was asserted per rank: the two below find nothing in the oracle either)
    assert wrong_all == BLIND_SPOT, f"channels without their cell: {sorted(wrong_all)} (expected the two blind-spot rotations)"
    assert all(t["gap"] is not None and t["gap"] <= oracle.ARGMAX_TIE_RTOL for t in ties_all), ties_all


def test_two_ranks_share_one_gpu():
    """`python bench.py --gpus 2` started by conftest.py before this process touched the GPU: two ranks over gloo on this
    box's one GPU -- the launcher, the rendezvous, the sharding by global channel, the aggregation of the line."""
    if not REHEARSAL:
        pytest.skip("no rehearsal was started (LT_SKIP_REHEARSAL, or no GPU when the session began)")
    assert REHEARSAL["returncode"] == 0, REHEARSAL["stderr"][-3000:]
    lines = [ln for ln in REHEARSAL["stdout"].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, REHEARSAL["stdout"][-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == REHEARSAL["steps"] and out["scaling"] == "weak"
    assert out["config"]["ranks_per_gpu"] == 2
    assert out["value"] > 0 and out["ms_per_step"] > 0
    # 128 channels: 64..127 hold global channel 106, the blind-spot rotation of rank 1
    assert out["cell_id_match_rate"] == pytest.approx(127 / 128)
    assert out["match_vs_oracle"] == 1.0
    assert out["oracle_check"]["records_match_oracle"] is True and out["oracle_check"]["chains"] == 384
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "two_ranks_one_gpu.json"), "w") as f:
        f.write(lines[0] + "\n")
