"""Oracle on synthetic cells (CPU): boundary cell ids, extended CP, CFO, subframe-5 ordering.
The GPU twin of this sweep (all 504 ids) lives in test_gpu_parity_synth.py."""
import numpy as np
import pytest

import lte_synth
import siggen


@pytest.mark.parametrize("cid,ext", [(0, False), (1, False), (2, True), (167 * 3 + 2, False), (90, True),
                                     (29 * 3, False), (30 * 3 + 1, True), (503, True)])
def test_oracle_recovers_synthetic_cell(oracle, cid, ext):
    f = lte_synth.make_frame(cid, cp_ext=ext, seed=cid)
    x = siggen.loop_to(f, 30 * 19200, rotate=(cid * 997) % 19200)
    for root in range(3):
        recs = oracle.chain_run(x, root)
        tag = recs[(recs["flags"] & oracle.F_TAG_CELL) != 0]
        if root == cid % 3:
            assert tag.size > 30
            assert np.all(tag["cell_id"] == cid)
            assert np.all(((tag["flags"] & oracle.F_CP_EXT) != 0) == ext)
            m0, m1 = oracle.sss_m0m1_of(cid // 3)
            assert set(zip(tag["m0"].tolist(), tag["m1"].tolist())) == {(m0, m1), (m1, m0)}
        else:
            assert tag.size == 0


def test_oracle_tracks_cfo(oracle):
    f = lte_synth.make_frame(210, seed=1)
    for cfo in (-0.3, 0.12, 0.2):  # beyond ~0.4 subcarriers the PSS correlation itself collapses
        x = siggen.apply_cfo(siggen.loop_to(f, 40 * 19200), cfo)
        recs = oracle.chain_run(x, 0)
        tag = recs[(recs["flags"] & oracle.F_TAG_CELL) != 0]
        assert tag.size and np.all(tag["cell_id"] == 210)
        assert float(tag["mean_cfo"][-1]) == pytest.approx(cfo, abs=0.02)
