"""GPU parity on synthetic cells: every one of the 504 physical cell ids (both CP types mixed in),
checked against the expected id and, record for record, against the oracle."""
import numpy as np
import pytest

import lte_synth
import siggen
from test_gpu_parity import assert_records_match

pytestmark = pytest.mark.gpu


def test_all_504_cell_ids(lt, oracle):
    n = 24 * 19200  # 0.24 s per cell
    iq = np.empty((504, n), np.complex64)
    ext = np.zeros(504, bool)
    for cid in range(504):
        ext[cid] = (cid % 5 == 3)
        f = lte_synth.make_frame(cid, cp_ext=bool(ext[cid]), seed=cid, fill_data=(cid % 2 == 0))
        # keep the PSS away from the window edges: a PSS that straddles the edge of the very first
        # window shows up at both ends of every later window, PSR stays ~1.2 and the reference
        # algorithm (oracle and kernel alike) never locks -- see DESIGN.md "blind spot"
        iq[cid] = siggen.loop_to(f, n, rotate=1200 + (cid * 7919) % 7000)
    b = lt.Batch(504, n)
    b.run_host(iq)
    got = b.collect()
    found = 0
    for cid in range(504):
        cells, exts = set(), set()
        for root in range(3):
            r = got[cid * 3 + root]
            t = r[(r["flags"] & lt.F_TAG_CELL) != 0]
            cells |= set(t["cell_id"].tolist())
            exts |= set(((t["flags"] & lt.F_CP_EXT) != 0).tolist())
            if root != cid % 3:
                assert t.size == 0, f"cell {cid}: root {root} reported {set(t['cell_id'].tolist())}"
        # never a wrong id or CP; a handful of loaded cells whose single-window PSR sits just under
        # the threshold never lock in the reference algorithm either ("livelock")
        assert cells <= {cid}, f"cell {cid}: got {cells}"
        if cells:
            assert exts == {bool(ext[cid])}, f"cell {cid}: CP"
            found += 1
    assert found >= 500
    # full record parity with the oracle on every chain
    want = oracle.batch_run(iq, nthreads=16)
    for c in range(504 * 3):
        assert_records_match(got[c], want[c], f"cell {c // 3} root {c % 3}")
