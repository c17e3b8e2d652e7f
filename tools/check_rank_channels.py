#!/usr/bin/env python3
"""The bench's channels of ranks 1..7 (global channels 64..511), one rank's worth at a time on one GPU: which channels'
set of found cells differs from the one cell that was put there (bench.py's cell_id_match_rate counts them), and whether
the product's records of such a channel are the CPU oracle's (then the 'mismatch' is what the algorithm does with that
channel's noise, not a defect of the device path).  Run on the GPU box."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import ltetrigger_amd as lt
    import bench
    from oracle import oracle as O
    dev = torch.device("cuda", 0)
    frames = bench.base_frames(lt)
    n = 1920000
    out = []
    for rank in range(1, 8):
        iq, expect = bench.make_channels(torch, frames, 64, n, 64 * rank, 20.0, dev)
        b = lt.Batch(64, n, psr_threshold=4.0)
        b.run_device(torch.view_as_real(iq).data_ptr(), n, torch.cuda.current_stream().cuda_stream)
        recs = b.collect()
        found = bench.cells_found(lt, recs, 64)
        for c in range(64):
            want = set() if expect[c] < 0 else {int(expect[c])}
            if found[c] != want:
                x = iq[c].cpu().numpy()
                o = O.batch_run(x[None, :], nthreads=3, psr_threshold=4.0)
                same = all(bench.records_equal(recs[3 * c + r], o[r]) for r in range(3))
                cells_o = set()
                for r in range(3):
                    cells_o |= set(o[r]["cell_id"][(o[r]["flags"] & lt.F_TAG_CELL) != 0].tolist())
                out.append({"global_channel": 64 * rank + c, "expected": sorted(want), "found": sorted(found[c]),
                            "oracle_found": sorted(cells_o), "records_equal_oracle": bool(same)})
        b.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
