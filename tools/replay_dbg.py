"""Replay a scene saved by tools/stress_parity.py (--out: scene_*.npz) and list the first records that differ."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ltetrigger_amd as lt
from oracle import oracle as O
z = np.load(sys.argv[1], allow_pickle=True)
iq = z["iq"]; kw = dict(z["kw"].item())
print(kw, iq.shape, file=sys.stderr)
want = O.batch_run(iq, nthreads=4, **kw)
b = lt.Batch(iq.shape[0], iq.shape[1], **kw)
b.run_host(iq)
got = b.collect()
for c in range(3):
    g, w = got[c], want[c]
    n = min(len(g), len(w))
    bad = [i for i in range(n) if g["peak_pos"][i] != w["peak_pos"][i] or g["flags"][i] != w["flags"][i] or g["in_pos"][i] != w["in_pos"][i] or abs(g["psr"][i]-w["psr"][i]) > 1e-3*abs(w["psr"][i])]
    print("chain", c, len(g), len(w), "first bad", bad[:5], file=sys.stderr)
    if bad:
        i0 = max(0, bad[0]-3)
        for i in range(i0, min(n, bad[0]+4)):
            print("  call", i, "gpu in_pos", g["in_pos"][i], "flags", hex(g["flags"][i]), "score", g["score"][i], "psr %.4f" % g["psr"][i], "peak", g["peak_pos"][i],
                  "| cpu in_pos", w["in_pos"][i], "flags", hex(w["flags"][i]), "score", w["score"][i], "psr %.4f" % w["psr"][i], "peak", w["peak_pos"][i], file=sys.stderr)
