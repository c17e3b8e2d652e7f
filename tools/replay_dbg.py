"""Replay a scene saved by tools/stress_parity.py (--out: scene_*.npz) and list the first records that differ.
usage: replay_dbg.py scene.npz [window_cache_level] [scan_segments first,later]"""
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
if len(sys.argv) > 2:
    b.set_window_cache(int(sys.argv[2]))
if len(sys.argv) > 3:
    b.set_scan_segments(*[int(v) for v in sys.argv[3].split(",")])
b.run_host(iq)
got = b.collect()
print("cache stats", b.cache_stats(), file=sys.stderr)
for c in range(3):
    g, w = got[c], want[c]
    n = min(len(g), len(w))
    bad = [i for i in range(n) if g["peak_pos"][i] != w["peak_pos"][i] or g["flags"][i] != w["flags"][i] or g["in_pos"][i] != w["in_pos"][i] or abs(g["psr"][i]-w["psr"][i]) > 1e-3*abs(w["psr"][i])]
    print("chain", c, len(g), len(w), "first bad", bad[:5], file=sys.stderr)
    lo = bad[0] - 3 if bad else (n - 4 if len(g) != len(w) else None)
    if lo is not None:
        for i in range(max(0, lo), max(len(g), len(w))):
            if i > max(0, lo) + 8:
                break
            gs = ("gpu in_pos %d out %d flags %s score %d psr %.4f peak %d timer %d" % (g["in_pos"][i], g["out_pos"][i], hex(g["flags"][i]), g["score"][i], g["psr"][i], g["peak_pos"][i], g["timer"][i])) if i < len(g) else "gpu --"
            ws = ("cpu in_pos %d out %d flags %s score %d psr %.4f peak %d timer %d" % (w["in_pos"][i], w["out_pos"][i], hex(w["flags"][i]), w["score"][i], w["psr"][i], w["peak_pos"][i], w["timer"][i])) if i < len(w) else "cpu --"
            print("  call", i, gs, "|", ws, file=sys.stderr)
