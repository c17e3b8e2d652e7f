"""Per-chain kernel with and without the window cache (lt_batch_set_window_cache) on noise, mode 0: time and records."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ltetrigger_amd as lt
dev = torch.device("cuda", 0)
n = 1920000
for nch in (2, 8):
    g = torch.Generator(device=dev); g.manual_seed(1)
    iq = torch.view_as_complex(torch.randn((nch, n, 2), generator=g, device=dev) * 0.5).contiguous()
    s = torch.cuda.Stream(dev); torch.cuda.synchronize()
    out = {}
    for cache in (0, 1):
        b = lt.Batch(nch, n)
        b.set_mode(0)
        b.set_window_cache(cache)
        for rep in range(2):
            b.reset(); b.run_device(torch.view_as_real(iq).data_ptr(), n, s.cuda_stream)
        recs = b.collect()
        print(f"noise {nch} ch mode 0 cache {cache}: {b.last_kernel_ms():.3f} ms, stages {[round(x,3) for x in b.phase_ms(0)]}, records {sum(len(r) for r in recs)}")
        out[cache] = recs
    for c, (a, d) in enumerate(zip(out[0], out[1])):
        if a.tobytes() != d.tobytes():
            for f in a.dtype.names:
                bad = np.nonzero(a[f] != d[f])[0] if len(a) == len(d) else [0]
                if len(bad):
                    print("chain", c, "field", f, "first bad", bad[:5], a[f][bad[:3]], d[f][bad[:3]], "count", len(bad), "lens", len(a), len(d))
            break
    else:
        print("records identical")
