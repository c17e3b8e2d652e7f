#!/usr/bin/env python3
"""Benchmark of the LTE cell-search hot path (PSS + SSS detect) on MI355X.

Workload (BASELINE.json configs[2]; with N GPUs it is configs[3] scaled, 64 channels per GPU):
each rank owns 64 independent 1.92 Msps capture channels of 1 s each, built from the four
bundled fixtures (decimated to 1.92 Msps, seeded rotation, looped, AWGN 20 dB, every 8th
channel noise only), resident in HBM before the timed region.  One step = one pass of the
PSS->SSS path over the whole batch (state reset + lag-metric, edge-lag and replay kernels).  Channels are
independent, so ranks never exchange data; torch.distributed is used for the barrier and the
max-over-ranks of the wall time only.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FS = 1_920_000
FIXTURES = [("lte_frame_6prb_cellid_123", 1, 123), ("lte_frame_25prb_cellid_124", 4, 124),
            ("lte_frame_50prb_cellid_125", 8, 125), ("lte_frame_100prb_cellid_369", 16, 369)]
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
ALGO_BYTES_PER_SAMPLE = 8.0  # SURVEY 8(d), whole path: cf32 read once, all three roots + SSS served from it
# per kernel of the bulk + replay execution (DESIGN.md "Kernels"): what each one must move at least
METRIC_BYTES_PER_SAMPLE = 8.0 + 3 * 4.0  # lt_metric_kernel: cf32 in, |conv|^2 of three roots out


def base_frames(lt):
    """One radio frame of each fixture at 1.92 Msps (19200 samples), steady state of the
    decimator (second of two looped frames), through the product's GPU decimator."""
    out = []
    gold = os.path.join(ROOT, "tests", "golden", "test_frames")
    for fname, dec, cid in FIXTURES:
        x = np.fromfile(os.path.join(gold, fname), dtype=np.complex64)
        if dec > 1:
            x = lt.Decim(dec).run_host(np.tile(x, 2))[19200:38400]
        out.append((np.ascontiguousarray(x), cid))
    return out


def make_channels(torch, frames, n_channels, n_samples, first_channel, snr_db, dev):
    """Seeded per global channel index; see tests/siggen.py for the numpy twin of this generator."""
    iq = torch.empty((n_channels, n_samples), dtype=torch.complex64, device=dev)
    expect = np.full(n_channels, -1, np.int64)
    t = torch.arange(n_samples, device=dev)
    for c in range(n_channels):
        gc = first_channel + c
        g = torch.Generator(device=dev)
        g.manual_seed(1000 + gc)
        frame, cid = frames[gc % len(frames)]
        fr = torch.from_numpy(frame).to(dev)
        p_sig = float(np.mean(np.abs(frame) ** 2))
        rot = int(np.random.default_rng(gc).integers(0, frame.size))
        noise = torch.randn((n_samples, 2), generator=g, device=dev, dtype=torch.float32)
        noise = torch.view_as_complex(noise)
        if gc % 8 == 7:
            iq[c] = noise * np.sqrt(p_sig / 2.0)
            continue
        iq[c] = fr[(t + rot) % frame.size] + noise * np.sqrt(p_sig / (10.0 ** (snr_db / 10.0)) / 2.0)
        expect[c] = cid
    return iq, expect


def cells_found(lt, recs, n_channels):
    found = []
    for ch in range(n_channels):
        cells = set()
        for root in range(3):
            r = recs[ch * 3 + root]
            cells |= set(r["cell_id"][(r["flags"] & lt.F_TAG_CELL) != 0].tolist())
        found.append(cells)
    return found


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--channels", type=int, default=64, help="channels per GPU")
    ap.add_argument("--seconds", type=float, default=1.0, help="capture length per channel")
    ap.add_argument("--snr-db", type=float, default=20.0)
    ap.add_argument("--cpu-channels", type=int, default=64, help="channels in the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import ltetrigger_amd as lt

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the hot path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from ltetrigger_amd import sharding
    n_ch = args.channels
    n = int(round(args.seconds * FS))
    frames = base_frames(lt)
    my_channels = sharding.channel_range(rank, world, n_ch)
    iq, expect = make_channels(torch, frames, n_ch, n, my_channels[0], args.snr_db, dev)
    iq_real = torch.view_as_real(iq)
    assert iq_real.is_contiguous()
    torch.cuda.synchronize()

    batch = lt.Batch(n_ch, n, psr_threshold=4.0, device=local_rank)
    # a dedicated non-default stream: the kernel, the events that time it and the state reset all
    # go onto it (a NULL stream handle would select the batch's own stream inside the library)
    tstream = torch.cuda.Stream(device=dev)
    stream = tstream.cuda_stream
    assert stream != 0

    def step():
        batch.reset()
        batch.run_device(iq_real.data_ptr(), n, stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    # kernel time: events on the launch stream around every run inside the timed region
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        batch.reset()
        ev[k][0].record(tstream)
        batch.run_device(iq_real.data_ptr(), n, stream)
        ev[k][1].record(tstream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    elapsed = sharding.max_over_ranks(dist, torch, elapsed, dev)
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps else float("nan")
    # per kernel: the library's own HIP events around each launch, of the last (up to 8) timed steps
    phases = np.array([batch.phase_ms(k) for k in range(min(args.steps, 8))], dtype=np.float64)
    phase_ms = phases.mean(axis=0) if len(phases) else np.full(4, np.nan)

    # correctness of what was timed: cell_id match on this rank's channels
    recs = batch.collect()
    found = cells_found(lt, recs, n_ch)
    ok = sum(1 for c in range(n_ch)
             if found[c] == (set() if expect[c] < 0 else {int(expect[c])}))
    ok_sum, ch_sum = sharding.sum_over_ranks(dist, torch, [ok, n_ch], dev)
    match_rate = ok_sum / ch_sum

    if rank == 0:
        value = sharding.aggregate_msamp_per_s(world, n_ch, n, args.steps, elapsed)
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    traffic = json.load(f).get("kernels", {})
            except (OSError, ValueError):
                traffic = {}
        # algorithmic bytes per launch of each kernel: the metric kernel reads every sample once
        # and writes three floats per sample; the edge kernel reads 256 samples and writes 3 x 253 floats per
        # window; the replay kernel reads the 9473 interior lags + 253 edge lags of every window it searches
        # and the 128-sample PSS symbol of every tracked half-frame, and writes one 96-byte record per call
        n_win = n // 9600
        searched = sum(int(((r["flags"] & lt.F_SEARCHED) != 0).sum()) for r in recs)
        sss_ran = sum(int(((r["flags"] & lt.F_SSS_RAN) != 0).sum()) for r in recs)
        calls = sum(len(r) for r in recs)
        algo = {
            "lt_metric_kernel": METRIC_BYTES_PER_SAMPLE * n_ch * n,
            "lt_edge_kernel": float(n_ch * n_win) * (256 * 8 + 3 * 253 * 4),
            "lt_chain_kernel": searched * 9726 * 4.0 + sss_ran * 128 * 8.0 + calls * 96.0,
            # lt_cp_kernel + lt_cp_scan_kernel + lt_sss_kernel: 480 + 128 samples per tracked half-frame, the
            # record read and written once each
            "post_pass": sss_ran * (480 + 128) * 8.0 + calls * 2 * 96.0,
        }
        kernels = []
        for name, ms in zip(("lt_metric_kernel", "lt_edge_kernel", "lt_chain_kernel", "post_pass"), phase_ms):
            if not ms > 0:
                continue
            t = traffic.get(name, {}).get("hbm_bytes_per_launch")
            kernels.append({"kernel": name, "ms": round(float(ms), 4), "algorithmic_bytes_per_launch": algo[name],
                            "achieved": round(algo[name] / (ms * 1e-3) / 1e9, 2), "traffic": t})
        dom = max(kernels, key=lambda k: k["ms"])
        out = {
            "metric": "IQ Msamp/s through PSS+SSS detect",
            "value": round(value, 3),
            "unit": "Msamp/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{n_ch}-channel band scan per GPU, {args.seconds:g} s @1.92 Msps each "
                            "(BASELINE configs[2]; 64/GPU x 8 GPUs = configs[3]); fixtures looped, "
                            f"seeded rotation, AWGN {args.snr_db:g} dB, every 8th channel noise only",
                "channels_per_gpu": n_ch,
                "samples_per_channel": n,
                "psr_threshold": 4.0,
                "sharding": "independent channels per rank, no collective on the data path",
            },
            "cell_id_match_rate": match_rate,
            "realtime_factor": round(value * 1e6 / (world * n_ch * FS), 2),
            # the dominant kernel = the one with the largest measured duration
            "roofline": {
                "bound": "hbm",
                "kernel": dom["kernel"],
                "achieved": dom["achieved"],
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(dom["achieved"] / HBM_PEAK_GBS, 5),
                "traffic": dom["traffic"],
                "kernel_ms": dom["ms"],
                "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                "kernels": kernels,
                # the whole path against its own floor (every sample read once, 8 B)
                "path_ms": round(kernel_ms, 4),
                "path_achieved": round(ALGO_BYTES_PER_SAMPLE * n_ch * n / (kernel_ms * 1e-3) / 1e9, 2),
            },
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(lt, iq, recs, min(args.cpu_channels, n_ch), value)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(lt, iq, gpu_recs, n_ch, gpu_value):
    """The CPU oracle (restatement of the srsLTE-18.06.1 algorithm the reference runs: FFT-9728
    convolution per root, three independent chains per channel) timed on this box's host cores
    on the first n_ch channels of the same batch; also used to check the GPU records."""
    from oracle import oracle as O
    cores = min(os.cpu_count() or 1, 16)  # the GPU box's CPU share for one GPU
    x = iq[:n_ch].cpu().numpy()
    O.lib()
    passes = 3  # ~20 s of CPU work in total; the fastest pass is reported
    dt = float("inf")
    for _ in range(passes):
        t0 = time.perf_counter()
        want = O.batch_run(x, nthreads=cores)
        dt = min(dt, time.perf_counter() - t0)
    same = True
    for c in range(n_ch * 3):
        g, w = gpu_recs[c], want[c]
        if len(g) != len(w):
            same = False
            continue
        for f in ("in_pos", "out_pos", "flags", "peak_pos", "score", "m0", "m1", "n_id_1", "cell_id"):
            same = same and bool(np.array_equal(g[f], w[f]))
    value = x.size / dt / 1e6
    return {
        "value": round(value, 3),
        "unit": "Msamp/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {n_ch} channels x {x.shape[1]} samples of the same batch, 3 chains each, "
                  f"OpenMP over chains on {cores} threads, best of {passes} passes ({dt:.2f} s wall each, "
                  f"~{dt * cores * passes:.0f} core-seconds in total)",
        "records_match_gpu": same,
        "gpu_over_cpu": round(gpu_value / value, 1),
    }


if __name__ == "__main__":
    main()
