#!/usr/bin/env python3
"""Benchmark of the LTE cell-search hot path (PSS + SSS detect) on MI355X.

Workload (BASELINE.json configs[2]; with N GPUs it is configs[3] scaled, 64 channels per GPU):
each rank owns 64 independent 1.92 Msps capture channels of 1 s each, built from the four
bundled fixtures (decimated to 1.92 Msps, seeded rotation, looped, AWGN 20 dB, every 8th
channel noise only), resident in HBM before the timed region.  One step = one pass of the
PSS->SSS path over the whole batch (state reset + grid scan + per-chain kernel + per-record kernels).
Channels are independent, so ranks never exchange data; torch.distributed (gloo, on host tensors: no RCCL) is used for the barrier and the
max-over-ranks of the wall time only.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python bench.py --gpus N ...          # WORLD_SIZE unset: starts the N ranks itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
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
VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector)
ALGO_BYTES_PER_SAMPLE = 8.0  # cf32 read exactly once, all three roots + SSS served from it
STAGES = ("grid_scan", None, "lt_chain_kernel", "per_record")  # lt_batch_phase_ms slots
# kernels rocprofv3 sees per stage (profiles/traffic.json is keyed by kernel name)
STAGE_KERNELS = {"grid_scan": ("lt_plan_kernel", "lt_scan_kernel", "lt_resolve_kernel"),
                 "lt_chain_kernel": ("lt_chain_kernel",),
                 "per_record": ("lt_cpsss_kernel", "lt_cp_scan_kernel")}
INT_FIELDS = ("in_pos", "out_pos", "flags", "peak_pos", "score", "m0", "m1", "n_id_1", "cell_id", "timer")
EXACT_FLOAT_FIELDS = ("cfo", "mean_cfo", "m_norm_avg", "m_ext_avg", "m0_value", "m1_value")
METRIC_FIELDS = ("psr", "peak_value", "psr_max")


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
    """Seeded per global channel index; see tests/siggen.py for the numpy twin of this generator.  The noise of every
    channel first (one generator each), then the channels of a fixture together: a few large launches rather than
    six small ones per channel."""
    iq = torch.empty((n_channels, n_samples), dtype=torch.complex64, device=dev)
    iq_real = torch.view_as_real(iq)
    expect = np.full(n_channels, -1, np.int64)
    for c in range(n_channels):
        g = torch.Generator(device=dev)
        g.manual_seed(1000 + first_channel + c)
        torch.randn((n_samples, 2), generator=g, device=dev, dtype=torch.float32, out=iq_real[c])
    t = torch.arange(n_samples, device=dev)
    by_fixture = {}
    for c in range(n_channels):
        by_fixture.setdefault((first_channel + c) % len(frames), []).append(c)
    for f, chans in by_fixture.items():
        frame, cid = frames[f]
        fr = torch.from_numpy(frame).to(dev)
        p_sig = float(np.mean(np.abs(frame) ** 2))
        idle = [c for c in chans if (first_channel + c) % 8 == 7]
        live = [c for c in chans if (first_channel + c) % 8 != 7]
        if idle:
            ii = torch.tensor(idle, device=dev)
            iq[ii] = iq[ii] * np.sqrt(p_sig / 2.0)
        if live:
            li = torch.tensor(live, device=dev)
            rot = torch.tensor([int(np.random.default_rng(first_channel + c).integers(0, frame.size)) for c in live], device=dev)
            sig = fr[(t[None, :] + rot[:, None]) % frame.size]
            iq[li] = sig + iq[li] * np.sqrt(p_sig / (10.0 ** (snr_db / 10.0)) / 2.0)
            for c in live:
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


def records_equal(got, want):
    """GPU records against the oracle's: index / flag fields and everything downstream of the lag metric bit for
    bit, the lag-metric quantities to 1e-4 (the parity bar of tests/test_gpu_parity.py)."""
    if len(got) != len(want):
        return False
    ok = True
    for f in INT_FIELDS:
        ok = ok and bool(np.array_equal(got[f], want[f]))
    for f in EXACT_FLOAT_FIELDS:
        ok = ok and bool(np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)))
    for f in METRIC_FIELDS:
        ok = ok and bool(np.allclose(got[f], want[f], rtol=1e-4, atol=0, equal_nan=True))
    return ok


def oracle_check(lt, iq, recs, n_ch, threads=0):
    """This rank's channels through the CPU oracle (the checker; outside the timed region): -> (channels whose set of
    found cells is the oracle's, chains whose records are the oracle's on the parity bar -- oracle.compare_chain --, the
    arg-max ties among them with their gaps)."""
    from oracle import oracle as O
    x = iq[:n_ch].cpu().numpy()
    want = O.batch_run(x, nthreads=cpu_threads_default(threads))
    found_g, found_o = cells_found(lt, recs, n_ch), cells_found(lt, want, n_ch)
    cells_ok = sum(1 for c in range(n_ch) if found_g[c] == found_o[c])
    chains_ok, ties = 0, []
    for c in range(3 * n_ch):
        ok, tie = O.compare_chain(recs[c], want[c], x[c // 3], c % 3)
        chains_ok += bool(ok)
        if tie is not None:
            ties.append(dict(tie, chain=c))
    return cells_ok, chains_ok, ties


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """--gpus N without a launcher: start the N ranks as children (torch.distributed.run) before this
    process has touched a GPU, pass their output through, exit with their code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults long enough for the GPU's clock to have settled: it climbs for ~20 ms after the first launches of a
    # loaded queue -- the same step takes 0.93 ms in a run of 2 + 10 steps, 0.87 ms in 5 + 20, 0.80 ms from step ~40 on)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--channels", type=int, default=64, help="channels per GPU")
    ap.add_argument("--seconds", type=float, default=1.0, help="capture length per channel")
    ap.add_argument("--snr-db", type=float, default=20.0)
    ap.add_argument("--cpu-channels", type=int, default=64, help="channels in the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oracle-check", action="store_true",
                    help="skip `match_vs_oracle` (every rank: the CPU oracle on its own channels, outside the timed region)")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-config2", action="store_true")
    ap.add_argument("--no-streaming", action="store_true", help="skip the streaming leg (lt_batch_push in 10 ms / 100 ms chunks)")
    ap.add_argument("--no-config5", action="store_true", help="skip the BASELINE configs[4] leg (SNR sweep at 30.72 Msps)")
    ap.add_argument("--config5-trials", type=int, default=50, help="seeded trials per SNR point of the config5 leg")
    ap.add_argument("--config5-snrs", default="", help="comma-separated SNRs in dB (default -10..10 in steps of 2)")
    ap.add_argument("--config5-window-cache", type=int, default=4,
                    help="lt_batch_set_window_cache level of the config5 leg (4 = lag-metric streams; the default level's time is reported beside it)")
    ap.add_argument("--config5-seconds", type=float, default=1.0, help="capture length per trial of the config5 leg")
    ap.add_argument("--mode", type=int, default=1, help="lt_batch_set_mode (1 = grid scan + per-chain kernel)")
    ap.add_argument("--window-cache", type=int, default=-1, help="lt_batch_set_window_cache level (default: the library's)")
    ap.add_argument("--overlap", action="store_true",
                    help="time the steps with lt_batch_set_overlap(1): a step's per-record kernels on a second stream beside the "
                         "next step's scan rounds (default: off, the library's default; the bench line carries that "
                         "measurement as an extra leg either way)")
    ap.add_argument("--no-overlap-leg", action="store_true", help="skip the extra overlapped measurement")
    ap.add_argument("--no-steady-leg", action="store_true", help="skip the extra measurement with the clock settled")
    ap.add_argument("--scan-segments", default="", help="first,later: lt_batch_set_scan_segments (windows per scan segment)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the cpu_baseline (0 = the box's CPU share for one GPU: min(available, 16))")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / aggregation plumbing only (gloo, no GPU, no hot path): for the CPU test suite")
    return ap.parse_args(argv)


def dry_run(args, rank, world):
    """No GPU work at all: every rank reports a fixed fake time, rank 0 prints the line.  Exercises the
    self-launch, the rendezvous, the max-over-ranks and the JSON contract on a CPU-only box."""
    import torch
    import torch.distributed as dist
    from ltetrigger_amd import sharding
    if world > 1:
        dist.init_process_group("gloo")
    dev = torch.device("cpu")
    n = int(round(args.seconds * FS))
    mine = sharding.channel_range(rank, world, args.channels)
    elapsed = sharding.max_over_ranks(dist, torch, 1.0 + 0.5 * rank, dev)
    ch_sum, = sharding.sum_over_ranks(dist, torch, [len(mine)], dev)
    if rank == 0:
        print(json.dumps({"metric": "IQ Msamp/s through PSS+SSS detect", "value": 0.0, "unit": "Msamp/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "none (dry run: no hot-path work was done)", "dry_run": True,
                          "config": {"workload": "dry run", "channels_total": int(ch_sum), "samples_per_channel": n},
                          "slowest_rank_s": elapsed}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, rank, world)

    import torch
    import torch.distributed as dist
    import ltetrigger_amd as lt
    from ltetrigger_amd import sharding

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the hot path)")
    # (a rehearsal of the N > 1 path on a box with fewer GPUs than ranks maps the ranks onto the GPUs there are, and the
    # line says so: `ranks_per_gpu`; on the node the driver uses there is one GPU per rank)
    n_dev = torch.cuda.device_count()
    ranks_per_gpu = (world + n_dev - 1) // n_dev
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # north_star: no RCCL -- the ranks never exchange data.  The barrier and the two scalar reductions of the report run
        # over gloo on host tensors.
        dist.init_process_group("gloo")
    cpu = torch.device("cpu")

    n_ch = args.channels
    n = int(round(args.seconds * FS))
    frames = base_frames(lt)
    my_channels = sharding.channel_range(rank, world, n_ch)
    # (the batch and its buffers first, the input last: the allocations leave the GPU idle for tens of milliseconds, and
    # what follows the input's generation is the warm-up)
    batch = lt.Batch(n_ch, n, psr_threshold=4.0, device=local_rank)
    batch.set_mode(args.mode)
    if args.window_cache >= 0:
        batch.set_window_cache(args.window_cache)
    batch.set_overlap(args.overlap)
    if args.scan_segments:
        batch.set_scan_segments(*[int(v) for v in args.scan_segments.split(",")])
    # a dedicated non-default stream: the kernels, the events that time them and the state reset all
    # go onto it (a NULL stream handle would select the batch's own stream inside the library)
    tstream = torch.cuda.Stream(device=dev)
    stream = tstream.cuda_stream
    assert stream != 0

    iq, expect = make_channels(torch, frames, n_ch, n, my_channels[0], args.snr_db, dev)
    iq_real = torch.view_as_real(iq)
    assert iq_real.is_contiguous()
    torch.cuda.synchronize()

    def step():
        batch.reset()
        batch.run_device(iq_real.data_ptr(), n, stream)

    # the timed steps run without the HIP events of lt_batch_phase_ms / lt_batch_last_kernel_ms (each costs the stream
    # ~5 us of idle time); the device times come from a few instrumented steps after the timed region
    batch.set_phase_timing(0)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()

    # (device time per run and per stage: the library's own HIP events on the launch stream, lt_batch_phase_ms below;
    # a second pair of events around every run would cost the timed region another ~10 us per step)
    # (... but one pair of HIP events around the whole timed region, on the stream the kernels are launched on, costs it
    # nothing to speak of: `roofline.timed_region_ms_per_step`, the GPU's own clock on the very steps `value` is the rate of)
    ev_t0, ev_t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_t0.record(tstream)
    for k in range(args.steps):
        batch.reset()
        batch.run_device(iq_real.data_ptr(), n, stream)
    ev_t1.record(tstream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timed_region_ms = ev_t0.elapsed_time(ev_t1) / max(args.steps, 1)

    elapsed = sharding.max_over_ranks(dist, torch, elapsed, cpu)
    if args.steps <= 0:
        if rank == 0:
            print(json.dumps({"metric": "IQ Msamp/s through PSS+SSS detect", "value": None, "unit": "Msamp/s",
                              "n_gpus": world, "steps": 0, "warmup": args.warmup, "error": "no timed steps"}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    # device time of a run and its split per stage: the library's own HIP events on the launch stream, of a few more
    # steps with the events on (outside the timed region)
    batch.set_phase_timing(2)
    n_inst = min(args.steps, 8)
    for _ in range(n_inst):
        step()
    torch.cuda.synchronize()
    phases = np.array([batch.phase_ms(k) for k in range(n_inst)], dtype=np.float64)
    phase_ms = phases.mean(axis=0)
    kernel_ms = float(phase_ms.sum())  # a run, first launch to last (with the ~10 us its two stage events cost)
    batch.set_phase_timing(0)
    if args.overlap:
        # a run's per-record kernels are on the batch's second stream, beside the next step: the path's time per step
        # is the timed region's
        kernel_ms = elapsed / args.steps * 1e3

    # correctness of what was timed: cell_id match on this rank's channels
    recs = batch.collect()
    # extra leg: the same steps once the clock has settled -- whatever --warmup / --steps were, the 50 steps timed here
    # start at least 150 steps after the first one of this process (the overlapped leg below runs after them)
    steady_leg = None
    if not args.overlap and not args.no_steady_leg and args.steps > 0:
        done = args.warmup + args.steps + n_inst
        for _ in range(max(0, 150 - done)):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t2 = time.perf_counter()
        for _ in range(50):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt_st = sharding.max_over_ranks(dist, torch, time.perf_counter() - t2, cpu)
        recs_st = batch.collect()
        v_st = sharding.aggregate_msamp_per_s(world, n_ch, n, 50, dt_st)
        steady_leg = {"value": round(v_st, 3), "unit": "Msamp/s", "steps": 50, "ms_per_step": round(dt_st / 50 * 1e3, 4),
                      "frac": round(v_st * 1e6 * ALGO_BYTES_PER_SAMPLE / world / 1e9 / HBM_PEAK_GBS, 5),
                      "steps_before": max(150, done),
                      "records_identical": all(a.tobytes() == b_.tobytes() for a, b_ in zip(recs, recs_st)),
                      "what": "the same steps, timed the same way, after at least 150 steps of this process: the GPU's clock climbs "
                              "for ~20 ms once its queue is loaded, and `value` is the rate of the steps --warmup / --steps name, "
                              "wherever on that climb they fall"}
    # extra leg: the same steps with the other overlap setting's counterpart (lt_batch_set_overlap(1)): per-record kernels
    # of step i beside the scan rounds of step i + 1; every step's records complete inside the timed region
    overlap_leg = None
    if not args.overlap and not args.no_overlap_leg and args.steps > 0:
        batch.set_overlap(True)
        # (switching allocates the second set of records: the queue runs dry and the clock falls back -- 50 untimed steps,
        # then 50 timed ones, like the steady-state leg this one is to be read against)
        for _ in range(50):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(50):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt_ov = sharding.max_over_ranks(dist, torch, time.perf_counter() - t1, cpu)
        recs_ov = batch.collect()
        same_ov = all(a.tobytes() == b_.tobytes() for a, b_ in zip(recs, recs_ov))
        batch.set_overlap(False)
        overlap_leg = {"value": round(sharding.aggregate_msamp_per_s(world, n_ch, n, 50, dt_ov), 3), "unit": "Msamp/s", "steps": 50,
                       "ms_per_step": round(dt_ov / 50 * 1e3, 4), "records_identical": bool(same_ov),
                       "what": "lt_batch_set_overlap(1): a step's per-record kernels (CFO, CP, SSS) on a second, low-priority stream "
                               "beside the next step's scan rounds; same steps, every step's records complete inside the timed region; "
                               "50 steps after 50 untimed ones: to be read against `steady_state`"}
    cache_stats = batch.cache_stats()
    found = cells_found(lt, recs, n_ch)
    ok = sum(1 for c in range(n_ch)
             if found[c] == (set() if expect[c] < 0 else {int(expect[c])}))
    ok_sum, ch_sum = sharding.sum_over_ranks(dist, torch, [ok, n_ch], cpu)
    match_rate = ok_sum / ch_sum
    # ... and against the CPU oracle on the very same samples, on every rank: `cell_id_match_rate` is defined against
    # construction and reads 0.992 / 0.996 at N = 2 / 8 (two seeded rotations fall into the reference algorithm's window-edge
    # blind spot); `match_vs_oracle` must be 1.0 at every N
    vs_oracle = None
    if not args.no_oracle_check:
        cells_ok, chains_ok, ties_o = oracle_check(lt, iq, recs, n_ch, args.cpu_threads)
        cells_sum, chains_sum, ties_sum = sharding.sum_over_ranks(dist, torch, [cells_ok, chains_ok, len(ties_o)], cpu)
        vs_oracle = {"match_vs_oracle": cells_sum / ch_sum, "chains_on_parity_bar": int(chains_sum), "chains": int(3 * ch_sum),
                     "records_match_oracle": bool(chains_sum == 3 * ch_sum), "arg_max_ties": int(ties_sum),
                     "arg_max_tie_gaps_rank0": ties_o}

    # End to end (pinned host -> H2D -> kernels -> records back): at N > 1 every rank feeds its GPU at the same time, behind a
    # barrier -- the host side of the feed (memory bandwidth, PCIe root complexes) is the one thing the ranks share, and the
    # only thing that can bend the scaling curve of a path without collectives.
    e2e = None
    if not args.no_end_to_end:
        if world > 1:
            dist.barrier()
        e2e = end_to_end(torch, lt, iq, n_ch, n, local_rank)
        if world > 1:
            dt_max = sharding.max_over_ranks(dist, torch, e2e["seconds"], cpu)
            h2d_min = sharding.min_over_ranks(dist, torch, e2e["h2d_gbs"], cpu)
            h2d_sum, = sharding.sum_over_ranks(dist, torch, [e2e["h2d_gbs"]], cpu)
            e2e = dict(e2e, value=round(world * n_ch * n * e2e["passes"] / dt_max / 1e6, 3), rank0_value=e2e["value"],
                       h2d_gbs=round(h2d_sum, 2), h2d_gbs_min_rank=round(h2d_min, 2), h2d_gbs_rank0=e2e["h2d_gbs"],
                       what=e2e["what"].replace("one GPU", f"{world} ranks at the same time: value = all ranks' samples / the "
                                                "slowest rank's time, h2d_gbs = sum over ranks, h2d_gbs_min_rank = the slowest feed"))
    if rank == 0:
        value = sharding.aggregate_msamp_per_s(world, n_ch, n, args.steps, elapsed)
        algo_bytes = ALGO_BYTES_PER_SAMPLE * n_ch * n  # per launch and GPU
        # what each stage has to read and write at least ("kernel I/O", not the path's algorithmic bytes): the scan
        # rounds read the 9600-sample window of every search once and write 12 x 32 B of partial results per searched
        # window plus one 96-byte record per call; the per-chain kernel only sees chains the rounds could not finish
        # (none in this workload) and reads the PSS symbol (128 samples) of every tracked half-frame for its CFO estimate;
        # the per-record kernels read 480 samples per tracked half-frame and rewrite the records
        searched = sum(int(((r["flags"] & lt.F_SEARCHED) != 0).sum()) for r in recs)
        sss_ran = sum(int(((r["flags"] & lt.F_SSS_RAN) != 0).sum()) for r in recs)
        calls = sum(len(r) for r in recs)
        io = {
            "grid_scan": searched * 9600 * 8.0 + searched * 12 * 32.0 + calls * 96.0,
            "lt_chain_kernel": sss_ran * 128 * 8.0 + calls * 96.0,
            "per_record": sss_ran * 480 * 8.0 + calls * 2 * 96.0,
        }
        # The arithmetic: a root wave's window is 935 flop per lane (last forward stage, filter spectrum,
        # inverse transform, |.|^2, average), a forward wave's block 450 (two transform stages): per searched window of a
        # chain 12 strips x 64 lanes x 935, per distinct window of a channel (the chains on one alignment share the
        # forward transform) 12 x 64 x 450.  Counted on what the records say was searched, not on what the rounds ran
        # speculatively.
        distinct = 0
        for ch in range(n_ch):
            pos = [recs[ch * 3 + r]["in_pos"][(recs[ch * 3 + r]["flags"] & lt.F_SEARCHED) != 0] for r in range(3)]
            distinct += int(np.unique(np.concatenate(pos)).size)
        flop_step = 12 * 64 * (935.0 * searched + 450.0 * distinct)
        traffic, traffic_src = load_traffic()
        kernels = []
        for name, ms in zip(STAGES, phase_ms):
            if name is None or not ms > 0:
                continue
            t = None
            if traffic is not None:
                t = sum(traffic.get(k, {}).get("hbm_bytes_per_launch", 0.0) for k in STAGE_KERNELS[name])
            kernels.append({"stage": name, "kernels": list(STAGE_KERNELS[name]), "ms": round(float(ms), 4),
                            "kernel_io_bytes_per_launch": io[name],
                            "kernel_io_gbs": round(io[name] / (ms * 1e-3) / 1e9, 2) if io[name] else None, "traffic": t})
        dom = max(kernels, key=lambda k: k["ms"])
        traffic_total = sum(k["traffic"] for k in kernels) if traffic is not None else None
        out = {
            "metric": "IQ Msamp/s through PSS+SSS detect",
            "value": round(value, 3),
            "unit": "Msamp/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
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
                "execution_mode": batch.mode(),
                "overlap": ("off" if not args.overlap else
                            "lt_batch_set_overlap(1): a step's per-record kernels (CFO, CP, SSS) run on a second stream beside "
                            "the next step's scan rounds; every step's records are complete inside the timed region"),
                "sharding": "independent channels per rank, no collective on the data path (barrier and scalar reductions of the "
                            "report over gloo on host tensors: no RCCL)",
                "ranks_per_gpu": ranks_per_gpu,
            },
            "cell_id_match_rate": match_rate,
            "cell_id_match_note": "the channel's set of found cells is exactly the one cell put there (none for the noise-only "
                                  "channels).  Of the 512 channels of 8 ranks two (global channels 106 and 346: ranks 1 and 5) "
                                  "find no cell at 20 dB -- and neither does the CPU oracle, record for record "
                                  "(tools/check_rank_channels.py): the reference algorithm on those rotations, not the device path",
            "match_vs_oracle": None if vs_oracle is None else vs_oracle["match_vs_oracle"],
            "oracle_check": vs_oracle,
            "window_cache": dict(zip(("windows_listed", "searches_cached", "searches_transformed", "chains_waiting", "moved_off_grid", "reruns_cached", "reruns_transformed"), cache_stats)),
            "realtime_factor": round(value * 1e6 / (world * n_ch * FS), 2),
            # algorithmic bytes = 8 B per input sample, whatever a kernel spills on top.
            # achieved / frac = those bytes / the wall time of a step of the timed region (= `value` x 8 B: reset, every launch of
            # the run, the gaps between runs); `timed_region_*` = the same steps by a pair of HIP events on the launch stream
            # around the whole timed region (in overlap mode the per-record kernels of the last step, on their own stream,
            # end after the second event); `device_*` = the same against the run's launches alone (first to last, by the
            # library's HIP events on the launch stream); `stage_*` = against the dominant stage alone (the scan rounds:
            # lt_plan_kernel + lt_scan_kernel + lt_resolve_kernel launches).
            "roofline": {
                "bound": "hbm",
                "binds": "valu",
                "bound_note": "reported against HBM as north_star asks (8 B per input sample); what binds the path is fp32 vector "
                              "issue (a matched filter of ~250 flop per 8-byte sample even as an FFT): see `valu`",
                "valu": {"flop_per_launch": flop_step, "achieved": round(flop_step / (elapsed / args.steps) / 1e12, 2),
                         "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(flop_step / (elapsed / args.steps) / 1e12 / VALU_PEAK_TFLOPS, 4),
                         "searched_windows": searched, "distinct_windows": distinct,
                         "note": "935 flop per lane and searched (chain, window, strip), 450 per distinct (channel, window, strip) "
                                 "[DESIGN 4.6]; peak = 157.3 TFLOP/s fp32 vector, which counts every slot as a fused multiply-add"},
                "kernel": "lt_scan_kernel (dominant; stage '%s' = %s)" % (dom["stage"], " + ".join(dom["kernels"])),
                "algorithmic_bytes_per_launch": algo_bytes,
                "achieved": round(algo_bytes / (elapsed / args.steps) / 1e9, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(algo_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5),
                "timed_region_ms_per_step": round(timed_region_ms, 4),
                "timed_region_frac": round(algo_bytes / (timed_region_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "device_ms": round(kernel_ms, 4),
                "device_achieved": round(algo_bytes / (kernel_ms * 1e-3) / 1e9, 2),
                "device_frac": round(algo_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "stage_ms": dom["ms"],
                "stage_achieved": round(algo_bytes / (dom["ms"] * 1e-3) / 1e9, 2),
                "stage_frac": round(algo_bytes / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                # HBM bytes per launch from the PMC counters of a profiled run of this same command
                "traffic": traffic_total,
                "traffic_ratio": round(traffic_total / algo_bytes, 3) if traffic_total else None,
                "traffic_source": traffic_src,
                "kernels": kernels,
            },
        }
        if overlap_leg is not None:
            out["overlapped"] = overlap_leg
        if steady_leg is not None:
            out["steady_state"] = steady_leg
        # the side measurements run at N = 1 only (rank 0 would keep the other ranks waiting at the last barrier)
        if e2e is not None:
            out["end_to_end"] = e2e
        if world == 1 and not args.no_streaming:
            out["streaming"] = {
                "what": "lt_batch_push: the same 64 channels as a stream, chunk by chunk from pinned host memory (H2D of the next "
                        "chunk under this chunk's kernels, records back per chunk); wall time per chunk on one GPU",
                "chunks_10ms": streaming(torch, lt, iq, n_ch, n, local_rank, 19200, recs),
                "chunks_100ms": streaming(torch, lt, iq, n_ch, n, local_rank, 192000, recs)}
        if world == 1 and not args.no_config2:
            out["config2"] = config2(torch, lt, dev, local_rank)
        if world == 1 and not args.no_config5:
            snrs = [float(v) for v in args.config5_snrs.split(",")] if args.config5_snrs else list(range(-10, 11, 2))
            out["config5"] = config5(torch, lt, dev, local_rank, snrs, args.config5_trials, args.config5_seconds, args.cpu_threads,
                                     args.config5_window_cache)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(lt, iq, recs, min(args.cpu_channels, n_ch), value / world, args.cpu_threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def load_traffic():
    """profiles/traffic.json: per-kernel HBM bytes per launch from rocprofv3 PMC passes over this workload
    (tools/pmc_traffic.sh).  Used only if it was taken with the kernels that exist now."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(tpath) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, "no profiles/traffic.json"
    k = t.get("kernels", {})
    if "lt_scan_kernel" not in k:
        return None, "profiles/traffic.json predates the current kernels"
    if t.get("kernel_source_sha16") != kernel_source_sha16():
        return None, ("profiles/traffic.json was taken with other kernel sources (its kernel_source_sha16 "
                      f"{t.get('kernel_source_sha16')} != {kernel_source_sha16()}): re-run tools/pmc_traffic.sh")
    return k, f"profiles/traffic.json ({t.get('taken', 'undated')}; {t.get('method', '')})"


def kernel_source_sha16():
    """Hash of the device sources (gr-ltetrigger_amd/csrc): ties profiles/traffic.json to the kernels it measured."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gr-ltetrigger_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip", ".cpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def end_to_end(torch, lt, iq, n_ch, n, device):
    """SURVEY 8(d) end-to-end figure: pinned host buffer -> hipMemcpyAsync -> kernels -> records back to the host,
    two batches on two streams so that the copy of one pass overlaps the kernels of the other."""
    host = torch.empty((n_ch, n), dtype=torch.complex64, pin_memory=True)
    host.copy_(iq)
    torch.cuda.synchronize()
    hnp = host.numpy()
    batches = [lt.Batch(n_ch, n, psr_threshold=4.0, device=device) for _ in range(2)]
    passes = 4
    for b in batches:  # warm-up: staging buffers, first-touch
        b.reset()
        b.run_host(hnp)
        b.collect()
    t0 = time.perf_counter()
    for i in range(passes):
        b = batches[i & 1]
        if i >= 2:
            b.collect()  # records of the pass this batch ran two passes ago
        b.reset()
        b.run_host(hnp)  # H2D + kernels, asynchronous on the batch's own stream
    for b in batches:
        b.collect()
    dt = time.perf_counter() - t0
    for b in batches:
        b.close()
    return {"value": round(n_ch * n * passes / dt / 1e6, 3), "unit": "Msamp/s", "passes": passes, "seconds": dt,
            "ms_per_pass": round(dt / passes * 1e3, 3),
            "h2d_gbs": round(n_ch * n * 8.0 * passes / dt / 1e9, 2),
            "what": "pinned host -> hipMemcpyAsync -> kernels -> records D2H, double-buffered over two streams; one GPU"}


def streaming(torch, lt, iq, n_ch, n, device, chunk_samples, ref_recs, seconds=None):
    """The reference is a streaming block (set_history(9600), consume_each: lib/pss_impl.cc:81-82,195): the same n_ch
    channels fed as a STREAM, in chunks of chunk_samples per channel from pinned host memory -- per chunk: H2D into one of
    two device buffers (on a copy stream, the next chunk's copy under this chunk's kernels), lt_batch_push on the device
    buffer, the chunk's records back to the host (lt_batch_collect).  Wall time per chunk, as a factor over real time, and
    the records of all chunks together against the single run's."""
    from oracle import oracle as O
    dev = iq.device
    n_chunks = n // chunk_samples
    if seconds is not None:
        n_chunks = min(n_chunks, max(2, int(round(seconds * FS / chunk_samples))))
    total = n_chunks * chunk_samples
    # chunk-major pinned memory: what a capture front end hands over every chunk_samples / 1.92 MHz seconds
    host = torch.empty((n_chunks, n_ch, chunk_samples), dtype=torch.complex64, pin_memory=True)
    host.copy_(iq[:, :total].reshape(n_ch, n_chunks, chunk_samples).permute(1, 0, 2))
    torch.cuda.synchronize()
    dbuf = [torch.empty((n_ch, chunk_samples), dtype=torch.complex64, device=dev) for _ in range(2)]
    b = lt.Batch(n_ch, chunk_samples, psr_threshold=4.0, device=device)
    b.set_phase_timing(0)
    s_run, s_copy = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    copied = [torch.cuda.Event() for _ in range(2)]
    freed = [torch.cuda.Event() for _ in range(2)]

    def feed(k):  # H2D of chunk k into buffer k & 1, once the push that read that buffer two chunks ago is through
        with torch.cuda.stream(s_copy):
            if k >= 2:
                s_copy.wait_event(freed[k & 1])
            dbuf[k & 1].copy_(host[k], non_blocking=True)
            copied[k & 1].record(s_copy)

    def one_pass(timed):
        b.reset()
        got = [[] for _ in range(3 * n_ch)]
        per_chunk = []
        feed(0)
        t_start = time.perf_counter()
        for k in range(n_chunks):
            t0 = time.perf_counter()
            if k + 1 < n_chunks:
                feed(k + 1)
            s_run.wait_event(copied[k & 1])
            b.push_device(torch.view_as_real(dbuf[k & 1]).data_ptr(), chunk_samples, end_of_stream=(k == n_chunks - 1), stream=s_run.cuda_stream)
            freed[k & 1].record(s_run)
            recs = b.collect()  # (synchronises the run's stream: the chunk's records are on the host when it returns)
            per_chunk.append(time.perf_counter() - t0)
            if not timed:
                for c, r in enumerate(recs):
                    if len(r):
                        got[c].append(r.copy())
        return time.perf_counter() - t_start, per_chunk, got
    _, _, got = one_pass(False)  # warm-up, and the records that are compared
    # (the warm-up pass leaves tens of thousands of small arrays alive: a collection of the oldest generation in the middle of
    # the timed pass was one chunk of 40 ms among a hundred of 0.2 -- the interpreter's pause, not the path's)
    import gc
    gc.collect()
    gc.disable()
    try:
        dt, per_chunk, _ = one_pass(True)
    finally:
        gc.enable()
    launches = b.last_launch_count()
    b.close()
    # the records of the stream = those of the single run up to where the stream ends (same parity bar as every comparison;
    # the stream's last call may be missing where the shorter capture ends inside it)
    ok = True
    if total == n:
        for c in range(3 * n_ch):
            g = np.concatenate(got[c]) if got[c] else np.zeros(0, lt.RECORD_DTYPE)
            w = ref_recs[c]
            m = min(len(g), len(w))
            ok = ok and len(g) == len(w) and O.first_difference(g[:m], w[:m]) is None
    pc = np.array(per_chunk) * 1e6
    chunk_s = chunk_samples / FS
    return {"chunk_ms": round(chunk_s * 1e3, 3), "chunks": n_chunks, "us_per_chunk": round(dt / n_chunks * 1e6, 1),
            "us_per_chunk_p50": round(float(np.median(pc)), 1), "us_per_chunk_max": round(float(pc.max()), 1),
            "realtime_factor": round(chunk_s / (dt / n_chunks), 1), "value": round(n_ch * total / dt / 1e6, 1), "unit": "Msamp/s",
            "h2d_gbs": round(n_ch * total * 8.0 / dt / 1e9, 2), "launches_per_chunk": launches,
            "records_match_single_run": bool(ok) if total == n else None}


def config2(torch, lt, dev, device):
    """BASELINE configs[1]: one 50-PRB channel at 15.36 Msps, 1 s, resident in HBM -> lt_decim_kernel<8> -> the
    batched path (three chains) -> cell 125.  Device time per pass and its roofline at 8 B per input-rate sample."""
    raw = np.fromfile(os.path.join(ROOT, "tests", "golden", "test_frames", "lte_frame_50prb_cellid_125"), dtype=np.complex64)
    xin = torch.from_numpy(np.tile(raw, 100)).to(dev)  # 15.36 M samples
    n_in, n_out = xin.numel(), xin.numel() // 8
    y = torch.empty((1, n_out), dtype=torch.complex64, device=dev)
    dec = lt.Decim(8, device=device)
    b = lt.Batch(1, n_out, psr_threshold=4.0, device=device)
    s = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    t_dec, t_all = [], []
    for _ in range(6):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        b.reset()
        e0.record(s)
        dec.run_device(torch.view_as_real(xin).data_ptr(), torch.view_as_real(y).data_ptr(), 1, n_in, s.cuda_stream)
        e1.record(s)
        b.run_device(torch.view_as_real(y).data_ptr(), n_out, s.cuda_stream)
        e2.record(s)
        torch.cuda.synchronize()
        t_dec.append(e0.elapsed_time(e1))
        t_all.append(e0.elapsed_time(e2))
    recs = b.collect()
    cells = set()
    for r in recs:
        cells |= set(r["cell_id"][(r["flags"] & lt.F_TAG_CELL) != 0].tolist())
    ms, dms = float(np.median(t_all[1:])), float(np.median(t_dec[1:]))
    algo = 8.0 * n_in
    return {"workload": "1 channel, lte_frame_50prb_cellid_125 looped x100 @15.36 Msps (BASELINE configs[1]), decimation 8",
            "device_ms": round(ms, 4), "decimator_ms": round(dms, 4), "path_ms": round(ms - dms, 4),
            "launches": 1 + int(b.last_launch_count()),  # the decimator's + lt_batch_run's
            "value": round(n_in / ms / 1e3, 1), "unit": "Msamp/s at the input rate",
            "roofline": {"bound": "hbm", "algorithmic_bytes": algo, "achieved": round(algo / (ms * 1e-3) / 1e9, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "note": "one channel: 15 small launches in a row (plan, then lag metrics / cached scan / resolver per "
                                 "round, per-chain and per-record kernels); a round's windows are dealt to eight waves per "
                                 "(strip, root) that take the lag metrics from the window cache (lt_scan_cached_kernel); "
                                 "the decimator is VALU-bound (33 taps per input sample, tap-ordered unfused sums)"},
            "cells": sorted(int(c) for c in cells), "ok": cells == {125}}


def cpu_threads_default(threads=0):
    """The GPU box hands one GPU's job 16 of the host's cores, whatever os.cpu_count() sees."""
    try:
        avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        avail = os.cpu_count() or 1
    return threads if threads > 0 else min(avail, 16)


def first_cell(recs3, halfframe, decode, F):
    """What a downlink_trigger_c with exit_on_success publishes first for one channel: every chain's mib stage walks its
    emitted half-frames in call order (lib/mib_impl.cc:97-183: a tracking_lost tag resets, a cell_id tag starts an
    attempt on the half-frame's first subframe, the first success publishes); the channel's first `track` is the one
    earliest in the stream.  recs3: the channel's three record arrays; halfframe(record) -> the pss block's output for
    it (9600 CFO-corrected samples); decode(subframe, cell_id, cp_ext) -> dict | None.  -> (cell_id, nof_prb, ports) | None."""
    best = None
    for r3 in recs3:
        em = r3[((r3["flags"] & F.F_EMITTED) != 0) & ((r3["flags"] & F.F_TAG_LOST) == 0) & ((r3["flags"] & F.F_TAG_CELL) != 0)]
        for r in em:
            m = decode(halfframe(r)[:1920], int(r["cell_id"]), bool(r["flags"] & F.F_CP_EXT))
            if m is not None:
                cand = (int(r["out_pos"]), int(r["n_id_2"]), (int(r["cell_id"]), int(m["nof_prb"]), int(m["nof_tx_ports"])))
                if best is None or cand[:2] < best[:2]:
                    best = cand
                break
    return None if best is None else best[2]


def config5(torch, lt, dev, device, snrs, trials, seconds, threads=0, window_cache=4):
    """BASELINE configs[4] in full: the 100-PRB fixture looped to `seconds` at 30.72 Msps + AWGN, SNR as in
    examples/snr_ltetrigger.grc:612-663 (mean signal power over mean noise power across the full sampled band, both
    before the sum), `trials` seeded host-drawn noise realisations (numpy PCG64: the same samples on every box; trial t's
    realisation is scaled to every SNR point -- common random numbers), lt_decim_kernel<16> -> one `trials`-channel batch
    per point -> MIB replay on the host (the product's host/pbch.cc).  Beside it the CPU oracle on the same decimated
    streams: records chain for chain, detection trial for trial (oracle/mib_oracle.py on the oracle's half-frames).
    Device time per point = decimator + path on HIP events of the stream they are launched on."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    from oracle import mib_oracle as M
    import ltetrigger_amd.host as host
    t_leg = time.perf_counter()
    raw = np.fromfile(os.path.join(ROOT, "tests", "golden", "test_frames", "lte_frame_100prb_cellid_369"), dtype=np.complex64)
    frames = int(round(seconds * 100))
    sig = torch.from_numpy(np.tile(raw, frames)).to(dev)
    p_sig = float(np.mean(np.abs(raw) ** 2))  # over the full 30.72 MHz band
    n_in = sig.numel()
    n_out = n_in // 16
    T = trials
    cores = cpu_threads_default(threads)
    noise = torch.empty((T, n_in), dtype=torch.complex64, device=dev)

    def draw(t):
        nz = np.random.default_rng(500000 + t).standard_normal((n_in, 2), dtype=np.float32)
        noise[t].copy_(torch.view_as_complex(torch.from_numpy(nz)))
        return t
    with ThreadPoolExecutor(max_workers=max(1, min(cores, 8))) as pool:
        list(pool.map(draw, range(T)))
    torch.cuda.synchronize()
    t_noise = time.perf_counter() - t_leg
    mixed = torch.empty((T, n_in), dtype=torch.complex64, device=dev)
    y = torch.empty((T, n_out), dtype=torch.complex64, device=dev)
    dec = lt.Decim(16, device=device)
    bt = lt.BatchTrigger(T, n_out, psr_threshold=4.0, exit_on_success=True, device=device)
    bt.batch.set_phase_timing(0)
    s = torch.cuda.Stream(device=dev)
    yr, mr = torch.view_as_real(y), torch.view_as_real(mixed)
    points = []
    all_trials = all_records = True
    ties_total = 0
    all_ties = []
    for snr in snrs:
        torch.add(sig[None, :], noise, alpha=float(np.sqrt(p_sig / (10.0 ** (snr / 10.0)) / 2.0)), out=mixed)
        torch.cuda.synchronize()
        ms, dms, ms_other = [], [], []
        # (the library's default level first -- timing only --, then the level of this leg: its records are the ones compared)
        for level in ([3, window_cache] if window_cache != 3 else [3]):
            bt.batch.set_window_cache(level)
            ms_l, dms_l = [], []
            for rep in range(3):  # the first pass warms caches and clocks; the median of the three is reported
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                bt.batch.reset()
                e0.record(s)
                dec.run_device(mr.data_ptr(), yr.data_ptr(), T, n_in, s.cuda_stream)
                e1.record(s)
                bt.batch.run_device(yr.data_ptr(), n_out, s.cuda_stream)
                e2.record(s)
                torch.cuda.synchronize()
                dms_l.append(e0.elapsed_time(e1))
                ms_l.append(e0.elapsed_time(e2))
                bt.batch.collect()  # (the host's cue for the next run: one pass or two, lt_batch_set_window_cache 3 / 4)
            if level == window_cache:
                ms, dms = ms_l, dms_l
            else:
                ms_other = ms_l
        recs = bt.batch.collect()
        stats = bt.batch.cache_stats()
        # ---- the product's detection: MIB replay on the host over the records
        cfo = bt.cfo

        def gpu_hf(r, ch):
            p0 = int(r["out_pos"])
            hf = np.zeros(lt.HALF_FRAME, np.complex64)
            lo, hi = max(p0, 0), min(p0 + lt.HALF_FRAME, n_out)
            hf[lo - p0:hi - p0] = y[ch, lo:hi].cpu().numpy()
            return cfo.correct(hf, float(np.float32(-r["mean_cfo"]) / np.float32(lt.SYMBOL_SZ)))
        gpu_cells = [first_cell(recs[3 * t:3 * t + 3], lambda r, t=t: gpu_hf(r, t), host.decode_mib, lt) for t in range(T)]
        # ---- the oracle on the same decimated streams
        yh = y.cpu().numpy()
        t0 = time.perf_counter()
        want = O.batch_run(yh, nthreads=cores)
        cpu_s = time.perf_counter() - t0
        same, tie_list = True, []
        for c in range(3 * T):
            ok, tie = O.compare_chain(recs[c], want[c], yh[c // 3], c % 3)
            if tie is not None:
                tie_list.append(dict(tie, chain=c, snr_db=snr))
            same = same and ok
        ties = len(tie_list)
        all_ties.extend(tie_list)

        def cpu_hf(r, ch):
            p0 = int(r["out_pos"])
            hf = np.zeros(O.HALF_FRAME, np.complex64)
            lo, hi = max(p0, 0), min(p0 + O.HALF_FRAME, n_out)
            hf[lo - p0:hi - p0] = yh[ch, lo:hi]
            return O.cfo_correct(hf, float(np.float32(-r["mean_cfo"]) / np.float32(128)))
        cpu_cells = [first_cell(want[3 * t:3 * t + 3], lambda r, t=t: cpu_hf(r, t), M.decode_mib, O) for t in range(T)]
        target = (369, 100, 1)
        point = {"snr_db": snr, "gpu_detected": sum(c == target for c in gpu_cells), "cpu_detected": sum(c == target for c in cpu_cells),
                 "gpu_wrong_cell": sum(c is not None and c != target for c in gpu_cells),
                 "trial_for_trial_equal": gpu_cells == cpu_cells, "records_equal": same, "arg_max_ties": ties,
                 "ms": round(float(np.median(ms)), 4), "decimator_ms": round(float(np.median(dms)), 4),
                 "ms_level3": round(float(np.median(ms_other)), 4) if ms_other else None,
                 "cpu_oracle_s": round(cpu_s, 3), "searches_cached": stats[1], "searches_transformed": stats[2], "moved_off_grid": stats[4]}
        points.append(point)
        all_trials = all_trials and point["trial_for_trial_equal"]
        all_records = all_records and same
        ties_total += ties
        print(json.dumps(point), file=sys.stderr, flush=True)
    del noise, mixed
    algo = 8.0 * T * n_in  # 8 B per input-rate sample (Msamp/s are counted at the rate the samples enter)
    ms_pts = [p["ms"] for p in points]
    ms_mean = float(np.mean(ms_pts))
    return {
        "workload": f"lte_frame_100prb_cellid_369 looped to {seconds:g} s @30.72 Msps + AWGN (BASELINE configs[4]), {T} seeded "
                    "host-drawn noise realisations scaled per SNR point (SNR over the full sampled band, "
                    "examples/snr_ltetrigger.grc:612-663), lt_decim_kernel<16> -> one batch of all trials -> host MIB replay; "
                    "threshold 4; detection = the channel's first track dict is cell 369 / 100 PRB / 1 port",
        "trials": T, "snr_db": [p["snr_db"] for p in points],
        "gpu_rate": [round(p["gpu_detected"] / T, 3) for p in points],
        "cpu_rate": [round(p["cpu_detected"] / T, 3) for p in points],
        "gpu_wrong_cell": [p["gpu_wrong_cell"] for p in points],
        "trial_for_trial_equal": bool(all_trials), "records_equal": bool(all_records), "arg_max_ties": ties_total,
        "arg_max_tie_gaps": all_ties, "arg_max_tie_rtol": O.ARGMAX_TIE_RTOL,
        "window_cache": window_cache,
        "window_cache_note": "lt_batch_set_window_cache level of this leg.  4 = the per-chain kernel's searches out of the chains' "
                             "lag-metric streams (made for marginal SNR, where every hit moves a chain's grid; same records as the "
                             "oracle's -- this leg checks it -- but not the bits of levels 0..3: tests/test_gpu_stream.py).  The "
                             "library's default is 3; its time per point is ms_per_point_level3",
        "ms_per_point": ms_pts, "ms_per_point_level3": [p["ms_level3"] for p in points], "decimator_ms_per_point": [p["decimator_ms"] for p in points],
        "path_ms_per_point": [round(p["ms"] - p["decimator_ms"], 4) for p in points],
        "searches_cached": [p["searches_cached"] for p in points],
        "searches_transformed": [p["searches_transformed"] for p in points],
        "moved_off_grid": [p["moved_off_grid"] for p in points],
        "msamp_per_s_at_input_rate": round(T * n_in / (ms_mean * 1e-3) / 1e6, 1),
        "msamp_per_s_per_point": [round(T * n_in / (m * 1e-3) / 1e6, 1) for m in ms_pts],
        "cpu_oracle_msamp_per_s_at_1p92": round(T * n_out * len(points) / sum(p["cpu_oracle_s"] for p in points) / 1e6, 1),
        "cpu_threads": cores,
        "roofline": {"bound": "hbm", "bytes_per_sample": 8.0, "algorithmic_bytes_per_point": algo,
                     "achieved": round(algo / (ms_mean * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(algo / (ms_mean * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                     "note": "device time of a point = lt_decim_kernel<16> over all trials' 30.72 Msps streams (where the bytes "
                             "are: 16 x the path's) + the batched path on the decimated streams; median of 3 passes per point, "
                             "mean over the points"},
        "noise_generation_s": round(t_noise, 2), "leg_s": round(time.perf_counter() - t_leg, 2),
    }


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(lt, iq, gpu_recs, n_ch, gpu_value_per_gpu, threads=0):
    """The CPU oracle (restatement of the srsLTE-18.06.1 algorithm the reference runs: FFT-9728
    convolution per root, three independent chains per channel) timed on this box's host cores
    on the first n_ch channels of the same batch; also used to check the GPU records."""
    from oracle import oracle as O
    nproc = os.cpu_count() or 1
    try:
        nproc_avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        nproc_avail = nproc
    # a GPU box hands one GPU's job 16 of the host's cores, whatever os.cpu_count() sees; more threads than that
    # only oversubscribe them (--cpu-threads overrides)
    cores = threads if threads > 0 else min(nproc_avail, 16)
    x = iq[:n_ch].cpu().numpy()
    O.lib()
    passes = 3  # the fastest pass is reported
    dt = float("inf")
    for _ in range(passes):
        t0 = time.perf_counter()
        want = O.batch_run(x, nthreads=cores)
        dt = min(dt, time.perf_counter() - t0)
    # A chain whose records differ is looked at once more (oracle.compare_chain): an arg-max between two lags whose averages
    # agree to 1e-5 in the oracle's own average is excused -- its relative gap is on the line -- if the chain's records up
    # to that call agree; any larger gap fails the comparison.
    same, ties = True, []
    for c in range(n_ch * 3):
        ok, tie = O.compare_chain(gpu_recs[c], want[c], x[c // 3], c % 3)
        if tie is not None:
            ties.append(dict(tie, chain=c))
        same = same and ok
    # all the host's cores: one pass with a thread per available core (3 x n_ch chains at most are
    # busy: OpenMP deals chains to threads)
    all_cores = None
    if nproc_avail > cores:
        t0 = time.perf_counter()
        O.batch_run(x, nthreads=nproc_avail)
        dta = time.perf_counter() - t0
        all_cores = {"value": round(x.size / dta / 1e6, 3), "unit": "Msamp/s", "threads": nproc_avail,
                     "busy_at_most": min(nproc_avail, 3 * n_ch),
                     "sample": f"the same {n_ch} channels, one pass ({dta:.2f} s), OpenMP over the {3 * n_ch} chains"}
    # single core: a smaller sample of the same batch, one thread
    n1 = min(n_ch, 8)
    t0 = time.perf_counter()
    O.batch_run(x[:n1], nthreads=1)
    dt1 = time.perf_counter() - t0
    value = x.size / dt / 1e6
    return {
        "value": round(value, 3),
        "unit": "Msamp/s",
        "cores": cores,
        "kind": "port",
        "what": "restatement of the srsLTE 18.06.1 algorithm (oracle/lte_oracle.c), not srsLTE itself",
        "sample": f"first {n_ch} channels x {x.shape[1]} samples of the same batch, 3 chains each, "
                  f"OpenMP over chains on {cores} threads, best of {passes} passes ({dt:.2f} s wall each, "
                  f"~{dt * cores * passes + dt1:.0f} core-seconds in total)",
        "single_core": {"value": round(x[:n1].size / dt1 / 1e6, 3), "unit": "Msamp/s",
                        "sample": f"first {n1} channels, one thread, one pass ({dt1:.2f} s)"},
        "all_cores": all_cores,
        "arg_max_ties": len(ties),
        "arg_max_tie_gaps": ties,
        "arg_max_tie_rtol": O.ARGMAX_TIE_RTOL,
        "nproc": nproc,
        "nproc_available": nproc_avail,
        "threads_rule": "--cpu-threads" if threads > 0 else "min(nproc_available, 16): the GPU box's CPU share for one GPU",
        "cpu_model": cpu_model(),
        "records_match_gpu": same,
        "gpu_over_cpu": round(gpu_value_per_gpu / value, 1),
    }


if __name__ == "__main__":
    main()
