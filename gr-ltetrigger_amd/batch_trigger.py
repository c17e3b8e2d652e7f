"""Batched counterpart of downlink_trigger_c: many channels at once on the device.

The pss -> sss part of every chain runs on the device (lt_batch_run: grid scan, per-chain and per-record kernels); the
mib stage replays lib/mib_impl.cc:97-183 on the host over the returned records, decoding PBCH on
the CFO-corrected first subframe of tagged half-frames (lt_cfo_correct reproduces the kernel's
rotation bit for bit)."""
import numpy as np

from . import _binding as B
from . import host

MIN_PSR_THRESHOLD = 1.5


class BatchTrigger:
    def __init__(self, n_channels, max_samples, psr_threshold=4.0, track_after=16, track_every=8,
                 exit_on_success=False, device=0):
        self.psr_threshold = max(psr_threshold, MIN_PSR_THRESHOLD)
        self.batch = B.Batch(n_channels, max_samples, self.psr_threshold, track_after, track_every, device)
        self.cfo = B.Cfo(B.HALF_FRAME, device)
        self.n_channels = n_channels
        self.exit_on_success = exit_on_success

    def set_psr_threshold(self, t):
        self.psr_threshold = max(t, MIN_PSR_THRESHOLD)
        self.batch.set_psr_threshold(self.psr_threshold)

    def process(self, iq):
        """iq: [n_channels, n] complex64.  Returns (records per chain, events) where events is a
        list over channels of [("track" | "drop", cell dict), ...] in stream order per chain."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        self.batch.reset()
        self.batch.run_host(iq)
        recs = self.batch.collect()
        n = iq.shape[1]
        return recs, self._events(recs, lambda ch, lo, hi: iq[ch, lo:hi], n)

    def process_device(self, iq_tensor, stream=None):
        """Same for a torch complex64 tensor [n_channels, n] already resident on the batch's device;
        only the half-frames the MIB stage looks at are copied to the host."""
        import torch
        if iq_tensor.dtype != torch.complex64 or not iq_tensor.is_contiguous():
            raise ValueError("need a contiguous complex64 tensor")
        n = iq_tensor.shape[1]
        if stream is None:
            # the stream the tensor's producer ran on: the library's own stream would not be ordered behind it
            cur = torch.cuda.current_stream(iq_tensor.device)
            stream = cur.cuda_stream
            if not stream:
                # torch's default stream has the handle 0, which the C-ABI reads as "use the batch's own (non-blocking)
                # stream": nothing would order that one behind the producer, so wait for the producer here
                cur.synchronize()
        self.batch.reset()
        self.batch.run_device(torch.view_as_real(iq_tensor).data_ptr(), n, stream)
        recs = self.batch.collect()
        return recs, self._events(recs, lambda ch, lo, hi: iq_tensor[ch, lo:hi].cpu().numpy(), n)

    def _events(self, recs, fetch, n):
        events = [[] for _ in range(self.n_channels)]
        for chain, r in enumerate(recs):
            ch = chain // 3
            events[ch].extend(self._mib_stage(lambda lo, hi, ch=ch: fetch(ch, lo, hi), n, r))
        return events

    def _mib_stage(self, fetch, n, recs):
        out = []
        published = None
        nof_ports_hint = 0
        for r in recs:
            if not (r["flags"] & B.F_EMITTED):
                continue
            if r["flags"] & B.F_TAG_LOST:
                if published is not None:
                    out.append(("drop", published))
                published = None
                continue
            if published is not None or not (r["flags"] & B.F_TAG_CELL):
                continue
            hf = np.zeros(B.HALF_FRAME, np.complex64)
            p0 = int(r["out_pos"])
            lo = max(p0, 0)
            hf[lo - p0:] = fetch(lo, p0 + B.HALF_FRAME)
            freq = float(np.float32(-r["mean_cfo"]) / np.float32(B.SYMBOL_SZ))
            hf = self.cfo.correct(hf, freq)
            cp_ext = bool(r["flags"] & B.F_CP_EXT)
            m = host.decode_mib(hf[:1920], int(r["cell_id"]), cp_ext, nof_ports_hint)
            if m is None:
                continue
            nof_ports_hint = m["nof_tx_ports"]
            cell = dict(cell_id=int(r["cell_id"]), nof_tx_ports=m["nof_tx_ports"],
                        cp_len="Extended" if cp_ext else "Normal", nof_prb=m["nof_prb"],
                        phich_len=m["phich_len"], nof_phich_resources=m["nof_phich_resources"],
                        sfn_offset=m["sfn_offset"], call_idx=int(r["call_idx"]))
            published = cell
            out.append(("track", cell))
            if self.exit_on_success:
                break
        return out
