// gfx950 kernels of the LTE cell-search hot path.
//
// lt_batch_run, bulk + replay mode (default):
//   lt_metric_kernel   -- |matched filter|^2 of every sample position for the three PSS roots, overlap-save
//                         FFT-1024, one forward transform per block shared by the roots (chip-wide).
//   lt_edge_kernel     -- the 253 zero-padding-affected lags of the windows on the 9600-sample grid.
//   lt_chain_kernel<*, true> -- one workgroup per (channel, N_id_2) chain replays pss_impl::general_work
//                         (lib/pss_impl.cc:155-223) over the capture: EMA of the lag metric in registers,
//                         arg-max / main-lobe walk / PSR, score and tracking logic, CFO ring; one lt_record
//                         per call.
//   lt_cp_kernel, lt_cp_scan_kernel, lt_sss_kernel -- sss_impl::work (lib/sss_impl.cc:83-156) of the tracked
//                         half-frames, per record, after the replay (its results feed records only).
// lt_batch_run, fused mode: lt_chain_kernel<*, false> does all of it in one persistent workgroup per chain.
// lt_find_kernel, lt_cfo_*, lt_sync_*  -- single-call forms behind the srslte_* shaped C-ABI.
// lt_decim_kernel -- decimating FIR in front of the path.
#include "lt_device.h"

namespace lt {

struct ChainArgs {
  const v2f *iq;          // [n_channels][ch_stride]
  int64_t n_samples;
  int64_t ch_stride;
  int n_channels;
  ChainState *state;      // [n_channels*3], index channel*3 + n_id_2
  lt_record *recs;        // [n_channels*3][rec_cap]
  int rec_cap;
  int *rec_count;         // [n_channels*3]
  v2f *out_hf;            // nullable: [n_channels*3][out_cap][9600]
  int out_cap;
  const DeviceTables *tab;
  float psr_threshold;
  int track_after;
  int track_every;
  const float *metric;    // bulk + replay mode: P[root][channel][pos] from lt_metric_kernel, else null
  int64_t metric_ch_stride;   // floats between channels
  int64_t metric_root_stride; // floats between roots
  const float *edge;          // bulk + replay mode: edge lags of the windows on the 9600-sample grid (lt_edge_kernel)
  int64_t edge_ch_stride;
  int64_t edge_root_stride;
  int ablate;             // diagnostics only (tools/probe.py --ablate): bits 0-7 select the diagnostics build of the
                          // chain kernel and switch parts of it off; 256 = per-chain clocks from the product
                          // build; 512 / 1024 = lag-metric kernel without its stores / loads
  unsigned long long *prof; // diagnostics only: per-phase shader-clock sums of chain 0 (ablate & 8)
};

// kDiag = true is the diagnostics build (tools/probe.py: ablation switches, in-kernel stamps, the
// direct-form filter); the product launches kDiag = false, where all of that folds away.
// Workgroup shape: the fused kernel needs 512 threads (16 half-wave FFT blocks in flight); the replay
// kernel runs serial-ish per-window code that every wave executes, so it takes ONE wave per SIMD
// (256 threads, 38 lags each): two waves sharing a SIMD would only take turns issuing the same work.
template <bool kDiag, bool kBulk> __global__ __launch_bounds__(kBulk ? kThreadsReplay : kThreads) void lt_chain_kernel(ChainArgs a_in)
{
  constexpr int NT = kBulk ? kThreadsReplay : kThreads; // threads
  constexpr int NL = kConvLen / NT;                     // lags per thread
  ChainArgs a = a_in;
  unsigned long long *elapsed_out = nullptr; // diagnostics (flag 256): this chain's shader clocks, product build included
  if (a.prof && (a.ablate & 256)) elapsed_out = a.prof + (size_t)(blockIdx.x % a.n_channels * 3 + blockIdx.x / a.n_channels) * 24 + 23;
  const unsigned long long clk0 = elapsed_out ? __builtin_amdgcn_s_memtime() : 0;
  a.ablate &= 0xFF;
  if (!kDiag) {
    a.ablate = 0;
    a.prof = nullptr;
  }
  // fused mode: window buffer (also FFT scratch), twiddles and filter spectrum; replay mode: only the
  // 2 x 128 edge samples of the window
  __shared__ v2f W[kBulk ? 256 : kWLen];
  // replay mode: two EMA mirrors used in turn, so that a wave still reading the previous search's mirror
  // and a wave already writing the next one never meet (no barrier between searches)
  __shared__ float AVG_[kBulk ? 2 * kAvgLen : kAvgLen];
  float *AVG = AVG_;
  __shared__ v2f S[kSssLen];
  __shared__ v2f RAW[kSssLen];
  __shared__ LdsTables ltab;
  __shared__ v2f TW[kBulk ? 1 : fft::kN];
  __shared__ v2f SPEC[kBulk ? 1 : fft::kN];
  __shared__ float psr_data[kMovingAvgSz];
  __shared__ float cfo_data[kMovingAvgSz];
  __shared__ Scratch sc;
  __shared__ BulkShared<NT> bs;
  // replay mode, batched tracked emits: one half-frame per wave
  constexpr int NW = NT / 64;
  __shared__ v2f RAWb[kBulk ? NW : 1][kBulk ? kSymbolSz : 1]; // the PSS symbol of each wave's half-frame
  __shared__ float batch_cfo[NW];

  const int tid = threadIdx.x;
  // blocks c, c + n_channels, c + 2 n_channels are the three roots of channel c: with n_channels a
  // multiple of 8 they land on the same XCD (round-robin dispatch) and share its L2 copy of the IQ.
  const int channel = blockIdx.x % a.n_channels;
  const int n_id_2 = blockIdx.x / a.n_channels;
  const int chain = channel * 3 + n_id_2;
  const v2f *__restrict__ x = a.iq + (int64_t)channel * a.ch_stride;
  const int64_t n = a.n_samples;
  const DeviceTables *__restrict__ tab = a.tab;
  const v2f *__restrict__ taps = reinterpret_cast<const v2f *>(tab->pss_taps[n_id_2]);
  ChainState *st = a.state + chain;
  lt_record *recs = a.recs + (int64_t)chain * a.rec_cap;

  // ---- load state
  float ema[NL];
#pragma unroll
  for (int t = 0; t < NL; ++t) ema[t] = st->avg[0][kBulk ? tid + t * NT : tid * NL + t];
  for (int i = tid; i < kMovingAvgSz; i += NT) {
    psr_data[i] = st->psr_data[i];
    cfo_data[i] = st->cfo_data[i];
  }
  if (tid < kAvgLen - kConvLen) {
    AVG_[kConvLen + tid] = 0.f;
    if (kBulk) AVG_[kAvgLen + kConvLen + tid] = 0.f;
  }
  load_lds_tables(ltab, tab, n_id_2);
  if (!kBulk) load_filter_lds(TW, SPEC, tab, n_id_2);
  const FilterLds fl = {TW, SPEC};
  const float *__restrict__ P = kBulk ? a.metric + n_id_2 * a.metric_root_stride + channel * a.metric_ch_stride : nullptr;
  const float *__restrict__ EP = kBulk ? a.edge + n_id_2 * a.edge_root_stride + channel * a.edge_ch_stride : nullptr;
  int grid_k = 0;   // replay mode: pos == 9600 * grid_k while the chain is on the grid, else -1
  int buf = 0;      // replay mode: which EMA mirror / maxima set the next search writes
  bool pf_edges = false; // replay mode: the prefetched slice carries its edge lags (window on the grid)
  int64_t pos = 0; // every run starts at sample 0 of the capture handed to it
  int is_tracking = st->is_tracking, score = st->score, timer = st->timer;
  int tracking_lost = st->tracking_lost, peak_pos = st->peak_pos;
  uint32_t psr_i = st->psr_i, cfo_i = st->cfo_i;
  int n_emits = 0;
  float psr = st->psr, psr_max = st->psr_max;
  float m_norm_avg = st->m_norm_avg, m_ext_avg = st->m_ext_avg;
  int nrec = 0;
  const int call_base = st->n_calls;
  v2f pf[kBulk ? 1 : NL]; // fused mode: next window, prefetched (19 * 512 = 9728 >= 9600 samples)
  float pfm[kBulk ? NL : 1]; // replay mode: next window's interior metric slice
  v2f pfe = {0.f, 0.f};      // replay mode: one of the next window's 256 edge samples (tid < 256)
  int64_t pf_pos = -1;       // stream position the prefetch belongs to, -1 = none
#pragma unroll
  for (int i = 0; i < (kBulk ? 1 : NL); ++i) pf[i] = v2f{0.f, 0.f};
#pragma unroll
  for (int i = 0; i < (kBulk ? NL : 1); ++i) pfm[i] = 0.f;
  unsigned long long prof[24] = {0};
  unsigned long long *profp = (kDiag && (a.ablate & 8) && a.prof) ? prof : nullptr;
  __syncthreads();

  // replay mode: thread's 19 lags of the window at q: interior from P, edge lags from the grid table
  // (window index k >= 0) or left alone (k < 0: the caller computes them)
  constexpr int kTailTid0 = kHalfFrame - (NL - 1) * NT; // thread owning lag 9600 in its last slot
  auto load_slice = [&](float (&m)[kBulk ? NL : 1], int64_t q, int k) {
    if constexpr (kBulk) {
      if (kDiag && (a.ablate & 1)) { // diagnostics: no HBM reads
#pragma unroll
        for (int i = 0; i < (kBulk ? NL : 1); ++i) m[i] = (float)((tid * 7 + i) & 15);
        return;
      }
      const float *__restrict__ Pq = P + q + tid;
#pragma unroll
      for (int i = 1; i < NL - 1; ++i) m[i] = __builtin_nontemporal_load(&Pq[i * NT]); // read once
      const float *__restrict__ Ek = EP + (int64_t)(k < 0 ? 0 : k) * 256;
      if (tid >= kSymbolSz - 1) m[0] = __builtin_nontemporal_load(&Pq[0]);
      else if (k >= 0) m[0] = Ek[tid];
      if (tid < kTailTid0) m[NL - 1] = __builtin_nontemporal_load(&Pq[(NL - 1) * NT]);
      else if (k >= 0 && tid < kTailTid0 + 126) m[NL - 1] = Ek[kSymbolSz - 1 + tid - kTailTid0];
    }
  };
  while (pos + kHalfFrame <= n && nrec < a.rec_cap) {
    if constexpr (kBulk) {
      // ---- tracked chain between two searches (lib/pss_impl.cc:163-214 with timer > 0): the next
      // `timer` calls neither search nor change the score; each emits the half-frame at its input
      // position (peak_pos is 960 since the emit of the last search) and runs the SSS stage on it.
      // Their heavy parts are independent, so they are done one per wave, NW at a time:
      //   A  stage samples [480, 960) and estimate the CFO                      (parallel)
      //   B  ring mean with this call's and the earlier calls' estimates entered (parallel, exact)
      //   C  rotate, CP metrics                                                   (parallel)
      //   D  CP averages in call order (a few flops), then FFT-128 / m0 / m1 / N_id_1 (parallel)
      if (is_tracking && timer > 0 && !tracking_lost && peak_pos == kSlotLen && psr > a.psr_threshold) {
        int64_t left = (n - pos) / kHalfFrame;
        int nb = timer;
        if (left < nb) nb = (int)left;
        if (a.rec_cap - nrec < nb) nb = a.rec_cap - nrec;
        const unsigned long long tb_ = profp ? __builtin_amdgcn_s_memtime() : 0;
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        for (int done = 0; done < nb; done += NW) {
          const int nr = (nb - done < NW) ? nb - done : NW;
          const bool act = w < nr;
          const int idx = done + w;                          // call within the batch
          const int64_t hf0 = pos + (int64_t)idx * kHalfFrame;
          float c = 0.f, mc = 0.f;
          if (act) {
            for (int i = lane; i < kSymbolSz; i += 64) RAWb[w][i] = ld_iq(x, n, hf0 + (kSlotLen - kSymbolSz) + i);
            c = cfo_estimate_wave(ltab.taps, RAWb[w]);
            if (lane == 0) batch_cfo[w] = c;
          }
          __syncthreads();
          const uint32_t slot0 = cfo_i % kMovingAvgSz;
          float ring_new = 0.f;
          if (tid < nr) ring_new = batch_cfo[tid];
          v2f *out = nullptr;
          if (act) {
            mc = moving_avg_wave(cfo_data, cfo_i + (uint32_t)w + 1u, slot0, batch_cfo, (uint32_t)w + 1u);
            if (a.out_hf && n_emits + idx < a.out_cap) out = a.out_hf + ((int64_t)chain * a.out_cap + n_emits + idx) * kHalfFrame;
          }
          __syncthreads(); // every wave has read the ring
          if (tid < nr) cfo_data[(slot0 + tid) % kMovingAvgSz] = ring_new;
          if (act) {
            // rotation, CP decision and SSS feed nothing here: lt_cp_kernel / lt_cp_scan_kernel / lt_sss_kernel
            if (lane == 0) {
              lt_record r;
              r.in_pos = hf0; r.out_pos = hf0; r.channel = channel; r.n_id_2 = n_id_2;
              r.call_idx = call_base + nrec + idx; r.flags = LT_F_OVER | LT_F_EMITTED | LT_F_SSS_RAN | LT_F_TRACKING;
              r.peak_pos = kSlotLen; r.score = score;
              r.psr = psr; r.peak_value = 0.f; r.cfo = c; r.mean_cfo = mc;
              r.m_norm_avg = 0.f; r.m_ext_avg = 0.f; r.m0_value = 0.f;
              r.m1_value = 0.f; r.m0 = -1; r.m1 = -1; r.n_id_1 = -1; r.cell_id = -1;
              r.psr_max = psr_max; r.timer = timer - 1 - idx;
              recs[nrec + idx] = r;
            }
            if (out) {
              const float freq = -mc / (float)kSymbolSz;
              const float phase_inc = __fmul_rn(freq, (float)kCexpTabSize);
              for (int i = lane; i < kHalfFrame; i += 64) out[i] = cfo_rotate(tab->cexptab, phase_inc, i, ld_iq(x, n, hf0 + i));
            }
          }
          cfo_i += (uint32_t)nr;
        }
        if (profp) {
          prof[16] += __builtin_amdgcn_s_memtime() - tb_;
          prof[17] += 1;
        }
        if (nb > 0) {
          timer -= nb;
          nrec += nb;
          n_emits += nb;
          if (grid_k >= 0) grid_k += nb;
          pos += (int64_t)nb * kHalfFrame;
          continue;
        }
      }
    }
    uint32_t flags = 0;
    float peak_value = 0.f;
    bool have_result = false; // replay mode: this call's search has been done by the idle loop below
    if constexpr (kBulk) {
      // ---- idle chain (nothing over the threshold lately, still on the capture's 9600-sample grid):
      // every call searches its window, misses, and moves on by 9600 samples -- general_work without
      // the emit branch (lib/pss_impl.cc:163-182, 217).  That is what nearly all chains do nearly all
      // the time, so it gets a loop of its own that keeps nothing but the EMA and the prefetched
      // slice in registers; the first window that fires is handed to the general code below with its
      // search result.
      if (!is_tracking && score == 0 && !tracking_lost && grid_k >= 0 && !(kDiag && a.ablate)) {
        bool fired = false;
        while (pos + kHalfFrame <= n && nrec < a.rec_cap) {
          float met[NL];
          if (pf_pos == pos && pf_edges) {
#pragma unroll
            for (int i = 0; i < NL; ++i) met[i] = pfm[i];
          } else {
            load_slice(met, pos, grid_k);
          }
          float *mirror = AVG_ + buf * kAvgLen;
          bulk_search_ema<NT, NL>(met, ema, mirror, bs.tmax[buf], bs.tidx[buf]);
          pf_pos = pos + kHalfFrame; // a miss leads there
          pf_edges = true;
          if (pf_pos + kHalfFrame <= n) load_slice(pfm, pf_pos, grid_k + 1);
          else pf_pos = -1;
          const SearchResult r = bulk_search_tail<NT, NL>(ema, mirror, bs.tmax[buf], bs.tidx[buf], bs.sl, nullptr);
          buf ^= 1;
          timer = a.track_every;
          peak_pos = r.peak_pos;
          psr = r.psr;
          peak_value = r.peak_value;
          if (tid == 0) psr_data[psr_i % kMovingAvgSz] = psr;
          psr_i++;
          if (psr > a.psr_threshold) {
            fired = true;
            break;
          }
          if (psr > psr_max) psr_max = psr;
          if (tid == 0) {
            lt_record rec;
            rec.in_pos = pos; rec.out_pos = -1; rec.channel = channel; rec.n_id_2 = n_id_2;
            rec.call_idx = call_base + nrec; rec.flags = LT_F_SEARCHED; rec.peak_pos = peak_pos; rec.score = 0;
            rec.psr = psr; rec.peak_value = peak_value; rec.cfo = 0.f; rec.mean_cfo = 0.f;
            rec.m_norm_avg = m_norm_avg; rec.m_ext_avg = m_ext_avg; rec.m0_value = 0.f;
            rec.m1_value = 0.f; rec.m0 = -1; rec.m1 = -1; rec.n_id_1 = -1; rec.cell_id = -1;
            rec.psr_max = psr_max; rec.timer = timer;
            recs[nrec] = rec;
          }
          nrec++;
          grid_k++;
          pos += kHalfFrame;
        }
        if (!fired) continue; // out of samples or of record space
        have_result = true;
        flags |= LT_F_SEARCHED;
      }
    }

    // ---- lib/pss_impl.cc:163-172
    if (have_result) {
      // searched above
    } else if (!is_tracking || timer == 0) {
      timer = a.track_every;
      unsigned long long t0_ = profp ? __builtin_amdgcn_s_memtime() : 0;
      SearchResult r;
      if constexpr (kBulk) {
        // lag tid + NT i.  Interior lags 127..9599 = position-independent metric P[pos + lag]; the 253
        // edge lags come from lt_edge_kernel's table while the chain is on the 9600-sample grid and are
        // recomputed from the window's first and last 128 samples once it has re-aligned.
        float met[NL];
        bool have_edges;
        if (pf_pos == pos) {
#pragma unroll
          for (int i = 0; i < NL; ++i) met[i] = pfm[i];
          have_edges = pf_edges;
          if (!have_edges && tid < 256) W[tid] = pfe; // (NT = 256: every thread carries one edge sample)
        } else {
          have_edges = grid_k >= 0;
          load_slice(met, pos, grid_k);
          if (!have_edges) {
            if (tid < 128) W[tid] = x[pos + kHalfFrame - 128 + tid]; // Z = [window tail | window head]
            else if (tid < 256) W[tid] = x[pos - 128 + tid];
          }
        }
        if (!have_edges) { // (block-uniform)
          __syncthreads();
          edge_lags(W, ltab.taps, bs.edge, bs.edge + 127, tid, NT);
          __syncthreads();
          if (tid < kSymbolSz - 1) met[0] = bs.edge[tid];
          if (tid >= kTailTid0 && tid < kTailTid0 + 126) met[NL - 1] = bs.edge[kSymbolSz - 1 + tid - kTailTid0];
        }
        if (profp) prof[6] += __builtin_amdgcn_s_memtime() - t0_; // metric slice (+ edges)
        unsigned long long t1_ = profp ? __builtin_amdgcn_s_memtime() : 0;
        AVG = AVG_ + buf * kAvgLen;
        bulk_search_ema<NT, NL>(met, ema, AVG, bs.tmax[buf], bs.tidx[buf]);
        // prefetch the next search's inputs: a miss leads to pos + 9600; a tracked chain that keeps its
        // timing searches again track_every + 1 half-frames on
        const int ahead = is_tracking ? a.track_every + 1 : 1;
        pf_pos = pos + (int64_t)kHalfFrame * ahead;
        if (pf_pos + kHalfFrame <= n) {
          pf_edges = grid_k >= 0;
          load_slice(pfm, pf_pos, pf_edges ? grid_k + ahead : -1);
          if (!pf_edges) {
            if (tid < 128) pfe = x[pf_pos + kHalfFrame - 128 + tid];
            else if (tid < 256) pfe = x[pf_pos - 128 + tid];
          }
        } else {
          pf_pos = -1;
        }
        if (profp) prof[1] += __builtin_amdgcn_s_memtime() - t1_; // EMA + mirror + prefetch issue
        r = bulk_search_tail<NT, NL>(ema, AVG, bs.tmax[buf], bs.tidx[buf], bs.sl, profp);
        buf ^= 1;
      } else {
        if (!(a.ablate & 1)) {
          if (pf_pos == pos) { // prefetched under the previous search's tail
#pragma unroll
            for (int i = 0; i < (kBulk ? 1 : NL); ++i) {
              const int k = tid + i * NT;
              if (k < kHalfFrame) W[kWPad + k] = pf[i];
            }
          } else {
            for (int i = tid; i < kHalfFrame; i += NT) W[kWPad + i] = x[pos + i];
          }
          // the zero pads around the window (the matched filter used W as scratch last time)
          for (int i = tid; i < kWPad; i += NT) W[i] = v2f{0.f, 0.f};
          for (int i = kWPad + kHalfFrame + tid; i < kWLen; i += NT) W[i] = v2f{0.f, 0.f};
        }
        __syncthreads();
        if (profp) prof[6] += __builtin_amdgcn_s_memtime() - t0_; // window load
        search_front(W, AVG, ema, taps, fl, sc, a.ablate, profp);
        // W is free from here on: start fetching the window a miss leads to (pos + 9600); the loads
        // fly under the tail reductions and the state machine, a realigning hit discards them.
        pf_pos = -1;
        if (!(a.ablate & 1) && pos + 2 * kHalfFrame <= n) {
          pf_pos = pos + kHalfFrame;
#pragma unroll
          for (int i = 0; i < (kBulk ? 1 : NL); ++i) {
            const int k = tid + i * NT;
            if (k < kHalfFrame) pf[i] = x[pf_pos + k];
          }
        }
        r = search_tail(AVG, sc, a.ablate, profp);
      }
      if (profp) prof[7] += 1;
      peak_pos = r.peak_pos;
      psr = r.psr;
      peak_value = r.peak_value;
      if (tid == 0) psr_data[psr_i % kMovingAvgSz] = psr;
      psr_i++;
      flags |= LT_F_SEARCHED;
    } else {
      timer--;
    }
    const int used_peak_pos = peak_pos;

    // ---- score, lib/pss_impl.cc:174-182 (incr_score :111-127, reset_score :129-152)
    const bool over = psr > a.psr_threshold;
    if (over) {
      if (!(is_tracking && score == a.track_after)) {
        score++;
        if (!is_tracking && score == a.track_after) {
          is_tracking = 1;
#pragma unroll
          for (int t = 0; t < NL; ++t) ema[t] = 0.f; // srslte_pss_reset
        }
      }
      flags |= LT_F_OVER;
    } else if (score != 0) {
      score = 0;
      timer = 0;
      is_tracking = 0;
#pragma unroll
      for (int t = 0; t < NL; ++t) ema[t] = 0.f;
      psr_i = 0;
      cfo_i = 0;
      tracking_lost = 1;
    }
    if (psr > psr_max) psr_max = psr;

    int64_t out_pos = -1, next_pos;
    float cfo = 0.f, mean_cfo = 0.f, m0_value = 0.f, m1_value = 0.f;
    int m0 = -1, m1 = -1, n_id_1 = -1, cell_id = -1;

    if (over || tracking_lost) { // ---- lib/pss_impl.cc:184-214
      const int frame_start = peak_pos - kSlotLen;
      peak_pos = kSlotLen;
      const int64_t hf0 = pos + frame_start;
      if (hf0 + kHalfFrame > n) break; // quirk q1 normalised: never read past the capture
      next_pos = hf0 + kHalfFrame;
      out_pos = hf0;
      flags |= LT_F_EMITTED;
      v2f *out = (a.out_hf && n_emits < a.out_cap)
                     ? a.out_hf + ((int64_t)chain * a.out_cap + n_emits) * kHalfFrame
                     : nullptr;

      if (is_tracking) {
        if constexpr (kBulk) {
          // replay mode: CFO estimate on the uncorrected PSS symbol (lib/pss_impl.cc:199-201), ring update and
          // mean_cfo() -- the chain's only emit-side state.  Rotation of the half-frame, CP decision and SSS
          // feed nothing in this kernel: lt_cp_kernel / lt_cp_scan_kernel / lt_sss_kernel run them afterwards
          // from out_pos and mean_cfo of the record.
          if (tid < kSymbolSz) RAW[tid] = ld_iq(x, n, hf0 + (kSlotLen - kSymbolSz) + tid);
          __syncthreads();
          if (tid < 64) {
            const float c = cfo_estimate_wave(ltab.taps, RAW);
            const uint32_t slot = cfo_i % kMovingAvgSz;
            const float mc = moving_avg_wave(cfo_data, cfo_i + 1, slot, c);
            if (tid == 0) {
              cfo_data[slot] = c;
              sc.cfo = c;
              sc.mean_cfo = mc;
            }
          }
          __syncthreads();
          cfo_i++;
          cfo = sc.cfo;
          mean_cfo = sc.mean_cfo;
          if (out) {
            const float freq = -mean_cfo / (float)kSymbolSz; // srslte_cfo_correct, lib/pss_impl.cc:204
            const float phase_inc = __fmul_rn(freq, (float)kCexpTabSize);
            for (int i = tid; i < kHalfFrame; i += NT) out[i] = cfo_rotate(tab->cexptab, phase_inc, i, ld_iq(x, n, hf0 + i));
          }
          flags |= LT_F_SSS_RAN;
        } else {
          unsigned long long tt_ = profp ? __builtin_amdgcn_s_memtime() : 0;
#define LT_TSTAMP(i)                                             \
    if (profp) {                                                   \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
      prof[i] += t_ - tt_;                                         \
      tt_ = t_;                                                    \
    }
          // stage half-frame samples [480, 960) once (coalesced); everything below reads LDS
          // (by waves 1..7: wave 0 arrives last from the previous emission's m0/m1 search and finds
          // the samples already in LDS)
          for (int i = tid - 64; i >= 0 && i < kSssLen; i += NT - 64) RAW[i] = ld_iq(x, n, hf0 + kSssBase + i);
          __syncthreads();
          LT_TSTAMP(8) // stage raw samples
          // srslte_pss_cfo_compute on the uncorrected PSS symbol (lib/pss_impl.cc:199-201), ring
          // update and mean_cfo(): every wave computes them redundantly, wave 0 keeps the ring
          if (tid < 64) { // wave 0 alone: the other waves wait at the barrier instead of competing for issue slots
            const float c = cfo_estimate_wave(ltab.taps, RAW + (kSlotLen - kSymbolSz - kSssBase));
            LT_TSTAMP(9) // cfo estimate (wave 0)
            const uint32_t slot = cfo_i % kMovingAvgSz;
            const float mc = (kDiag && (a.ablate & 32))
                                 ? c
                                 : moving_avg_wave(cfo_data, cfo_i + 1, slot, c, profp ? &prof[22] : nullptr);
            LT_TSTAMP(21) // ring mean (wave 0)
            if (tid == 0) {
              cfo_data[slot] = c;
              sc.cfo = c;
              sc.mean_cfo = mc;
            }
          }
          __syncthreads();
          cfo_i++;
          cfo = sc.cfo;
          mean_cfo = sc.mean_cfo;
          LT_TSTAMP(10) // cfo estimate + ring mean
          // srslte_cfo_correct(out, out, -mean_cfo()/symbol_sz), lib/pss_impl.cc:204
          const float freq = -mean_cfo / (float)kSymbolSz;
          const float phase_inc = __fmul_rn(freq, (float)kCexpTabSize);
          for (int i = tid; i < kSssLen; i += NT) S[i] = cfo_rotate(tab->cexptab, phase_inc, kSssBase + i, RAW[i]);
          if (out)
            for (int i = tid; i < kHalfFrame; i += NT)
              out[i] = cfo_rotate(tab->cexptab, phase_inc, i, ld_iq(x, n, hf0 + i));
          __syncthreads();
          LT_TSTAMP(11) // rotation
          // srslte_pss_chest (lib/pss_impl.cc:206-209): its output is never consumed -- dropped.

          // ---- sss_impl::work, no tracking_lost tag on this half-frame (lib/sss_impl.cc:102-150)
          // (wave 0 only: m0, m1, n_id_1 and the CP averages feed nothing but the record and the
          // state thread 0 stores; the other waves run ahead to the next barrier)
          SssResult s;
          s.cp = LT_CP_NORM; s.m0 = -1; s.m1 = -1; s.n_id_1 = -1; s.m0_value = s.m1_value = 0.f;
          if (tid < 64 && !(kDiag && (a.ablate & 64))) {
            if (kBulk) s.cp = cp_detect_wave(S, m_norm_avg, m_ext_avg); // the SSS itself: lt_sss_kernel, afterwards
            else s = sss_stage_wave(S, ltab, tab, m_norm_avg, m_ext_avg, sc);
          }
          LT_TSTAMP(12) // sss stage
          if (profp) prof[13] += 1;
          flags |= LT_F_SSS_RAN;
          if (s.cp == LT_CP_EXT) flags |= LT_F_CP_EXT;
          m0 = s.m0; m1 = s.m1; m0_value = s.m0_value; m1_value = s.m1_value;
          n_id_1 = s.n_id_1;
          if (n_id_1 >= 0) { // (m0, m1, n_id_1 are valid in wave 0, whose thread 0 writes the record)
            cell_id = 3 * n_id_1 + n_id_2; // srslte_sync_get_cell_id
            flags |= LT_F_TAG_CELL;
          }
        }
      } else {
        // tracking lost: tag the half-frame (lib/pss_impl.cc:210-214); sss resets on the tag
        // (lib/sss_impl.cc:91-100)
        flags |= LT_F_TAG_LOST;
        tracking_lost = 0;
        m_norm_avg = 0.f;
        m_ext_avg = 0.f;
        if (out)
          for (int i = tid; i < kHalfFrame; i += NT) out[i] = ld_iq(x, n, hf0 + i);
      }
      n_emits++;
    } else {
      next_pos = pos + kHalfFrame; // lib/pss_impl.cc:217
    }

    if (is_tracking) flags |= LT_F_TRACKING;
    if (tid == 0) {
      lt_record r;
      r.in_pos = pos; r.out_pos = out_pos; r.channel = channel; r.n_id_2 = n_id_2;
      r.call_idx = call_base + nrec; r.flags = flags; r.peak_pos = used_peak_pos; r.score = score;
      r.psr = psr; r.peak_value = peak_value; r.cfo = cfo; r.mean_cfo = mean_cfo;
      r.m_norm_avg = m_norm_avg; r.m_ext_avg = m_ext_avg; r.m0_value = m0_value;
      r.m1_value = m1_value; r.m0 = m0; r.m1 = m1; r.n_id_1 = n_id_1; r.cell_id = cell_id;
      r.psr_max = psr_max; r.timer = timer;
      recs[nrec] = r;
    }
    nrec++;
    if (kBulk) grid_k = (grid_k >= 0 && next_pos == pos + kHalfFrame) ? grid_k + 1 : -1;
    pos = next_pos;
  }

  if (profp && tid == 0)
    for (int i = 0; i < 23; ++i) a.prof[(size_t)chain * 24 + i] = prof[i];
  if (elapsed_out && tid == 0) *elapsed_out = __builtin_amdgcn_s_memtime() - clk0;
  // ---- store state
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NL; ++t) st->avg[0][kBulk ? tid + t * NT : tid * NL + t] = ema[t];
  for (int i = tid; i < kMovingAvgSz; i += NT) {
    st->psr_data[i] = psr_data[i];
    st->cfo_data[i] = cfo_data[i];
  }
  if (tid == 0) {
    st->pos = pos;
    st->is_tracking = is_tracking; st->score = score; st->timer = timer;
    st->tracking_lost = tracking_lost; st->peak_pos = peak_pos;
    st->psr_i = psr_i; st->cfo_i = cfo_i;
    st->n_calls += nrec; st->n_emits += n_emits;
    st->psr = psr; st->psr_max = psr_max;
    if (!kBulk) { // (replay mode: lt_cp_scan_kernel owns the CP averages)
      st->m_norm_avg = m_norm_avg;
      st->m_ext_avg = m_ext_avg;
    }
    a.rec_count[chain] = nrec;
  }
}

// ---------------------------------------------------------------- single-call kernels

struct FindResult {
  int peak_pos;
  float psr;
  float peak_value;
};

// srslte_pss_find_pss for one handle: window = 9600 device samples, avg = 9728-float EMA state.
__global__ __launch_bounds__(kThreads) void lt_find_kernel(const v2f *__restrict__ win, float *avg,
                                                           const DeviceTables *__restrict__ tab,
                                                           int n_id_2, FindResult *res)
{
  __shared__ v2f W[kWLen];
  __shared__ float AVG[kAvgLen];
  __shared__ v2f TW[fft::kN];
  __shared__ v2f SPEC[fft::kN];
  __shared__ Scratch sc;
  const int tid = threadIdx.x;
  float ema[kT];
#pragma unroll
  for (int t = 0; t < kT; ++t) ema[t] = avg[tid * kT + t];
  for (int i = tid; i < kWPad; i += kThreads) W[i] = v2f{0.f, 0.f};
  for (int i = kWPad + kHalfFrame + tid; i < kWLen; i += kThreads) W[i] = v2f{0.f, 0.f};
  if (tid < kAvgLen - kConvLen) AVG[kConvLen + tid] = 0.f;
  for (int i = tid; i < kHalfFrame; i += kThreads) W[kWPad + i] = win[i];
  load_filter_lds(TW, SPEC, tab, n_id_2);
  const FilterLds fl = {TW, SPEC};
  __syncthreads();
  const v2f *taps = reinterpret_cast<const v2f *>(tab->pss_taps[n_id_2]);
  const SearchResult r = search_window(W, AVG, ema, taps, fl, sc, 0);
#pragma unroll
  for (int t = 0; t < kT; ++t) avg[tid * kT + t] = ema[t];
  if (tid == 0) {
    res->peak_pos = r.peak_pos;
    res->psr = r.psr;
    res->peak_value = r.peak_value;
  }
}

__global__ void lt_cfo_compute_kernel(const v2f *__restrict__ sym, const DeviceTables *__restrict__ tab,
                                      int n_id_2, float *cfo)
{
  const v2f *taps = reinterpret_cast<const v2f *>(tab->pss_taps[n_id_2]);
  const float c = cfo_estimate_wave(taps, sym); // one wave
  if (threadIdx.x == 0) *cfo = c;
}

__global__ void lt_cfo_correct_kernel(const v2f *__restrict__ in, v2f *__restrict__ out, int n,
                                      float freq, const DeviceTables *__restrict__ tab)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float phase_inc = __fmul_rn(freq, (float)kCexpTabSize);
  out[i] = cfo_rotate(tab->cexptab, phase_inc, i, in[i]);
}

struct SyncState {
  float m_norm_avg, m_ext_avg;
};

// srslte_sync_detect_cp; in = first peak_pos samples, one wave.  The reference only ever passes
// peak_pos = slot_length; three symbols per hypothesis are available for every peak_pos >= 480.
__global__ void lt_detect_cp_kernel(const v2f *__restrict__ in, int peak_pos, SyncState *st, int *cp_out)
{
  if (peak_pos < 480) { // fewer than three extended-CP symbols: outside what the path uses
    if (threadIdx.x == 0) *cp_out = LT_CP_NORM;
    return;
  }
  float mn = st->m_norm_avg, me = st->m_ext_avg;
  const int cp = cp_detect_wave(in + (peak_pos - 480), mn, me);
  if (threadIdx.x == 0) {
    *cp_out = cp;
    st->m_norm_avg = mn;
    st->m_ext_avg = me;
  }
}

struct M0M1Result {
  int m0, m1;
  float m0_value, m1_value;
};

__global__ __launch_bounds__(128) void lt_sss_m0m1_kernel(const v2f *__restrict__ sym,
                                                          const DeviceTables *__restrict__ tab, int n_id_2,
                                                          M0M1Result *res)
{
  __shared__ Scratch sc;
  __shared__ LdsTables ltab;
  __shared__ v2f S[kSymbolSz];
  const int tid = threadIdx.x;
  load_lds_tables(ltab, tab, n_id_2);
  if (tid < kSymbolSz) S[tid] = sym[tid];
  __syncthreads();
  SssResult r;
  if (tid < 64) {
    sss_fft128_wave(S, ltab.tw128, sc.Y);
    sss_m0m1_wave(sc, ltab, r);
  }
  if (tid == 0) {
    res->m0 = r.m0; res->m1 = r.m1; res->m0_value = r.m0_value; res->m1_value = r.m1_value;
  }
}

// ---------------------------------------------------------------- decimating FIR
// y[c][k] = sum_j taps[j] * x[c][k*D - j].  One thread per output sample; a block stages the
// (64*D + ntaps) input samples it needs in LDS with coalesced loads.
constexpr int kDecimThreads = 256;
constexpr int kDecimOutPerThread = 2;
constexpr int kDecimOutPerBlock = kDecimThreads * kDecimOutPerThread;

// Row length (in samples) of the polyphase staging area: >= cols, and such that the D rows a 16-lane
// group of the staging store touches start on distinct even banks (R = 16/D mod 32 for D | 16, else odd).
__host__ __device__ inline int decim_row_len(int cols, int D)
{
  const int want = (D <= 16 && (16 % D) == 0) ? 16 / D : 1;
  int R = cols;
  while ((R & 31) != (want & 31)) ++R;
  return R;
}

// y[c][k] = sum_j taps[j] x[c][k D - j], accumulated in tap order with an unfused multiply-add (the
// oracle's and GNU Radio's generic kernel order).  The block's input span is staged in LDS phase by
// phase (sample i at row i mod D, column i / D): for a given tap all lanes then read one row at
// consecutive columns -- conflict-free, where the natural layout would put every D-th sample of a wave
// on the same banks.  Two outputs per thread run as independent add chains.
// kD = decimation known at compile time (4, 8, 16: the reference's rates), or 0: read it from the argument.
template <int kD>
__global__ __launch_bounds__(kDecimThreads) void lt_decim_kernel(const v2f *__restrict__ x, v2f *__restrict__ y,
                                                                 int64_t n_in, int64_t n_out, int D_arg,
                                                                 const float *__restrict__ taps, int ntaps, int R)
{
  extern __shared__ v2f xs[]; // D rows of R samples
  const int D = kD ? kD : D_arg;
  const int c = blockIdx.y, tid = threadIdx.x;
  const int64_t k0 = (int64_t)blockIdx.x * kDecimOutPerBlock;
  const v2f *xc = x + (int64_t)c * n_in;
  // oldest input sample needed, moved down to a multiple of D so that sample `first + i` has phase i mod D
  const int lead = ((ntaps - 1 + D - 1) / D) * D; // >= ntaps - 1
  const int64_t first = k0 * D - lead;
  const int span = kDecimOutPerBlock * D + lead;
  for (int i = tid; i < span; i += kDecimThreads) xs[(i % D) * R + i / D] = ld_iq(xc, n_in, first + i);
  __syncthreads();
  // newest sample of output k0 + o sits at i = lead + o D: row 0, column lead / D + o
  float re[kDecimOutPerThread], im[kDecimOutPerThread];
#pragma unroll
  for (int o = 0; o < kDecimOutPerThread; ++o) re[o] = im[o] = 0.f;
  const v2f *base = xs + lead / D + tid;
  // tap j = q D + r reads sample i = lead + o D - j: row (-j) mod D = (D - r) mod D, column lead / D + o - q - (r > 0)
  auto tap = [&](int j, const v2f *p) { // same order and rounding as the oracle: product, then add
    const float t = taps[j];
#pragma unroll
    for (int o = 0; o < kDecimOutPerThread; ++o) {
      const v2f v = p[o * kDecimThreads];
      re[o] = __fadd_rn(re[o], __fmul_rn(t, v.x));
      im[o] = __fadd_rn(im[o], __fmul_rn(t, v.y));
    }
  };
  if (kD) {
    const int nq = ntaps / kD; // whole groups of D taps
    for (int q = 0; q < nq; ++q) {
      const v2f *pq = base - q;
#pragma unroll
      for (int r = 0; r < kD; ++r) tap(q * kD + r, pq + ((kD - r) % kD) * R - (r > 0));
    }
    const v2f *pq = base - nq;
#pragma unroll
    for (int r = 0; r < kD; ++r)
      if (nq * kD + r < ntaps) tap(nq * kD + r, pq + ((kD - r) % kD) * R - (r > 0));
  } else {
    int row = 0, col = 0;
    for (int j = 0; j < ntaps; ++j) {
      tap(j, base + row * R + col);
      if (--row < 0) {
        row += D;
        --col;
      }
    }
  }
#pragma unroll
  for (int o = 0; o < kDecimOutPerThread; ++o) {
    const int64_t k = k0 + tid + o * kDecimThreads;
    if (k < n_out) y[(int64_t)c * n_out + k] = v2f{re[o], im[o]};
  }
}

// ---------------------------------------------------------------- bulk lag metric (bulk + replay mode)
// P[root][channel][p] = | sum_m h_root[m] x[channel][p - m] |^2 for every sample position p (x[<0] = 0):
// the position-independent part of srslte_pss_find_pss's convolution, for all three roots from ONE
// forward FFT per 1024-sample overlap-save block (897 new positions per block).  One wave per block
// (csrc/lt_fft.h, full-wave form: 16 values per lane), one persistent workgroup of 12 waves per CU; inputs come straight
// from HBM (16 x 512-byte rows per block), outputs go back as 3 x 897 floats.  Algorithmic bytes: 8 B read + 12 B written per sample.
struct MetricArgs {
  const v2f *iq;   // [n_channels][ch_stride]
  int64_t n_samples;
  int64_t ch_stride;
  int n_channels;
  int blocks_per_channel; // ceil(n_samples / 897)
  float *metric;   // [3][n_channels][metric_ch_stride]
  int64_t metric_ch_stride;
  int64_t metric_root_stride;
  const DeviceTables *tab;
};

#define LT_SB __builtin_amdgcn_sched_barrier(0)
constexpr int kMetricWaves = 12;                  // one 1024-sample block per wave at a time, three waves per SIMD
constexpr int kMetricThreads = 64 * kMetricWaves; // ONE workgroup per CU: the tables below are shared by all 12 waves

// kAbl: diagnostics only (tools/probe.py --ablate 512 / 1024): 1 = no stores, 2 = no loads
template <int kAbl> __global__ __launch_bounds__(kMetricThreads) void lt_metric_kernel(MetricArgs a)
{
  __shared__ v2f E[kMetricWaves * fft::kScratchW]; // one exchange area per wave (12 x 8.5 KB)
  __shared__ fft::c32x2 TW1[16 * 64];              // 16 KB
  __shared__ fft::c32x2 TW2[16 * 4];               // 1 KB
  __shared__ v2f SPEC[3 * fft::kN];                // 24 KB: the three filter spectra / 1024
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 16 * 64; i += kMetricThreads) TW1[i] = *reinterpret_cast<const fft::c32x2 *>(a.tab->fft_tw1[i]);
  if (tid < 64) TW2[tid] = *reinterpret_cast<const fft::c32x2 *>(a.tab->fft_tw2[tid]);
  for (int i = tid; i < 3 * fft::kN; i += kMetricThreads) {
    const cf32 h = a.tab->pss_spec[i >> 10][i & 1023];
    SPEC[i] = v2f{h.re, h.im};
  }
  __syncthreads(); // the only barrier: from here on every wave works through its own blocks alone
  const uint32_t total = (uint32_t)a.n_channels * (uint32_t)a.blocks_per_channel;
  const uint32_t stride = gridDim.x * kMetricWaves;
  uint32_t g = blockIdx.x * kMetricWaves + w; // block index: channel-major, consecutive waves = consecutive blocks
  if (g >= total) return;
  fft::c32 *e = E + w * fft::kScratchW;
  fft::c32 xf[16], v[16];

  // input samples base .. base+1023 of block g (first 127 = overlap with the previous block)
  auto load_block = [&](uint32_t gb) {
    const uint32_t ch = gb / (uint32_t)a.blocks_per_channel;
    const uint32_t jb = gb - ch * (uint32_t)a.blocks_per_channel;
    const v2f *__restrict__ x = a.iq + (int64_t)ch * a.ch_stride;
    const int64_t base = (int64_t)jb * kBlkValid - (kSymbolSz - 1);
    if (kAbl & 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xf[r] = fft::c32{(float)(l + r), (float)gb};
    } else if (base >= 0 && base + fft::kN <= a.n_samples) { // interior block: no bounds checks
#pragma unroll
      for (int r = 0; r < 16; ++r) xf[r] = __builtin_nontemporal_load(&x[base + l + 64 * r]);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) xf[r] = ld_iq(x, a.n_samples, base + l + 64 * r);
    }
  };
  // |v|^2 of the 897 valid positions of block gb, root `root`
  auto store_block = [&](uint32_t gb, int root) {
    const uint32_t ch = gb / (uint32_t)a.blocks_per_channel;
    const uint32_t jb = gb - ch * (uint32_t)a.blocks_per_channel;
    const int64_t base = (int64_t)jb * kBlkValid - (kSymbolSz - 1);
    float *__restrict__ out = a.metric + root * a.metric_root_stride + (int64_t)ch * a.metric_ch_stride + base;
    if (kAbl & 1) {
      float s = 0.f;
#pragma unroll
      for (int r = 1; r < 16; ++r) s += v[r].x + v[r].y;
      if (s == 1234.5f) out[l + 64] = s;
    } else if (base + fft::kN <= a.n_samples) {
      if (l == 63) __builtin_nontemporal_store(__fadd_rn(__fmul_rn(v[1].x, v[1].x), __fmul_rn(v[1].y, v[1].y)), &out[127]); // reg 1 = positions 64..127
#pragma unroll
      for (int r = 2; r < 16; ++r) __builtin_nontemporal_store(__fadd_rn(__fmul_rn(v[r].x, v[r].x), __fmul_rn(v[r].y, v[r].y)), &out[l + 64 * r]);
    } else {
#pragma unroll
      for (int r = 1; r < 16; ++r) {
        const int i = l + 64 * r;
        if (i >= kSymbolSz - 1 && base + i < a.n_samples)
          out[i] = __fadd_rn(__fmul_rn(v[r].x, v[r].x), __fmul_rn(v[r].y, v[r].y));
      }
    }
  };
  auto inverse = [&]() {
    LT_SB;
    fft::fft1024w_stageA<true>(l, v, TW1, e);
    LT_SB;
    fft::fft1024w_stageB<true>(l, v, TW2, e, e);
    LT_SB;
    fft::fft1024w_stageC<true>(l, v, e);
    LT_SB;
  };

  load_block(g);
  for (;;) {
    fft::fft1024w_stageA<false>(l, xf, TW1, e);
    LT_SB;
    fft::fft1024w_stageB<false>(l, xf, TW2, e, e);
    LT_SB;
    fft::fft1024w_stageC<false>(l, xf, e);
    LT_SB;
#pragma unroll
    for (int root = 0; root < 2; ++root) {
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = fft::cmul<false>(xf[r], SPEC[root * fft::kN + r * 64 + l]);
      inverse();
      store_block(g, root);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = fft::cmul<false>(xf[r], SPEC[2 * fft::kN + r * 64 + l]);
    LT_SB;
    // the spectrum of this block is no longer needed: fetch the next block under the last transform
    const uint32_t g_next = g + stride;
    const bool more = g_next < total;
    if (more) load_block(g_next);
    inverse();
    store_block(g, 2);
    if (!more) break;
    g = g_next;
  }
}

// ---------------------------------------------------------------- CP length of the tracked half-frames (bulk + replay mode)
// srslte_sync_detect_cp (lib/sss_impl.cc:104-108) in two steps.  Its correlations (per half-frame,
// independent) run one wave per record on the half-frame rotated exactly as srslte_cfo_correct did; its
// two exponential averages and the decision are a short scan per chain in call order, which also
// resets them where the reference's sss block sees the tracking_lost tag (lib/sss_impl.cc:91-100) and
// writes the averages into every record.  Neither feeds the PSS state machine of the replay kernel.
struct CpArgs {
  const v2f *iq;
  int64_t n_samples;
  int64_t ch_stride;
  lt_record *recs;      // [n_chains][rec_cap]
  int rec_cap;
  const int *rec_count; // [n_chains]
  CpMetrics *metrics;   // [n_chains][rec_cap] scratch
  ChainState *state;    // [n_chains]
  const DeviceTables *tab;
  int n_chains;
};

__global__ __launch_bounds__(256) void lt_cp_kernel(CpArgs a)
{
  __shared__ v2f S[4][kSssLen];
  const int chain = blockIdx.x, channel = chain / 3; // (chains along x: up to 2^31 - 1 of them)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = blockIdx.y * 4 + w;
  if (idx >= a.rec_count[chain]) return;
  const lt_record *rec = a.recs + (int64_t)chain * a.rec_cap + idx;
  if (!(rec->flags & LT_F_SSS_RAN)) return;
  const int64_t hf0 = rec->out_pos;
  const v2f *__restrict__ x = a.iq + (int64_t)channel * a.ch_stride;
  const float freq = -rec->mean_cfo / (float)kSymbolSz; // lib/pss_impl.cc:204
  const float phase_inc = __fmul_rn(freq, (float)kCexpTabSize);
  for (int i = lane; i < kSssLen; i += 64)
    S[w][i] = cfo_rotate(a.tab->cexptab, phase_inc, kSssBase + i, ld_iq(x, a.n_samples, hf0 + kSssBase + i));
  const CpMetrics cm = cp_metrics_wave(S[w]);
  if (lane == 0) a.metrics[(int64_t)chain * a.rec_cap + idx] = cm;
}

// One wave per chain: 64 records at a time are fetched one per lane and everything that does not depend on
// the running averages is formed in parallel; the wave then walks only the records that change the
// averages (tracked half-frames, tracking_lost tags) in call order with wave-uniform arithmetic, every lane
// keeping the state as of its own record; decisions and stores are per lane again.
__global__ __launch_bounds__(64) void lt_cp_scan_kernel(CpArgs a)
{
  const int chain = blockIdx.x, lane = threadIdx.x;
  ChainState *st = a.state + chain;
  float m_norm_avg = st->m_norm_avg, m_ext_avg = st->m_ext_avg;
  lt_record *recs = a.recs + (int64_t)chain * a.rec_cap;
  const CpMetrics *cm = a.metrics + (int64_t)chain * a.rec_cap;
  const int count = a.rec_count[chain];
  for (int base = 0; base < count; base += 64) {
    const int i = base + lane;
    const bool valid = i < count;
    const uint32_t flags = valid ? recs[i].flags : 0u;
    const bool ran = (flags & LT_F_SSS_RAN) != 0;
    CpMetrics c;
    c.m[0] = c.m[1] = c.r[0] = c.r[1] = 0.f;
    if (ran) c = cm[i];
    const float in_norm = cp_avg_input(c.m[0]), in_ext = cp_avg_input(c.m[1]);
    const unsigned long long run = __ballot(ran), lost = __ballot((flags & LT_F_TAG_LOST) != 0);
    float my_norm = m_norm_avg, my_ext = m_ext_avg; // state as of this lane's record
    unsigned long long events = run | lost;
    while (events) {
      const int j = __builtin_ctzll(events);
      events &= events - 1;
      if ((lost >> j) & 1) { // the sss block resets on the tracking_lost tag (lib/sss_impl.cc:91-100)
        m_norm_avg = 0.f;
        m_ext_avg = 0.f;
      }
      if ((run >> j) & 1) {
        m_norm_avg = cp_avg_step(rlane(in_norm, j), m_norm_avg);
        m_ext_avg = cp_avg_step(rlane(in_ext, j), m_ext_avg);
      }
      if (lane >= j) {
        my_norm = m_norm_avg;
        my_ext = m_ext_avg;
      }
    }
    if (valid) {
      recs[i].m_norm_avg = my_norm;
      recs[i].m_ext_avg = my_ext;
      if (ran && cp_choose(my_norm, my_ext, c.r[0], c.r[1]) == LT_CP_EXT) recs[i].flags = flags | LT_F_CP_EXT;
    }
  }
  if (lane == 0) {
    st->m_norm_avg = m_norm_avg;
    st->m_ext_avg = m_ext_avg;
  }
}

// ---------------------------------------------------------------- SSS of the tracked half-frames (bulk + replay mode)
// srslte_sss_m0m1_partial + srslte_sss_N_id_1 (lib/sss_impl.cc:110-124) of every record whose half-frame
// went through the SSS stage.  The replay kernel has done the sequential part (CFO ring mean, CP
// averages and decision) and left mean_cfo and the CP flag in the record; m0 / m1 / N_id_1 / cell_id feed
// nothing but the record, so they are computed here for all chains at once, one wave per record: rotate
// the SSS symbol exactly as srslte_cfo_correct did (same table index, same products), FFT-128,
// 31 + 31 correlations, table look-up -- the same device functions the fused kernel runs in line.
struct SssArgs {
  const v2f *iq;
  int64_t n_samples;
  int64_t ch_stride;
  lt_record *recs;  // [n_chains][rec_cap]
  int rec_cap;
  const int *rec_count; // [n_chains]
  const DeviceTables *tab;
};

__global__ __launch_bounds__(256) void lt_sss_kernel(SssArgs a)
{
  __shared__ LdsTables ltab;
  __shared__ v2f SYM[4][kSymbolSz];
  __shared__ v2f Y[4][62];
  const int chain = blockIdx.x, n_id_2 = chain % 3, channel = chain / 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int idx = blockIdx.y * 4 + w;
  const int count = a.rec_count[chain];
  if ((int)blockIdx.y * 4 >= count) return; // whole block beyond this chain's records
  load_lds_tables(ltab, a.tab, n_id_2);
  __syncthreads();
  if (idx >= count) return;
  lt_record *rec = a.recs + (int64_t)chain * a.rec_cap + idx;
  const uint32_t flags = rec->flags;
  if (!(flags & LT_F_SSS_RAN)) return;
  const int64_t hf0 = rec->out_pos;
  const float mean_cfo = rec->mean_cfo;
  const v2f *__restrict__ x = a.iq + (int64_t)channel * a.ch_stride;
  const float freq = -mean_cfo / (float)kSymbolSz; // lib/pss_impl.cc:204
  const float phase_inc = __fmul_rn(freq, (float)kCexpTabSize);
  const int cp_len = (flags & LT_F_CP_EXT) ? 32 : 9;              // srslte_sync_set_cp
  const int sss_idx = kSlotLen - 2 * kSymbolSz - cp_len;           // lib/sss_impl.cc:110
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = sss_idx + lane + 64 * k;
    SYM[w][lane + 64 * k] = cfo_rotate(a.tab->cexptab, phase_inc, i, ld_iq(x, a.n_samples, hf0 + i));
  }
  SssResult r;
  r.cp = 0;
  sss_fft128_wave(SYM[w], ltab.tw128, Y[w]);
  sss_m0m1_wave(Y[w], ltab, r);
  r.n_id_1 = sss_n_id_1(a.tab, (uint32_t)r.m0, (uint32_t)r.m1);
  if (lane == 0) {
    rec->m0 = r.m0;
    rec->m1 = r.m1;
    rec->m0_value = r.m0_value;
    rec->m1_value = r.m1_value;
    rec->n_id_1 = r.n_id_1;
    if (r.n_id_1 >= 0) {
      rec->cell_id = 3 * r.n_id_1 + n_id_2; // srslte_sync_get_cell_id
      rec->flags = flags | LT_F_TAG_CELL;
    }
  }
}

// ---------------------------------------------------------------- window-edge lags on the 9600-sample grid
// A window's first 127 and last 126 lags see its zero padding, so they are not in P.  A chain that has
// not re-aligned yet (every chain that never fires, i.e. almost all of them) searches the windows
// [9600 k, 9600 k + 9600) of the capture: their edge lags are computed here for all three roots, chip
// wide, and the replay kernel picks them up with its metric prefetch (chains off the grid compute their
// own, device function edge_lags).
//   edge[root][channel][k][0..126]   = | sum_{m<=j} h[m] x[9600 k + j - m] |^2          (lag j)
//   edge[root][channel][k][127+j]    = | sum_{m>j}  h[m] x[9600 k + 9600 + j - m] |^2   (lag 9600 + j, j < 126)
// One lane per (channel, window), 64 consecutive ones per wave; a wave takes a tile of 8 units j0..j0+7
// (j0 = 8 * tile) through all 128 taps in 16 steps of 8 taps.  With j0 wave-uniform the split of a step's
// products into head and tail sums is known at compile time (steps before tap j0: all head; the step
// at tap j0: a fixed staircase; steps after it: all tail), so the work is the exact triangle: no
// predicates, no zero padding.  Taps are wave-uniform (scalar loads), samples come per lane from L1/L2.
struct EdgeArgs {
  const v2f *iq;
  int64_t ch_stride;
  int n_channels;
  int n_windows;       // windows per channel on the grid: floor(n_samples / 9600)
  float *edge;         // [3][n_channels][edge_win_stride / 256][256]
  int64_t edge_ch_stride;   // floats between channels
  int64_t edge_root_stride; // floats between roots
  const DeviceTables *tab;
};

// kMode 0: all 64 products of the step go to the head sums, 2: to the tail sums, 1: tap j0 + d feeds
// the head sum of unit u when u >= d and its tail sum otherwise.
template <int kMode>
__device__ __forceinline__ void edge_step(const v2f *__restrict__ zp, const cf32 *__restrict__ h0, const cf32 *__restrict__ h1,
                                          const cf32 *__restrict__ h2, int m0, float (&H)[3][8][2], float (&T)[3][8][2])
{
  // samples zs[i] = Z[z0 - 7 + i], i = 0..14, where unit u / tap m0 + t reads Z[z0 + u - t]
  v2f zs[15];
#pragma unroll
  for (int i = 0; i < 15; ++i) zs[i] = zp[i - 7];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const cf32 hh[3] = {h0[m0 + t], h1[m0 + t], h2[m0 + t]};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const v2f z = zs[u - t + 7];
      const bool head = kMode == 0 || (kMode == 1 && u >= t);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float &re = head ? H[r][u][0] : T[r][u][0];
        float &im = head ? H[r][u][1] : T[r][u][1];
        re = fmaf(hh[r].re, z.x, re);
        re = fmaf(-hh[r].im, z.y, re);
        im = fmaf(hh[r].re, z.y, im);
        im = fmaf(hh[r].im, z.x, im);
      }
    }
  }
}

__global__ __launch_bounds__(256) void lt_edge_kernel(EdgeArgs a)
{
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int tile = __builtin_amdgcn_readfirstlane(wave & 15); // units 8 tile .. 8 tile + 7
  const int group = __builtin_amdgcn_readfirstlane(wave >> 4); // 64 (channel, window) pairs
  const int total = a.n_channels * a.n_windows;
  int pair = group * 64 + lane;
  const bool live = pair < total;
  if (!live) pair = total - 1; // keep the lane busy on valid addresses, drop its stores
  const int ch = pair / a.n_windows, k = pair - ch * a.n_windows;
  const v2f *__restrict__ x = a.iq + (int64_t)ch * a.ch_stride + (int64_t)k * kHalfFrame;
  // Z = [ x[9472 .. 9600) | x[0 .. 128) ] of the window: Z[i >= 128] = x[i - 128], Z[i < 128] = x[9472 + i]
  const v2f *__restrict__ zh = x - 128;               // head half, indexed by Z index
  const v2f *__restrict__ zt = x + (kHalfFrame - 128); // tail half, indexed by Z index
  const cf32 *__restrict__ h0 = a.tab->pss_taps[0];
  const cf32 *__restrict__ h1 = a.tab->pss_taps[1];
  const cf32 *__restrict__ h2 = a.tab->pss_taps[2];
  const int j0 = 8 * tile;
  float H[3][8][2], T[3][8][2];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int u = 0; u < 8; ++u) H[r][u][0] = H[r][u][1] = T[r][u][0] = T[r][u][1] = 0.f;
  // unit j0 + u, tap m reads Z[128 + j0 + u - m]
  for (int m0 = 0; m0 < j0; m0 += 8) edge_step<0>(zh + (128 + j0 - m0), h0, h1, h2, m0, H, T);
  { // taps j0 .. j0 + 7: Z[128 + u - d] with d = m - j0: head half iff u >= d
    v2f zs[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) zs[i] = (i >= 7) ? zh[128 + i - 7] : zt[128 + i - 7];
    // (the generic step reads one pointer; give it a private copy instead)
    edge_step<1>(zs + 7, h0, h1, h2, j0, H, T);
  }
  for (int m0 = j0 + 8; m0 < 128; m0 += 8) edge_step<2>(zt + (128 + j0 - m0), h0, h1, h2, m0, H, T);
  if (!live) return;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float *__restrict__ out = a.edge + r * a.edge_root_stride + (int64_t)ch * a.edge_ch_stride + (int64_t)k * 256;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u;
      if (j < 127) out[j] = __fadd_rn(__fmul_rn(H[r][u][0], H[r][u][0]), __fmul_rn(H[r][u][1], H[r][u][1]));
      if (j < 126) out[127 + j] = __fadd_rn(__fmul_rn(T[r][u][0], T[r][u][0]), __fmul_rn(T[r][u][1], T[r][u][1]));
    }
  }
}

} // namespace lt

// Host-callable launchers (the C-ABI in lt_api.cpp only sees these).
#include "lt_launch.h"

namespace lt {

hipError_t launch_chain(const ChainLaunch &L, hipStream_t stream)
{
  ChainArgs a;
  a.iq = reinterpret_cast<const v2f *>(L.iq);
  a.n_samples = L.n_samples;
  a.ch_stride = L.ch_stride;
  a.n_channels = L.n_channels;
  a.state = reinterpret_cast<ChainState *>(L.state);
  a.recs = L.recs;
  a.rec_cap = L.rec_cap;
  a.rec_count = L.rec_count;
  a.out_hf = reinterpret_cast<v2f *>(L.out_hf);
  a.out_cap = L.out_cap;
  a.tab = L.tab;
  a.psr_threshold = L.psr_threshold;
  a.track_after = L.track_after;
  a.track_every = L.track_every;
  a.ablate = L.ablate;
  a.prof = reinterpret_cast<unsigned long long *>(L.prof);
  a.metric = L.metric;
  a.metric_ch_stride = L.metric_ch_stride;
  a.metric_root_stride = L.metric_root_stride;
  a.edge = L.edge;
  a.edge_ch_stride = L.edge_ch_stride;
  a.edge_root_stride = L.edge_root_stride;
  if (L.metric) {
    MetricArgs m;
    m.iq = a.iq;
    m.n_samples = L.n_samples;
    m.ch_stride = L.ch_stride;
    m.n_channels = L.n_channels;
    m.blocks_per_channel = (int)((L.n_samples + kBlkValid - 1) / kBlkValid);
    m.metric = L.metric;
    m.metric_ch_stride = L.metric_ch_stride;
    m.metric_root_stride = L.metric_root_stride;
    m.tab = L.tab;
    const int64_t total = (int64_t)m.n_channels * m.blocks_per_channel;
    if (total > 0 && total < (int64_t)1 << 31) {
      // persistent: one workgroup per CU (its LDS footprint allows no second one), blocks strided over all waves
      const int64_t need = (total + kMetricWaves - 1) / kMetricWaves;
      const unsigned grid = (unsigned)(need < L.n_cus ? need : L.n_cus);
      const int abl = ((L.ablate & 512) ? 1 : 0) | ((L.ablate & 1024) ? 2 : 0); // diagnostics only (tools/probe.py)
      switch (abl) {
      case 1: hipLaunchKernelGGL(lt_metric_kernel<1>, dim3(grid), dim3(kMetricThreads), 0, stream, m); break;
      case 2: hipLaunchKernelGGL(lt_metric_kernel<2>, dim3(grid), dim3(kMetricThreads), 0, stream, m); break;
      case 3: hipLaunchKernelGGL(lt_metric_kernel<3>, dim3(grid), dim3(kMetricThreads), 0, stream, m); break;
      default: hipLaunchKernelGGL(lt_metric_kernel<0>, dim3(grid), dim3(kMetricThreads), 0, stream, m);
      }
    } else if (total > 0) {
      return hipErrorInvalidValue;
    }
    if (L.ev_metric_done) (void)hipEventRecord(L.ev_metric_done, stream);
    {
      EdgeArgs e;
      e.iq = a.iq;
      e.ch_stride = L.ch_stride;
      e.n_channels = L.n_channels;
      e.n_windows = (int)(L.n_samples / kHalfFrame);
      e.edge = L.edge;
      e.edge_ch_stride = L.edge_ch_stride;
      e.edge_root_stride = L.edge_root_stride;
      e.tab = L.tab;
      const int64_t pairs = (int64_t)e.n_channels * e.n_windows;
      if (pairs > 0) {
        const int64_t waves = ((pairs + 63) / 64) * 16;
        hipLaunchKernelGGL(lt_edge_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, e);
      }
    }
    if (L.ev_edge_done) (void)hipEventRecord(L.ev_edge_done, stream);
    if (L.ablate & 0xFF) hipLaunchKernelGGL((lt_chain_kernel<true, true>), dim3(L.n_channels * 3), dim3(kThreadsReplay), 0, stream, a);
    else hipLaunchKernelGGL((lt_chain_kernel<false, true>), dim3(L.n_channels * 3), dim3(kThreadsReplay), 0, stream, a);
    if (L.ev_replay_done) (void)hipEventRecord(L.ev_replay_done, stream);
    {
      SssArgs q;
      q.iq = a.iq;
      q.n_samples = L.n_samples;
      q.ch_stride = L.ch_stride;
      q.recs = L.recs;
      q.rec_cap = L.rec_cap;
      q.rec_count = L.rec_count;
      q.tab = L.tab;
      CpArgs c;
      c.iq = a.iq;
      c.n_samples = L.n_samples;
      c.ch_stride = L.ch_stride;
      c.recs = L.recs;
      c.rec_cap = L.rec_cap;
      c.rec_count = L.rec_count;
      c.metrics = reinterpret_cast<CpMetrics *>(L.cp_scratch);
      c.state = a.state;
      c.tab = L.tab;
      c.n_chains = L.n_channels * 3;
      if (L.rec_cap > 0) {
        if ((L.rec_cap + 3) / 4 > 65535) return hipErrorInvalidValue; // (a capture of > 37 minutes per run)
        const dim3 per_record((unsigned)(L.n_channels * 3), (unsigned)((L.rec_cap + 3) / 4));
        hipLaunchKernelGGL(lt_cp_kernel, per_record, dim3(256), 0, stream, c);
        hipLaunchKernelGGL(lt_cp_scan_kernel, dim3((unsigned)c.n_chains), dim3(64), 0, stream, c);
        hipLaunchKernelGGL(lt_sss_kernel, per_record, dim3(256), 0, stream, q);
      }
    }
  } else if (L.ablate & 0xFF) {
    a.metric = nullptr;
    hipLaunchKernelGGL((lt_chain_kernel<true, false>), dim3(L.n_channels * 3), dim3(kThreads), 0, stream, a);
  } else {
    hipLaunchKernelGGL((lt_chain_kernel<false, false>), dim3(L.n_channels * 3), dim3(kThreads), 0, stream, a);
  }
  return hipGetLastError();
}

size_t chain_state_bytes() { return sizeof(ChainState); }
size_t chain_state_avg_offset() { return offsetof(ChainState, avg); }

hipError_t launch_find(const void *win, float *avg, const DeviceTables *tab, int n_id_2, void *res,
                       hipStream_t stream)
{
  hipLaunchKernelGGL(lt_find_kernel, dim3(1), dim3(kThreads), 0, stream,
                     reinterpret_cast<const v2f *>(win), avg, tab, n_id_2,
                     reinterpret_cast<FindResult *>(res));
  return hipGetLastError();
}

hipError_t launch_cfo_compute(const void *sym, const DeviceTables *tab, int n_id_2, float *cfo,
                              hipStream_t stream)
{
  hipLaunchKernelGGL(lt_cfo_compute_kernel, dim3(1), dim3(64), 0, stream,
                     reinterpret_cast<const v2f *>(sym), tab, n_id_2, cfo);
  return hipGetLastError();
}

hipError_t launch_cfo_correct(const void *in, void *out, int n, float freq, const DeviceTables *tab,
                              hipStream_t stream)
{
  hipLaunchKernelGGL(lt_cfo_correct_kernel, dim3((n + 255) / 256), dim3(256), 0, stream,
                     reinterpret_cast<const v2f *>(in), reinterpret_cast<v2f *>(out), n, freq, tab);
  return hipGetLastError();
}

hipError_t launch_detect_cp(const void *in, int peak_pos, void *sync_state, int *cp_out,
                            hipStream_t stream)
{
  hipLaunchKernelGGL(lt_detect_cp_kernel, dim3(1), dim3(64), 0, stream,
                     reinterpret_cast<const v2f *>(in), peak_pos,
                     reinterpret_cast<SyncState *>(sync_state), cp_out);
  return hipGetLastError();
}

hipError_t launch_sss_m0m1(const void *sym, const DeviceTables *tab, int n_id_2, void *res,
                           hipStream_t stream)
{
  hipLaunchKernelGGL(lt_sss_m0m1_kernel, dim3(1), dim3(128), 0, stream,
                     reinterpret_cast<const v2f *>(sym), tab, n_id_2,
                     reinterpret_cast<M0M1Result *>(res));
  return hipGetLastError();
}

hipError_t launch_decim(const void *x, void *y, int n_channels, int64_t n_in, int D, const float *taps,
                        int ntaps, hipStream_t stream)
{
  const int64_t n_out = n_in / D;
  if (n_out <= 0) return hipSuccess;
  const int lead = ((ntaps - 1 + D - 1) / D) * D;
  const int R = decim_row_len((kDecimOutPerBlock * D + lead + D - 1) / D, D);
  const size_t lds = sizeof(v2f) * (size_t)D * (size_t)R;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  void (*kern)(const v2f *, v2f *, int64_t, int64_t, int, const float *, int, int) =
      D == 4 ? lt_decim_kernel<4> : D == 8 ? lt_decim_kernel<8> : D == 16 ? lt_decim_kernel<16> : lt_decim_kernel<0>;
  if (lds > 64 * 1024) { // (decimation 16: 70 KB)
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  dim3 grid((unsigned)((n_out + kDecimOutPerBlock - 1) / kDecimOutPerBlock), (unsigned)n_channels);
  hipLaunchKernelGGL(kern, grid, dim3(kDecimThreads), lds, stream, reinterpret_cast<const v2f *>(x),
                     reinterpret_cast<v2f *>(y), n_in, n_out, D, taps, ntaps, R);
  return hipGetLastError();
}

} // namespace lt
