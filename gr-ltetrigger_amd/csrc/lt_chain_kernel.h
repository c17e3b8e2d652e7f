// Per-chain kernel: pss_impl::general_work (lib/pss_impl.cc:155-223) call by call for one (channel, N_id_2)
// chain per workgroup, from wherever the chain stands: a chain the grid scan (lt_scan_kernels.h) has carried up
// to the window that fired, a chain that is building its score or tracking, or -- with the scan switched
// off (lt_batch_set_mode 0) -- any chain at all.  One record per call.  Everything the emitted half-frames go
// through -- CFO estimate and correction of the pss block, CP decision and SSS of the sss block -- feeds records
// only, never the state machine, and is added afterwards by the per-record kernels in lt_kernels.hip.
//
// 12 waves, wave s = strip s of the window being searched (lt_scan.h): it fetches the strip's 1024 samples,
// runs the forward transform, multiplies by the root's filter spectrum, transforms back and keeps the strip's
// exponential average in 14 registers per lane -- the same instruction sequence as the scan's root waves, so both
// give the same bits.  The 12 StripPartials meet in LDS and every wave resolves them redundantly
// (resolve_window): one workgroup barrier per search.  A lobe too wide for the strips' 64-lag neighbourhoods
// takes the exact path: the average goes to an LDS mirror and the walk of oracle/lte_oracle.c:300-316 runs on it.
#pragma once
#include <cstddef>

#include "lt_records.h"
#include "lt_scan_kernels.h"

namespace lt {

constexpr int kChainWaves = kStrips;
constexpr int kChainThreads = 64 * kChainWaves;
// farthest sample a call starting at pos can touch is pos + kCallReach - 1 (peak at lag 9725: the half-frame
// [pos + 8765, pos + 18365)); lib/pss_impl.cc:185-195
constexpr int kCallReach = (kNLags - 1 - kSlotLen) + kHalfFrame;

struct ChainArgs {
  const v2f *iq;       // [n_channels][ch_stride]
  int64_t ch_stride;
  const v2f *hist;     // [n_channels][hist_stride] or null
  int64_t hist_stride;
  int64_t t0, n, hist_len;
  int flush;           // 1: the stream ends with this chunk (a chain runs while a whole window is left, and stops
                       // at an emission that would reach past the end: quirk q1 normalised); 0: a call starts only
                       // when everything it could touch is in the chunk
  int n_channels;
  ChainState *state;   // [n_channels*3], index channel*3 + n_id_2
  lt_record *recs;     // [n_channels*3][rec_cap]
  int rec_cap;
  int *rec_count;      // [n_channels*3]: in = records the resolver wrote, out = records of the run
  const DeviceTables *tab;
  float psr_threshold;
  int track_after;
  int track_every;
  int2 *list;          // work list of the per-record kernels: (chain, record) of every tracked half-frame of the run
  int *list_count;
  // ---- window cache (lt_metric_kernel below).  pass 0: a chain with many searches ahead of it asks for the lag
  // metric of its coming windows and leaves the rest to pass 1; pass 1: every chain runs to the end of the chunk,
  // taking the lag metric of a window from the cache where it is there.  cache == null: one pass, no cache.
  float *cache;        // [slot_cap][12][896]
  int2 *slot_owner;    // [slot_cap]: (chain, k) of every slot handed out
  int64_t *slot_pos;   // [slot_cap]: the window's position
  int *slot_next;      // slots handed out so far in this run (may run past slot_cap: the excess does not exist)
  int *launch_slots;   // ... those of them that pass 0 hands out (lt_metric_kernel's share of the work)
  int slot_cap;
  int pass;
  int *wait_hint;      // two words of pinned host memory (or null): [0] set to run_id by a chain that has enough searches ahead
  int run_id;          // of it to ask for the cache -- whether the run has one or not --, [1] by the kernel as such (pass 0):
                       // the host's cue to give the next runs two passes or one
};
constexpr int kCacheStripLags = 64 * kStripRegs;   // 896 floats per (slot, strip)
constexpr int kCacheMinSearches = 6;               // fewer searches ahead than this: not worth a pass of its own
constexpr int kCacheMaxPerChain = 256;
__host__ __device__ constexpr size_t cache_slot_floats() { return (size_t)kStrips * kCacheStripLags; }

struct SearchResult {
  int peak_pos;
  float peak_value;
  float psr;
};

// LDS of the chain kernel / the single-call find kernel
struct ChainLds {
  alignas(16) v2f E[kChainWaves * fft::kScratchW]; // exchange areas; the exact path's mirror of the average lives here too
  alignas(16) fft::c32 TW1[fft::kTw1Len];
  alignas(16) fft::c32 TW2[fft::kTw2Len];
  v2f SPEC[fft::kN];
  StripPartial parts[2][kStrips];      // two sets used in turn: no barrier between a resolve and the next search
  float wave_out[kChainWaves];
};
static_assert(sizeof(v2f) * kChainWaves * fft::kScratchW >= sizeof(float) * (kConvLen + 8), "mirror must fit in the exchange areas");

__device__ __forceinline__ void chain_lds_init(ChainLds &L, const DeviceTables *__restrict__ tab, int n_id_2)
{
  for (int i = threadIdx.x; i < fft::kTw1Len; i += blockDim.x) L.TW1[i] = *reinterpret_cast<const fft::c32 *>(tab->fft_tw1[i]);
  for (int i = threadIdx.x; i < fft::kTw2Len; i += blockDim.x) L.TW2[i] = *reinterpret_cast<const fft::c32 *>(tab->fft_tw2[i]);
  for (int i = threadIdx.x; i < fft::kN; i += blockDim.x) {
    const cf32 h = tab->pss_spec[n_id_2][i];
    L.SPEC[i] = v2f{h.re, h.im};
  }
}

// srslte_pss_find_pss on the window whose strip samples are in x (strip_load).  resolve = false: lag metric and
// exponential average only (a window the scan has already turned into a record).  Contains workgroup
// barriers; `buf` alternates between calls.
// cached: the window's lag metric as lt_metric_kernel left it (12 x 896 floats) -- the transforms are skipped, the
// average and everything after it run as ever (x is not looked at then).
// Round 3 pulled the NEXT slot's lines into the L2 during a cached search, with fourteen loads into one register that
// nobody reads, left in flight past their block of assembly -- a register the compiler is free to reuse under them
// (an instrumented build of this kernel faulted).  The loads are gone; measured on one box against the
// version with them, per-chain kernel of a 64-channel step at -4 / 0 dB: 1.92 / 1.21 ms without, 1.91 / 1.21 with --
// they bought nothing.  What is left of the mechanism is the plain wait at the top of the search after a cached one.
// (Safe prefetches that were measured and do not pay: the same loads closed by a wait of their own, +15 %: the search
// then waits for the next slot's HBM latency; one or two LDS-DMA loads, a sector per lane, into a dump area, +8 %; LDS-DMA
// of the whole strip into the wave's idle exchange area, +3 %: an LDS-DMA instruction costs the wave ~300 cycles to issue.)
struct WarmUp {
  float sink = 0.f;
  bool on = false;
};
__device__ __forceinline__ void warm_up_landed(WarmUp &wu)
{
  if (wu.on) asm volatile("s_waitcnt vmcnt(0)" : "+v"(wu.sink)::"memory");
  wu.on = false;
}
__device__ __forceinline__ SearchResult chain_search(ChainLds &L, fft::c32 (&x)[16], int w, int lane, ema_t &ema, int &buf,
                                                     bool resolve, WarmUp &wu, const float *cached = nullptr,
                                                     const float *cache_end = nullptr)
{
  fft::c32 *e = L.E + w * fft::kScratchW;
  warm_up_landed(wu);
  if (cached) {
    float met[kStripRegs];
    const float *m = cached + (size_t)w * kCacheStripLags + lane;
#pragma unroll
    for (int r = 0; r < kStripRegs; ++r) met[r] = m[64 * r];
    // The slot after this one is what the chain will want next if it stays on its grid: its lines are pulled into
    // this XCD's L2 now, so that the search after this one finds them there instead of in HBM (WarmUp above).
    if (cached + cache_slot_floats() < cache_end) {
      const float *n = m + cache_slot_floats();
      wu.on = true;
      (void)n; // (round 3 issued fourteen loads into wu.sink here and left them in flight: removed, see below)
    }
    strip_ema_apply(w, lane, met, ema);
  } else {
    strip_fft<false>(lane, x, L.TW1, L.TW2, e);
    strip_root_update(w, lane, x, L.SPEC, L.TW1, L.TW2, e, ema);
  }
  SearchResult out;
  out.peak_pos = 0;
  out.peak_value = 0.f;
  out.psr = 0.f;
  if (!resolve) return out;
  StripPartial p;
  strip_resolve(ema, w, lane, p);
  StripPartial *parts = L.parts[buf];
  buf ^= 1;
  if (lane == 0) parts[w] = p;
  __syncthreads();
  const WindowResult r = resolve_window_unrolled(parts); // (the reads of a loop in flight together, not in turn)
  out.peak_pos = __builtin_amdgcn_readfirstlane(r.peak_pos);
  out.peak_value = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(r.peak_value)));
  out.psr = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(r.psr)));
  if (__builtin_amdgcn_readfirstlane((int)r.exact)) return out;
  // ---- exact path (oracle/lte_oracle.c:295-316 on the whole vector): the peak is right as it is; the lobe
  // walk and the side-lobe maximum run on a mirror of the average in LDS (the exchange areas are idle now)
  float *M = reinterpret_cast<float *>(L.E);
  strip_ema_store(M, w, lane, ema);
  if (threadIdx.x < 8) M[kNLags + threadIdx.x] = 0.f;
  __syncthreads();
  const int peak = out.peak_pos;
  int pl_ub = peak + 1;
  for (int base = peak + 1;; base += 64) { // while (avg[pl_ub + 1] <= avg[pl_ub] && pl_ub < 9727) pl_ub++
    const int u = base + lane;
    bool stop = true;
    if (u < kConvLen - 1) stop = !(M[u + 1] <= M[u]);
    const unsigned long long m = __ballot(stop);
    if (m) {
      pl_ub = base + __builtin_ctzll(m);
      break;
    }
  }
  int pl_lb = 0;
  if (peak > 2) {
    for (int base = peak - 1;; base -= 64) { // while (avg[pl_lb - 1] <= avg[pl_lb] && pl_lb > 1) pl_lb--
      const int u = base - lane;
      bool stop = true;
      if (u > 1) stop = !(M[u - 1] <= M[u]);
      const unsigned long long m = __ballot(stop);
      if (m) {
        pl_lb = base - __builtin_ctzll(m);
        break;
      }
    }
  }
  float o = -FLT_MAX; // maximum over the valid lags outside [pl_lb, pl_ub)
  {
    const int c0 = kStripPitch * w - kStripHalo;
#pragma unroll
    for (int r = 0; r < kStripRegs; ++r) {
      const int off = lane + 64 * r, lag = c0 + off;
      const bool own = off >= kStripHalo && off < kStripHalo + kStripPitch && lag < kNLags;
      if (own && (lag < pl_lb || lag >= pl_ub) && ema[r] > o) o = ema[r];
    }
  }
  o = wave_max_f(o);
  if (lane == 0) L.wave_out[w] = o;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kChainWaves; ++k) o = fmaxf(o, L.wave_out[k]);
  const float sl = (pl_lb == 0) ? M[0] : o;               // srslte_vec_max_fi of an empty range: its first element
  const float sr = (kNLags - pl_ub <= 0) ? M[pl_ub] : o;
  const float side = sl > sr ? sl : sr;
  out.psr = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(out.peak_value / side)));
  __syncthreads(); // the mirror is read: the exchange areas may be used again
  return out;
}

__device__ __forceinline__ void chain_run(const ChainArgs &a)
{
  __shared__ ChainLds L;
  __shared__ float psr_data[kMovingAvgSz];
  constexpr int NT = kChainThreads;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chain = blockIdx.x, channel = chain / 3, n_id_2 = chain % 3;
  ChainState *st = a.state + chain;
  if (st->run_done) return; // the resolver has seen the chain stop for this run

  Capture cap;
  cap.iq = a.iq + (int64_t)channel * a.ch_stride;
  cap.hist = a.hist ? a.hist + (int64_t)channel * a.hist_stride : nullptr;
  cap.t0 = a.t0;
  cap.n = a.n;
  cap.hist_len = a.hist ? a.hist_len : 0;
  const int64_t t1 = a.t0 + a.n;
  lt_record *recs = a.recs + (int64_t)chain * a.rec_cap;
  auto can_call = [&](int64_t p) { return a.flush ? p + kHalfFrame <= t1 : p + kCallReach <= t1; };

  int64_t pos = st->pos;
  int nrec = a.rec_count[chain]; // the calls the scan rounds have turned into records
  // a chain the scan rounds have carried as far as the chunk allows has nothing left to do in this run -- unless its
  // average is still described through windows of this chunk (pre_n): those samples will be gone by the next one
  if (!(can_call(pos) && nrec < a.rec_cap) && st->pre_n == 0) return;

  // ---- pass 0 of two: a chain with many searches ahead of it -- all on the grid it stands on, unless a hit moves it --
  // asks for their lag metric (lt_metric_kernel forms them for all such chains at once, chip-wide) and leaves
  // everything to pass 1.  An idle or acquiring chain searches every window from pos on; a tracked one every
  // (track_every + 1)-th from its next search on.  (Windows its average still has to be run through -- pre_n -- are
  // found in the cache if they lie on the same grid, and go through the transforms if not.)
  if ((a.cache || a.wait_hint) && a.pass == 0 && can_call(pos) && nrec < a.rec_cap) {
    const int stride = st->is_tracking ? a.track_every + 1 : 1;
    const int64_t first = pos + (st->is_tracking ? (int64_t)st->timer * kHalfFrame : 0);
    const int64_t last_start = a.flush ? t1 - kHalfFrame : t1 - kCallReach; // last position a call may start at
    int64_t ahead = first <= last_start ? (last_start - first) / ((int64_t)stride * kHalfFrame) + 1 : 0;
    if (ahead > kCacheMaxPerChain) ahead = kCacheMaxPerChain;
    if (ahead >= kCacheMinSearches && a.wait_hint && tid == 0) *reinterpret_cast<volatile int *>(a.wait_hint) = a.run_id;
    if (a.cache && ahead >= kCacheMinSearches) {
      __shared__ int s_base;
      if (tid == 0) s_base = atomicAdd(a.slot_next, (int)ahead);
      __syncthreads();
      const int base = s_base;
      int got = a.slot_cap - base < (int)ahead ? a.slot_cap - base : (int)ahead;
      if (got >= kCacheMinSearches) {
        for (int i = tid; i < got; i += NT) {
          a.slot_owner[base + i] = make_int2(chain, i);
          a.slot_pos[base + i] = first + (int64_t)i * stride * kHalfFrame;
        }
        if (tid == 0) {
          atomicAdd(a.launch_slots, got);
          atomicAdd(a.slot_next + 3, 1); // diagnostics
          st->cache_pos = first;
          st->cache_base = base;
          st->cache_n = got;
          st->cache_stride = stride;
        }
        return; // (untouched: pass 1 takes the chain from here)
      }
    }
  }
  const float *cache0 = nullptr;
  int64_t cache_pos = 0, cache_step = 1;
  int cache_n = 0;
  if (a.cache && a.pass == 1 && st->cache_n > 0) {
    cache0 = a.cache + (size_t)st->cache_base * cache_slot_floats();
    cache_pos = st->cache_pos;
    cache_step = (int64_t)st->cache_stride * kHalfFrame;
    cache_n = st->cache_n;
  }
  const float *cache_end = cache0 ? cache0 + (size_t)cache_n * cache_slot_floats() : nullptr;
  WarmUp warm;
  int64_t cache_next_pos = cache_pos; // the window of slot cache_next_k: where a chain that stays on its grid searches next
  int cache_next_k = 0;

  // ---- load state.  The average is avg[ema_sel] (or all zeros after srslte_pss_reset) run through the pre_n
  // windows of a scan round the chain did not follow to its end: brought up to date before the first search,
  // by the same search call as every other window.
  ema_t ema;
  float *avg = st->avg[st->ema_sel];
  if (st->ema_zero) {
#pragma unroll
    for (int r = 0; r < 16; ++r) ema[r] = r < kStripRegs ? 0.f : -FLT_MAX;
  } else {
    strip_ema_load(avg, w, lane, ema);
  }
  int redo_k = 0;
  const int redo_end = st->pre_n;
  const int64_t redo_pos0 = st->pre_pos;
  const int redo_stride = st->pre_stride;
  for (int i = tid; i < kMovingAvgSz; i += NT) psr_data[i] = st->psr_data[i];
  chain_lds_init(L, a.tab, n_id_2);
  int is_tracking = st->is_tracking, score = st->score, timer = st->timer;
  int tracking_lost = st->tracking_lost, peak_pos = st->peak_pos;
  uint32_t psr_i = st->psr_i;
  float psr = st->psr, psr_max = st->psr_max;
  const int call_base = st->n_calls - nrec;
  int n_emits = 0;
  int buf = 0;
  bool searched = false; // the registers hold the chain's average (a search has run)
  bool stop_run = false; // the stream ends inside the half-frame a call would emit: no further call in this run
  __syncthreads();
  // One record: everything the sss side adds (CFO, CP, SSS: lt_kernels.hip, per record) starts out empty.
  auto put_record = [&](int idx, int64_t in_pos, int64_t out_pos, uint32_t flags, int used_peak, int sc, float ps, float pv, float pmax, int tm) {
    // six 16-byte stores, one after the other: a whole lt_record in registers (24 per lane) on top of the strip's
    // average and the prefetched block is what pushed this kernel into scratch
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 *dst = reinterpret_cast<u4 *>(recs + idx);
    static_assert(sizeof(lt_record) == 96 && offsetof(lt_record, channel) == 16 && offsetof(lt_record, peak_pos) == 32 &&
                  offsetof(lt_record, cfo) == 48 && offsetof(lt_record, m0_value) == 64 && offsetof(lt_record, n_id_1) == 80,
                  "lt_record layout");
    dst[0] = u4{(uint32_t)in_pos, (uint32_t)(in_pos >> 32), (uint32_t)out_pos, (uint32_t)(out_pos >> 32)};
    asm volatile("" ::: "memory");
    dst[1] = u4{(uint32_t)channel, (uint32_t)n_id_2, (uint32_t)(call_base + idx), flags};
    asm volatile("" ::: "memory");
    dst[2] = u4{(uint32_t)used_peak, (uint32_t)sc, __float_as_uint(ps), __float_as_uint(pv)};
    asm volatile("" ::: "memory");
    dst[3] = u4{0u, 0u, 0u, 0u};                     // cfo, mean_cfo, m_norm_avg, m_ext_avg
    asm volatile("" ::: "memory");
    dst[4] = u4{0u, 0u, 0xffffffffu, 0xffffffffu};   // m0_value, m1_value, m0 = m1 = -1
    asm volatile("" ::: "memory");
    dst[5] = u4{0xffffffffu, 0xffffffffu, __float_as_uint(pmax), (uint32_t)tm}; // n_id_1 = cell_id = -1
  };

  for (;;) {
    const bool callable = can_call(pos) && nrec < a.rec_cap;
    // the windows the average still has to go through come first: before the first search, or -- if no call fits
    // into this chunk any more -- before the kernel ends
    const bool search_due = !(is_tracking && timer > 0 && !tracking_lost && peak_pos == kSlotLen && psr > a.psr_threshold) && (!is_tracking || timer == 0);
    const bool redo = redo_k < redo_end && (search_due || !callable);
    if (!callable && !redo) break;
    // ---- tracked chain between two searches (lib/pss_impl.cc:163-214 with timer > 0): the next `timer` calls
    // neither search nor change the score; each emits the half-frame at its input position (peak_pos is 960
    // since the emit of the last search).  Nothing of what they compute (CFO, CP, SSS) feeds the state machine:
    // one record per call, one lane each.
    if (!redo && is_tracking && timer > 0 && !tracking_lost && peak_pos == kSlotLen && psr > a.psr_threshold) {
      int nb = timer;
      while (nb > 0 && !can_call(pos + (int64_t)(nb - 1) * kHalfFrame)) --nb;
      if (a.rec_cap - nrec < nb) nb = a.rec_cap - nrec;
      for (int idx = tid; idx < nb; idx += NT) {
        const int64_t hf0 = pos + (int64_t)idx * kHalfFrame;
        put_record(nrec + idx, hf0, hf0, LT_F_OVER | LT_F_EMITTED | LT_F_SSS_RAN | LT_F_TRACKING, kSlotLen, score, psr, 0.f,
                   psr_max, timer - 1 - idx);
      }
      timer -= nb;
      nrec += nb;
      n_emits += nb;
      pos += (int64_t)nb * kHalfFrame;
      continue;
    }

    uint32_t flags = 0;
    float peak_value = 0.f;
    // ---- lib/pss_impl.cc:163-172
    if (redo || !is_tracking || timer == 0) {
      const int64_t wpos = redo ? redo_pos0 + (int64_t)redo_k * redo_stride * kHalfFrame : pos;
      // (the lane index is made opaque per iteration: otherwise the compiler hoists every lane-dependent address
      // and mask of the search out of the loop and spills them across it)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      fft::c32 x[16];
      // the window's lag metric may be in the cache (pass 1, a window on the grid the chain stood on in pass 0)
      const float *cached = nullptr;
      if (cache_n > 0 && wpos >= cache_pos) {
        int64_t kq;
        if (wpos == cache_next_pos) { // (the usual case: no division)
          kq = cache_next_k;
        } else {
          const int64_t d = wpos - cache_pos;
          kq = d / cache_step;
          if (kq * cache_step != d) kq = -1;
        }
        if (kq >= 0 && kq < cache_n) {
          cached = cache0 + (size_t)kq * cache_slot_floats();
          cache_next_k = (int)kq + 1;
          cache_next_pos = wpos + cache_step;
        }
      }
      if (!cached) strip_load(cap, wpos, w, ln, x);
      if (a.cache && a.pass == 1 && tid == 0) atomicAdd(a.slot_next + (redo ? (cached ? 5 : 6) : (cached ? 1 : 2)), 1); // diagnostics
      const SearchResult r = chain_search(L, x, w, ln, ema, buf, !redo, warm, cached, cache_end);
      searched = true;
      if (redo) {
        redo_k++;
        continue;
      }
      timer = a.track_every;
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
          for (int t = 0; t < kStripRegs; ++t) ema[t] = 0.f; // srslte_pss_reset
        }
      }
      flags |= LT_F_OVER;
    } else if (score != 0) {
      score = 0;
      timer = 0;
      is_tracking = 0;
#pragma unroll
      for (int t = 0; t < kStripRegs; ++t) ema[t] = 0.f;
      psr_i = 0; // (d_cfo_i restarts too: the CFO ring of this kernel's tail sees the tracking_lost tag this call puts on its half-frame)
      tracking_lost = 1;
    }
    if (psr > psr_max) psr_max = psr;

    int64_t out_pos = -1, next_pos;
    if (over || tracking_lost) { // ---- lib/pss_impl.cc:184-214
      const int frame_start = peak_pos - kSlotLen;
      peak_pos = kSlotLen;
      const int64_t hf0 = pos + frame_start;
      if (cache_n > 0 && frame_start != 0 && tid == 0) atomicAdd(a.slot_next + 4, 1); // diagnostics
      if (hf0 + kHalfFrame > t1) { // (flush only) quirk q1 normalised: never read past the capture
        // (the chain is through for this run -- and must stay so: the second pass of a two-pass run would take the call
        // up again, from an average this search has already gone into)
        stop_run = true;
        break;
      }
      next_pos = hf0 + kHalfFrame;
      out_pos = hf0;
      flags |= LT_F_EMITTED;
      if (is_tracking) {
        // CFO estimate and correction (lib/pss_impl.cc:199-204), CP decision and SSS (lib/sss_impl.cc:102-150) of
        // this half-frame: per record, after this kernel
        flags |= LT_F_SSS_RAN;
      } else {
        // tracking lost: tag the half-frame (lib/pss_impl.cc:210-214); sss resets on the tag (lib/sss_impl.cc:91-100)
        flags |= LT_F_TAG_LOST;
        tracking_lost = 0;
      }
      n_emits++;
    } else {
      next_pos = pos + kHalfFrame; // lib/pss_impl.cc:217
    }

    if (is_tracking) flags |= LT_F_TRACKING;
    if (tid == 0) put_record(nrec, pos, out_pos, flags, used_peak_pos, score, psr, peak_value, psr_max, timer);
    nrec++;
    pos = next_pos;
  }

  warm_up_landed(warm);
  // ---- store state
  __syncthreads();
  if (searched && redo_k >= redo_end) strip_ema_store(avg, w, lane, ema); // (else: the average is described as before)
  for (int i = tid; i < kMovingAvgSz; i += NT) st->psr_data[i] = psr_data[i];
  if (tid == 0) {
    st->pos = pos;
    st->is_tracking = is_tracking; st->score = score; st->timer = timer;
    st->tracking_lost = tracking_lost; st->peak_pos = peak_pos;
    st->psr_i = psr_i;
    st->n_calls = call_base + nrec; st->n_emits += n_emits;
    st->psr = psr; st->psr_max = psr_max;
    if (searched && redo_k >= redo_end) {
      st->ema_zero = 0;
      st->pre_n = 0;
    }
    a.rec_count[chain] = nrec;
    if (stop_run) st->run_done = 1;
  }
}

// One workgroup per chain: general_work call by call for whatever the scan rounds have left of it (chain_run), then
// -- every record of the chain is final now -- what the pss block does with the tracked half-frames besides emitting
// them (lib/pss_impl.cc:199-204): the CFO estimate of each (eight records per wave at a time) and, in call order, the
// ring with mean_cfo() (one wave), 768 records per turn; and the chain's tracked half-frames go onto the work list of
// the per-record kernels (one atomic per 64 records), which rotate them by that mean (lt_kernels.hip).
__global__ __launch_bounds__(kChainThreads) void lt_chain_kernel(ChainArgs a)
{
  chain_run(a);
  // (the host's cue, lt_api.cpp: this run's per-chain kernel has run -- so every launch of the run before is through)
  if (a.pass == 0 && a.wait_hint && blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<volatile int *>(a.wait_hint)[1] = a.run_id;
  __threadfence_block();
  __syncthreads();
  const int chain = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (with two passes: a chain whose records pass 0 left final is listed then; one that waits for the cache, in pass 1)
  ChainState *st = a.state + chain;
  if (st->listed || (a.cache && a.pass == 0 && st->cache_n > 0)) return;
  __syncthreads();
  if (tid == 0) st->listed = 1;
  const int count = a.rec_count[chain];
  lt_record *recs = a.recs + (int64_t)chain * a.rec_cap;
  __shared__ uint32_t t_flags[kChainThreads];
  __shared__ float t_cfo[kChainThreads];
  __shared__ int t_idx[kChainThreads];      // the turn's tracked records, compacted: index within the turn ...
  __shared__ int64_t t_pos[kChainThreads];  // ... and where the half-frame starts
  __shared__ int t_wave_n[kChainWaves];
  __shared__ float t_ring[kMovingAvgSz];
  Capture cap;
  cap.iq = a.iq + (int64_t)(chain / 3) * a.ch_stride;
  cap.hist = a.hist ? a.hist + (int64_t)(chain / 3) * a.hist_stride : nullptr;
  cap.t0 = a.t0;
  cap.n = a.n;
  cap.hist_len = a.hist ? a.hist_len : 0;
  __shared__ v2f t_taps[kSymbolSz];
  if (tid < kSymbolSz) {
    const cf32 h = a.tab->pss_taps[chain % 3][tid];
    t_taps[tid] = v2f{h.re, h.im};
  }
  const v2f *taps = t_taps; // (visible to every wave after the first barrier of the first turn)
  CfoRing ring;
  bool ring_open = false; // (wave 0) the ring has been fetched: only chains with tracked half-frames or lost tracking touch it
  for (int b0 = 0; b0 < count; b0 += kChainThreads) {
    const int i = b0 + tid;
    uint32_t flags = 0u;
    int64_t out_pos = 0;
    if (i < count) {
      flags = recs[i].flags;
      out_pos = recs[i].out_pos;
    }
    const bool ran = (flags & LT_F_SSS_RAN) != 0;
    const unsigned long long m = __ballot(ran);
    const int n_w = __builtin_popcountll(m), rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
    t_flags[tid] = flags;
    if (lane == 0) t_wave_n[w] = n_w;
    if (m) { // the work list of the per-record kernels
      int base = 0;
      if (lane == 0) base = atomicAdd(a.list_count, n_w);
      base = __builtin_amdgcn_readfirstlane(base);
      if (ran) a.list[base + rank] = make_int2(chain, i);
    }
    __syncthreads();
    int before = 0, n_ran = 0;
#pragma unroll
    for (int k = 0; k < kChainWaves; ++k) {
      const int c = t_wave_n[k];
      before += k < w ? c : 0;
      n_ran += c;
    }
    const unsigned long long any_lost = __ballot((flags & LT_F_TAG_LOST) != 0);
    if (ran) {
      t_idx[before + rank] = tid;
      t_pos[before + rank] = out_pos;
    }
    __shared__ int t_lost;
    if (tid == 0) t_lost = 0;
    __syncthreads();
    if (any_lost && lane == 0) t_lost = 1; // (benign race: every writer writes 1)
    // ---- srslte_pss_cfo_compute on the uncorrected PSS symbol of every tracked half-frame (lib/pss_impl.cc:199-201)
    for (int g = w * 8 + (lane >> 3); g < n_ran; g += kChainWaves * 8) {
      const int local = t_idx[g];
      const float c = cfo_estimate_group8(taps, cap, t_pos[g] + (kSlotLen - kSymbolSz), lane);
      if ((lane & 7) == 0) {
        t_cfo[local] = c;
        recs[b0 + local].cfo = c;
      }
    }
    __syncthreads();
    // ---- d_cfo_data ring, mean_cfo() (lib/pss_impl.cc:94-109,200-204): in call order, one wave
    if (w == 0 && (n_ran > 0 || t_lost)) {
      // (the lane index made opaque: otherwise the compiler forms the ring's lane-dependent addresses in HBM at the top of the
      // kernel's tail and spills them across the CFO loop)
      int ln = lane;
      asm volatile("" : "+v"(ln));
      if (!ring_open) {
        cfo_ring_begin(st, t_ring, ring, ln);
        ring_open = true;
      }
      const int nb = count - b0 < kChainThreads ? count - b0 : kChainThreads;
      cfo_ring_records(t_ring, ring, nb, ln,
                       [&](int k, uint32_t &f, float &c) { f = t_flags[k]; c = t_cfo[k]; },
                       [&](int k, float mean) { recs[b0 + k].mean_cfo = mean; });
    }
    __syncthreads();
  }
  if (w == 0 && ring_open) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    cfo_ring_end(st, t_ring, ring, ln);
  }
}

// ---------------------------------------------------------------- window cache: the lag metric in bulk
// The lag metric of a window does not depend on the chain's state, only the average that follows it does: for the
// chains pass 0 of the per-chain kernel has put on the list, the metric of every coming window is formed here, one
// wave per (window, strip), with nothing sequential about it -- 16 waves per CU, every CU of the chip -- and the
// per-chain kernel's pass 1 is left with the average, the peak search and general_work.  Same instruction sequence
// as the search itself (strip_load, strip_fft, strip_root_lags, strip_metric): same bits.
struct MetricArgs {
  const v2f *iq;
  int64_t ch_stride;
  const v2f *hist;
  int64_t hist_stride;
  int64_t t0, n, hist_len;
  const int2 *slot_owner;
  const int64_t *slot_pos;
  const int *launch_slots; // [n_before + 1]: slots handed out for the launches before this one, then for this one
  int n_before;
  int slot_cap;
  float *cache;
  const DeviceTables *tab;
};
constexpr int kMetricWaves = 8;
__global__ __launch_bounds__(64 * kMetricWaves) void lt_metric_kernel(MetricArgs a)
{
  __shared__ __attribute__((aligned(16))) v2f E[kMetricWaves * fft::kScratchW];
  __shared__ __attribute__((aligned(16))) fft::c32 TW1[fft::kTw1Len];
  __shared__ __attribute__((aligned(16))) fft::c32 TW2[fft::kTw2Len];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // this launch's slots: those handed out since the launch before (the slots in use are contiguous from 0)
  int slot0 = 0;
  for (int i = 0; i < a.n_before; ++i) slot0 += a.launch_slots[i];
  int n_slots = a.launch_slots[a.n_before];
  if (slot0 + n_slots > a.slot_cap) n_slots = a.slot_cap - slot0;
  const int n_items = n_slots * kStrips;
  if ((int)blockIdx.x * kMetricWaves >= n_items) return;
  for (int i = tid; i < fft::kTw1Len; i += 64 * kMetricWaves) TW1[i] = *reinterpret_cast<const fft::c32 *>(a.tab->fft_tw1[i]);
  if (tid < fft::kTw2Len) TW2[tid] = *reinterpret_cast<const fft::c32 *>(a.tab->fft_tw2[tid]);
  __syncthreads();
  fft::c32 *e = E + w * fft::kScratchW;
  for (int item = blockIdx.x * kMetricWaves + w; item < n_items; item += gridDim.x * kMetricWaves) {
    const int slot = slot0 + item / kStrips, strip = item % kStrips;
    const int chain = __builtin_amdgcn_readfirstlane(a.slot_owner[slot].x);
    const int64_t wpos = uniform_i64(a.slot_pos[slot]);
    Capture cap;
    cap.iq = a.iq + (int64_t)(chain / 3) * a.ch_stride;
    cap.hist = a.hist ? a.hist + (int64_t)(chain / 3) * a.hist_stride : nullptr;
    cap.t0 = a.t0;
    cap.n = a.n;
    cap.hist_len = a.hist ? a.hist_len : 0;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    fft::c32 x[16];
    strip_load(cap, wpos, strip, ln, x);
    strip_fft<false>(ln, x, TW1, TW2, e);
    v2f spec[16];
#pragma unroll
    for (int R = 0; R < 16; ++R) {
      const cf32 hh = a.tab->pss_spec[chain % 3][ln + 64 * R];
      spec[R] = v2f{hh.re, hh.im};
    }
    fft::Twiddles tw;
    fft::stageA_twiddles(ln, TW1, tw);
    strip_root_lags(ln, x, spec, tw, TW2, e);
    float met[kStripRegs];
    strip_metric(x, met);
    float *out = a.cache + (size_t)slot * cache_slot_floats() + (size_t)strip * kCacheStripLags + ln;
#pragma unroll
    for (int r = 0; r < kStripRegs; ++r) out[64 * r] = met[r];
  }
}

// srslte_pss_find_pss for one handle (lt_pss_find_pss): window = 9600 device samples, avg = 9728-float state.
struct FindResult {
  int peak_pos;
  float psr;
  float peak_value;
};
__global__ __launch_bounds__(kChainThreads) void lt_find_kernel(const v2f *__restrict__ win, float *avg,
                                                                const DeviceTables *__restrict__ tab, int n_id_2, FindResult *res)
{
  __shared__ ChainLds L;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  Capture cap;
  cap.iq = win; cap.hist = nullptr; cap.t0 = 0; cap.n = kHalfFrame; cap.hist_len = 0;
  ema_t ema;
  strip_ema_load(avg, w, lane, ema);
  chain_lds_init(L, tab, n_id_2);
  __syncthreads();
  int buf = 0;
  fft::c32 x[16];
  strip_load(cap, 0, w, lane, x);
  WarmUp none;
  const SearchResult r = chain_search(L, x, w, lane, ema, buf, true, none);
  strip_ema_store(avg, w, lane, ema);
  if (tid == 0) {
    res->peak_pos = r.peak_pos;
    res->psr = r.psr;
    res->peak_value = r.peak_value;
  }
}

} // namespace lt
