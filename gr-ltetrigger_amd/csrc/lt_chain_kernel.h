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
#include <type_traits>

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
  int slot_quota;      // slots of the cache that are this chain's and nobody else's: [chain * slot_quota, (chain + 1) * slot_quota).
                       // What a chain gets out of the cache is then a function of its own state alone -- not of the order in
                       // which the workgroups of a launch reach an atomic --, and so are the bits of its records where the
                       // cached form rounds differently from the transforms (level 4).  slot_owner / slot_pos stay a work list
                       // in whatever order the chains came (lt_metric_kernel does not care)
  int pass;
  int *wait_hint;      // two words of pinned host memory (or null): [0] set to run_id by a chain that has enough searches ahead
  int run_id;          // of it to ask for the cache -- whether the run has one or not --, [1] by the kernel as such (pass 0):
                       // the host's cue to give the next runs two passes or one
  int stream;          // window cache level 4: an untracked chain asks for its lag-metric STREAM instead of windows (below)
};
// (kCacheStripLags, kCacheMinSearches, kCacheMaxPerChain, kStreamBlock / kStreamSlot / kStreamBack: lt_scan_kernels.h -- the last
// resolve launch of a run makes the chains' cache requests with them, chain_request_cache)
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

// The second half of a search: the average is up to date in the strips' registers; peak, lobe and side-lobe maximum
// (srslte_pss_find_pss after the average, oracle/lte_oracle.c:295-316).
#ifdef LT_PROF
__device__ unsigned int lt_prof_walks; // searches that took the walk over the whole average (all chains, all launches)
__device__ unsigned long long lt_prof_q[4]; // wave 0's cycles in strip_resolve / at the barrier / in resolve_window, and the searches
#endif
template <class Lds>
__device__ __forceinline__ SearchResult chain_search_resolve(Lds &L, int w, int lane, ema_t &ema, int &buf, bool resolve)
{
  SearchResult out;
  out.peak_pos = 0;
  out.peak_value = 0.f;
  out.psr = 0.f;
  if (!resolve) return out;
  StripPartial p;
#ifdef LT_PROF
  const unsigned long long q0 = __builtin_amdgcn_s_memtime();
#endif
  strip_resolve(ema, w, lane, p);
#ifdef LT_PROF
  const unsigned long long q1 = __builtin_amdgcn_s_memtime();
#endif
  StripPartial *parts = L.parts[buf];
  buf ^= 1;
  if (lane == 0) parts[w] = p;
  __syncthreads();
#ifdef LT_PROF
  const unsigned long long q2 = __builtin_amdgcn_s_memtime();
#endif
  WindowResult r;
  r = resolve_window_lanes(parts, lane); // (one LDS round trip and two wave reductions instead of two scalar loops over LDS: lt_scan.h)
#ifdef LT_PROF
  if (threadIdx.x == 0) {
    atomicAdd(&lt_prof_q[0], q1 - q0);
    atomicAdd(&lt_prof_q[1], q2 - q1);
    atomicAdd(&lt_prof_q[2], (unsigned long long)__builtin_amdgcn_s_memtime() - q2 + (r.peak_pos & 0)); // (after the window's results exist)
    atomicAdd(&lt_prof_q[3], 1ull);
  }
#endif
  out.peak_pos = __builtin_amdgcn_readfirstlane(r.peak_pos);
  out.peak_value = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(r.peak_value)));
  out.psr = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(r.psr)));
  if (__builtin_amdgcn_readfirstlane((int)r.exact)) return out;
#ifdef LT_PROF
  if (threadIdx.x == 0) atomicAdd(&lt_prof_walks, 1u);
#endif
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

// ---------------------------------------------------------------- lag-metric stream (window cache level 4)
// A window's lag metric at lag l, 127 <= l < 9600, is |sum_k h[k] x[w0 + l - k]|^2 with every sample inside the window:
// it depends on the absolute position w0 + l alone, not on where the window starts.  So for a chain whose grid may move
// with every hit (an untracked chain: idle, or building its score at marginal SNR) the window cache holds not windows
// but the chain's STREAM, S[i] = |sum_k h[k] x[o + i - k]|^2 for every sample position from o on -- overlap-save blocks
// of 896 outputs, 12 to a slot, formed by lt_metric_kernel like the windows' strips -- and whatever grid the chain is
// on, a search reads its interior lags out of it (one unaligned, coalesced load per register) and forms the 253 edge
// lags, whose sums stop at the window's first or last sample (lags 0..126 and 9600..9725), directly: 12 waves x 64 lanes,
// 22 taps of a pair of edge lags each.  The chains with a stream run in a kernel of their own between the two passes of
// lt_chain_kernel (lt_chain_stream_kernel = chain_run<true>: the same general_work, no transforms -- 57 KB of LDS
// instead of 140, nothing spilled); a window the stream does not hold ends their part, and pass 1 takes them up.
// Where a search's time goes there (cycle stamps of wave 0, -4 dB, 162 searches of one chain, 17.5 k cycles = 7.3 us each):
// edge lags 34 %, peak / lobe / side-lobe maximum 31 %, state machine and record 18 %, average 9 %, the stream's loads 9 %.
// (Those shares are of the kernel's first version: edge sums of sixteen masked vector instructions a tap, taps and samples
// from LDS, and the twelve partials combined by every wave in two scalar loops.)  Twelve waves of one workgroup on one CU are
// bound by what they issue and by what they move through LDS, so what paid was less of both: the edge sums without masks
// (zeros in LDS around the samples), the taps in scalar registers and two packed FMAs a tap (-7 % at -4 dB; without the
// scalar taps the unmasked sums read twice the LDS bytes and were 4 % slower), the twelve partials combined by twelve lanes
// (resolve_window_lanes: -6 %; in lt_chain_kernel too since: -3 % / -4 % at -4 / 0 dB, level 3).  What did not, three times over: loads taken a search early on the guess that the chain stays
// on its timing (right 63 % of the time at -4 dB) -- the stream's values parked in LDS (+4 %), the same as plain loads into
// registers nobody reads, to warm this XCD's L2 (+5 %), the edge samples alone, straight into their place in LDS (+3 %).
// The stream's blocks are not the window's strips: the same sums in another order of rounding.  Level 4 is therefore
// NOT bit-identical to levels 0..3 (which are, among themselves); it matches them, and the oracle, to the tolerance the
// parity tests give the device's own transform against the oracle's (tests/test_gpu_stream.py).
constexpr int kEdgeHead = kSymbolSz - 1;                // lags 0..126: the sum stops at the window's first sample
constexpr int kEdgeTail = kNLags - kHalfFrame;          // lags 9600..9725: ... at its last
constexpr int kEdgeShare = kChainWaves / 2;             // lanes (in as many waves) that share the 128 products of a pair of edge lags
constexpr int kEdgeTaps = (kSymbolSz + kEdgeShare - 1) / kEdgeShare; // 22 taps each
constexpr int kPadH = 72, kPadT0 = 8, kPadT = 88;
constexpr int kXh = kPadH, kXt = kPadH + kSymbolSz + kPadT0, kXsLen = kXt + kSymbolSz + kPadT;
struct StreamLds {
  // the window's first and last 127 samples, with zeros where a lane's taps reach past them: a head sum's taps beyond its
  // lag (k > l) read the zeros in front of the head samples, a tail sum's taps up to its lag (k <= l) those behind the
  // tail's -- the sums need no masks.  (The pads are as wide as the lanes of one wave are apart from their taps.)
  v2f xs[kXsLen]; // xs[kXh + i] = window sample i, xs[kXt + i] = window sample 9473 + i, i < 127; zeros elsewhere
  v2f part[2][kChainWaves / 2][kSymbolSz]; // [head, tail][share of the taps][unit]
};
typedef const __attribute__((address_space(4))) unsigned long long *const_taps_t; // (re, im) pairs in constant memory

// acc += sum over kEdgeTaps taps of tap * sample, the taps wave-uniform (scalar registers), the samples at p[kEdgeTaps - 1 - i]
// in LDS: per tap two packed FMAs -- (re, im) += h.re * (x.re, x.im); (re, im) += (-h.im, h.im) * (x.im, x.re) -- which is the
// per-component order of the oracle's complex product (re: h.re x.re - h.im x.im; im: h.re x.im + h.im x.re).
__device__ __forceinline__ v2f edge_sum(const_taps_t taps, const v2f *p)
{
  v2f acc = {0.f, 0.f};
  constexpr int kNB = 2, kB = (kEdgeTaps + kNB - 1) / kNB;
#pragma unroll
  for (int b = 0; b < kNB; ++b) {
    unsigned long long h[kB];
    v2f x[kB];
#pragma unroll
    for (int i = 0; i < kB; ++i) {
      if (kB * b + i >= kEdgeTaps) continue;
      h[i] = taps[kB * b + i];
      x[i] = p[kEdgeTaps - 1 - (kB * b + i)];
    }
#pragma unroll
    for (int i = 0; i < kB; ++i) {
      if (kB * b + i >= kEdgeTaps) continue;
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(h[i]), "v"(x[i]));
      asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "s"(h[i]), "v"(x[i]));
    }
  }
  return acc;
}

// the 1024 samples of the overlap-save block that starts at absolute sample a0 (zero outside the capture)
__device__ __forceinline__ void stream_block_load(const Capture &c, int64_t a0, int lane, fft::c32 (&x)[16])
{
  const int64_t lo = a0 - c.t0;
  if (lo >= 0 && lo + fft::kN <= c.n) {
    const v2f *p = c.iq + lo + lane;
#pragma unroll
    for (int R = 0; R < 16; ++R) x[R] = p[64 * R];
  } else {
#pragma unroll
    for (int R = 0; R < 16; ++R) x[R] = ld_abs(c, a0 + lane + 64 * R);
  }
}

// met[r] = lag metric of lag c0 + lane + 64 r of the window at w0; s0 = the stream element of the window's lag 0 (all of
// [s0, s0 + 9726) exists).  Two workgroup barriers.
__device__ __forceinline__ void stream_metric(StreamLds &S, const Capture &cap, int64_t w0, const float *__restrict__ s0,
                                              const_taps_t taps, int w, int lane, float (&met)[kStripRegs])
{
  const int c0 = kStripPitch * w - kStripHalo;
#pragma unroll
  for (int r = 0; r < kStripRegs; ++r) {
    int idx = c0 + lane + 64 * r;
    idx = idx < 0 ? 0 : idx > kNLags - 1 ? kNLags - 1 : idx; // (lags outside [0, 9726) are dropped by strip_ema_apply)
    met[r] = s0[idx];
  }
  const int tid = 64 * w + lane;
  if (tid < 2 * kEdgeHead) {
    const bool head = tid < kEdgeHead;
    const int j = head ? tid : tid - kEdgeHead;
    const int i = head ? j : kHalfFrame - kEdgeHead + j; // the window's sample
    const int64_t lo = w0 - cap.t0;
    S.xs[(head ? kXh : kXt) + j] = (lo >= 0 && lo + kHalfFrame <= cap.n) ? cap.iq[lo + i] : ld_abs(cap, w0 + i);
  }
  __syncthreads();
  {
    // The partial sums of the workgroup's 253 edge lags, wave w's share (between the two barriers of stream_metric).
    // Head lag l (taps 0..l) and tail lag 9600 + l (taps l + 1..127) make one unit of exactly 128 products, l = 0..126;
    // six lanes -- in six waves -- share a unit, 22 consecutive taps each (the last 18): every lane of the workgroup
    // has the same work.  Wave w: units 64 (w / 6) + lane, taps from 22 (w % 6) on.
    const int half = w / kEdgeShare, j = w - kEdgeShare * half;
    const int l = 64 * half + lane, k0 = kEdgeTaps * j;
    // Tap k belongs to the head sum of the lanes with l >= k and to the tail sum of the others.  Seven of the twelve waves
    // have lanes of both kinds and form both sums over all their taps -- the taps of the wrong kind meet zeros --; in the
    // other five every tap is a tail tap (units 0..63, taps from 66 on) or a head tap (units 64..127, taps up to 43).
    const int l_lo = 64 * half, k_hi = k0 + kEdgeTaps - 1 < kSymbolSz - 1 ? k0 + kEdgeTaps - 1 : kSymbolSz - 1;
    const int d = l - k0 - (kEdgeTaps - 1);
    v2f hs = {0.f, 0.f}, ts = {0.f, 0.f};
    if (k0 <= l_lo + 63) hs = edge_sum(taps + k0, S.xs + kXh + d);            // sample l - k
    if (k_hi > l_lo) ts = edge_sum(taps + k0, S.xs + kXt + kEdgeHead + d);    // sample 9473 + (127 + l - k)
    const float hr = hs.x, hi = hs.y, tr = ts.x, ti = ts.y;
    S.part[0][j][l] = v2f{hr, hi};
    S.part[1][j][l] = v2f{tr, ti};
  }
  __syncthreads();
  if (w == 0 || w == kStrips - 1) {
    const int first = w == 0 ? 0 : kHalfFrame, n = w == 0 ? kEdgeHead : kEdgeTail;
    const v2f(*part)[kSymbolSz] = S.part[w == 0 ? 0 : 1];
#pragma unroll
    for (int r = 0; r < kStripRegs; ++r) {
      if (!((w == 0 && r < 3) || (w != 0 && r >= kStripRegs - 3))) continue;
      const int e = c0 + lane + 64 * r - first; // index among the edge lags = the unit
      if (e >= 0 && e < n) {
        const v2f p0 = part[0][e], p1 = part[1][e], p2 = part[2][e], p3 = part[3][e], p4 = part[4][e], p5 = part[5][e];
        const float yr = ((p0.x + p1.x) + (p2.x + p3.x)) + (p4.x + p5.x), yi = ((p0.y + p1.y) + (p2.y + p3.y)) + (p4.y + p5.y);
        met[r] = yr * yr + yi * yi;
      }
    }
  }
}

// srslte_pss_find_pss on the window whose strip samples are in x (strip_load).  resolve = false: lag metric and
// exponential average only (a window the scan has already turned into a record).  Contains workgroup
// barriers; `buf` alternates between calls.
// cached: the window's lag metric as lt_metric_kernel left it (12 x 896 floats) -- the transforms are skipped, the
// average and everything after it run as ever (x is not looked at then).
// (Round 3 pulled the NEXT slot's lines into the L2 during a cached search, with fourteen loads into one register that
// nobody reads, left in flight past their block of assembly -- a register the compiler is free to reuse under them
// (an instrumented build of this kernel faulted).  Gone without a trace in round 4: measured on one box against
// the version with them, per-chain kernel of a 64-channel step at -4 / 0 dB: 1.92 / 1.21 ms without, 1.91 / 1.21 with.  Safe
// prefetches that were measured and do not pay either: the same loads closed by a wait of their own, +15 %: the search then
// waits for the next slot's HBM latency; one or two LDS-DMA loads, a sector per lane, into a dump area, +8 %; LDS-DMA of the
// whole strip into the wave's idle exchange area, +3 %: an LDS-DMA instruction costs the wave ~300 cycles to issue.)
__device__ __forceinline__ SearchResult chain_search(ChainLds &L, fft::c32 (&x)[16], int w, int lane, ema_t &ema, int &buf,
                                                     bool resolve, const float *cached = nullptr)
{
  fft::c32 *e = L.E + w * fft::kScratchW;
  if (cached) {
    float met[kStripRegs];
    const float *m = cached + (size_t)w * kCacheStripLags + lane;
#pragma unroll
    for (int r = 0; r < kStripRegs; ++r) met[r] = m[64 * r];
    strip_ema_apply(w, lane, met, ema);
  } else {
    strip_fft<false>(lane, x, L.TW1, L.TW2, e);
    strip_root_update(w, lane, x, L.SPEC, L.TW1, L.TW2, e, ema);
  }
  return chain_search_resolve(L, w, lane, ema, buf, resolve);
}

// LDS of the stream kernel (lt_chain_stream_kernel): no transforms, so no exchange areas, twiddles or spectrum -- the
// mirror of the exact path, the strips' partial results and what the edge lags need
struct LeanLds {
  alignas(16) v2f E[(kConvLen + 8 + 1) / 2]; // (named as in ChainLds: chain_search_resolve's mirror of the average)
  StripPartial parts[2][kStrips];
  float wave_out[kChainWaves];
  StreamLds stream;
};
// Diagnostics build (-DLT_PROF): shader-clock stamps of wave 0 around the phases of a search out of the stream, summed over
// the chain's searches and printed per chain (device printf) -- where DESIGN 4.3's shares come from.
#ifdef LT_PROF
#define LT_STAMP(i) (prof_t[i] = __builtin_amdgcn_s_memtime())
#else
#define LT_STAMP(i) ((void)0)
#endif
// kLean: the stream kernel's version (window cache level 4, between pass 0 and pass 1 of the kernel proper) -- the chains
// with a stream only, every search out of the stream, no transforms (and none of their registers and LDS); at a window the
// stream does not hold it puts the chain down as it stands, for pass 1 to take up
template <bool kLean> __device__ __forceinline__ void chain_run(const ChainArgs &a)
{
  __shared__ std::conditional_t<kLean, LeanLds, ChainLds> L;
  __shared__ float psr_data[kMovingAvgSz];
  constexpr int NT = kChainThreads;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chain = blockIdx.x, channel = chain / 3, n_id_2 = chain % 3;
  ChainState *st = a.state + chain;
  if (st->run_done) return; // the resolver has seen the chain stop for this run
  if constexpr (kLean) {
    if (!(a.cache && st->cache_n > 0 && st->cache_stride == 0)) return;
  }

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
  if (!kLean && (a.cache || a.wait_hint) && a.pass == 0 && can_call(pos) && nrec < a.rec_cap) {
    const int stride = st->is_tracking ? a.track_every + 1 : 1;
    const int64_t first = pos + (st->is_tracking ? (int64_t)st->timer * kHalfFrame : 0);
    const int64_t last_start = a.flush ? t1 - kHalfFrame : t1 - kCallReach; // last position a call may start at
    int64_t ahead = first <= last_start ? (last_start - first) / ((int64_t)stride * kHalfFrame) + 1 : 0;
    if (ahead > kCacheMaxPerChain) ahead = kCacheMaxPerChain;
    if (ahead >= kCacheMinSearches && a.wait_hint && tid == 0) *reinterpret_cast<volatile int *>(a.wait_hint) = a.run_id;
    {
      // level 4, an untracked chain: its stream from a little in front of where it stands to the last lag of the last
      // window it can search -- as far as kCacheMaxPerChain slots reach (beyond them: the transforms)
      if (a.cache && a.stream && ahead >= kCacheMinSearches && !st->is_tracking) {
        __shared__ int s_sbase;
        // (from the first of the windows its average still has to go through, if there are any: they come first)
        int64_t o = first - kStreamBack;
        if (st->pre_n > 0 && st->pre_pos < o) o = st->pre_pos;
        int64_t want = (last_start + kNLags - o + kStreamSlot - 1) / kStreamSlot;
        if (want > kCacheMaxPerChain) want = kCacheMaxPerChain;
        const int got = a.slot_quota < (int)want ? a.slot_quota : (int)want; // (beyond its slots: pass 1, through the transforms)
        if (got >= 2) {
          if (tid == 0) s_sbase = atomicAdd(a.slot_next, got); // a place on the work list (never short: the quotas add up to slot_cap)
          __syncthreads();
          const int wi = s_sbase;
          for (int i = tid; i < got; i += NT) {
            a.slot_owner[wi + i] = make_int2(chain, -1 - i); // (second member negative: a slot of a stream)
            a.slot_pos[wi + i] = o + (int64_t)i * kStreamSlot;
          }
          if (tid == 0) {
            atomicAdd(a.launch_slots, got);
            atomicAdd(a.slot_next + 3, 1); // diagnostics
            st->cache_pos = o;
            st->cache_base = chain * a.slot_quota;
            st->cache_n = got;
            st->cache_stride = 0; // (0: a stream)
          }
          return;
        }
        ahead = 0; // (no room: nothing to ask for)
      }
    }
    if (a.cache && ahead >= kCacheMinSearches) {
      __shared__ int s_base;
      const int got = a.slot_quota < (int)ahead ? a.slot_quota : (int)ahead;
      if (got >= kCacheMinSearches) {
        if (tid == 0) s_base = atomicAdd(a.slot_next, got);
        __syncthreads();
        const int wi = s_base;
        for (int i = tid; i < got; i += NT) {
          a.slot_owner[wi + i] = make_int2(chain, i);
          a.slot_pos[wi + i] = first + (int64_t)i * stride * kHalfFrame;
        }
        if (tid == 0) {
          atomicAdd(a.launch_slots, got);
          atomicAdd(a.slot_next + 3, 1); // diagnostics
          st->cache_pos = first;
          st->cache_base = chain * a.slot_quota;
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
  // (a chain with a stream, level 4: the same variables, cache_step = 0 -- cache_pos is where the stream starts, cache_n its slots)
  if (a.cache && a.pass == 1 && st->cache_n > 0 && (kLean || st->cache_stride > 0)) {
    cache0 = a.cache + (size_t)st->cache_base * cache_slot_floats();
    cache_pos = st->cache_pos;
    cache_step = (int64_t)st->cache_stride * kHalfFrame;
    cache_n = st->cache_n;
  }
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
  if constexpr (kLean) {
    if (tid < kXsLen) L.stream.xs[tid] = v2f{0.f, 0.f}; // (the pads: no search writes them)
  } else {
    chain_lds_init(L, a.tab, n_id_2);
  }
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

#ifdef LT_PROF
  unsigned long long prof_t[5] = {0, 0, 0, 0, 0}, prof_sum[5] = {0, 0, 0, 0, 0}, prof_first = 0;
#endif
  for (;;) {
#ifdef LT_PROF
    if (kLean && prof_t[3]) { // the search before: its phases, and what followed it up to here
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      prof_sum[0] += prof_t[1] - prof_t[0]; prof_sum[1] += prof_t[2] - prof_t[1]; prof_sum[2] += prof_t[3] - prof_t[2];
      prof_sum[3] += now - prof_t[3]; prof_sum[4] += 1;
      if (!prof_first) prof_first = prof_t[0];
      prof_t[4] = now; prof_t[3] = 0;
    }
#endif
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
      if constexpr (!kLean) asm volatile("" : "+v"(ln)); // (the stream kernel has the registers: its lane-dependent offsets stay)
      SearchResult r;
      if constexpr (kLean) {
        const int64_t d = wpos - cache_pos;
        if (!(d >= 0 && d + kNLags <= (int64_t)cache_n * kStreamSlot)) break; // not in the stream: pass 1's, through the transforms
        if (tid == 0) atomicAdd(a.slot_next + (redo ? 5 : 1), 1); // diagnostics
        float met[kStripRegs];
        LT_STAMP(0);
        stream_metric(L.stream, cap, wpos, cache0 + d, (const_taps_t)(uintptr_t)a.tab->pss_taps_pad[n_id_2], w, ln, met);
        LT_STAMP(1);
        strip_ema_apply(w, ln, met, ema);
        LT_STAMP(2);
        r = chain_search_resolve(L, w, ln, ema, buf, !redo);
        LT_STAMP(3);
      } else {
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
        r = chain_search(L, x, w, ln, ema, buf, !redo, cached);
      }
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

#ifdef LT_PROF
  if (kLean && tid == 0 && prof_sum[4] > 100)
    printf("PROF chain %d searches %llu: stream+edges %llu average %llu resolve %llu rest %llu of %llu cycles; walks so far (all chains) %u; all chains so far: strip_resolve %llu barrier %llu resolve_window %llu of %llu searches\n", chain, prof_sum[4],
           prof_sum[0], prof_sum[1], prof_sum[2], prof_sum[3], prof_t[4] - prof_first, lt_prof_walks, lt_prof_q[0], lt_prof_q[1], lt_prof_q[2], lt_prof_q[3]);
#endif
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
__device__ __forceinline__ void chain_kernel_body(const ChainArgs &a)
{
  chain_run<false>(a);
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
    int lc = lane; // (opaque per turn, like the ring's below: nothing lane-dependent of the estimate is kept across the turns)
    asm volatile("" : "+v"(lc));
    for (int g = w * 8 + (lc >> 3); g < n_ran; g += kChainWaves * 8) {
      const int local = t_idx[g];
      const float c = cfo_estimate_group8(taps, cap, t_pos[g] + (kSlotLen - kSymbolSz), lc);
      if ((lc & 7) == 0) {
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
__global__ __launch_bounds__(kChainThreads) void lt_chain_kernel(ChainArgs a) { chain_kernel_body(a); }
// Window cache level 4, between the two passes of the kernel above: the chains that asked for their lag-metric stream run
// general_work on it as far as it reaches (chain_run<true>); everything else -- their tails too -- is pass 1's.
__global__ __launch_bounds__(kChainThreads) void lt_chain_stream_kernel(ChainArgs a) { chain_run<true>(a); }

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
  int quota;               // > 0: the list's entry (chain, k) is stored in the chain's own slots, chain * quota + k (the per-chain
                           // kernel's requests); 0: in the slot of its place on the list (the scan rounds of small batches)
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
    const int2 owner = a.slot_owner[slot];
    const int chain = __builtin_amdgcn_readfirstlane(owner.x);
    const bool of_stream = __builtin_amdgcn_readfirstlane(owner.y) < 0; // (level 4: twelve blocks of a chain's stream)
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
    // (a stream's block: outputs 128..1023 are the stream elements wpos + 896 strip + 0..895)
    if (of_stream) stream_block_load(cap, wpos + (int64_t)strip * kStreamBlock - (fft::kN - kStreamBlock), ln, x);
    else strip_load(cap, wpos, strip, ln, x);
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
    const int k_own = __builtin_amdgcn_readfirstlane(owner.y);
    const int store = a.quota > 0 ? chain * a.quota + (of_stream ? -1 - k_own : k_own) : slot;
    float *out = a.cache + (size_t)store * cache_slot_floats() + (size_t)strip * kCacheStripLags + ln;
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
  const SearchResult r = chain_search(L, x, w, lane, ema, buf, true);
  strip_ema_store(avg, w, lane, ema);
  if (tid == 0) {
    res->peak_pos = r.peak_pos;
    res->psr = r.psr;
    res->peak_value = r.peak_value;
  }
}

} // namespace lt
