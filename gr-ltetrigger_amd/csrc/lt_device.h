// Device-side building blocks of the gfx950 LTE cell-search path.
//
// Work decomposition (one workgroup = one pss->sss chain = one (channel, N_id_2)):
//   fused kernel: 512 threads (8 wave64), thread `tid` owns the 19 consecutive lags [19*tid, 19*tid+19) of
//   the 9728-lag linear convolution of a 9600-sample half-frame window with the 128-tap PSS filter
//   (512 * 19 = 9728 exactly); the window sits zero-padded in LDS.
//   replay kernel: 256 threads, thread `tid` owns lags tid, tid + 256, ... (38 of them), the order the
//   precomputed lag metric arrives in.
//   Either way the EMA of the lag metric lives in registers and never touches HBM between windows.
//
// Arithmetic rules: the library is built with -ffp-contract=off; the only fused operations are
// the explicit FMAs of the matched filter and of the CFO rotation.  Everything downstream of the
// lag metric is evaluated in the oracle's operation order so that it is bit-exact against it.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>
#include <cstdint>

#include "../../include/ltetrigger_hip.h"
#include "lt_fft.h"
#include "lt_tables.h"

namespace lt {

typedef float v2f __attribute__((ext_vector_type(2)));

// Every device function is inlined.  (Out-of-line functions taking generic pointers to LDS were tried and
// dropped: the compiler may rebase such a pointer below the start of the array and add the difference
// back as an instruction offset; for an array at LDS offset 0 the rebased flat address leaves the LDS
// aperture and the access faults.)

constexpr int kThreads = 512;
constexpr int kThreadsReplay = 256; // workgroup of the replay kernel (bulk + replay mode)
constexpr int kWaves = kThreads / 64;
constexpr int kT = 19;                              // lags per thread
constexpr int kWPad = 128;                          // zero samples in front of the window in LDS
constexpr int kWLen = 10240;                        // 81920 B: window + zero pads; doubles as FFT scratch (9 x 8448 B)
constexpr int kAvgLen = kConvLen + 4;               // EMA mirror in LDS, [9728] is the always-zero lag
constexpr int kSssBase = 480;                       // first half-frame sample the sss stage touches
constexpr int kSssLen = kSlotLen - kSssBase;        // 480 samples [480, 960)
constexpr int kBlkValid = fft::kN - (kSymbolSz - 1);  // 897 valid lags per 1024-point overlap-save block
constexpr int kBlocks = (kConvLen + kBlkValid - 1) / kBlkValid; // 11
static_assert(kThreads * kT == kConvLen, "lag tiling must cover the convolution exactly");
static_assert(kWPad + (kBlocks - 1) * kBlkValid - (kSymbolSz - 1) + fft::kN <= kWLen, "window buffer too short");
constexpr int kScratchInW = kWLen / fft::kScratch;    // 9 transforms' scratch fit in the window buffer
static_assert(kBlocks <= 2 * kWaves, "one half-wave per overlap-save block");
static_assert((kBlocks - kScratchInW) * fft::kScratch * 2 <= kAvgLen, "remaining scratch must fit in the metric buffer");

// Persistent state of one chain (pss_impl + sss_impl members, lib/pss_impl.h:41-81,
// lib/sss_impl.h:42-44), kept in HBM between launches.
struct ChainState {
  int64_t pos;          // absolute index of the first unconsumed sample
  int32_t is_tracking;  // tracking_t::is_tracking
  int32_t score;        // tracking_t::score
  int32_t timer;        // tracking_t::timer
  int32_t tracking_lost;
  int32_t peak_pos;     // d_peak_pos
  uint32_t psr_i;       // d_psr_i
  uint32_t cfo_i;       // d_cfo_i
  int32_t n_calls;      // general_work calls so far
  int32_t n_emits;      // half-frames produced so far
  float psr;            // d_psr
  float psr_max;        // d_psr_max
  float m_norm_avg;     // srslte_sync_t::M_norm_avg
  float m_ext_avg;      // srslte_sync_t::M_ext_avg
  int32_t ema_sel;      // which of avg[2] is current (the grid scan writes the other one, lt_scan_kernels.h)
  int32_t scan_live;    // this run: the chain is (still) served by the grid scan
  int32_t scan_job;     // this run: its job, -1 = none
  int32_t scan_done;    // this run: windows of the job the resolver has turned into records
  int32_t scan_seg0;    // this run: first window of the scan segment avg[ema_sel] is the state before
  float psr_data[kMovingAvgSz];
  float cfo_data[kMovingAvgSz];
  float avg[2][kConvLen]; // srslte_pss_t::conv_output_avg (lags >= 9726 stay 0), natural lag order
};

// LDS scratch for block reductions and the sss stage.
struct Scratch {
  float tmax[kThreads]; // per-thread maximum of the EMA over its valid lags
  int tidx[kThreads];   // lag of that maximum
  v2f Y[62];           // occupied bins of the SSS symbol's FFT
  float cfo, mean_cfo; // wave 0 -> everyone
};

struct SearchResult {
  int peak_pos;
  float peak_value;
  float psr;
};

__device__ __forceinline__ v2f ld_iq(const v2f *__restrict__ x, int64_t n, int64_t i)
{
  // "zero history at stream start" (set_history, lib/pss_impl.cc:81); nothing past the capture
  v2f z = {0.f, 0.f};
  return (i < 0 || i >= n) ? z : x[i];
}

// acc += h * u (complex), two packed FMAs: (hx,hx)*(ux,uy) and (-hy,hy)*(uy,ux)
__device__ __forceinline__ v2f cmac(v2f acc, v2f h, v2f u)
{
  v2f hxx = {h.x, h.x};
  v2f hyy = {-h.y, h.y};
  v2f uyx = {u.y, u.x};
  acc = __builtin_elementwise_fma(hxx, u, acc);
  acc = __builtin_elementwise_fma(hyy, uyx, acc);
  return acc;
}

// 19 lags of the 128-tap matched filter, using the even symmetry h[m] = h[128-m] of the PSS
// filter (d[n] = d[61-n] in 36.211 6.11.1.1):
//   conv[n] = h[0] w[n] + h[64] w[n-64] + sum_{m=1..63} h[m] (w[n-m] + w[n-128+m])
// Wp points at w[n0] inside the zero-padded LDS window; taps = h[0..64] (wave-uniform).
__device__ __forceinline__ void conv19(const v2f *Wp, const v2f *__restrict__ taps, v2f acc[kT])
{
  {
    const v2f h0 = taps[0], h64 = taps[64];
#pragma unroll
    for (int t = 0; t < kT; ++t) {
      v2f z = {0.f, 0.f};
      z = cmac(z, h0, Wp[t]);
      acc[t] = cmac(z, h64, Wp[t - 64]);
    }
  }
  for (int c = 0; c < 7; ++c) { // taps m0 .. m0+8
    const int m0 = 1 + 9 * c;
    const v2f *A = Wp - m0 - 8;   // A[i] = w[n0 - m0 - 8 + i];  w[n0+t-(m0+j)]     = A[t-j+8]
    const v2f *B = Wp - 128 + m0; // B[i] = w[n0 - 128 + m0 + i]; w[n0+t-128+(m0+j)] = B[t+j]
    v2f a[kT + 8], b[kT + 8];
#pragma unroll
    for (int i = 0; i < kT + 8; ++i) {
      a[i] = A[i];
      b[i] = B[i];
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const v2f h = taps[m0 + j];
#pragma unroll
      for (int t = 0; t < kT; ++t) acc[t] = cmac(acc[t], h, a[t - j + 8] + b[t + j]);
    }
  }
}

__device__ __forceinline__ void argmax_combine(float &v, int &i, float ov, int oi)
{
  if (ov > v || (ov == v && oi < i)) {
    v = ov;
    i = oi;
  }
}

// ---- wavefront reductions on the VALU cross-lane path (DPP row rotates + 4 readlanes) instead of
// ds_bpermute: every lane of a 16-lane row ends up with the row's result, the four rows are then
// combined through scalar registers, so the result is wave-uniform.
template <int kCtrl> __device__ __forceinline__ int dpp_i(int v)
{
  return __builtin_amdgcn_update_dpp(v, v, kCtrl, 0xf, 0xf, false);
}
#define LT_ROW_ALLREDUCE(v, OP)                 \
  v = OP(v, dpp_i<0x128>(v)); /* row_ror:8 */  \
  v = OP(v, dpp_i<0x124>(v)); /* row_ror:4 */  \
  v = OP(v, dpp_i<0x122>(v)); /* row_ror:2 */  \
  v = OP(v, dpp_i<0x121>(v)); /* row_ror:1 */

__device__ __forceinline__ float rlane_f(float v, int lane)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int imin_(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax_(int a, int b) { return a > b ? a : b; }
// max of non-NaN floats >= -FLT_MAX, through their ordered bit patterns is not possible for negative
// values, so compare as floats on reinterpreted lanes
__device__ __forceinline__ int fmax_bits_(int a, int b) { return __int_as_float(a) >= __int_as_float(b) ? a : b; }

__device__ __forceinline__ int wave_min_i(int v)
{
  LT_ROW_ALLREDUCE(v, imin_)
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return imin_(imin_(a, b), imin_(c, d));
}
__device__ __forceinline__ int wave_max_i(int v)
{
  LT_ROW_ALLREDUCE(v, imax_)
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return imax_(imax_(a, b), imax_(c, d));
}
// maximum of floats none of which is NaN (callers pass -FLT_MAX for "nothing"): v_max_f32 with a DPP
// operand per step, the four row results through v_readlane
__device__ __forceinline__ float wave_max_f(float f)
{
  f = fmaxf(f, __int_as_float(dpp_i<0x128>(__float_as_int(f)))); // row_ror:8
  f = fmaxf(f, __int_as_float(dpp_i<0x124>(__float_as_int(f)))); // row_ror:4
  f = fmaxf(f, __int_as_float(dpp_i<0x122>(__float_as_int(f)))); // row_ror:2
  f = fmaxf(f, __int_as_float(dpp_i<0x121>(__float_as_int(f)))); // row_ror:1
  const float a = rlane_f(f, 0), b = rlane_f(f, 16), c = rlane_f(f, 32), d = rlane_f(f, 48);
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// srslte_pss_find_pss on the window held in W (LDS, zero padded): lag metric, EMA(0.2),
// arg-max, main-lobe walk, peak-to-side-lobe ratio (call site lib/pss_impl.cc:165).
// `ema` is this thread's slice of conv_output_avg.  Contains __syncthreads().
// value of lane (l ^ kMask): DPP quad permutes for 1 and 2, ds_swizzle (bit mode) up to 16,
// ds_bpermute for 32
template <int kMask> __device__ __forceinline__ int lane_xor_i(int v)
{
  if (kMask == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false); // quad_perm [1,0,3,2]
  if (kMask == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false); // quad_perm [2,3,0,1]
  if (kMask == 32) return __shfl_xor(v, 32);
  return __builtin_amdgcn_ds_swizzle(v, (kMask << 10) | 0x1F); // and 0x1f, or 0, xor kMask
}
template <int kMask> __device__ __forceinline__ float lane_xor(float v)
{
  return __int_as_float(lane_xor_i<kMask>(__float_as_int(v)));
}
template <int kMask> __device__ __forceinline__ double lane_xor_d(double v)
{
  return __hiloint2double(lane_xor_i<kMask>(__double2hiint(v)), lane_xor_i<kMask>(__double2loint(v)));
}

// Overlap-save matched filter: |conv[lag]|^2 for all 9728 lags of the window in W, written to
// `metric` (LDS).  Block b (1024 samples starting 127 before lag 897 b) is transformed by one
// half-wave (csrc/lt_fft.h): FFT-1024, multiply by the filter spectrum, inverse FFT, keep the
// 897 alias-free outputs; 11 blocks = 11 of the 16 half-waves, one
// round.  Once every half-wave holds its 1024 inputs in registers, the window buffer (and, for the
// last two blocks, the still unused metric buffer) serves as the FFT exchange scratch.  Three
// __syncthreads() inside; W is garbage afterwards, `metric` complete.
__device__ __forceinline__ void conv_metric_fft(v2f *W, float *metric, const v2f *tw, const v2f *spec,
                                                unsigned long long *prof = nullptr)
{
  unsigned long long tq_ = prof ? __builtin_amdgcn_s_memtime() : 0;
#define LT_FSTAMP(i)                                             \
  if (prof) {                                                    \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
    prof[i] += t_ - tq_;                                         \
    tq_ = t_;                                                    \
  }
  const int l = threadIdx.x & 31, b = threadIdx.x >> 5; // lane within the half-wave, block index
  const bool active = b < kBlocks;
  fft::c32 v[32];
  if (active) {
    const v2f *p = W + kWPad + b * kBlkValid - (kSymbolSz - 1) + l;
#pragma unroll
    for (int r = 0; r < 32; ++r) v[r] = p[32 * r];
  }
  __syncthreads(); // every half-wave holds its inputs: W may now be overwritten
  LT_FSTAMP(14)
  if (active) {
    fft::c32 *e = (b < kScratchInW) ? W + b * fft::kScratch
                                    : reinterpret_cast<fft::c32 *>(metric) + (b - kScratchInW) * fft::kScratch;
    fft::fft1024_stage1<false>(l, v, tw, e);
    fft::fft1024_stage2<false>(l, v, e);
#pragma unroll
    for (int r = 0; r < 32; ++r) v[r] = fft::cmul<false>(v[r], spec[r * 32 + l]);
    fft::fft1024_stage1<true>(l, v, tw, e);
    fft::fft1024_stage2<true>(l, v, e);
#pragma unroll
    for (int r = 0; r < 32; ++r) v[r].x = __fadd_rn(__fmul_rn(v[r].x, v[r].x), __fmul_rn(v[r].y, v[r].y));
  }
  LT_FSTAMP(15)
  __syncthreads(); // no scratch in use any more: the metric buffer may be written
  if (active) {
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const int i = l + 32 * r;
      const int lag = b * kBlkValid + i - (kSymbolSz - 1);
      if (i >= kSymbolSz - 1 && lag < kConvLen) metric[lag] = v[r].x;
    }
  }
  __syncthreads(); // metric[] complete
  LT_FSTAMP(16)
#undef LT_FSTAMP
}

#define LT_STAMP(i)                                              \
  if (prof) {                                                    \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
    prof[i] += t_ - t_prev;                                      \
    t_prev = t_;                                                 \
  }
// Part 1 (search_front): matched filter, EMA, LDS mirror, ends with the block-wide barrier; after it
// the window buffer W is no longer read.  Part 2 (search_tail): arg-max, walk, side lobes, PSR.
struct FilterLds {
  const v2f *tw;   // FFT twiddles W1024^(reg*lane), [32][32], in LDS
  const v2f *spec; // filter spectrum / 1024, natural order, in LDS
};

__device__ __forceinline__ void search_ema(const float met[kT], float *AVG, float ema[kT], Scratch &sc, int ablate,
                                           unsigned long long *prof);

__device__ __forceinline__ void search_front(v2f *W, float *AVG, float ema[kT],
                                             const v2f *__restrict__ taps, const FilterLds &fl, Scratch &sc,
                                             int ablate, unsigned long long *prof = nullptr)
{
  unsigned long long t_prev = prof ? __builtin_amdgcn_s_memtime() : 0;
  const int tid = threadIdx.x;
  const int n0 = tid * kT;

  // lag metric |conv|^2 of this thread's 19 lags
  float met[kT];
  if (ablate & 2) { // diagnostics: no matched filter
#pragma unroll
    for (int t = 0; t < kT; ++t) met[t] = W[kWPad + n0 + t].x;
  } else if (ablate & 16) { // direct-form reference path (folded 128-tap FIR), kept for A/B checks
    v2f acc[kT];
    conv19(W + kWPad + n0, taps, acc);
#pragma unroll
    for (int t = 0; t < kT; ++t) met[t] = __fadd_rn(__fmul_rn(acc[t].x, acc[t].x), __fmul_rn(acc[t].y, acc[t].y));
  } else {
    conv_metric_fft(W, AVG, fl.tw, fl.spec, prof); // AVG doubles as the staging area of the metric
#pragma unroll
    for (int t = 0; t < kT; ++t) met[t] = AVG[n0 + t];
  }
  LT_STAMP(0) // matched filter
  search_ema(met, AVG, ema, sc, ablate, prof);
}

// EMA of the lag metric (three roundings like srsLTE's three vector passes), LDS mirror, per-thread
// maxima; ends with the block-wide barrier.  met[t] = |conv|^2 of lag 19 tid + t.
__device__ __forceinline__ void search_ema(const float met[kT], float *AVG, float ema[kT], Scratch &sc, int ablate,
                                           unsigned long long *prof)
{
  unsigned long long t_prev = prof ? __builtin_amdgcn_s_memtime() : 0;
  const int tid = threadIdx.x;
  const int n0 = tid * kT;
  if (ablate & 4) { // diagnostics: no reductions
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < kT; ++t) {
      ema[t] = met[t];
      s += ema[t];
    }
    AVG[n0] = s;
    __syncthreads();
    return;
  }
  // EMA with three roundings like srsLTE's three vector passes, local first-maximum
  float bv = -FLT_MAX;
  int bi = INT_MAX;
#pragma unroll
  for (int t = 0; t < kT; ++t) {
    const int lag = n0 + t;
    const float a = met[t];
    if (lag < kNLags) {
      ema[t] = __fadd_rn(__fmul_rn(a, 0.2f), __fmul_rn(ema[t], 1.0f - 0.2f));
      if (ema[t] > bv) {
        bv = ema[t];
        bi = lag;
      }
    }
    AVG[lag] = ema[t];
  }
  sc.tmax[tid] = bv; // per-thread maximum over its 19 lags (valid lags only) and where it sits
  sc.tidx[tid] = bi;
  LT_STAMP(1) // EMA + LDS mirror
  __syncthreads(); // the only block-wide synchronisation of the search: AVG, tmax, tidx are visible
  LT_STAMP(2) // barrier wait
}

__device__ __forceinline__ SearchResult search_tail(const float *AVG, const Scratch &sc, int ablate,
                                                   unsigned long long *prof = nullptr)
{
  const int lane = threadIdx.x & 63;
  unsigned long long t_prev = prof ? __builtin_amdgcn_s_memtime() : 0;
  if (ablate & 4) {
    SearchResult r0;
    r0.peak_pos = __builtin_amdgcn_readfirstlane((int)AVG[4] & 1023);
    r0.peak_value = 1.f;
    r0.psr = 1.f;
    return r0;
  }
  // Every wave redundantly derives the same (wave-uniform) results from LDS, so no further
  // barrier is needed to broadcast them.
  // ---- arg-max: first maximum = largest value, smallest lag among equals
  float bv = -FLT_MAX;
  int bi = INT_MAX;
#pragma unroll
  for (int k = 0; k < kThreads / 64; ++k) argmax_combine(bv, bi, sc.tmax[lane + 64 * k], sc.tidx[lane + 64 * k]);
  {
    const float wv = wave_max_f(bv);
    bi = wave_min_i(bv == wv ? bi : INT_MAX);
    bv = wv;
  }
  const int peak = (bi == INT_MAX) ? 0 : bi; // srslte_vec_max_fi returns 0 when nothing beats -FLT_MAX
  const float peak_value = AVG[peak];
  LT_STAMP(3) // arg-max

  // ---- main-lobe walk, 64 lags per step in each direction:
  //   pl_ub = first u >= peak+1 with !(avg[u+1] <= avg[u]) or u >= 9727
  //   pl_lb = last  l <= peak-1 with !(avg[l-1] <= avg[l]) or l <= 1      (0 when peak <= 2)
  int pl_ub = 0, pl_lb = 0;
  for (int base = peak + 1;; base += 64) {
    const int u = base + lane;
    bool stop = false;
    if (u <= kConvLen - 1) stop = !(AVG[u + 1] <= AVG[u]) || u >= kConvLen - 1;
    const unsigned long long m = __ballot(stop);
    if (m) {
      pl_ub = base + __builtin_ctzll(m);
      break;
    }
  }
  if (peak > 2) {
    for (int base = peak - 1;; base -= 64) {
      const int u = base - lane;
      bool stop = false;
      if (u >= 1) stop = !(AVG[u - 1] <= AVG[u]) || u <= 1;
      const unsigned long long m = __ballot(stop);
      if (m) {
        pl_lb = base - __builtin_ctzll(m);
        break;
      }
    }
  }

  LT_STAMP(4) // walk
  // ---- side lobes: max over [0, pl_lb) and over [pl_ub, 9726).  Threads whose 19 lags lie
  // entirely on one side contribute their stored maximum; the two threads cut by the lobe
  // contribute their outside elements one by one.
  const int tL = pl_lb / kT, tU = pl_ub / kT;
  float sl = -FLT_MAX, sr = -FLT_MAX;
#pragma unroll
  for (int k = 0; k < kThreads / 64; ++k) {
    const int t = lane + 64 * k;
    const float v = sc.tmax[t];
    if (t < tL && v > sl) sl = v;
    if (t > tU && v > sr) sr = v;
  }
  if (lane < kT) {
    const int u = tL * kT + lane;
    if (u < pl_lb) {
      const float v = AVG[u];
      if (v > sl) sl = v;
    }
  } else if (lane < 2 * kT) {
    const int u = pl_ub + (lane - kT);
    if (u < (tU + 1) * kT && u < kNLags) {
      const float v = AVG[u];
      if (v > sr) sr = v;
    }
  }
  sl = wave_max_f(sl);
  sr = wave_max_f(sr);
  // an empty range makes srslte_vec_max_fi return index 0 of that range
  if (pl_lb == 0) sl = AVG[0];
  if (kNLags - pl_ub <= 0) sr = AVG[pl_ub];
  const float side = sl > sr ? sl : sr;

  LT_STAMP(5) // side lobes
  // wave-uniform values: pin them to scalar registers
  SearchResult r;
  r.peak_pos = __builtin_amdgcn_readfirstlane(peak);
  r.peak_value = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(peak_value)));
  r.psr = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(peak_value / side)));
  return r;
}
#undef LT_STAMP

__device__ __forceinline__ SearchResult search_window(v2f *W, float *AVG, float ema[kT],
                                                     const v2f *__restrict__ taps, const FilterLds &fl,
                                                     Scratch &sc, int ablate)
{
  search_front(W, AVG, ema, taps, fl, sc, ablate);
  return search_tail(AVG, sc, ablate);
}

// ---- bulk + replay mode: the interior lags 127..9599 of a window come from the position-independent
// metric P[pos + lag] computed once per channel by lt_metric_kernel; only the 253 edge lags, whose
// convolution sums are cut by the window's zero padding, are recomputed here.
//   head lag j      (j = 0..126): sum_{m=0}^{j}     h[m] x[pos + j - m]
//   tail lag 9600+j (j = 0..125): sum_{m=j+1}^{127} h[m] x[pos + 9600 + j - m]
// With Z = [ x[pos+9472 .. pos+9600) | x[pos .. pos+128) ] (256 samples staged in LDS by the caller)
// both sums read Z[128 + j - m]: taps m <= j fall into the head half, the others into the tail half,
// so "unit" j walks all 128 taps once and yields head lag j and tail lag 9600+j.  Four lanes share a
// unit, taps m = q, q+4, q+8, ... each (fixed trip count: the LDS loads batch; neighbouring lanes
// read neighbouring or identical words: no bank conflicts); a lane's accumulator is moved aside
// where its taps cross from head to tail; the four partial sums meet in an xor butterfly.
// Items first, first+stride, ... < 512 are processed (item = 4 * unit + q).
__device__ __forceinline__ void edge_lags(const v2f *Z, const v2f *taps, float *head, float *tail, int first, int stride)
{
  for (int item = first; item < 512; item += stride) { // trip count must be wave-uniform (DPP below)
    const int j_raw = item >> 2, q = item & 3;
    const int j = j_raw < 127 ? j_raw : 126; // items 508..511 are padding
    const int b = (j - q + 4) >> 2;          // leading taps of this lane (m = q + 4 i <= j) that are head taps
    const v2f *zp = Z + 128 + j - q;         // sample of tap q + 4 i = zp[-4 i]
    const v2f *hp = taps + q;
    float re = 0.f, im = 0.f, hre = 0.f, him = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const v2f h = hp[4 * i], z = zp[-4 * i];
      re = fmaf(h.x, z.x, re);
      re = fmaf(-h.y, z.y, re);
      im = fmaf(h.x, z.y, im);
      im = fmaf(h.y, z.x, im);
      if (i + 1 == b) { // last head tap of the range: what follows belongs to the tail lag
        hre = re; him = im;
        re = 0.f; im = 0.f;
      }
    }
    hre += lane_xor<1>(hre); him += lane_xor<1>(him); re += lane_xor<1>(re); im += lane_xor<1>(im);
    hre += lane_xor<2>(hre); him += lane_xor<2>(him); re += lane_xor<2>(re); im += lane_xor<2>(im);
    if (q == 0 && j_raw < 127) {
      head[j] = __fadd_rn(__fmul_rn(hre, hre), __fmul_rn(him, him));                 // lag j
      if (j < 126) tail[j] = __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));        // lag 9600 + j
    }
  }
}

// ---- bulk + replay mode search: srslte_pss_find_pss downstream of the lag metric with the lags
// dealt out round-robin (thread t owns lags t, t + NT, t + 2 NT, ...), which is the order the
// metric slice arrives in from HBM, so nothing is staged through LDS on the way in.
template <int NT> struct BulkShared {
  float tmax[2][NT]; // per-thread maximum of the EMA over its valid lags (two sets used in turn)
  int tidx[2][NT];   // lag of that maximum
  float edge[256];   // edge lags computed in the kernel: [0,127) head lags, [127,253) lags 9600..9725
  float sl[NT / 64]; // wide main lobes only: per-wave side-lobe maxima
};

// EMA (three roundings like srsLTE's three vector passes), LDS mirror, per-thread first maximum.
// No barrier inside: the caller issues its prefetch, then calls bulk_search_tail.
template <int NT, int NL>
__device__ __forceinline__ void bulk_search_ema(const float (&met)[NL], float (&ema)[NL], float *AVG, float *tmax, int *tidx)
{
  const int tid = threadIdx.x;
  float bv = -FLT_MAX;
  int bi = INT_MAX;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int lag = tid + i * NT;
    if (lag < kNLags) {
      ema[i] = __fadd_rn(__fmul_rn(met[i], 0.2f), __fmul_rn(ema[i], 1.0f - 0.2f));
      if (ema[i] > bv) { // lags ascend with i: the first maximum of this thread
        bv = ema[i];
        bi = lag;
      }
    }
    AVG[lag] = ema[i];
  }
  tmax[tid] = bv;
  tidx[tid] = bi;
}

// arg-max, main-lobe walk, side lobes, PSR.  One block-wide barrier, after which every wave derives
// the same (wave-uniform) results from LDS on its own.  `ema` = this thread's EMA values.
//
// Side lobes: srsLTE takes max(max over [0, pl_lb), max over [pl_ub, 9726)), an empty range standing
// for the first element of where it would start (srslte_vec_max_fi returns index 0).  The maximum over
// the lags outside the main lobe is assembled from the per-thread maxima: a thread whose maximum lies
// outside the lobe contributes it as it is; the lags inside the lobe belong to pl_ub - pl_lb threads,
// whose remaining lags are looked up in the LDS mirror one by one (a handful for the usual narrow
// lobe; lobes wider than 64 lags take a second barrier and a masked pass over the registers instead).
// (Superseded in part: only threads whose own maximum lies inside the lobe are looked up.)
template <int NT, int NL>
__device__ __forceinline__ SearchResult bulk_search_tail(const float (&ema)[NL], const float *AVG, const float *tmax,
                                                         const int *tidx, float *wave_sl,
                                                         unsigned long long *prof = nullptr)
{
  unsigned long long t_prev = prof ? __builtin_amdgcn_s_memtime() : 0;
#define LT_BSTAMP(i)                                             \
  if (prof) {                                                    \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
    prof[i] += t_ - t_prev;                                      \
    t_prev = t_;                                                 \
  }
  const int tid = threadIdx.x, lane = tid & 63;
  __syncthreads(); // AVG mirror and per-thread maxima visible
  LT_BSTAMP(2) // barrier
  float tv[NT / 64];
  int ti[NT / 64];
  float bv = -FLT_MAX;
  int bi = INT_MAX;
#pragma unroll
  for (int k = 0; k < NT / 64; ++k) {
    tv[k] = tmax[lane + 64 * k];
    ti[k] = tidx[lane + 64 * k];
    argmax_combine(bv, bi, tv[k], ti[k]);
  }
  {
    const float wv = wave_max_f(bv);
    const unsigned long long holders = __ballot(bv == wv);
    if (__builtin_popcountll(holders) == 1) bi = __builtin_amdgcn_readlane(bi, __builtin_ctzll(holders)); // the usual case
    else bi = wave_min_i(bv == wv ? bi : INT_MAX);
  }
  const int peak = (bi == INT_MAX) ? 0 : bi; // srslte_vec_max_fi returns 0 when nothing beats -FLT_MAX
  const float peak_value = AVG[peak];
  LT_BSTAMP(3) // arg-max

  // main-lobe walk, 64 lags per step, both directions in the same turn (their LDS round trips overlap):
  //   pl_ub = first u >= peak+1 with !(avg[u+1] <= avg[u]) or u >= 9727
  //   pl_lb = last  l <= peak-1 with !(avg[l-1] <= avg[l]) or l <= 1      (0 when peak <= 2)
  const float q_left = AVG[0];
  int pl_ub = 0, pl_lb = 0, up_base = peak + 1, dn_base = peak - 1;
  bool up_done = false, dn_done = !(peak > 2);
  for (;;) {
    unsigned long long mu = 0, md = 0;
    if (!up_done) {
      const int u = up_base + lane;
      bool stop = false;
      if (u <= kConvLen - 1) stop = !(AVG[u + 1] <= AVG[u]) || u >= kConvLen - 1;
      mu = __ballot(stop);
    }
    if (!dn_done) {
      const int u = dn_base - lane;
      bool stop = false;
      if (u >= 1) stop = !(AVG[u - 1] <= AVG[u]) || u <= 1;
      md = __ballot(stop);
    }
    if (!up_done) {
      if (mu) {
        pl_ub = up_base + __builtin_ctzll(mu);
        up_done = true;
      } else {
        up_base += 64;
      }
    }
    if (!dn_done) {
      if (md) {
        pl_lb = dn_base - __builtin_ctzll(md);
        dn_done = true;
      } else {
        dn_base -= 64;
      }
    }
    if (up_done && dn_done) break;
  }
  const float q_right = AVG[pl_ub];
  LT_BSTAMP(4) // walk
  float out = -FLT_MAX; // maximum over the valid lags outside [pl_lb, pl_ub)
  const int width = pl_ub - pl_lb;
  if (width <= 64) {
    // A thread whose maximum lies outside the lobe contributes it as it is (its other lags are not above
    // it); a thread whose maximum lies inside has its NL lags looked up in the mirror, one per lane.
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) {
      const bool inside = ti[k] >= pl_lb && ti[k] < pl_ub;
      if (!inside && tv[k] > out) out = tv[k]; // (a thread without a maximum holds -FLT_MAX, lag INT_MAX)
      unsigned long long owners = __ballot(inside);
      while (owners) {
        const int t = __builtin_ctzll(owners) + 64 * k;
        owners &= owners - 1;
        const int lag = t + lane * NT;
        if (lane < NL && (lag < pl_lb || lag >= pl_ub) && lag < kNLags) {
          const float v = AVG[lag];
          if (v > out) out = v;
        }
      }
    }
    out = wave_max_f(out);
  } else {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int lag = tid + i * NT;
      if ((lag < pl_lb || lag >= pl_ub) && lag < kNLags && ema[i] > out) out = ema[i];
    }
    out = wave_max_f(out);
    if (lane == 0) wave_sl[tid >> 6] = out;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) {
      const float o = wave_sl[k];
      if (o > out) out = o;
    }
    __syncthreads(); // bs.sl may be rewritten by the next wide lobe
  }
  LT_BSTAMP(5) // side lobes
  float sl = (pl_lb == 0) ? q_left : out, sr = (kNLags - pl_ub <= 0) ? q_right : out;
  const float side = sl > sr ? sl : sr;
  SearchResult r;
  r.peak_pos = __builtin_amdgcn_readfirstlane(peak);
  r.peak_value = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(peak_value)));
  r.psr = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(peak_value / side)));
  return r;
}
#undef LT_BSTAMP

// copy the FFT twiddles and the filter spectrum of one root into LDS (2 x 8 KB)
__device__ __forceinline__ void load_filter_lds(v2f *tw, v2f *spec, const DeviceTables *__restrict__ tab, int n_id_2)
{
  for (int i = threadIdx.x; i < fft::kN; i += blockDim.x) {
    tw[i] = v2f{tab->fft_tw[i].re, tab->fft_tw[i].im};
    spec[i] = v2f{tab->pss_spec[n_id_2][i].re, tab->pss_spec[n_id_2][i].im};
  }
}

// ---- pieces of the tracked-emit path.  Every sum is evaluated in the oracle's operation order
// (interleaved partial sums like the SIMD dot products of srsLTE, combined pairwise), so the
// results are bit-exact against it.  The *_wave functions are executed by every lane of a wave,
// read their inputs from LDS, exchange partial results with v_readlane and return wave-uniform
// values: several waves can run them redundantly instead of waiting on a broadcast.

__device__ __forceinline__ float rlane(float v, int lane)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__device__ __forceinline__ v2f cmul_plain(v2f a, v2f b)
{
  v2f r;
  r.x = __fsub_rn(__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y));
  r.y = __fadd_rn(__fmul_rn(a.x, b.y), __fmul_rn(a.y, b.x));
  return r;
}

// (p0+p1)+(p2+p3) of four consecutive lanes starting at `base`
__device__ __forceinline__ float sum4_lanes(float v, int base)
{
  return __fadd_rn(__fadd_rn(rlane(v, base), rlane(v, base + 1)), __fadd_rn(rlane(v, base + 2), rlane(v, base + 3)));
}

// srslte_pss_cfo_compute: y_half = sum_{m in half} h[m] r[m] as eight interleaved
// partial sums (m mod 8) combined as ((p0+p1)+(p2+p3))+((p4+p5)+(p6+p7)) -- lanes 0..7 / 8..15 hold the
// partial sums of the two halves and meet in an xor butterfly; cfo = arg(conj(y0) y1) / pi.
// h, r: 128 entries each (LDS or global).
__device__ __forceinline__ float cfo_estimate_wave(const v2f *h, const v2f *r)
{
  const int lane = threadIdx.x & 63;
  v2f y = {0.f, 0.f};
  if (lane < 16) {
    const int base = 64 * (lane >> 3) + (lane & 7);
    v2f hh[8], rr[8]; // all reads first: one round trip
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      hh[j] = h[base + 8 * j];
      rr[j] = r[base + 8 * j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const v2f p = cmul_plain(hh[j], rr[j]);
      y.x = __fadd_rn(y.x, p.x);
      y.y = __fadd_rn(y.y, p.y);
    }
  }
  y.x = __fadd_rn(y.x, lane_xor<1>(y.x)); y.y = __fadd_rn(y.y, lane_xor<1>(y.y));
  y.x = __fadd_rn(y.x, lane_xor<2>(y.x)); y.y = __fadd_rn(y.y, lane_xor<2>(y.y));
  y.x = __fadd_rn(y.x, lane_xor<4>(y.x)); y.y = __fadd_rn(y.y, lane_xor<4>(y.y));
  const v2f y0 = {rlane(y.x, 0), rlane(y.y, 0)};
  const v2f y1 = {rlane(y.x, 8), rlane(y.y, 8)};
  const v2f c = {y0.x, -y0.y};
  const v2f z = cmul_plain(c, y1);
  return (float)(atan2((double)z.y, (double)z.x) / 3.14159265358979323846);
}

// compute_moving_avg (lib/pss_impl.cc:94-109): sequential double sum of the first min(npts, 200)
// ring entries, the entry `new_idx` taken as `new_val` (it is being written by this call).
// Lanes fetch four entries each.  Fast path: if the exponents of the non-zero entries lie within
// 20 binades (and none is subnormal), every partial sum of any subset is a multiple of the smallest
// ulp and below 2^53 of them, i.e. exactly representable in double -- the sequential sum rounds
// nowhere, so any association gives the same bits and an xor butterfly over the wave is used.
// Otherwise the sum runs strictly in index order through v_readlane.
// `n_new` entries starting at ring index `new_idx` (wrapping) are taken from new_val(0..n_new-1) instead of
// the ring: the entry that is being written by this call and, in the batched tracked path, those of the
// calls before it in the batch.  (A functor, not a pointer: a pointer that may be LDS or a local would
// turn the reads into flat loads.)
template <typename NewVal>
__device__ __forceinline__ float moving_avg_wave_f(const float *ring, uint32_t npts, uint32_t new_idx, uint32_t n_new,
                                                  NewVal new_val, unsigned long long *slow_count)
{
  if (!npts) return 0.0f;
  if (npts > (uint32_t)kMovingAvgSz) npts = kMovingAvgSz;
  const uint32_t lane = threadIdx.x & 63;
  float v[4];
  uint32_t emax = 0, emin = 255, any_sub = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t i = lane + 64 * k;
    float f = (i < (uint32_t)kMovingAvgSz) ? ring[i] : 0.f;
    const uint32_t d = i >= new_idx ? i - new_idx : i + (uint32_t)kMovingAvgSz - new_idx; // distance past new_idx, wrapping
    if (i < (uint32_t)kMovingAvgSz && d < n_new) f = new_val(d);
    v[k] = (i < npts) ? f : 0.f; // x + 0.0 leaves the running sum unchanged
    const uint32_t bits = __float_as_uint(v[k]) & 0x7fffffffu;
    const uint32_t e = bits >> 23;
    if (bits != 0u) {
      any_sub |= (e == 0u);
      emax = e > emax ? e : emax;
      emin = e < emin ? e : emin;
    }
  }
  // Exponent window [lo, lo+20] around the first lane's largest exponent; one ballot decides
  // whether every non-zero entry of the wave lies inside it.
  const uint32_t ref_max = __builtin_amdgcn_readfirstlane(emax);
  const uint32_t lo = ref_max > 10u ? ref_max - 10u : 1u, hi = lo + 20u;
  const bool outside = any_sub || (emax != 0u && (emax > hi || emax < lo)) || (emin != 255u && emin < lo);
  const bool fast = (__ballot(outside) == 0ull) && hi < 255u;
  double acc;
  if (fast) {
    // no partial sum of these values can round: any association gives the reference's bits
    acc = ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
    acc += lane_xor_d<1>(acc);
    acc += lane_xor_d<2>(acc);
    acc += lane_xor_d<4>(acc);
    acc += lane_xor_d<8>(acc);
    acc += lane_xor_d<16>(acc);
    acc += lane_xor_d<32>(acc);
    acc = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(acc)),
                           __builtin_amdgcn_readfirstlane(__double2loint(acc)));
  } else {
    if (slow_count) *slow_count += 1;
    acc = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int l = 0; l < 64; ++l)
        if (64 * k + l < kMovingAvgSz) acc = __dadd_rn(acc, (double)rlane(v[k], l));
  }
  return (float)(acc / (double)npts);
}
__device__ __forceinline__ float moving_avg_wave(const float *ring, uint32_t npts, uint32_t new_idx, float new_val,
                                                 unsigned long long *slow_count = nullptr)
{
  return moving_avg_wave_f(ring, npts, new_idx, 1u, [new_val](uint32_t) { return new_val; }, slow_count);
}
// new values in an LDS array
__device__ __forceinline__ float moving_avg_wave(const float *ring, uint32_t npts, uint32_t new_idx, const float *new_vals,
                                                 uint32_t n_new, unsigned long long *slow_count = nullptr)
{
  return moving_avg_wave_f(ring, npts, new_idx, n_new, [new_vals](uint32_t d) { return new_vals[d]; }, slow_count);
}

// srslte_cfo_correct for sample i: tab[floor(i*freq*4096 mod 4096)] * v  (quirk q6)
__device__ __forceinline__ v2f cfo_rotate(const cf32 *__restrict__ tab, float phase_inc, int i, v2f v)
{
  double p = __dmul_rn((double)phase_inc, (double)i);
  p = __dsub_rn(p, __dmul_rn(4096.0, floor(p / 4096.0)));
  uint32_t idx = (uint32_t)p;
  if (idx > 4095u) idx = 4095u;
  const cf32 e = tab[idx];
  v2f r;
  r.x = fmaf(e.re, v.x, -__fmul_rn(e.im, v.y));
  r.y = fmaf(e.re, v.y, __fmul_rn(e.im, v.x));
  return r;
}

// srslte_sync_detect_cp for peak_pos >= 480 (three symbols per hypothesis).
// in: samples such that in[peak_pos - 480 ...] is readable; `base` = index of sample (peak_pos-480).
// Lane (hyp, sym, l) = lanes 0..23 sums every 4th term; returns LT_CP_NORM / LT_CP_EXT and updates
// the two EMAs (wave-uniform).
struct CpMetrics {
  float m[2]; // M_norm, M_ext of this half-frame (already the per-hypothesis ratios)
  float r[2]; // R_norm, R_ext (tie break)
};
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f cp_metrics_wave_v(const v2f *win480); // (M_norm, M_ext, R_norm, R_ext): in registers, no stack object
__device__ __forceinline__ CpMetrics cp_metrics_wave(const v2f *win480)
{
  const v4f v = cp_metrics_wave_v(win480);
  CpMetrics c;
  c.m[0] = v.x; c.m[1] = v.y; c.r[0] = v.z; c.r[1] = v.w;
  return c;
}
// EMA update and decision of srslte_sync_detect_cp from one half-frame's metrics, in three pieces so that
// lt_cp_scan_kernel can form the state-independent products for 64 records at once:
//   avg' = alpha * (M / 3) + (1 - alpha) * avg;   CP = avg_norm > avg_ext ? normal : avg_norm < avg_ext ? ext : by R
constexpr float kCpAlpha = 0.1f;
__device__ __forceinline__ float cp_avg_input(float m) { return __fmul_rn(kCpAlpha, m / 3.0f); }
__device__ __forceinline__ float cp_avg_step(float input, float avg) { return __fadd_rn(input, __fmul_rn(1.0f - kCpAlpha, avg)); }
__device__ __forceinline__ int cp_choose(float m_norm_avg, float m_ext_avg, float r_norm, float r_ext)
{
  if (m_norm_avg > m_ext_avg) return LT_CP_NORM;
  if (m_norm_avg < m_ext_avg) return LT_CP_EXT;
  return (r_norm > r_ext) ? LT_CP_NORM : LT_CP_EXT;
}
__device__ __forceinline__ int cp_decide(const CpMetrics &c, float &m_norm_avg, float &m_ext_avg)
{
  m_norm_avg = cp_avg_step(cp_avg_input(c.m[0]), m_norm_avg);
  m_ext_avg = cp_avg_step(cp_avg_input(c.m[1]), m_ext_avg);
  return cp_choose(m_norm_avg, m_ext_avg, c.r[0], c.r[1]);
}
__device__ __forceinline__ int cp_detect_wave(const v2f *win480, float &m_norm_avg, float &m_ext_avg)
{
  const CpMetrics c = cp_metrics_wave(win480);
  return cp_decide(c, m_norm_avg, m_ext_avg);
}
__device__ __forceinline__ v4f cp_metrics_wave_v(const v2f *win480)
{
  const int lane = threadIdx.x & 63;
  float pr = 0.f, pp = 0.f;
  if (lane < 24) {
    const int hs = lane >> 2, l = lane & 3, hyp = hs / 3, sym = hs % 3;
    const int L = hyp ? 32 : 9;
    const v2f *p = win480 + (480 - 3 * (kSymbolSz + L)) + sym * (kSymbolSz + L);
    v2f aa[8], bb[8]; // all LDS reads first
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = min(l + 4 * j, L - 1);
      aa[j] = p[kSymbolSz + i];
      bb[j] = p[i];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (l + 4 * j < L) {
        const v2f a = aa[j], b = bb[j];
        pr = __fadd_rn(pr, __fadd_rn(__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y)));
        pp = __fadd_rn(pp, __fadd_rn(__fmul_rn(b.x, b.x), __fmul_rn(b.y, b.y)));
      }
  }
  // (p0+p1)+(p2+p3) of each group of four lanes, left in every lane of the group
  pr = __fadd_rn(pr, lane_xor<1>(pr)); pp = __fadd_rn(pp, lane_xor<1>(pp));
  pr = __fadd_rn(pr, lane_xor<2>(pr)); pp = __fadd_rn(pp, lane_xor<2>(pp));
  float R[2], M[2];
#pragma unroll
  for (int hyp = 0; hyp < 2; ++hyp) {
    const float L = hyp ? 32.f : 9.f;
    float r = 0.f, c = 0.f;
#pragma unroll
    for (int sym = 0; sym < 3; ++sym) {
      const int base = 4 * (3 * hyp + sym);
      r = __fadd_rn(r, rlane(pr, base));
      c = __fadd_rn(c, __fmul_rn(L, rlane(pp, base) / L));
    }
    R[hyp] = r;
    M[hyp] = (c > 0.f) ? r / c : 0.f;
  }
  return v4f{M[0], M[1], R[0], R[1]};
}

// 128-point forward FFT of the SSS symbol on one wave: iterative radix-2 decimation in time with the
// oracle's butterfly, t = w b = (b.re w.re - b.im w.im, b.re w.im + b.im w.re), a' = a + t, b' = a - t,
// w = W128^(k 128/len) from a float table tw[64].  Lane l owns elements 2l and 2l+1 of the
// bit-reversed working array; stage len = 2 is lane-local, the later stages pair lane l with
// l ^ (len/4) through an xor exchange.  The 62 occupied bins (-31..-1, +1..+31) go to Y[0..61] in LDS.
template <int kHalfLanes> // = len / 4: the lane distance of this stage
__device__ __forceinline__ void fft128_stage(int lane, v2f &a0, v2f &a1, const v2f *tw)
{
  constexpr int half = 2 * kHalfLanes; // element distance
  constexpr int step = 64 / half;      // twiddle stride
  const v2f o0 = {lane_xor<kHalfLanes>(a0.x), lane_xor<kHalfLanes>(a0.y)};
  const v2f o1 = {lane_xor<kHalfLanes>(a1.x), lane_xor<kHalfLanes>(a1.y)};
  const bool top = (lane & kHalfLanes) == 0;
  const int k0 = (2 * lane) & (half - 1);
  const v2f w0 = tw[k0 * step], w1 = tw[(k0 + 1) * step];
  const v2f u0 = top ? a0 : o0, v0 = top ? o0 : a0;
  const v2f u1 = top ? a1 : o1, v1 = top ? o1 : a1;
  const float t0r = __fsub_rn(__fmul_rn(v0.x, w0.x), __fmul_rn(v0.y, w0.y));
  const float t0i = __fadd_rn(__fmul_rn(v0.x, w0.y), __fmul_rn(v0.y, w0.x));
  const float t1r = __fsub_rn(__fmul_rn(v1.x, w1.x), __fmul_rn(v1.y, w1.y));
  const float t1i = __fadd_rn(__fmul_rn(v1.x, w1.y), __fmul_rn(v1.y, w1.x));
  a0.x = top ? __fadd_rn(u0.x, t0r) : __fsub_rn(u0.x, t0r);
  a0.y = top ? __fadd_rn(u0.y, t0i) : __fsub_rn(u0.y, t0i);
  a1.x = top ? __fadd_rn(u1.x, t1r) : __fsub_rn(u1.x, t1r);
  a1.y = top ? __fadd_rn(u1.y, t1i) : __fsub_rn(u1.y, t1i);
}

__device__ __forceinline__ void sss_fft128_wave(const v2f *sym, const v2f *tw, v2f *Y)
{
  const int lane = threadIdx.x & 63;
  const int br = (int)(__brev((unsigned)lane) >> 26); // 6-bit reversal: bitrev7(2l) = br, bitrev7(2l+1) = 64 + br
  v2f a0 = sym[br], a1 = sym[64 + br];
  { // len = 2: both points in this lane, w = W128^0
    const v2f w = tw[0];
    const float tr = __fsub_rn(__fmul_rn(a1.x, w.x), __fmul_rn(a1.y, w.y));
    const float ti = __fadd_rn(__fmul_rn(a1.x, w.y), __fmul_rn(a1.y, w.x));
    const v2f u = a0;
    a0.x = __fadd_rn(u.x, tr); a0.y = __fadd_rn(u.y, ti);
    a1.x = __fsub_rn(u.x, tr); a1.y = __fsub_rn(u.y, ti);
  }
  fft128_stage<1>(lane, a0, a1, tw);
  fft128_stage<2>(lane, a0, a1, tw);
  fft128_stage<4>(lane, a0, a1, tw);
  fft128_stage<8>(lane, a0, a1, tw);
  fft128_stage<16>(lane, a0, a1, tw);
  fft128_stage<32>(lane, a0, a1, tw);
  // lane l now holds X[2l], X[2l+1]
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int e = 2 * lane + c;
    const v2f x = c ? a1 : a0;
    if (e >= 1 && e <= 31) Y[e + 30] = x;
    if (e >= 97) Y[e - 97] = x;
  }
}

// rotate a 31-bit mask: result bit i = mask bit (i + sh) % 31
__device__ __forceinline__ uint32_t rot31(uint32_t mask, int sh)
{
  return ((mask >> sh) | (mask << (31 - sh))) & 0x7fffffffu;
}

// |sum_i Y[2i+odd] c[i] z1[i] s~[(i+m)%31]|^2 for shift m (corr_all_sz_partial, M = 1), four
// interleaved partial sums (i mod 4) per component, combined as (p0+p1)+(p2+p3).  The +-1 factors
// only flip signs, so they are applied as one XOR of the sign bit: `signs` bit i = sign of
// c[i] z1[i] s~[(i+m)%31].  Lane m (half = 0) sums the terms with i mod 4 in {0,1}, lane m + 32
// (half = 1) those with i mod 4 in {2,3}; the halves meet through one cross-lane exchange.
// Returns the correlation in lanes 0..30 (garbage elsewhere).
__device__ __forceinline__ float sss_corr(const v2f *Y, int odd, uint32_t signs, int half)
{
  float ar = 0.f, ai = 0.f, br = 0.f, bi = 0.f; // partial sums p(2 half), p(2 half + 1)
  v2f ya[8], yb[8]; // all LDS reads first (index 61 at most: Y has 62 entries)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i0 = 4 * j + 2 * half, i1 = i0 + 1;
    ya[j] = Y[min(2 * i0 + odd, 61)];
    yb[j] = Y[min(2 * i1 + odd, 61)];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i0 = 4 * j + 2 * half, i1 = i0 + 1;
    if (i0 < 31) {
      const v2f y = ya[j];
      const uint32_t sg = ((signs >> i0) & 1u) << 31;
      ar = __fadd_rn(ar, __uint_as_float(__float_as_uint(y.x) ^ sg));
      ai = __fadd_rn(ai, __uint_as_float(__float_as_uint(y.y) ^ sg));
    }
    if (i1 < 31) {
      const v2f y = yb[j];
      const uint32_t sg = ((signs >> i1) & 1u) << 31;
      br = __fadd_rn(br, __uint_as_float(__float_as_uint(y.x) ^ sg));
      bi = __fadd_rn(bi, __uint_as_float(__float_as_uint(y.y) ^ sg));
    }
  }
  const float sr = __fadd_rn(ar, br), si = __fadd_rn(ai, bi); // p0+p1 (half 0) or p2+p3 (half 1)
  const float re = __fadd_rn(sr, __shfl_xor(sr, 32));          // lanes < 32: (p0+p1)+(p2+p3)
  const float im = __fadd_rn(si, __shfl_xor(si, 32));
  return __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
}

// first maximum over lanes 0..30 of a wave (srslte_vec_max_fi): lane index and value, wave-uniform
__device__ __forceinline__ int first_max31_wave(float corr, float *value)
{
  const int lane = threadIdx.x & 63;
  const float v = (lane < 31) ? corr : -FLT_MAX;
  const float mx = wave_max_f(v);
  const unsigned long long hit = __ballot(lane < 31 && v == mx && v > -FLT_MAX);
  *value = mx;
  return hit ? __builtin_ctzll(hit) : 0;
}

// srslte_sss_N_id_1 with its unsigned wrap
__device__ __forceinline__ int sss_n_id_1(const DeviceTables *__restrict__ tab, uint32_t m0, uint32_t m1)
{
  int n = -1;
  if (m1 > m0) {
    if (m0 < 30u && m1 - 1u < 30u) n = tab->nid1_table[m0][m1 - 1u];
  } else {
    if (m1 < 30u && m0 - 1u < 30u) n = tab->nid1_table[m1][m0 - 1u];
  }
  return n;
}

// Per-workgroup LDS copy of the small tables the tracked-emit path walks through.
struct LdsTables {
  v2f tw128[64];            // W128^e = (cos, sin)(-2 pi e / 128), e < 64, single precision
  v2f taps[128];
  uint32_t s_mask, z_mask, c0_mask, c1_mask; // m-sequence signs (bit i set <=> -1)
};

__device__ __forceinline__ void load_lds_tables(LdsTables &lt_, const DeviceTables *__restrict__ tab, int n_id_2)
{
  for (int i = threadIdx.x; i < 128; i += blockDim.x) {
    if (i < 64) lt_.tw128[i] = v2f{(float)tab->dft128_cos[i], (float)tab->dft128_sin[i]};
    v2f h = {tab->pss_taps[n_id_2][i].re, tab->pss_taps[n_id_2][i].im};
    lt_.taps[i] = h;
  }
  if (threadIdx.x == 0) {
    lt_.s_mask = tab->sss_s_mask;
    lt_.z_mask = tab->sss_z_mask;
    lt_.c0_mask = tab->sss_c0_mask[n_id_2];
    lt_.c1_mask = tab->sss_c1_mask[n_id_2];
  }
}

struct SssResult {
  int cp, m0, m1, n_id_1;
  float m0_value, m1_value;
};

// srslte_sss_m0m1_partial after the FFT (bins in sc.Y): two lanes per shift correlate against the 31
// shifts of s~.  Wave-uniform result.
__device__ __forceinline__ void sss_m0m1_wave(const v2f *Y, const LdsTables &lt_, SssResult &r)
{
  const int lane = threadIdx.x & 63;
  const int m = lane & 31, half = lane >> 5;
  const uint32_t s_rot = rot31(lt_.s_mask, m < 31 ? m : 0);
  float corr = sss_corr(Y, 0, s_rot ^ lt_.c0_mask, half);
  r.m0 = first_max31_wave(corr, &r.m0_value);
  corr = sss_corr(Y, 1, s_rot ^ lt_.c1_mask ^ rot31(lt_.z_mask, r.m0 % 8), half);
  r.m1 = first_max31_wave(corr, &r.m1_value);
}
__device__ __forceinline__ void sss_m0m1_wave(Scratch &sc, const LdsTables &lt_, SssResult &r) { sss_m0m1_wave(sc.Y, lt_, r); }

// sss_impl::work after the CP decision: FFT-128 of the SSS symbol, m0/m1 search, N_id_1 (one wave,
// Y = 62 bins of LDS scratch owned by that wave)
__device__ __forceinline__ void sss_after_cp_wave(const v2f *S, int cp, const LdsTables &lt_, const DeviceTables *__restrict__ tab,
                                                  v2f *Y, SssResult &r)
{
  r.cp = cp;
  const int cp_len = (cp == LT_CP_NORM) ? 9 : 32;                   // srslte_sync_set_cp
  const int sss_idx = kSlotLen - 2 * kSymbolSz - cp_len - kSssBase; // lib/sss_impl.cc:110
  sss_fft128_wave(S + sss_idx, lt_.tw128, Y);
  sss_m0m1_wave(Y, lt_, r);
  r.n_id_1 = sss_n_id_1(tab, (uint32_t)r.m0, (uint32_t)r.m1);
}

// sss_impl::work body after the tag check (lib/sss_impl.cc:102-150) on S = corrected half-frame
// samples [480, 960) in LDS, executed by ONE wave (no block barrier inside): CP decision, FFT-128 of
// the SSS symbol, m0/m1 search, N_id_1.  Its results feed nothing but the record and the CP
// averages, both owned by that wave.
__device__ __forceinline__ SssResult sss_stage_wave(const v2f *S, const LdsTables &lt_, const DeviceTables *__restrict__ tab,
                                                   float &m_norm_avg, float &m_ext_avg, Scratch &sc)
{
  SssResult r;
  r.cp = cp_detect_wave(S, m_norm_avg, m_ext_avg);
  const int cp_len = (r.cp == LT_CP_NORM) ? 9 : 32;                 // srslte_sync_set_cp
  const int sss_idx = kSlotLen - 2 * kSymbolSz - cp_len - kSssBase; // lib/sss_impl.cc:110
  sss_fft128_wave(S + sss_idx, lt_.tw128, sc.Y);
  sss_m0m1_wave(sc, lt_, r);
  r.n_id_1 = sss_n_id_1(tab, (uint32_t)r.m0, (uint32_t)r.m1);
  return r;
}

} // namespace lt
