// Device-side building blocks of the gfx950 LTE cell-search path shared by all kernels: the chain state, wave
// reductions, and the emit-side stages (CFO estimate and correction, CP detection, SSS) of
// pss_impl::general_work / sss_impl::work.  The search itself (srslte_pss_find_pss) lives in lt_scan.h.
//
// Arithmetic rules: the library is built with -ffp-contract=off; the only fused operations are the explicit
// FMAs of the transforms and of the CFO rotation.  Everything downstream of the lag metric is evaluated in the
// oracle's operation order so that it is bit-exact against it.
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

constexpr int kSssBase = 480;                       // first half-frame sample the sss stage touches
constexpr int kSssLen = kSlotLen - kSssBase;        // 480 samples [480, 960)

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
  // ---- the exponential average and how the grid scan hands it on (lt_scan_kernels.h).  avg[ema_sel] is a valid
  // state of conv_output_avg; the chain's current average is that state run through the pre_n windows
  // pre_pos + k * pre_stride * 9600 (windows of a scan round whose tail the chain did not follow), or all
  // zeros when ema_zero is set (1: srslte_pss_reset; 2: lt_batch_reset, which leaves the 117 KB of buffers alone).
  // A scan round writes into the other two buffers.
  int32_t ema_sel;
  int32_t ema_zero;
  int32_t pre_n;
  int32_t pre_stride;
  int64_t pre_pos;
  // ---- this run / this scan round
  int32_t run_done;     // the chain has stopped for this run (nothing more fits into the chunk, or records are full)
  int32_t need_chain;   // the per-chain kernel takes the chain from here (a window the strips could not resolve)
  int32_t scan_job;     // this round: its job, -1 = none
  int32_t listed;       // this run: the chain's tracked half-frames are on the per-record work list
  // ---- this run's window cache (lt_metric_kernel): the lag metric of the windows cache_pos + k * cache_stride * 9600,
  // k < cache_n, sits in cache slots cache_base + k (12 strips x 896 lags each)
  int64_t cache_pos;
  int32_t cache_base;
  int32_t cache_n;
  int32_t cache_stride;
  int32_t cache_pad;
  float psr_data[kMovingAvgSz];
  float cfo_data[kMovingAvgSz];
  float avg[3][kConvLen]; // srslte_pss_t::conv_output_avg (lags >= 9726 stay 0), natural lag order
};

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

} // namespace lt
