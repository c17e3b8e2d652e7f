// Device-side building blocks of the gfx950 LTE cell-search path.
//
// Work decomposition (one workgroup = one pss->sss chain = one (channel, N_id_2)):
//   512 threads (8 wave64), thread `tid` owns the 19 consecutive lags [19*tid, 19*tid+19) of the
//   9728-lag linear convolution of a 9600-sample half-frame window with the 128-tap PSS filter
//   (512 * 19 = 9728 exactly).  The window sits zero-padded in LDS; the EMA of the lag metric
//   lives in 19 registers per thread and never touches HBM between windows.
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
#include "lt_tables.h"

namespace lt {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kT = 19;                              // lags per thread
constexpr int kWPad = 128;                          // zero samples in front of the window in LDS
constexpr int kWLen = kWPad + kHalfFrame + 128;     // 9856 samples = 78848 B
constexpr int kAvgLen = kConvLen + 4;               // EMA mirror in LDS, [9728] is the always-zero lag
constexpr int kSssBase = 480;                       // first half-frame sample the sss stage touches
constexpr int kSssLen = kSlotLen - kSssBase;        // 480 samples [480, 960)
static_assert(kThreads * kT == kConvLen, "lag tiling must cover the convolution exactly");

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
  float pad0;
  float psr_data[kMovingAvgSz];
  float cfo_data[kMovingAvgSz];
  float avg[kConvLen];  // srslte_pss_t::conv_output_avg (lags >= 9726 stay 0)
};

// LDS scratch for block reductions and the sss stage.
struct Scratch {
  float am_v[kWaves];
  int am_i[kWaves];
  int ub[kWaves];
  int lb[kWaves];
  float sl[kWaves];
  float sr[kWaves];
  v2f y[2];
  float cfo, mean_cfo, freq;
  float cpR[2][3], cpC[2][3];
  v2f Y[62];
  float corr[32];
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

// srslte_pss_find_pss on the window held in W (LDS, zero padded): lag metric, EMA(0.2),
// arg-max, main-lobe walk, peak-to-side-lobe ratio (call site lib/pss_impl.cc:165).
// `ema` is this thread's slice of conv_output_avg.  Contains __syncthreads().
__device__ __forceinline__ SearchResult search_window(const v2f *W, float *AVG, float ema[kT],
                                                     const v2f *__restrict__ taps, Scratch &sc,
                                                     int ablate)
{
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = tid * kT;

  v2f acc[kT];
  if (ablate & 2) { // diagnostics: no matched filter
#pragma unroll
    for (int t = 0; t < kT; ++t) acc[t] = W[kWPad + n0 + t];
  } else {
    conv19(W + kWPad + n0, taps, acc);
  }
  if (ablate & 4) { // diagnostics: no reductions
    SearchResult r0;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < kT; ++t) {
      ema[t] = acc[t].x * acc[t].x + acc[t].y * acc[t].y;
      s += ema[t];
    }
    AVG[n0] = s;
    __syncthreads();
    r0.peak_pos = __builtin_amdgcn_readfirstlane((int)AVG[4] & 1023);
    r0.peak_value = 1.f;
    r0.psr = 1.f;
    return r0;
  }

  // |conv|^2, EMA with three roundings like srsLTE's three vector passes, local first-maximum
  float bv = -FLT_MAX;
  int bi = INT_MAX;
#pragma unroll
  for (int t = 0; t < kT; ++t) {
    const int lag = n0 + t;
    const float a = __fadd_rn(__fmul_rn(acc[t].x, acc[t].x), __fmul_rn(acc[t].y, acc[t].y));
    if (lag < kNLags) {
      ema[t] = __fadd_rn(__fmul_rn(a, 0.2f), __fmul_rn(ema[t], 1.0f - 0.2f));
      if (ema[t] > bv) {
        bv = ema[t];
        bi = lag;
      }
    }
    AVG[lag] = ema[t];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(bv, off);
    const int oi = __shfl_xor(bi, off);
    argmax_combine(bv, bi, ov, oi);
  }
  if (lane == 0) {
    sc.am_v[wave] = bv;
    sc.am_i[wave] = bi;
  }
  __syncthreads(); // AVG and the per-wave maxima are visible
  bv = sc.am_v[0];
  bi = sc.am_i[0];
#pragma unroll
  for (int w = 1; w < kWaves; ++w) argmax_combine(bv, bi, sc.am_v[w], sc.am_i[w]);
  const int peak = (bi == INT_MAX) ? 0 : bi; // srslte_vec_max_fi returns 0 when nothing beats -FLT_MAX
  const float peak_value = AVG[peak];

  // main-lobe walk, both directions at once:
  //   pl_ub = first u >= peak+1 with !(avg[u+1] <= avg[u]) or u >= 9727
  //   pl_lb = last  l <= peak-1 with !(avg[l-1] <= avg[l]) or l <= 1      (0 when peak <= 2)
  int ub = INT_MAX, lb = -1;
  {
    const float nxt = AVG[n0 + kT];
    const float prv = (n0 > 0) ? AVG[n0 - 1] : 0.f;
#pragma unroll
    for (int t = 0; t < kT; ++t) {
      const int u = n0 + t;
      const float cur = ema[t];
      const float up = (t < kT - 1) ? ema[t + 1] : nxt;
      const float dn = (t > 0) ? ema[t - 1] : prv;
      if (u > peak && (!(up <= cur) || u >= kConvLen - 1)) ub = min(ub, u);
      if (peak > 2 && u < peak && u >= 1 && (!(dn <= cur) || u <= 1)) lb = max(lb, u);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ub = min(ub, __shfl_xor(ub, off));
    lb = max(lb, __shfl_xor(lb, off));
  }
  if (lane == 0) {
    sc.ub[wave] = ub;
    sc.lb[wave] = lb;
  }
  __syncthreads();
  ub = sc.ub[0];
  lb = sc.lb[0];
#pragma unroll
  for (int w = 1; w < kWaves; ++w) {
    ub = min(ub, sc.ub[w]);
    lb = max(lb, sc.lb[w]);
  }
  const int pl_ub = ub;
  const int pl_lb = (peak > 2) ? lb : 0;

  // side lobes: max over [0, pl_lb) and over [pl_ub, 9726)
  float sl = -FLT_MAX, sr = -FLT_MAX;
#pragma unroll
  for (int t = 0; t < kT; ++t) {
    const int u = n0 + t;
    if (u < pl_lb && ema[t] > sl) sl = ema[t];
    if (u >= pl_ub && u < kNLags && ema[t] > sr) sr = ema[t];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sl = fmaxf(sl, __shfl_xor(sl, off));
    sr = fmaxf(sr, __shfl_xor(sr, off));
  }
  if (lane == 0) {
    sc.sl[wave] = sl;
    sc.sr[wave] = sr;
  }
  __syncthreads();
  sl = sc.sl[0];
  sr = sc.sr[0];
#pragma unroll
  for (int w = 1; w < kWaves; ++w) {
    sl = fmaxf(sl, sc.sl[w]);
    sr = fmaxf(sr, sc.sr[w]);
  }
  // an empty range makes srslte_vec_max_fi return index 0 of that range
  if (pl_lb == 0) sl = AVG[0];
  if (kNLags - pl_ub <= 0) sr = AVG[pl_ub];
  const float side = sl > sr ? sl : sr;

  // every thread holds the same values; pin them to scalar registers
  SearchResult r;
  r.peak_pos = __builtin_amdgcn_readfirstlane(peak);
  r.peak_value = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(peak_value)));
  r.psr = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(peak_value / side)));
  return r;
}

// ---- pieces of the tracked-emit path, each evaluated in the oracle's operation order ----

__device__ __forceinline__ v2f cmul_plain(v2f a, v2f b)
{
  v2f r;
  r.x = __fsub_rn(__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y));
  r.y = __fadd_rn(__fmul_rn(a.x, b.y), __fmul_rn(a.y, b.x));
  return r;
}

// half of srslte_pss_cfo_compute: sum_{m in [m_begin, m_begin+64)} h[m] * r[m]   
template <typename LoadFn>
__device__ __forceinline__ v2f cfo_half_dot(const v2f *__restrict__ h, int m_begin, LoadFn ld)
{
  v2f y = {0.f, 0.f};
#pragma unroll 1
  for (int m = m_begin; m < m_begin + 64; ++m) {
    const v2f p = cmul_plain(h[m], ld(m));
    y.x = __fadd_rn(y.x, p.x);
    y.y = __fadd_rn(y.y, p.y);
  }
  return y;
}

__device__ __forceinline__ float cfo_from_halves(v2f y0, v2f y1)
{
  v2f c = {y0.x, -y0.y};
  const v2f z = cmul_plain(c, y1);
  return (float)(atan2((double)z.y, (double)z.x) / 3.14159265358979323846);
}

// compute_moving_avg, lib/pss_impl.cc:94-109 (double accumulator)
__device__ __forceinline__ float moving_avg(const float *data, uint32_t npts)
{
  if (!npts) return 0.0f;
  if (npts > (uint32_t)kMovingAvgSz) npts = kMovingAvgSz;
  double acc = 0.0;
#pragma unroll 1
  for (uint32_t i = 0; i < npts; ++i) acc = __dadd_rn(acc, (double)data[i]);
  return (float)(acc / (double)npts);
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

// one (hypothesis, symbol) term of srslte_sync_detect_cp: R = Re sum tail*conj(cp), C = L*avg_power(cp)
__device__ __forceinline__ void cp_term(const v2f *p, int L, float *R, float *C)
{
  float dre = 0.f, pw = 0.f;
#pragma unroll 1
  for (int i = 0; i < L; ++i) {
    const v2f a = p[kSymbolSz + i], b = p[i];
    dre = __fadd_rn(dre, __fadd_rn(__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y)));
    pw = __fadd_rn(pw, __fadd_rn(__fmul_rn(b.x, b.x), __fmul_rn(b.y, b.y)));
  }
  *R = dre;
  *C = __fmul_rn((float)L, pw / (float)L);
}

// decision part of srslte_sync_detect_cp from the six terms; updates the two EMAs
__device__ __forceinline__ int cp_decide(const float R3[2][3], const float C3[2][3], float &m_norm_avg,
                                         float &m_ext_avg)
{
  float R[2], M[2];
  for (int h = 0; h < 2; ++h) {
    float r = 0.f, c = 0.f;
    for (int s = 0; s < 3; ++s) {
      r = __fadd_rn(r, R3[h][s]);
      c = __fadd_rn(c, C3[h][s]);
    }
    R[h] = r;
    M[h] = (c > 0.f) ? r / c : 0.f;
  }
  const float alpha = 0.1f;
  m_norm_avg = __fadd_rn(__fmul_rn(alpha, M[0] / 3.0f), __fmul_rn(1.0f - alpha, m_norm_avg));
  m_ext_avg = __fadd_rn(__fmul_rn(alpha, M[1] / 3.0f), __fmul_rn(1.0f - alpha, m_ext_avg));
  if (m_norm_avg > m_ext_avg) return LT_CP_NORM;
  if (m_norm_avg < m_ext_avg) return LT_CP_EXT;
  return (R[0] > R[1]) ? LT_CP_NORM : LT_CP_EXT;
}

// one of the 62 occupied bins of the 128-point DFT of the SSS symbol (double accumulation)
__device__ __forceinline__ v2f sss_dft_bin(const v2f *sym, int k, const double *dcos, const double *dsin)
{
  const int bin = (k < 31) ? (k - 31) : (k - 30);
  double sr = 0.0, si = 0.0;
#pragma unroll 1
  for (int n = 0; n < kSymbolSz; ++n) {
    const int e = ((bin * n) % 128 + 128) % 128;
    const double c = dcos[e], s = dsin[e];
    const double re = (double)sym[n].x, im = (double)sym[n].y;
    sr = __dadd_rn(sr, __dsub_rn(__dmul_rn(re, c), __dmul_rn(im, s)));
    si = __dadd_rn(si, __dadd_rn(__dmul_rn(re, s), __dmul_rn(im, c)));
  }
  v2f r = {(float)sr, (float)si};
  return r;
}

// |sum_i Y[2i+odd] * c[i] * z[i] * s~[(i+m)%31]|^2 for shift m  (corr_all_sz_partial, M = 1)
__device__ __forceinline__ float sss_corr(const v2f *Y, int odd, const float *__restrict__ c,
                                          const float *__restrict__ z, int zshift,
                                          const float *__restrict__ s, int m)
{
  float re = 0.f, im = 0.f;
#pragma unroll 1
  for (int i = 0; i < 31; ++i) {
    const v2f y = Y[2 * i + odd];
    float yr = __fmul_rn(y.x, c[i]), yi = __fmul_rn(y.y, c[i]);
    if (z) {
      const float zz = z[(i + zshift) % 31];
      yr = __fmul_rn(yr, zz);
      yi = __fmul_rn(yi, zz);
    }
    const float sv = s[(i + m) % 31];
    re = __fadd_rn(re, __fmul_rn(yr, sv));
    im = __fadd_rn(im, __fmul_rn(yi, sv));
  }
  return __fadd_rn(__fmul_rn(re, re), __fmul_rn(im, im));
}

__device__ __forceinline__ int first_max31(const float *corr)
{
  float m = -FLT_MAX;
  int p = 0;
#pragma unroll 1
  for (int i = 0; i < 31; ++i)
    if (corr[i] > m) {
      m = corr[i];
      p = i;
    }
  return p;
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

// Per-workgroup LDS copy of the small tables the sequential (bit-exact) sums walk through, so
// that their dependent loads are LDS reads rather than global-memory round trips.
struct LdsTables {
  double dcos[128], dsin[128];
  v2f taps[128];
  float c0[32], c1[32], s[32], z[32];
};

__device__ __forceinline__ void load_lds_tables(LdsTables &lt_, const DeviceTables *__restrict__ tab, int n_id_2)
{
  for (int i = threadIdx.x; i < 128; i += blockDim.x) {
    lt_.dcos[i] = tab->dft128_cos[i];
    lt_.dsin[i] = tab->dft128_sin[i];
    v2f h = {tab->pss_taps[n_id_2][i].re, tab->pss_taps[n_id_2][i].im};
    lt_.taps[i] = h;
    if (i < 32) {
      lt_.c0[i] = tab->sss_c0[n_id_2][i];
      lt_.c1[i] = tab->sss_c1[n_id_2][i];
      lt_.s[i] = tab->sss_s[i];
      lt_.z[i] = tab->sss_z[i];
    }
  }
}

// sss_impl::work body after the tag check (lib/sss_impl.cc:102-150) on S = half-frame samples
// [480, 960) in LDS.  All threads of the block call it; results are uniform.
struct SssResult {
  int cp, m0, m1, n_id_1;
  float m0_value, m1_value;
};

__device__ __forceinline__ SssResult sss_stage(const v2f *S, const LdsTables &lt_, const DeviceTables *__restrict__ tab,
                                              float &m_norm_avg, float &m_ext_avg, Scratch &sc)
{
  const int tid = threadIdx.x;
  if (tid < 6) { // 2 hypotheses x 3 symbols, each summed sequentially by one lane
    const int hyp = tid / 3, sym = tid % 3;
    const int L = hyp ? 32 : 9;
    const int p = kSlotLen - 3 * (kSymbolSz + L) + sym * (kSymbolSz + L) - kSssBase;
    cp_term(S + p, L, &sc.cpR[hyp][sym], &sc.cpC[hyp][sym]);
  }
  __syncthreads();
  SssResult r;
  r.cp = cp_decide(sc.cpR, sc.cpC, m_norm_avg, m_ext_avg);
  const int cp_len = (r.cp == LT_CP_NORM) ? 9 : 32;               // srslte_sync_set_cp
  const int sss_idx = kSlotLen - 2 * kSymbolSz - cp_len - kSssBase; // lib/sss_impl.cc:110
  if (tid < 62) sc.Y[tid] = sss_dft_bin(S + sss_idx, tid, lt_.dcos, lt_.dsin);
  __syncthreads();
  if (tid < 31) sc.corr[tid] = sss_corr(sc.Y, 0, lt_.c0, nullptr, 0, lt_.s, tid);
  __syncthreads();
  r.m0 = first_max31(sc.corr);
  r.m0_value = sc.corr[r.m0];
  __syncthreads();
  if (tid < 31)
    sc.corr[tid] = sss_corr(sc.Y, 1, lt_.c1, lt_.z, r.m0 % 8, lt_.s, tid);
  __syncthreads();
  r.m1 = first_max31(sc.corr);
  r.m1_value = sc.corr[r.m1];
  r.n_id_1 = sss_n_id_1(tab, (uint32_t)r.m0, (uint32_t)r.m1);
  return r;
}

} // namespace lt
