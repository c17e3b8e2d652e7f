// gfx950 kernels of the LTE cell-search hot path.
//
// lt_batch_run / lt_batch_push:
//   lt_reset_kernel    -- lt_batch_reset.
//   lt_plan_kernel, lt_scan_kernel, lt_resolve_kernel (lt_scan_kernels.h) -- the searches of all chains in rounds,
//                         strip-wise, exponential averages in registers, one forward transform per block shared by the
//                         chains of a window sequence; pss_impl::general_work (lib/pss_impl.cc:155-223) on the results.
//   lt_chain_kernel (lt_chain_kernel.h) -- one workgroup per chain: general_work call by call for whatever the rounds
//                         leave over (and for every chain in mode 0), one lt_record per call; then, the chain's records
//                         being final, CFO estimate and mean_cfo() of its tracked half-frames (lib/pss_impl.cc:199-204).
//   (the per-record kernels walk a work list of the tracked half-frames that lt_chain_kernel leaves behind:)
//   lt_cpsss_kernel, lt_cp_scan_kernel -- sss_impl::work (lib/sss_impl.cc:83-156) on them, lt_hf_kernel -- the
//                         pss block's output stream (optional capture): per record, afterwards (they feed records only).
//   lt_hist_kernel     -- keeps the tail of the chunk for the next lt_batch_push.
// lt_find_kernel, lt_cfo_*, lt_sync_*  -- single-call forms behind the srslte_* shaped C-ABI.
// lt_decim_kernel -- decimating FIR in front of the path.
#include "lt_chain_kernel.h"

namespace lt {

template <typename Args> __device__ __forceinline__ Capture make_capture(const Args &a, int channel)
{
  Capture c;
  c.iq = a.iq + (int64_t)channel * a.ch_stride;
  c.hist = a.hist ? a.hist + (int64_t)channel * a.hist_stride : nullptr;
  c.t0 = a.t0;
  c.n = a.n;
  c.hist_len = a.hist ? a.hist_len : 0;
  return c;
}

// x[i] for 0 <= i < n, else 0
__device__ __forceinline__ v2f ld_clip(const v2f *__restrict__ x, int64_t n, int64_t i)
{
  v2f z = {0.f, 0.f};
  return (i < 0 || i >= n) ? z : x[i];
}

// ---------------------------------------------------------------- single-call kernels

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
  __shared__ v2f Y[62];
  __shared__ LdsTables ltab;
  __shared__ v2f S[kSymbolSz];
  const int tid = threadIdx.x;
  load_lds_tables(ltab, tab, n_id_2);
  if (tid < kSymbolSz) S[tid] = sym[tid];
  __syncthreads();
  SssResult r;
  if (tid < 64) {
    sss_fft128_wave(S, ltab.tw128, Y);
    sss_m0m1_wave(Y, ltab, r);
  }
  if (tid == 0) {
    res->m0 = r.m0; res->m1 = r.m1; res->m0_value = r.m0_value; res->m1_value = r.m1_value;
  }
}

// ---------------------------------------------------------------- decimating FIR
// y[c][k] = sum_j taps[j] * x[c][k*D - j], accumulated in tap order as a chain of fused multiply-adds (the oracle's
// order, oracle/lte_oracle.c orc_decimate: one rounding per tap).
//
// lt_decim_kernel<D, NTAPS> (the reference's rates, D = 4 / 8 / 16 with GNU Radio's default 131 / 263 / 525 taps): a thread
// owns B = 32 / D ADJACENT outputs, so that one sample read serves B taps -- sample i is tap j of output k and tap j + D of
// output k + 1 --: per step s one 8-byte LDS read and B packed FMAs, acc[b] += taps[s + b D] * X_s, where X_s is
// the sample D B l + const - s for lane l.  Every accumulator sees its taps in ascending order.  The block's input span is
// staged in LDS in pairs of samples, 16 rows (decim_row2 below: sample u in entry (u + 1) / 2, entry v at row v mod 16,
// column v / 16): for a given pair of steps all lanes of a wave read one row at consecutive columns (conflict-free 16-byte
// reads).  16 rows x 275 entries x 16 B = 70 KB: two
// blocks per CU, one staging its samples while the other computes.  The taps are wave-uniform: scalar loads, 32 at a time
// from a table laid out in the order the steps use them (decim_tap_table).
// Algorithmic bytes: 8 B in + 8/D B out per input sample; arithmetic: 2 NTAPS / D FMAs per input sample.
//
// lt_decim_generic_kernel: any other rate, one thread per output, same arithmetic.
#ifndef LT_DECIM_P
#define LT_DECIM_P 32
#endif
constexpr int kDecimPhases = LT_DECIM_P;
constexpr int kDecimThreads = 8192 / kDecimPhases; // 32 phases x 256 threads (128 and 512 measured: 3 % and 15 % slower)
constexpr int kDecimGeneric = 256;

__host__ __device__ constexpr int decim_lead(int ntaps) { return ((ntaps - 1 + kDecimPhases - 1) / kDecimPhases) * kDecimPhases; }
// row length in samples: >= threads + lead / phases, = 1 mod 16
__host__ __device__ constexpr int decim_row(int ntaps)
{
  int r = kDecimThreads + decim_lead(ntaps) / kDecimPhases;
  while (r % 16 != 1) ++r;
  return r;
}

// Samples in pairs (round 5): sample u sits in the 16-byte LDS entry v = (u + 1) >> 1, half (u + 1) & 1; entry v at row
// v mod 16, column v / 16 (rows of R2 entries, R2 odd: the staging stores of a 16-lane group -- eight or nine entries in
// consecutive rows -- fall on distinct banks).  Two consecutive steps n (even), n + 1 of a lane read the samples u, u - 1 =
// both halves of ONE entry: one ds_read_b128 per two steps instead of two ds_read_b64 -- half the LDS instructions for the
// same bytes.  Same sums in the same order: bit-exact, 2-4 % faster (profiles/r05_decim_probe.txt).
__host__ __device__ constexpr int decim_row2(int ntaps) // row length in 16-byte entries: >= threads + lead / phases + 2, odd
{
  int r = kDecimThreads + decim_lead(ntaps) / kDecimPhases + 2;
  while (r % 2 != 1) ++r;
  return r;
}

__device__ __forceinline__ v2f fma_tap(float t, v2f x, v2f acc)
{
  const v2f tt = {t, t};
  return __builtin_elementwise_fma(tt, x, acc);
}

template <int kD, int kNtaps>
__global__ __launch_bounds__(kDecimThreads) void lt_decim_kernel(const v2f *__restrict__ x, v2f *__restrict__ y,
                                                                    int64_t n_in, int64_t n_out, const float *__restrict__ tt)
{
  constexpr int P = kDecimPhases, B = P / kD, T = kDecimThreads, R = decim_row(kNtaps);
  static_assert(P % kD == 0 && B >= 1 && (P & (P - 1)) == 0, "the decimation must divide the phases");
  constexpr int lead = decim_lead(kNtaps);
  constexpr int span = P * T + lead;
  static_assert(T + lead / P <= R, "row too short");
  constexpr int n_steps = kNtaps + (B - 1) * kD; // s = -(B - 1) D .. NTAPS - 1
  constexpr int W0 = lead + P - kD; // u = 32 tid + W0 - n for step n
  static_assert(P == 32 && (lead % 2) == 0 && (kD % 2) == 0 && (T % 32) == 0, "the pair layout assumes 32 phases and an even alignment");
  constexpr int R2 = decim_row2(kNtaps);
  typedef float v4f __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) v4f xs4[16 * R2];
  v2f *xs = reinterpret_cast<v2f *>(xs4);
  static_assert(((P * T + lead + 1) / 2) / 16 + 1 <= R2 && R <= 2 * R2, "row too short for the pairs");
  const int tid = threadIdx.x, c = blockIdx.y;
  const int64_t k0 = (int64_t)blockIdx.x * (T * B);
  const int64_t first = k0 * kD - lead; // input sample of u = 0
  // ---- stage the span: the channel's samples [first, first + span) as a buffer resource (what lies outside the channel
  // -- in front of the stream, behind its end -- is out of range and reads as zero)
  {
    const int64_t base_i = first > 0 ? first : 0;
    int64_t avail = n_in - base_i;
    avail = avail < 0 ? 0 : avail > span ? span : avail;
    const v2f *bp = x + (int64_t)c * n_in + base_i;
    const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(bp), 0, (int)avail * (int)sizeof(v2f), 0x00020000);
    const int shift = (int)(first - base_i); // <= 0: offsets of samples in front of the stream wrap to huge unsigned values
    // sample u = tid + i T: entry v = ((tid + 1) >> 1) + i T / 2 -- row and half fixed per thread, 8 columns on per load
    const int v0 = (tid + 1) >> 1;
    v2f *dst = xs + 2 * ((v0 & 15) * R2 + (v0 >> 4)) + ((tid + 1) & 1);
    constexpr int kDstStep = 2 * (T / 32);
    constexpr int n_it = (span + T - 1) / T;
    constexpr int kBatch = n_it; // all of a thread's loads in flight together: one memory round trip per block
#pragma unroll
    for (int i0 = 0; i0 < n_it; i0 += kBatch) {
      v2f v[kBatch];
#pragma unroll
      for (int i = 0; i < kBatch; ++i) {
        const int u = tid + (i0 + i) * T;
        if (i0 + i < n_it) v[i] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(win, (shift + u) * (int)sizeof(v2f), 0, 0));
      }
#pragma unroll
      for (int i = 0; i < kBatch; ++i)
        if (i0 + i < n_it && (tid + (i0 + i) * T < span)) dst[(i0 + i) * kDstStep] = v[i];
    }
  }
  __syncthreads();
  v2f acc[B];
#pragma unroll
  for (int b = 0; b < B; ++b) acc[b] = v2f{0.f, 0.f};
  const v4f *xl4 = xs4 + tid;
  // Step n (= s + (B - 1) D) reads sample u = 32 tid + W0 - n: row (W0 - n) mod 32, column tid + (W0 - n) / 32; output b
  // takes it with tap j = n + (b - (B - 1)) D.  The steps run in units of U = D (32 taps each: two scalar loads of 16
  // from the table `tt`, laid out [unit][step][output] by decim_tap_table below); a unit's LDS reads and taps are fetched
  // while the unit before it computes.  Everything is unrolled: which (step, output) pairs have a tap is known here.
  constexpr int U = 32 / B, G = P / U; // G units per chunk of P steps (even, so a unit's buffer is its index's parity)
  static_assert(U * B == 32 && G % 2 == 0, "a unit is 32 taps; the two buffers alternate in step with the chunks");
  constexpr int m_last = (n_steps - 1) / P; // chunks 0 and m_last have (step, output) pairs without a tap: unrolled apart
  static_assert(m_last >= 2 && P * (m_last - 1) + P - 1 <= kNtaps - 1 && P * 1 - (B - 1) * kD >= 0, "the steady chunks must lie inside the taps");
  static_assert(U % 2 == 0 && (W0 % 2) == 0, "steps pair up inside a unit");
  v4f xq[2][U / 2];
  float tv[2][32];
  // unit g of chunk m: steps n = 32 m + g U + u
  // (steps n = 32 m + r and n + 1, r even: the samples 32 tid + e and 32 tid + e - 1, e = W0 - n even -- the two halves of
  // entry 16 tid + e / 2: row (e / 2) mod 16, column tid + e / 32)
#define LT_DECIM_X(g, u) (((u) & 1) ? v2f{xq[(g) & 1][(u) / 2].x, xq[(g) & 1][(u) / 2].y} : v2f{xq[(g) & 1][(u) / 2].z, xq[(g) & 1][(u) / 2].w})
#define LT_DECIM_LOAD(m, g, kPeel)                                                                       \
  {                                                                                                      \
    _Pragma("unroll") for (int u = 0; u < U; u += 2)                                                     \
    {                                                                                                    \
      const int r = (g) * U + u;                                                                         \
      if (!kPeel || P * (m) + r < n_steps) xq[(g) & 1][u / 2] = xl4[(((W0 - r) / 2) & 15) * R2 + ((W0 - r) / P) - (m)]; \
    }                                                                                                    \
    const float *tp = tt + ((m) * G + (g)) * 32;                                                         \
    _Pragma("unroll") for (int i = 0; i < 32; ++i) tv[(g) & 1][i] = tp[i];                               \
  }
#define LT_DECIM_FMA(m, g, kPeel)                                                                        \
  {                                                                                                      \
    _Pragma("unroll") for (int u = 0; u < U; ++u)                                                        \
    {                                                                                                    \
      const int n = P * (m) + (g) * U + u;                                                               \
      _Pragma("unroll") for (int b = 0; b < B; ++b)                                                      \
      {                                                                                                  \
        const int j = n + (b - (B - 1)) * kD;                                                            \
        if (!kPeel || (n < n_steps && j >= 0 && j < kNtaps)) acc[b] = fma_tap(tv[(g) & 1][u * B + b], LT_DECIM_X(g, u), acc[b]); \
      }                                                                                                  \
    }                                                                                                    \
  }
  // one chunk: while unit g computes, unit g + 1 (or the next chunk's first) is on its way.  Scalar loads return in any order,
  // so a wait for a unit's taps is a wait for everything the counter they share with the LDS reads still holds: taken BEFORE the
  // next unit's loads are issued it waits for loads that have had a whole unit's FMAs to land; left to the compiler it comes in
  // front of the unit's first FMA, behind the next unit's loads, and drains those too -- a full trip to the LDS and the scalar
  // cache in the open, every other unit.
#ifndef LT_DECIM_EARLY_WAIT
#define LT_DECIM_EARLY_WAIT 1
#endif
#if LT_DECIM_EARLY_WAIT
#define LT_DECIM_WAIT { __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */ __builtin_amdgcn_sched_barrier(0); }
#else
#define LT_DECIM_WAIT
#endif
#define LT_DECIM_CHUNK(m, kPeel, kMore)                                                                  \
  {                                                                                                      \
    _Pragma("unroll") for (int g = 0; g < G; ++g)                                                        \
    {                                                                                                    \
      LT_DECIM_WAIT                                                                                      \
      if (g + 1 < G) LT_DECIM_LOAD(m, g + 1, kPeel)                                                      \
      else if (kMore) LT_DECIM_LOAD((m) + 1, 0, ((m) + 1 == m_last))                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
      LT_DECIM_FMA(m, g, kPeel)                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
    }                                                                                                    \
  }
  LT_DECIM_LOAD(0, 0, true)
  LT_DECIM_CHUNK(0, true, true)
#pragma unroll 1
  for (int m = 1; m < m_last; ++m) LT_DECIM_CHUNK(m, false, true)
  LT_DECIM_CHUNK(m_last, true, false)
#undef LT_DECIM_CHUNK
#undef LT_DECIM_WAIT
#undef LT_DECIM_FMA
#undef LT_DECIM_LOAD
#undef LT_DECIM_X
  const int64_t k = k0 + (int64_t)B * tid;
  v2f *yo = y + (int64_t)c * n_out + k;
  if (k + B <= n_out) {
#pragma unroll
    for (int b = 0; b < B; ++b) yo[b] = acc[b];
  } else {
#pragma unroll
    for (int b = 0; b < B; ++b)
      if (k + b < n_out) yo[b] = acc[b];
  }
}

// any decimation: one thread per output, samples from global memory (the taps walk down the stream: a lane's reads
// are consecutive and stay in its cache lines)
__global__ __launch_bounds__(kDecimGeneric) void lt_decim_generic_kernel(const v2f *__restrict__ x, v2f *__restrict__ y,
                                                                         int64_t n_in, int64_t n_out, int D,
                                                                         const float *__restrict__ taps, int ntaps)
{
  const int c = blockIdx.y;
  const int64_t k = (int64_t)blockIdx.x * kDecimGeneric + threadIdx.x;
  if (k >= n_out) return;
  const v2f *xc = x + (int64_t)c * n_in;
  v2f acc = {0.f, 0.f};
  const int64_t base = k * D;
  for (int j = 0; j < ntaps; ++j) acc = fma_tap(taps[j], ld_clip(xc, n_in, base - j), acc);
  y[(int64_t)c * n_out + k] = acc;
}

// ---------------------------------------------------------------- the sss block on the tracked half-frames
// sss_impl::work (lib/sss_impl.cc:83-156) in two launches.  (The pss block's share -- CFO estimate, ring, mean_cfo() --
// is the per-chain kernel's tail, lt_chain_kernel.h.)
//   lt_cpsss_kernel   per record, chip-wide (the work list the per-chain kernel left): the half-frame's samples
//                     [480, 960) rotated exactly as srslte_cfo_correct did (lib/pss_impl.cc:204), the CP correlations
//                     of srslte_sync_detect_cp (lib/sss_impl.cc:104-106) and srslte_sss_m0m1_partial (:110-116) under
//                     BOTH CP hypotheses -- which of the two SSS positions counts is only known once the CP averages
//                     have been run through the chain's records in call order --: metrics and the two (m0, m1) results go
//                     to the per-record scratch.
//   lt_cp_scan_kernel per chain, in call order: the two exponential averages of srslte_sync_detect_cp and its decision
//                     (reset where the sss block sees the tracking_lost tag, lib/sss_impl.cc:91-100), then the SSS result
//                     of the chosen CP, srslte_sss_N_id_1, cell_id and the tags (lib/sss_impl.cc:118-150).
// Neither feeds the PSS state machine of the per-chain kernel.
struct SssPair {
  int m0, m1;
  float m0_value, m1_value;
};
struct RecordScratch {
  CpMetrics cp;      // M_norm, M_ext, R_norm, R_ext of this half-frame
  SssPair sss[2];    // srslte_sss_m0m1_partial with the SSS symbol where a normal / an extended CP puts it
};
static_assert(sizeof(RecordScratch) == 48, "RecordScratch layout");

struct CpArgs {
  const v2f *iq;
  int64_t ch_stride;
  const v2f *hist;
  int64_t hist_stride;
  int64_t t0, n, hist_len;
  lt_record *recs;      // [n_chains][rec_cap]
  int rec_cap;
  const int *rec_count; // [n_chains]
  RecordScratch *scratch; // [n_chains][rec_cap]
  ChainState *state;    // [n_chains]
  const DeviceTables *tab;
  int n_chains;
  const int2 *list;     // (chain, record) of every half-frame that went through the SSS stage in this run
  const int *list_count;
};

constexpr int kListBlocks = 2048; // workgroups (x 4 waves) that walk the work list of the tracked half-frames (built by lt_chain_kernel)

// The pss block's output stream (optional capture, lt_batch_enable_halfframes): every emitted half-frame,
// CFO-corrected when tracking (lib/pss_impl.cc:188-193,204).  One workgroup per record.
struct HfArgs {
  const v2f *iq;
  int64_t ch_stride;
  const v2f *hist;
  int64_t hist_stride;
  int64_t t0, n, hist_len;
  const lt_record *recs;
  int rec_cap;
  const int *rec_count;
  v2f *out_hf; // [n_chains][out_cap][9600]
  int out_cap;
  const DeviceTables *tab;
};
__global__ __launch_bounds__(256) void lt_hf_kernel(HfArgs a)
{
  __shared__ int n_before;
  const int chain = blockIdx.x, idx = blockIdx.y, tid = threadIdx.x;
  if (idx >= a.rec_count[chain]) return;
  const lt_record *recs = a.recs + (int64_t)chain * a.rec_cap;
  const lt_record rec = recs[idx];
  if (!(rec.flags & LT_F_EMITTED)) return;
  if (tid == 0) n_before = 0;
  __syncthreads();
  int cnt = 0;
  for (int i = tid; i < idx; i += 256) cnt += (recs[i].flags & LT_F_EMITTED) ? 1 : 0;
  if (cnt) atomicAdd(&n_before, cnt);
  __syncthreads();
  const int slot = n_before;
  if (slot >= a.out_cap) return;
  v2f *out = a.out_hf + ((int64_t)chain * a.out_cap + slot) * kHalfFrame;
  const Capture cap = make_capture(a, chain / 3);
  if (rec.flags & LT_F_SSS_RAN) {
    const float freq = -rec.mean_cfo / (float)kSymbolSz; // srslte_cfo_correct, lib/pss_impl.cc:204
    const float phase_inc = __fmul_rn(freq, (float)kCexpTabSize);
    for (int i = tid; i < kHalfFrame; i += 256) out[i] = cfo_rotate(a.tab->cexptab, phase_inc, i, ld_abs(cap, rec.out_pos + i));
  } else {
    for (int i = tid; i < kHalfFrame; i += 256) out[i] = ld_abs(cap, rec.out_pos + i);
  }
}

__global__ __launch_bounds__(256) void lt_cpsss_kernel(CpArgs a)
{
  __shared__ LdsTables ltab[4]; // (the m-sequence masks depend on the chain's N_id_2: one set per wave)
  __shared__ v2f S[4][kSssLen];
  __shared__ v2f Y[4][62];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_items = *a.list_count;
  if ((int)blockIdx.x * 4 >= n_items) return;
  for (int i = lane; i < 64; i += 64) ltab[w].tw128[i] = v2f{(float)a.tab->dft128_cos[i], (float)a.tab->dft128_sin[i]};
  if (lane == 0) {
    ltab[w].s_mask = a.tab->sss_s_mask;
    ltab[w].z_mask = a.tab->sss_z_mask;
  }
  for (int item = blockIdx.x * 4 + w; item < n_items; item += gridDim.x * 4) {
    const int2 it = a.list[item];
    const int chain = __builtin_amdgcn_readfirstlane(it.x), idx = __builtin_amdgcn_readfirstlane(it.y);
    const int n_id_2 = chain % 3;
    const lt_record *rec = a.recs + (int64_t)chain * a.rec_cap + idx;
    const int64_t hf0 = rec->out_pos;
    const Capture cap = make_capture(a, chain / 3);
    const float freq = -rec->mean_cfo / (float)kSymbolSz; // lib/pss_impl.cc:204
    const float phase_inc = __fmul_rn(freq, (float)kCexpTabSize);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      ltab[w].c0_mask = a.tab->sss_c0_mask[n_id_2];
      ltab[w].c1_mask = a.tab->sss_c1_mask[n_id_2];
    }
    {
      const int64_t i0 = hf0 + kSssBase - cap.t0;
      if (i0 >= 0 && i0 + kSssLen <= cap.n) { // inside the chunk (always, but for a half-frame across the start of a pushed chunk)
        const v2f *__restrict__ p = cap.iq + i0 + lane;
        v2f v[(kSssLen + 63) / 64];
#pragma unroll
        for (int k = 0; k < (kSssLen + 63) / 64; ++k) v[k] = (lane + 64 * k < kSssLen) ? p[64 * k] : v2f{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < (kSssLen + 63) / 64; ++k)
          if (lane + 64 * k < kSssLen) S[w][lane + 64 * k] = cfo_rotate(a.tab->cexptab, phase_inc, kSssBase + lane + 64 * k, v[k]);
      } else {
        for (int i = lane; i < kSssLen; i += 64)
          S[w][i] = cfo_rotate(a.tab->cexptab, phase_inc, kSssBase + i, ld_abs(cap, hf0 + kSssBase + i));
      }
    }
    __builtin_amdgcn_wave_barrier();
    RecordScratch out;
    out.cp = cp_metrics_wave(S[w]);
#pragma unroll
    for (int hyp = 0; hyp < 2; ++hyp) {
      const int cp_len = hyp ? 32 : 9;                                   // srslte_sync_set_cp
      const int sss_idx = kSlotLen - 2 * kSymbolSz - cp_len - kSssBase;  // lib/sss_impl.cc:110
      SssResult r;
      r.cp = hyp;
      __builtin_amdgcn_wave_barrier();
      sss_fft128_wave(S[w] + sss_idx, ltab[w].tw128, Y[w]);
      __builtin_amdgcn_wave_barrier();
      sss_m0m1_wave(Y[w], ltab[w], r);
      out.sss[hyp].m0 = r.m0;
      out.sss[hyp].m1 = r.m1;
      out.sss[hyp].m0_value = r.m0_value;
      out.sss[hyp].m1_value = r.m1_value;
    }
    if (lane == 0) a.scratch[(int64_t)chain * a.rec_cap + idx] = out;
  }
}

// One wave per chain: 64 records at a time are fetched one per lane and everything that does not depend on
// the running averages is formed in parallel; two lanes then run the two averages through the block's records in call
// order; decisions, the choice between the two SSS results and the stores are per lane again.
__global__ __launch_bounds__(64) void lt_cp_scan_kernel(CpArgs a)
{
  __shared__ float s_in[2][64], s_out[2][64];
  __shared__ uint32_t s_ev[64];
  const int chain = blockIdx.x, lane = threadIdx.x;
  const int n_id_2 = chain % 3;
  ChainState *st = a.state + chain;
  float m_norm_avg = st->m_norm_avg, m_ext_avg = st->m_ext_avg;
  lt_record *recs = a.recs + (int64_t)chain * a.rec_cap;
  const RecordScratch *sc = a.scratch + (int64_t)chain * a.rec_cap;
  const int count = a.rec_count[chain];
  // (a record's flags and its scratch entry are fetched together, and the next 64 records' while this block is walked: the
  // scratch entry of a record that is no tracked half-frame holds whatever an earlier run left there and is not looked at)
  uint32_t flags_n = lane < count ? recs[lane].flags : 0u;
  RecordScratch c_n = lane < count ? sc[lane] : RecordScratch{};
  for (int base = 0; base < count; base += 64) {
    const int i = base + lane;
    const bool valid = i < count;
    const uint32_t flags = flags_n;
    const bool ran = (flags & LT_F_SSS_RAN) != 0;
    RecordScratch c = c_n;
    if (!ran) c.cp.m[0] = c.cp.m[1] = c.cp.r[0] = c.cp.r[1] = 0.f;
    {
      const int in = i + 64;
      flags_n = in < count ? recs[in].flags : 0u;
      if (in < count) c_n = sc[in];
    }
    const float in_norm = cp_avg_input(c.cp.m[0]), in_ext = cp_avg_input(c.cp.m[1]);
    const unsigned long long run = __ballot(ran), lost = __ballot((flags & LT_F_TAG_LOST) != 0);
    float my_norm = m_norm_avg, my_ext = m_ext_avg; // state as of this lane's record
    if (run | lost) {
      // The two averages are sequential in the records; everything else is not.  Inputs and event bits go to LDS, lane 0
      // runs M_norm_avg and lane 1 M_ext_avg through the 64 records (eight at a time: the reads of a group are in flight
      // together, the recurrence itself is two dependent operations per record), every lane then picks up the state as
      // of its own record.  A tracking_lost tag resets first (lib/sss_impl.cc:91-100), a tracked half-frame then steps.
      s_in[0][lane] = in_norm;
      s_in[1][lane] = in_ext;
      s_ev[lane] = (ran ? 1u : 0u) | ((flags & LT_F_TAG_LOST) ? 2u : 0u);
      __builtin_amdgcn_wave_barrier();
      if (lane < 2) {
        float avg = lane ? m_ext_avg : m_norm_avg;
        const float *in = s_in[lane];
        float *out = s_out[lane];
#pragma unroll 1
        for (int j0 = 0; j0 < 64; j0 += 8) {
          float x[8];
          uint32_t e[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            x[k] = in[j0 + k];
            e[k] = s_ev[j0 + k];
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            avg = (e[k] & 2u) ? 0.f : avg;
            const float stepped = cp_avg_step(x[k], avg);
            avg = (e[k] & 1u) ? stepped : avg;
            out[j0 + k] = avg;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      my_norm = s_out[0][lane];
      my_ext = s_out[1][lane];
      m_norm_avg = s_out[0][63];
      m_ext_avg = s_out[1][63];
      __builtin_amdgcn_wave_barrier();
    }
    if (valid) {
      recs[i].m_norm_avg = my_norm;
      recs[i].m_ext_avg = my_ext;
      if (ran) {
        const bool ext = cp_choose(my_norm, my_ext, c.cp.r[0], c.cp.r[1]) == LT_CP_EXT;
        const SssPair r = ext ? c.sss[1] : c.sss[0];
        const int n_id_1 = sss_n_id_1(a.tab, (uint32_t)r.m0, (uint32_t)r.m1); // lib/sss_impl.cc:118
        uint32_t f = flags | (ext ? LT_F_CP_EXT : 0u);
        recs[i].m0 = r.m0;
        recs[i].m1 = r.m1;
        recs[i].m0_value = r.m0_value;
        recs[i].m1_value = r.m1_value;
        recs[i].n_id_1 = n_id_1;
        if (n_id_1 >= 0) {
          recs[i].cell_id = 3 * n_id_1 + n_id_2; // srslte_sync_get_cell_id
          f |= LT_F_TAG_CELL;
        }
        if (f != flags) recs[i].flags = f;
      }
    }
  }
  if (lane == 0) {
    st->m_norm_avg = m_norm_avg;
    st->m_ext_avg = m_ext_avg;
  }
}

// ---------------------------------------------------------------- history for the next chunk
// new_hist = last hist_cap samples of (old history ++ chunk), per channel; the chains never reach further back
// than kCallReach + kSlotLen samples from the end of a chunk (lt_chain_kernel.h).
struct HistArgs {
  const v2f *iq;
  int64_t ch_stride;
  const v2f *hist_old;
  v2f *hist_new;
  int64_t hist_stride; // = capacity
  int64_t n, hist_len_old, hist_len_new;
};
__global__ __launch_bounds__(256) void lt_hist_kernel(HistArgs a)
{
  const int c = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.hist_len_new) return;
  // new history sample i is stream sample (end - hist_len_new + i); end = old history ++ chunk
  const int64_t j = a.n - a.hist_len_new + i; // index into the chunk (negative: into the old history)
  v2f v;
  if (j >= 0) v = a.iq[(int64_t)c * a.ch_stride + j];
  else v = a.hist_old[(int64_t)c * a.hist_stride + a.hist_len_old + j];
  a.hist_new[(int64_t)c * a.hist_stride + i] = v;
}

// Diagnostics (lt_batch_debug_hold_records): one wave that sits on its stream for `ticks` of the 100 MHz wall clock,
// so that a test can hold a run's per-record kernels back until the next run's launches are through.
__global__ __launch_bounds__(64) void lt_delay_kernel(long long ticks)
{
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
}

} // namespace lt

// Host-callable launchers (the C-ABI in lt_api.cpp only sees these).
#include "lt_launch.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace lt {

constexpr int kMaxRounds = 256; // scan rounds per run at most (the per-chain kernel takes what is left)
// Batches up to this size scan from the window cache (lt_batch_set_window_cache >= 2): lt_metric_kernel forms a round's lag
// metrics chip-wide, lt_scan_cached_kernel runs the averages and heads in segments.  Through the transforms a round is a serial
// chain of windows per strip whatever the batch (2.3 us per window with one strip per workgroup); from the cache it costs 1.5 x
// the transforms (every root its own forward one) but all of them at once.  One lt_batch_run of N channels x 1 s, us (one box,
// tools/cached_crossover.py), transforms / cache: N = 5: 518 / 298, 8: 527 / 361, 12: 536 / 466, 16: 546 / 570, 24: 689 / 788.
constexpr int kCachedScanMaxChannels = 12;
int cached_scan_max_channels() { return kCachedScanMaxChannels; }

int call_reach() { return kCallReach; }
int scan_job_bytes() { return (int)sizeof(ScanJob); }
int scan_parts_bytes_per_chain() { return (int)((sizeof(StripHead) + sizeof(float) * kRowLags) * kStrips * kScanMaxRound); } // heads, then rows
int scan_max_round() { return kScanMaxRound; }
int scan_max_rounds() { return kMaxRounds; }
// the run's counters: job counters of every round, the two work lists', the window cache's slot counter with six
// diagnostic counts behind it (lt_batch_cache_stats), the slots handed out per lt_metric_kernel launch
constexpr int kCtrList = 2 * (kMaxRounds + 1), kCtrSlots = kCtrList + 2, kCtrLaunchSlots = kCtrSlots + 7;
int scan_counter_ints() { return kCtrLaunchSlots + kMaxRounds + 2; }
int cache_counter_index() { return kCtrSlots; } // job counters of every round + the per-record work lists' + the window cache's
int cache_slot_bytes() { return (int)(cache_slot_floats() * sizeof(float)); }
int cache_max_per_chain() { return kCacheMaxPerChain; }
int per_record_scratch_bytes() { return (int)(sizeof(RecordScratch) + 2 * sizeof(int2)); } // CP metrics and SSS results + one work-list entry for each of the two list slots

// The work lists sit behind the CP metrics and SSS results in the per-record scratch buffer, one per list slot (their counters
// behind the job counters): with lt_batch_set_overlap the per-chain kernel of run k + 1 fills its list on the main
// stream while the per-record kernels of run k still walk theirs on the side stream.
static int2 *work_list(const RunLaunch &R, int n_chains)
{
  const size_t per_slot = (size_t)n_chains * (size_t)R.rec_cap;
  return reinterpret_cast<int2 *>(reinterpret_cast<char *>(R.cp_scratch) + sizeof(RecordScratch) * per_slot) + (R.list_slot ? per_slot : 0);
}

hipError_t launch_run(const RunLaunch &R, hipStream_t stream, hipStream_t records_stream)
{
  int n_launch = 0; // kernel launches of this run (lt_batch_last_launch_count)
#define LT_LAUNCH(...) do { ++n_launch; hipLaunchKernelGGL(__VA_ARGS__); } while (0)
  const int n_chains = R.n_channels * 3;
  if (R.rec_cap <= 0 || n_chains <= 0) return hipErrorInvalidValue;
  if ((R.rec_cap + 3) / 4 > 65535) return hipErrorInvalidValue; // (a chunk of > 37 minutes)
  // (lt_hf_kernel: one grid row per record slot; checked here, before anything is enqueued -- lt_batch_enable_halfframes
  // refuses the combination with a message of its own)
  if (R.out_hf && R.out_cap > 0 && R.rec_cap > 65535) return hipErrorInvalidValue;
  static const bool debug_jobs = getenv("LT_DEBUG_JOBS") != nullptr; // (read once: this is the launch path of every run)
  const v2f *iq = reinterpret_cast<const v2f *>(R.iq);
  const v2f *hist = reinterpret_cast<const v2f *>(R.hist);
  ChainState *state = reinterpret_cast<ChainState *>(R.state);
  ScanJob *jobs = reinterpret_cast<ScanJob *>(R.jobs);
  // (what the scan's strips leave for the resolver: every chain's heads first, the rows behind them)
  StripHead *parts = reinterpret_cast<StripHead *>(R.parts);
  float *part_rows = parts ? reinterpret_cast<float *>(parts + (size_t)n_chains * kScanMaxRound * kStrips) : nullptr;
  const int64_t t1 = R.t0 + R.n;
  bool ask_early = false; // the last resolve launch makes the chains' cache requests (below)
  // ---- rounds of plan, scan, resolve.  An idle chain needs one round per `seg_len` windows; a chain that fires
  // needs a few more (a round ends for it where it re-aligns, where tracking starts, where it is lost).  Rounds
  // with nothing left to do cost three empty launches; whatever a chain has not got through after the last one --
  // and every window the strips cannot resolve -- is the per-chain kernel's.
  {
    // jobs and their counters alternate between two buffers: a round's resolve launch reads this round's jobs
    // and writes the next round's
    ScanJob *jobs2[2] = {jobs, jobs + n_chains};
    PlanArgs p;
    p.state = state; p.rec_count = R.rec_count; p.n_channels = R.n_channels;
    p.rec_cap = R.rec_cap; p.t1 = t1; p.flush = R.flush; p.enable = R.scan && parts ? 1 : 0;
    p.track_after = R.track_after; p.track_every = R.track_every; p.call_reach = kCallReach;
    ScanArgs s;
    s.iq = iq; s.ch_stride = R.ch_stride; s.hist = hist; s.hist_stride = R.hist_stride; s.t0 = R.t0; s.n = R.n;
    s.hist_len = R.hist_len; s.state = state; s.heads = parts; s.rows = part_rows;
    s.tab = R.tab; s.prof = nullptr;
    ResolveArgs q;
    q.state = state; q.heads = parts; q.rows = part_rows; q.recs = R.recs; q.rec_cap = R.rec_cap;
    q.rec_count = R.rec_count; q.t1 = t1; q.flush = R.flush; q.call_reach = kCallReach; q.psr_threshold = R.psr_threshold;
    q.track_after = R.track_after; q.track_every = R.track_every;
    // the chains' requests for the window cache by the last resolve launch instead of a first pass of the per-chain kernel
    // (levels 1, 2, 4 with the cache there and the rounds on; level 3 -- R.two_pass from the host's cue -- keeps its pass 0)
    const bool cache_ok = R.cache && R.slot_owner && R.slot_pos && R.slot_cap > 0 && R.slot_quota >= kCacheMinSearches && R.cache_level >= 1 && R.two_pass;
    ask_early = cache_ok && R.ask_early && R.scan && parts;
    q.ask = 0; q.ask_stream = R.stream ? 1 : 0; q.slot_quota = R.slot_quota;
    q.slot_owner = reinterpret_cast<int2 *>(R.slot_owner); q.slot_pos = reinterpret_cast<int64_t *>(R.slot_pos);
    q.slot_next = R.n_jobs + kCtrSlots; q.launch_slots = R.n_jobs + kCtrLaunchSlots + kMaxRounds + 1;
    const int seg_first = R.seg_first < 1 ? 1 : R.seg_first > kScanMaxRound ? kScanMaxRound : R.seg_first;
    const int64_t windows = (R.n + R.hist_len) / kHalfFrame + 1;
    int seg_len = R.seg_len > kScanMaxRound ? kScanMaxRound : R.seg_len;
    if (seg_len < 1) { // as few rounds as the windows need, all of one length
      const int64_t rest = windows > seg_first ? windows - seg_first : 1;
      const int64_t n_rounds = (rest + kScanMaxRound - 1) / kScanMaxRound;
      seg_len = (int)((rest + n_rounds - 1) / n_rounds);
    }
    int rounds = 1;
    if (p.enable) rounds = 1 + (int)((windows - seg_first + seg_len - 1) / seg_len) + R.extra_rounds;
    if (rounds > kMaxRounds) rounds = kMaxRounds;
    // Strips per workgroup: what fills the chip's CUs in the fewest turns at the cost per window measured for each
    // shape (tools/scan_bench.hip: 3.0 / 3.5 / 4.7 us per window of a three-root sequence with 1 / 2 / 3 strips per
    // workgroup); RunLaunch::scan_groups overrides.
    int G = R.scan_groups;
    if (G < 1 || G > 3) {
      // us per window of a three-root sequence (tools/scan_bench.hip): one strip per workgroup 2.3 (one workgroup per
      // CU) or 3.4 (two: 60 KB of LDS each), two strips 3.3, three strips 4.7; a shape's workgroups go through the
      // chip in turns of n_cus (x 2 for one strip)
      const int n_cus = R.n_cus > 0 ? R.n_cus : 256;
      double best = 0.0;
      for (int g = 1; g <= 3; ++g) {
        const int wgs = R.n_channels * scan_wg_per_job(g);
        double cost;
        if (g == 1) cost = wgs <= n_cus ? 2.3 : (double)((wgs + 2 * n_cus - 1) / (2 * n_cus)) * 3.4;
        else cost = (double)((wgs + n_cus - 1) / n_cus) * (g == 2 ? 3.3 : 4.7);
        if (g == 1 || cost < best - 1e-9) {
          best = cost;
          G = g;
        }
      }
    }
    // one job per channel at most: enough workgroups for all their parts, a multiple of the 8 XCDs
    const int grid = ((R.n_channels * scan_wg_per_job(G) + 7) / 8) * 8;
    p.first_round = 1;
    p.round_windows = seg_first;
    p.jobs = jobs2[0];
    p.n_jobs = R.n_jobs;
    p.jobs_cap = s.jobs_cap = n_chains;
    p.n_counters = scan_counter_ints();
    p.keep_counter = kCtrList + (R.list_slot ? 0 : 1); // the other run's work-list counter (the cache's, behind them, are this run's)
    p.rebase = R.rebase;
    // window cache for the scan rounds themselves: batches so small that a channel's strips are alone on their CUs
    // and a round is bound by a root wave's instruction stream -- without the transforms that stream is 330
    // instructions instead of 840
    const bool cached_scan = p.enable && R.cache && R.slot_owner && R.slot_pos && R.slot_cap > 0 && R.cache_level >= 2 &&
                             R.n_channels <= cached_scan_max_channels();
    p.cached = cached_scan ? 1 : 0;
    p.slot_owner = reinterpret_cast<int2 *>(R.slot_owner);
    p.slot_pos = reinterpret_cast<int64_t *>(R.slot_pos);
    p.slot_next = R.n_jobs + kCtrSlots;
    p.round_slots = R.n_jobs + kCtrLaunchSlots;
    p.slot_cap = R.slot_cap;
    s.cache = cached_scan ? reinterpret_cast<const float *>(R.cache) : nullptr;
    MetricArgs mm;
    mm.iq = iq; mm.ch_stride = R.ch_stride; mm.hist = hist; mm.hist_stride = R.hist_stride; mm.t0 = R.t0; mm.n = R.n;
    mm.hist_len = R.hist_len; mm.slot_owner = p.slot_owner; mm.slot_pos = p.slot_pos; mm.launch_slots = R.n_jobs + kCtrLaunchSlots;
    mm.slot_cap = R.slot_cap; mm.quota = 0; mm.cache = reinterpret_cast<float *>(R.cache); mm.tab = R.tab;
    const int metric_grid = 4 * (R.n_cus > 0 ? R.n_cus : 256);
    // cached jobs go through lt_scan_cached_kernel (a round's windows in segments); a job that got no slots -- possible only
    // when the cache holds fewer than a round's windows for every chain -- stays with lt_scan_kernel, which skips the cached ones
    CachedScanArgs cs;
    cs.jobs = nullptr; cs.n_jobs = nullptr; cs.jobs_cap = n_chains; cs.state = state; cs.heads = parts; cs.rows = part_rows;
    cs.cache = reinterpret_cast<const float *>(R.cache);
    cs.n_seg = R.cached_segments; cs.light = R.cached_light;
    if (cs.n_seg > kCachedSegMax) cs.n_seg = kCachedSegMax;
    if (!(cs.light > 0.01f && cs.light < 0.99f)) cs.light = 0.15f;
    s.skip_cached = cached_scan && cs.n_seg > 0 ? 1 : 0;
    const bool all_cached = (int64_t)R.slot_cap >= (int64_t)n_chains * kScanMaxRound;
    const int cached_grid = ((R.n_channels * kStrips * 3 + 7) / 8) * 8 * (cs.n_seg > 0 ? cs.n_seg : 1);
    LT_LAUNCH(lt_plan_kernel, dim3(1), dim3(256), 0, stream, p);
    for (int r = 0; r < rounds && p.enable; ++r) {
      s.jobs = q.jobs = cs.jobs = jobs2[r & 1];
      s.n_jobs = cs.n_jobs = R.n_jobs + 2 * r;
      q.plan_next = r + 1 < rounds;
      q.ask = (!q.plan_next && ask_early) ? 1 : 0;
      q.next = p;
      q.next.first_round = 0;
      q.next.round_windows = seg_len;
      q.next.jobs = jobs2[(r + 1) & 1];
      q.next.n_jobs = R.n_jobs + 2 * (r + 1);
      q.next.round_slots = R.n_jobs + kCtrLaunchSlots + (r + 1);
      if (cached_scan) { // the lag metrics of this round's windows, before its scan (the round's slots: from 0)
        mm.n_before = 0;
        mm.launch_slots = R.n_jobs + kCtrLaunchSlots + r;
        LT_LAUNCH(lt_metric_kernel, dim3((unsigned)metric_grid), dim3(64 * kMetricWaves), 0, stream, mm);
      }
      if (cached_scan && cs.n_seg > 0) LT_LAUNCH(lt_scan_cached_kernel, dim3((unsigned)cached_grid), dim3(64), 0, stream, cs);
      if (cached_scan && cs.n_seg > 0 && all_cached) { // (every job of a round has its slots: nothing is left for the transforms)
      } else if (G == 1) LT_LAUNCH(lt_scan_kernel<1>, dim3(grid), dim3(scan_threads(1)), 0, stream, s);
      else if (G == 2) LT_LAUNCH(lt_scan_kernel<2>, dim3(grid), dim3(scan_threads(2)), 0, stream, s);
      else LT_LAUNCH(lt_scan_kernel<3>, dim3(grid), dim3(scan_threads(3)), 0, stream, s);
      LT_LAUNCH(lt_resolve_kernel, dim3(R.n_channels), dim3(192), 0, stream, q);
      if (debug_jobs) { // diagnostics: the jobs of this round
        (void)hipStreamSynchronize(stream);
        int njj[2] = {0, 0};
        (void)hipMemcpy(njj, R.n_jobs + 2 * r, sizeof(njj), hipMemcpyDeviceToHost);
        std::vector<ScanJob> all((size_t)n_chains), hj;
        (void)hipMemcpy(all.data(), jobs2[r & 1], sizeof(ScanJob) * (size_t)n_chains, hipMemcpyDeviceToHost);
        for (int i = 0; i < njj[0]; ++i) hj.push_back(all[(size_t)i]);
        for (int i = 0; i < njj[1]; ++i) hj.push_back(all[(size_t)(n_chains - 1 - i)]);
        const int nj = njj[0] + njj[1];
        int hist[4][4] = {};
        long wsum[4] = {};
        for (const ScanJob &j : hj)
          for (int q = 0; q < j.n_seq; ++q) {
            const int nr = __builtin_popcount(j.seq[q].roots & 7);
            hist[nr][j.seq[q].stride_b > 1 ? 2 : (j.seq[q].n_a > 0 ? 1 : 0)]++;
            wsum[nr] += j.seq[q].n_a + j.seq[q].n_b + j.seq[q].n_c;
          }
        if (nj <= 4)
          for (const ScanJob &j : hj)
            for (int q = 0; q < j.n_seq; ++q)
              fprintf(stderr, "  round %d channel %d seq %d roots %d zero %d: A %lld x%d n%d | B %lld x%d n%d | C %lld x%d n%d\n", r, j.channel, q,
                      j.seq[q].roots, j.seq[q].zero, (long long)j.seq[q].pos_a, j.seq[q].stride_a, j.seq[q].n_a, (long long)j.seq[q].pos_b,
                      j.seq[q].stride_b, j.seq[q].n_b, (long long)j.seq[q].pos_c, j.seq[q].stride_c, j.seq[q].n_c);
        fprintf(stderr, "round %d: %d jobs;", r, nj);
        for (int nr = 1; nr <= 3; ++nr)
          fprintf(stderr, "  %d-root sequences: %d plain, %d with prefix, %d strided, %ld windows;", nr, hist[nr][0], hist[nr][1], hist[nr][2], wsum[nr]);
        fprintf(stderr, "\n");
      }
    }
  }
  if (debug_jobs) { // diagnostics: what the rounds leave to the per-chain kernel
    (void)hipStreamSynchronize(stream);
    std::vector<char> hs((size_t)n_chains * sizeof(ChainState));
    (void)hipMemcpy(hs.data(), state, hs.size(), hipMemcpyDeviceToHost);
    const int64_t limit = t1 - (R.flush ? kHalfFrame : kCallReach);
    int need = 0, done = 0, behind = 0;
    long windows_left = 0;
    for (int c = 0; c < n_chains; ++c) {
      const ChainState *st = reinterpret_cast<const ChainState *>(hs.data() + (size_t)c * sizeof(ChainState));
      need += st->need_chain != 0;
      done += st->run_done != 0;
      if (!st->run_done && st->pos <= limit) {
        behind++;
        windows_left += (long)((limit - st->pos) / kHalfFrame + 1);
      }
    }
    fprintf(stderr, "after the rounds: %d chains need the exact path, %d stopped, %d not through (%ld windows left)\n", need, done, behind, windows_left);
  }
  if (R.ev_scan_done) (void)hipEventRecord(R.ev_scan_done, stream);
  // ---- per-chain kernel
  {
    ChainArgs c;
    c.iq = iq; c.ch_stride = R.ch_stride; c.hist = hist; c.hist_stride = R.hist_stride; c.t0 = R.t0; c.n = R.n;
    c.hist_len = R.hist_len; c.flush = R.flush; c.n_channels = R.n_channels; c.state = state;
    c.recs = R.recs; c.rec_cap = R.rec_cap; c.rec_count = R.rec_count;
    c.tab = R.tab; c.psr_threshold = R.psr_threshold; c.track_after = R.track_after;
    c.track_every = R.track_every;
    c.list = work_list(R, n_chains);
    c.list_count = R.n_jobs + kCtrList + (R.list_slot ? 1 : 0);
    c.cache = reinterpret_cast<float *>(R.cache);
    c.slot_owner = reinterpret_cast<int2 *>(R.slot_owner);
    c.slot_pos = reinterpret_cast<int64_t *>(R.slot_pos);
    c.slot_next = R.n_jobs + kCtrSlots; // (cleared with the run's other counters by its first launch)
    c.launch_slots = R.n_jobs + kCtrLaunchSlots + kMaxRounds + 1; // (its own count: the rounds are over, their slots free)
    c.slot_cap = R.slot_cap;
    c.slot_quota = R.slot_quota;
    c.pass = 0;
    c.wait_hint = R.wait_hint;
    c.run_id = R.run_id;
    if (!(c.cache && c.slot_owner && c.slot_pos && c.slot_cap > 0 && c.slot_quota >= kCacheMinSearches && R.cache_level >= 1 && R.two_pass)) c.cache = nullptr;
    c.stream = R.stream && c.cache;
    if (!(ask_early && c.cache)) LT_LAUNCH(lt_chain_kernel, dim3(n_chains), dim3(kChainThreads), 0, stream, c); // pass 0 (or the only pass)
    if (c.cache) {
      // the lag metric of the windows pass 0 has asked for, chip-wide; then the chains that waited for it
      MetricArgs m;
      m.iq = iq; m.ch_stride = R.ch_stride; m.hist = hist; m.hist_stride = R.hist_stride; m.t0 = R.t0; m.n = R.n;
      m.hist_len = R.hist_len; m.slot_owner = c.slot_owner; m.slot_pos = c.slot_pos; m.launch_slots = c.launch_slots;
      m.n_before = 0; m.slot_cap = c.slot_cap; m.quota = c.slot_quota;
      m.cache = c.cache; m.tab = R.tab;
      const int n_cus = R.n_cus > 0 ? R.n_cus : 256;
      LT_LAUNCH(lt_metric_kernel, dim3((unsigned)(4 * n_cus)), dim3(64 * kMetricWaves), 0, stream, m);
      c.pass = 1;
      // (level 4: the chains with a stream first, in the kernel that has nothing but the stream to search in)
      if (c.stream) LT_LAUNCH(lt_chain_stream_kernel, dim3(n_chains), dim3(kChainThreads), 0, stream, c);
      LT_LAUNCH(lt_chain_kernel, dim3(n_chains), dim3(kChainThreads), 0, stream, c);
    }
  }
  if (R.ev_chain_done) (void)hipEventRecord(R.ev_chain_done, stream);
  // ---- per-record kernels: CP length and SSS of the tracked half-frames.  They feed nothing but the records, so they
  // may run on a stream of their own beside the next run's scan rounds (lt_batch_set_overlap).
  {
    hipStream_t rs = stream;
    if (records_stream) {
      if (!R.ev_chain_done) return hipErrorInvalidValue;
      hipError_t we = hipStreamWaitEvent(records_stream, R.ev_chain_done, 0);
      if (we != hipSuccess) return we;
      rs = records_stream;
    }
    CpArgs c;
    c.iq = iq; c.ch_stride = R.ch_stride; c.hist = hist; c.hist_stride = R.hist_stride; c.t0 = R.t0; c.n = R.n;
    c.hist_len = R.hist_len; c.recs = R.recs; c.rec_cap = R.rec_cap; c.rec_count = R.rec_count;
    c.scratch = reinterpret_cast<RecordScratch *>(R.cp_scratch); c.state = state; c.tab = R.tab; c.n_chains = n_chains;
    c.list = work_list(R, n_chains);
    c.list_count = R.n_jobs + kCtrList + (R.list_slot ? 1 : 0);
    if (R.records_delay_us > 0) LT_LAUNCH(lt_delay_kernel, dim3(1), dim3(64), 0, rs, (long long)R.records_delay_us * 100);
    LT_LAUNCH(lt_cpsss_kernel, dim3((unsigned)kListBlocks), dim3(256), 0, rs, c);
    LT_LAUNCH(lt_cp_scan_kernel, dim3((unsigned)n_chains), dim3(64), 0, rs, c);
    if (R.out_hf && R.out_cap > 0) {
      HfArgs h;
      h.iq = iq; h.ch_stride = R.ch_stride; h.hist = hist; h.hist_stride = R.hist_stride; h.t0 = R.t0; h.n = R.n;
      h.hist_len = R.hist_len; h.recs = R.recs; h.rec_cap = R.rec_cap; h.rec_count = R.rec_count;
      h.out_hf = reinterpret_cast<v2f *>(R.out_hf); h.out_cap = R.out_cap; h.tab = R.tab;
      LT_LAUNCH(lt_hf_kernel, dim3((unsigned)n_chains, (unsigned)R.rec_cap), dim3(256), 0, rs, h);
    }
  }
  // ---- history for the next chunk
  if (R.hist_new && R.hist_len_new > 0) {
    HistArgs h;
    h.iq = iq; h.ch_stride = R.ch_stride; h.hist_old = hist; h.hist_new = reinterpret_cast<v2f *>(R.hist_new);
    h.hist_stride = R.hist_stride; h.n = R.n; h.hist_len_old = R.hist_len; h.hist_len_new = R.hist_len_new;
    LT_LAUNCH(lt_hist_kernel, dim3((unsigned)((R.hist_len_new + 255) / 256), (unsigned)R.n_channels), dim3(256), 0, stream, h);
  }
#undef LT_LAUNCH
  if (R.n_launches) *R.n_launches = n_launch;
  return hipGetLastError();
}

// lt_batch_reset: the chains back to their constructor state.  The members only the per-record kernels
// touch (the CP averages; the CFO ring is the per-chain kernel's) are cleared separately, so that with lt_batch_set_overlap they can be cleared on the
// stream those kernels run on, behind the previous run's.  parts: 1 = everything else, 2 = those members, 3 = all.
__global__ __launch_bounds__(256) void lt_reset_kernel(ChainState *state, int parts, int *rec_count)
{
  if (rec_count && threadIdx.x == 0) rec_count[blockIdx.x] = 0;
  reset_chain_states(state + blockIdx.x, 1, parts, threadIdx.x, 256);
}
hipError_t launch_reset(void *state, int n_chains, int parts, int *rec_count, hipStream_t stream)
{
  hipLaunchKernelGGL(lt_reset_kernel, dim3((unsigned)n_chains), dim3(256), 0, stream, reinterpret_cast<ChainState *>(state), parts, rec_count);
  return hipGetLastError();
}

size_t chain_state_bytes() { return sizeof(ChainState); }
size_t chain_state_avg_offset() { return offsetof(ChainState, avg); }

hipError_t launch_find(const void *win, float *avg, const DeviceTables *tab, int n_id_2, void *res,
                       hipStream_t stream)
{
  hipLaunchKernelGGL(lt_find_kernel, dim3(1), dim3(kChainThreads), 0, stream,
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

// The taps in the order lt_decim_kernel<D, NTAPS> walks them: [unit q][step u < U][output b < B] = taps[j],
// B = phases / D, U = 32 / B, j = q U + u + (b - (B - 1)) D (0 where the pair has no tap: never used).
std::vector<float> decim_tap_table(int D, const float *taps, int ntaps)
{
  const int B = kDecimPhases / D, U = 32 / B, n_steps = ntaps + (B - 1) * D, Q = (n_steps + U - 1) / U;
  std::vector<float> t((size_t)Q * 32, 0.f);
  for (int q = 0; q < Q; ++q)
    for (int u = 0; u < U; ++u)
      for (int b = 0; b < B; ++b) {
        const int j = q * U + u + (b - (B - 1)) * D;
        if (j >= 0 && j < ntaps) t[(size_t)q * 32 + u * B + b] = taps[j];
      }
  return t;
}
bool decim_has_tap_table(int D, int ntaps) { return (D == 4 && ntaps == 131) || (D == 8 && ntaps == 263) || (D == 16 && ntaps == 525); }

hipError_t launch_decim(const void *x, void *y, int n_channels, int64_t n_in, int D, const float *taps,
                        int ntaps, const float *tap_table, hipStream_t stream)
{
  const int64_t n_out = n_in / D;
  if (n_out <= 0) return hipSuccess;
  const v2f *xi = reinterpret_cast<const v2f *>(x);
  v2f *yo = reinterpret_cast<v2f *>(y);
  const int B = (D == 4 || D == 8 || D == 16) ? kDecimPhases / D : 0;
  if (B && tap_table && n_channels <= 65535) {
    const int64_t per_block = (int64_t)kDecimThreads * B;
    const dim3 grid((unsigned)((n_out + per_block - 1) / per_block), (unsigned)n_channels);
    if (D == 4 && ntaps == 131) {
      hipLaunchKernelGGL((lt_decim_kernel<4, 131>), grid, dim3(kDecimThreads), 0, stream, xi, yo, n_in, n_out, tap_table);
      return hipGetLastError();
    }
    if (D == 8 && ntaps == 263) {
      hipLaunchKernelGGL((lt_decim_kernel<8, 263>), grid, dim3(kDecimThreads), 0, stream, xi, yo, n_in, n_out, tap_table);
      return hipGetLastError();
    }
    if (D == 16 && ntaps == 525) {
      hipLaunchKernelGGL((lt_decim_kernel<16, 525>), grid, dim3(kDecimThreads), 0, stream, xi, yo, n_in, n_out, tap_table);
      return hipGetLastError();
    }
  }
  if (n_channels > 65535) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((n_out + kDecimGeneric - 1) / kDecimGeneric), (unsigned)n_channels);
  hipLaunchKernelGGL(lt_decim_generic_kernel, grid, dim3(kDecimGeneric), 0, stream, xi, yo, n_in, n_out, D, taps, ntaps);
  return hipGetLastError();
}

} // namespace lt
