// Strip-wise form of srslte_pss_find_pss (call site lib/pss_impl.cc:165-167) for gfx950.
//
// One search window = 9600 samples between zero pads = 9726 lags of the 128-tap matched filter.  The lags
// are cut into 12 strips of 811; strip s is one 1024-point overlap-save block whose 896 usable outputs are
// the lags c0 .. c0 + 895, c0 = 811 s - 41 (41 lags of halo on the left, 44 on the right), so the window's
// head and tail lags -- whose convolution sums run into the zero pads -- need no special case: a block's
// samples outside [0, 9600) of the window are zero, exactly srsLTE's zero-padded input.
//
//   block element q = lane + 64 R (R = 0..15)  <->  window sample 811 s - 169 + q
//   lag c0 + lane + 64 r (r = 0..13)            <->  output element q with R = r + 2
//
// A wave owns a strip: its 14 registers per lane hold the exponential average (alpha = 0.2) of those 896
// lags across windows, so the average never leaves the register file between windows.  The strip reduces
// itself to a StripPartial (its maximum over the lags it owns, the main lobe around it walked inside a
// 64-lag neighbourhood, the maximum over its lags outside that lobe) and resolve_window() combines the 12
// partials of a window into peak position, peak value and peak-to-side-lobe ratio.  tests/strip_model.py is
// the numpy twin of strip_resolve() + resolve_window(); tests/test_strip_model.py checks the decomposition
// against oracle/lte_oracle.c:295-316.
#pragma once
#include "lt_device.h"

#ifndef LT_LOAD_AUX
#define LT_LOAD_AUX 0 // cache policy of the sample loads (2 = nt)
#endif

namespace lt {

constexpr int kStrips = 12;
constexpr int kStripPitch = 811;   // lags owned by a strip (12 * 811 = 9732 >= 9726)
constexpr int kStripHalo = 41;     // computed lags in front of the owned ones
constexpr int kStripRegs = 14;     // computed lags per lane
constexpr int kStripIn0 = -(kStripHalo + kSymbolSz); // window sample of block element 0, relative to 811 s
static_assert(kStrips * kStripPitch >= kNLags, "strips must cover every lag");
static_assert(kStripHalo >= 32 && 64 * kStripRegs - kStripHalo - kStripPitch >= 32, "halo must hold the lobe walk");

// A strip's exponential average: 14 live registers per lane, held as one 16-register tuple so that "the register
// the peak sits in" -- a wave-uniform but run-time index -- is a single indexed register move, not a chain of selects.
typedef float ema_t __attribute__((ext_vector_type(16)));

constexpr uint32_t kPartWide = 1u;  // lobe not closed inside the strip's 64-lag neighbourhood: exact path needed

// What a strip tells the resolver about one window (32 bytes, one per (chain, window, strip)).
struct StripPartial {
  float m1;       // maximum of the average over the lags the strip owns
  int32_t p1;     // its lag (first one in lag order)
  float s;        // maximum over the owned lags outside [lb, ub)
  int32_t lb;     // main lobe around p1 as srslte_pss_find_pss walks it (window-global stops folded in)
  int32_t ub;
  float a0;       // strip 0: average of lag 0 (srslte_vec_max_fi of an empty range returns index 0)
  uint32_t flags; // kPartWide
  uint32_t pad;
};
static_assert(sizeof(StripPartial) == 32, "StripPartial layout");

__device__ __forceinline__ uint64_t rotr64(uint64_t v, int n) { n &= 63; return n ? (v >> n) | (v << (64 - n)) : v; }
__device__ __forceinline__ uint64_t rotl64(uint64_t v, int n) { n &= 63; return n ? (v << n) | (v >> (64 - n)) : v; }

// value of lane (l + 1) % 64 / (l - 1) % 64: whole-wave DPP rotates
__device__ __forceinline__ float wave_next(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x134, 0xf, 0xf, false)); // wave_rol:1
}
__device__ __forceinline__ float wave_prev(float v)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x13C, 0xf, 0xf, false)); // wave_ror:1
}

// maximum over the wave, in every lane's hands as a wave-uniform value: four rotations inside the rows of 16, two
// row broadcasts, one readlane -- the DPP permutation rides on the v_max itself (six vector instructions; the
// generic wave_max_f of lt_device.h takes twenty-four)
__device__ __forceinline__ float wave_max_dpp(float f)
{
  // (a DPP operand written by the instruction before needs two wait states, and the compiler does not look
  // inside the block: the leading s_nop covers the instruction that produced f)
  asm("s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
      : "+v"(f));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f), 63));
}

// minimum of a non-negative key over the wave (same shape as wave_max_dpp)
__device__ __forceinline__ int wave_min_key_dpp(int k)
{
  asm("s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
      : "+v"(k));
  return __builtin_amdgcn_readlane(k, 63);
}

// e[r] = average of lag c0 + lane + 64 r (r < 14; e[14], e[15] are spare: the tuple has a power-of-two size).  Wave-uniform result in `out` (every
// lane holds it).  The tuple is only read -- two of its registers through a run-time, wave-uniform index (the rows the
// peak and its neighbourhood sit in) -- so the average stays where the update left it.
//
// One pass over the 14 registers gives every lane its maximum m, where it sits (a: the first register) and its second
// largest value m2 (v_med3: the median of the old maximum, the new value and the old second is the new second).  The
// strip's maximum is the wave maximum of m.  The maximum outside the main lobe needs no second pass: the 64 lags
// around the peak are one value per lane (`row`); outside that window a lane contributes m -- or m2 where its
// maximum IS its window value --, inside it the window value unless the lobe covers it.
// ... in two steps, so that a caller that sees all twelve strips of a window (the per-chain kernel) can stop after the
// first for every strip but the one that holds the window's peak: strip_peak = the strip's maximum and where it sits,
// strip_lobe = the main lobe around it and the maximum outside.  strip_resolve = both.
struct StripPeak {
  float m, m2;    // per lane: maximum over its owned lags, second largest
  int a;          // per lane: the register the maximum sits in
  float M1;       // the strip's maximum (wave-uniform, like everything below)
  int l_p, r_p;   // lane and register of its first occurrence
  int p1;         // its lag
  uint32_t flags; // kPartWide: no maximum found (a NaN)
};
__device__ __forceinline__ void strip_peak(const ema_t &e, int s, int lane, StripPeak &k)
{
  const int c0 = kStripPitch * s - kStripHalo;
  // owned and valid lags: offsets [41, 852) of the computed range, lag < 9726 (strip 11 ends at offset 845):
  // the halo lags in registers 0 and 13 sit out the two maxima
  const int own13_end = (s == kStrips - 1) ? (kNLags - (kStripPitch * (kStrips - 1) - kStripHalo) - 64 * 13)
                                           : (kStripHalo + kStripPitch - 64 * 13);
  const bool own0 = lane >= kStripHalo, own13 = lane < own13_end;
  // 1. per lane: maximum, its register (the first), second largest
  float m = own0 ? e[0] : -FLT_MAX, m2 = -FLT_MAX;
  int a = 0;
#pragma unroll
  for (int r = 1; r < kStripRegs; ++r) {
    const float v = (r == kStripRegs - 1) ? (own13 ? e[r] : -FLT_MAX) : e[r];
    // a = v > m ? r : a (strictly greater: the first register keeps a tie); m2 = med3(m, v, m2); m = max(m, v).
    // (As assembly: written with fmaxf / fmed3f every value first goes through a canonicalising v_max x, x.)
    asm("v_cmp_gt_f32 vcc, %[v], %[m]\n\t"
        "v_cndmask_b32_e64 %[a], %[a], %[r], vcc\n\t"
        "v_med3_f32 %[m2], %[m], %[v], %[m2]\n\t"
        "v_max_f32 %[m], %[m], %[v]"
        : [a] "+v"(a), [m2] "+v"(m2), [m] "+v"(m)
        : [v] "v"(v), [r] "n"(r)
        : "vcc");
  }
  const float M1 = wave_max_dpp(m);
  // where it sits, first in lag order: smallest register, then lane
  const uint64_t hit = __ballot(m == M1);
  uint32_t flags = 0;
  int l_p, r_p;
  if (__builtin_popcountll(hit) == 1) {
    l_p = __builtin_ctzll(hit);
    r_p = __builtin_amdgcn_readlane(a, l_p);
  } else if (hit) { // (an exact tie between lanes: degenerate inputs)
    const int key = wave_min_key_dpp(m == M1 ? a * 64 + lane : 0x7FFFFFFF);
    l_p = key & 63;
    r_p = key >> 6;
  } else { // only a NaN can do that: outside the contract, but stay in bounds
    l_p = kStripHalo;
    r_p = 0;
    flags = kPartWide;
  }
  k.m = m; k.m2 = m2; k.a = a; k.M1 = M1; k.l_p = l_p; k.r_p = r_p; k.p1 = c0 + l_p + 64 * r_p; k.flags = flags;
}
__device__ __forceinline__ void strip_lobe(const ema_t &e, int s, int lane, const StripPeak &k, StripPartial &out)
{
  const int own13_end = (s == kStrips - 1) ? (kNLags - (kStripPitch * (kStrips - 1) - kStripHalo) - 64 * 13)
                                           : (kStripHalo + kStripPitch - 64 * 13);
  const uint64_t halo0 = (1ull << kStripHalo) - 1ull, halo13 = ~((1ull << own13_end) - 1ull); // lanes of register 0 / 13 that sit out
  const float m = k.m, m2 = k.m2, M1 = k.M1;
  const int l_p = k.l_p, r_p = k.r_p, p1 = k.p1;
  uint32_t flags = k.flags;
  // 2. the lags p1 - 32 .. p1 + 31 as one rotated row: lane L holds offset (L - l_p) wrapped to [-32, 32).
  // Offsets that fall off this register's 64 lanes come from the register before (l_p < 32) or after it
  // (which exists whenever a lane needs it: the peak is an owned lag).
  const int r_b = (l_p < 32) ? (r_p > 0 ? r_p - 1 : 1) : (r_p < kStripRegs - 1 ? r_p + 1 : kStripRegs - 2);
  const float rowA = e[r_p], rowB = e[r_b]; // (run-time register index, wave-uniform)
  const int d = lane - l_p;
  const int off = ((d + 32) & 63) - 32;
  const bool use_b = d != off; // the offset wrapped: the lag sits in the neighbouring register
  const float row = use_b ? rowB : rowA;
  // the same with the halo lanes out (the window's share of the second maximum): which lanes of the row sit out is a
  // wave-uniform mask, handed to v_cndmask as the scalar pair it is (written as a test of bit `lane` it becomes a 64-bit
  // vector shift)
  const uint64_t outA = r_p == 0 ? halo0 : r_p == kStripRegs - 1 ? halo13 : 0ull;
  const uint64_t outB = r_b == 0 ? halo0 : r_b == kStripRegs - 1 ? halo13 : 0ull;
  const uint64_t wrapped = __ballot(use_b);
  const uint64_t sits_out = (wrapped & outB) | (~wrapped & outA);
  float row_own;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(row_own) : "v"(row), "v"(-FLT_MAX), "s"(sits_out));
  const float nxt = wave_next(row), prv = wave_prev(row);
  const int lag_row = p1 + off;
  // main-lobe walk (oracle/lte_oracle.c:300-308): right while avg[u+1] <= avg[u] and u < 9727, left while
  // avg[l-1] <= avg[l] and l > 1; no left walk for a peak at lag <= 2
  const uint64_t stop_r = rotr64(__ballot(!(nxt <= row) || lag_row >= kConvLen - 1), l_p); // bit j: offset +j
  const uint64_t stop_l = rotr64(__ballot(!(prv <= row) || lag_row <= 1), l_p);            // bit 64 - j: offset -j
  const uint64_t mr = stop_r & 0x7FFFFFFEull;         // offsets 1 .. 30
  const uint64_t ml = stop_l & 0xFFFFFFFE00000000ull; // offsets -31 .. -1
  int j_r = 1, j_l = 1;
  if (mr) j_r = __builtin_ctzll(mr);
  else flags |= kPartWide;
  if (p1 <= 2) j_l = p1; // pl_lb = 0
  else if (ml) j_l = 1 + __builtin_clzll(ml);
  else flags |= kPartWide;
  const int lb = p1 - j_l, ub = p1 + j_r;
  // 3. maximum over the owned lags outside [lb, ub): the lobe is the window's offsets [-j_l, j_r)
  const bool in_lobe = (unsigned)(off + j_l) < (unsigned)(j_l + j_r);
  const float far = (row_own == m) ? m2 : m; // (a tie: m2 == m, either is right)
  const float near = in_lobe ? -FLT_MAX : row_own;
  float both; // (as assembly: fmaxf would canonicalise both operands first)
  asm("v_max_f32 %0, %1, %2" : "=v"(both) : "v"(far), "v"(near));
  const float S = wave_max_dpp(both);
  out.m1 = M1;
  out.p1 = p1;
  out.s = S;
  out.lb = lb;
  out.ub = ub;
  out.a0 = (s == 0) ? rlane(e[0], kStripHalo) : 0.f;
  out.flags = flags;
  out.pad = 0;
}

__device__ __forceinline__ void strip_resolve(const ema_t &e, int s, int lane, StripPartial &out)
{
  StripPeak k;
  strip_peak(e, s, lane, k);
  strip_lobe(e, s, lane, k, out);
}

// ---- the deferred form (scan rounds, round 5).  Eleven of a window's twelve lobe walks are thrown away: only the strip that
// holds the window's peak needs its lobe.  A root wave of the scan therefore stops after the strip's maximum: it writes
//   StripHead  m1, p1 -- the maximum over the owned lags and its lag --, far = the maximum over the owned lags OUTSIDE the 32
//              lags around p1, a0 (strip 0: the average of lag 0), 16 bytes;
//   a row      the 32 lags p1 - 16 .. p1 + 15 themselves (halo lags included: the walk may cross the strip's boundary), 128 bytes
//              (a lobe that does not close within 14 lags to the right or 15 to the left -- the strips' own walk allowed 30 / 31 --
//              sends the window to the exact path; lobes of the lag metric are 2-6 lags wide.  64-lag rows were measured first:
//              twice the bytes written, HBM traffic 1.37 x algorithmic instead of 1.27 x, and a slower resolver)
// and the resolver (one lane per window) walks the lobe of the ONE strip that holds the window's maximum, in the oracle's own
// terms (oracle/lte_oracle.c:300-316): side-lobe maximum = the largest value outside [lb, ub) = max over the peak strip's far
// and the owned lags of its row outside the lobe, and over the other strips' maxima -- or, for a neighbour whose maximum lies
// on the slope of this lobe, over ITS far and the owned lags of ITS row outside the lobe (every lag the lobe covers in the
// neighbour lies within 30 lags of the neighbour's maximum: both sit between the strip boundary and the lobe's end).
struct StripHead {
  float m1;
  int32_t p1f;   // p1 | kHeadNoMax
  float far;
  float a0;
};
static_assert(sizeof(StripHead) == 16, "StripHead layout");
constexpr int32_t kHeadNoMax = 1 << 30; // no maximum found (a NaN in the average): the window goes to the exact path
constexpr int kRowLags = 32, kRowHalf = kRowLags / 2;

// -> the strip's head (wave-uniform), this lane's value of the row and its place in it (index i <-> lag p1 - 16 + i; -1: none)
__device__ __forceinline__ void strip_head_row(const ema_t &e, int s, int lane, StripHead &h, float &row, int &idx)
{
  StripPeak k;
  strip_peak(e, s, lane, k);
  const int own13_end = (s == kStrips - 1) ? (kNLags - (kStripPitch * (kStrips - 1) - kStripHalo) - 64 * 13)
                                           : (kStripHalo + kStripPitch - 64 * 13);
  const uint64_t halo0 = (1ull << kStripHalo) - 1ull, halo13 = ~((1ull << own13_end) - 1ull);
  const int l_p = k.l_p, r_p = k.r_p;
  const int r_b = (l_p < 32) ? (r_p > 0 ? r_p - 1 : 1) : (r_p < kStripRegs - 1 ? r_p + 1 : kStripRegs - 2);
  const float rowA = e[r_p], rowB = e[r_b]; // (run-time register index, wave-uniform)
  const int d = lane - l_p;
  const int off = ((d + 32) & 63) - 32;
  const bool use_b = d != off;
  row = use_b ? rowB : rowA;
  const uint64_t outA = r_p == 0 ? halo0 : r_p == kStripRegs - 1 ? halo13 : 0ull;
  const uint64_t outB = r_b == 0 ? halo0 : r_b == kStripRegs - 1 ? halo13 : 0ull;
  const uint64_t wrapped = __ballot(use_b);
  const uint64_t sits_out = (wrapped & outB) | (~wrapped & outA);
  float row_own;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(row_own) : "v"(row), "v"(-FLT_MAX), "s"(sits_out));
  // outside the row a lane contributes its maximum -- or its second largest value where its maximum IS its row value
  // (the row: the lanes whose offset lies in [-16, 16); the other lanes' values of these two registers are not in it)
  const bool in_row = (unsigned)(off + kRowHalf) < (unsigned)kRowLags;
  const float far_lane = (in_row && row_own == k.m) ? k.m2 : k.m;
  h.m1 = k.M1;
  h.p1f = k.p1 | (k.flags ? kHeadNoMax : 0);
  h.far = wave_max_dpp(far_lane);
  h.a0 = (s == 0) ? rlane(e[0], kStripHalo) : 0.f;
  idx = in_row ? off + kRowHalf : -1; // (-1: not part of the row)
}

struct WindowResult {
  int peak_pos;
  float peak_value;
  float psr;
  bool exact; // false: a lobe did not close inside its strip's neighbourhood -- the window needs the exact path
};

// The 12 partials of one window -> srslte_pss_find_pss's results.  Plain scalar code (one lane per window in
// the resolver kernel, wave-uniform in the chain kernel).  `parts` may be global or LDS memory.
template <typename Ptr> __device__ __forceinline__ WindowResult resolve_window(Ptr parts)
{
  int best = 0;
  float bm = parts[0].m1;
#pragma unroll 1
  for (int s = 1; s < kStrips; ++s) { // strips ascend in lag: strict > keeps the first maximum
    const float v = parts[s].m1;
    if (v > bm) {
      bm = v;
      best = s;
    }
  }
  WindowResult w;
  const uint32_t bflags = parts[best].flags;
  w.exact = !(bflags & kPartWide);
  w.peak_value = bm;
  w.peak_pos = (bm > -FLT_MAX) ? parts[best].p1 : 0; // srslte_vec_max_fi returns 0 when nothing beats -FLT_MAX
  const int lb = parts[best].lb, ub = parts[best].ub;
  float out = parts[best].s;
#pragma unroll 1
  for (int s = 0; s < kStrips; ++s) {
    if (s == best) continue;
    const int p = parts[s].p1;
    const bool inside = p >= lb && p < ub; // that strip's maximum lies on the slope of this lobe
    if (inside && (parts[s].flags & kPartWide)) w.exact = false;
    const float v = inside ? parts[s].s : parts[s].m1;
    if (v > out) out = v;
  }
  // empty ranges stand for their first element: the left one is empty exactly when the peak sits at lag <= 2,
  // the right one when the lobe reaches lag 9726 (avg[9726..9727] = 0)
  const float sl = (w.peak_pos <= 2) ? parts[0].a0 : out;
  const float sr = (ub >= kNLags) ? 0.f : out;
  const float side = sl > sr ? sl : sr;
  w.psr = w.peak_value / side;
  return w;
}

// The same by twelve lanes of a wave, one partial each (the per-chain kernels, every wave for itself): one LDS round trip, the
// two maxima by the wave reductions the strips use, the peak strip's fields by lane reads -- instead of thirty-odd
// dependent LDS reads and two scalar loops per wave (rounds 2 to 4's resolve_window_unrolled).  Same decisions, same values: the first maximum in strip order is the
// lowest lane that holds the wave's maximum; the side-lobe maximum is a maximum, in whatever order.
__device__ __forceinline__ WindowResult resolve_window_lanes(const StripPartial *parts, int lane)
{
  const bool on = lane < kStrips;
  const StripPartial q = parts[on ? lane : 0];
  const float m1 = on ? q.m1 : -FLT_MAX;
  const float bm = wave_max_dpp(m1);
  const uint64_t hit = __ballot(on && m1 == bm);
  const int best = hit ? __builtin_ctzll(hit) : 0; // (no lane equal: only a NaN can do that -- stay in bounds)
  WindowResult w;
  const uint32_t bflags = (uint32_t)__builtin_amdgcn_readlane((int)q.flags, best);
  const int p1b = __builtin_amdgcn_readlane(q.p1, best);
  const int lb = __builtin_amdgcn_readlane(q.lb, best), ub = __builtin_amdgcn_readlane(q.ub, best);
  const float sb = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(q.s), best));
  const float bmv = hit ? bm : __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(q.m1), 0));
  w.exact = !(bflags & kPartWide);
  w.peak_value = bmv;
  w.peak_pos = (bmv > -FLT_MAX) ? p1b : 0;
  const bool other = on && lane != best;
  const bool inside = q.p1 >= lb && q.p1 < ub; // that strip's maximum lies on the slope of this lobe
  if (__ballot(other && inside && (q.flags & kPartWide))) w.exact = false;
  const float v = other ? (inside ? q.s : q.m1) : -FLT_MAX;
  const float ov = wave_max_dpp(v);
  const float out = ov > sb ? ov : sb;
  const float a0 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(q.a0), 0));
  const float sl = (w.peak_pos <= 2) ? a0 : out;
  const float sr = (ub >= kNLags) ? 0.f : out;
  const float side = sl > sr ? sl : sr;
  w.psr = w.peak_value / side;
  return w;
}

// The deferred form's resolver, one lane per window: heads = the window's 12 StripHeads, rows = its 12 rows (global memory).
__device__ __forceinline__ bool lag_owned(int s, int lag) { return lag >= kStripPitch * s && lag < kStripPitch * (s + 1) && lag < kNLags; }
// maximum over strip s's far and the owned lags of its row outside [lb, ub)  (the rare path: a neighbour on the lobe's slope)
__device__ __forceinline__ float strip_outside_lobe(const StripHead *__restrict__ heads, const float *__restrict__ rows, int s, int lb, int ub)
{
  const StripHead h = heads[s];
  const int p = h.p1f & ~kHeadNoMax;
  float v = h.far;
#pragma unroll 1
  for (int i = 0; i < kRowLags; ++i) {
    const int lag = p - kRowHalf + i;
    const float x = rows[s * kRowLags + i];
    if (lag_owned(s, lag) && (lag < lb || lag >= ub) && x > v) v = x;
  }
  return v;
}
__device__ __forceinline__ WindowResult resolve_window_deferred(const StripHead *__restrict__ heads, const float *__restrict__ rows)
{
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f h[kStrips];
#pragma unroll
  for (int s = 0; s < kStrips; ++s) h[s] = *reinterpret_cast<const v4f *>(heads + s);
  int best = 0;
  float bm = h[0].x;
#pragma unroll
  for (int s = 1; s < kStrips; ++s) { // strips ascend in lag: strict > keeps the first maximum
    const float v = h[s].x;
    if (v > bm) {
      bm = v;
      best = s;
    }
  }
  // the peak strip's row, 32 values in registers (eight 16-byte loads in flight together)
  float r[kRowLags];
  {
    const v4f *rp = reinterpret_cast<const v4f *>(rows + best * kRowLags);
#pragma unroll
    for (int q = 0; q < kRowLags / 4; ++q) {
      const v4f t = rp[q];
      r[4 * q] = t.x; r[4 * q + 1] = t.y; r[4 * q + 2] = t.z; r[4 * q + 3] = t.w;
    }
  }
  // the peak strip's head out of the registers (a second trip to memory for heads[best] costs the wave ~2 us of latency)
  v4f hb = h[0];
#pragma unroll
  for (int s = 1; s < kStrips; ++s) hb = best == s ? h[s] : hb;
  const int p1f = __float_as_int(hb.y);
  const int p1 = p1f & ~kHeadNoMax;
  WindowResult w;
  w.exact = !(p1f & kHeadNoMax);
  w.peak_value = bm;
  w.peak_pos = (bm > -FLT_MAX) ? p1 : 0; // srslte_vec_max_fi returns 0 when nothing beats -FLT_MAX
  // main-lobe walk (oracle/lte_oracle.c:300-308): right while avg[u+1] <= avg[u] and u < 9727, left while avg[l-1] <= avg[l]
  // and l > 1; no left walk for a peak at lag <= 2.  Row index 16 is the peak; offsets beyond +14 / -15: not closed here.
  constexpr int C = kRowHalf;
  uint32_t mr = 0u, ml = 0u;
#pragma unroll
  for (int j = 1; j <= C - 2; ++j)
    if (!(r[C + j + 1] <= r[C + j]) || p1 + j >= kConvLen - 1) mr |= 1u << j;
#pragma unroll
  for (int j = 1; j <= C - 1; ++j)
    if (!(r[C - j - 1] <= r[C - j]) || p1 - j <= 1) ml |= 1u << j;
  int j_r = 1, j_l = 1;
  if (mr) j_r = __builtin_ctz(mr);
  else w.exact = false;
  if (p1 <= 2) j_l = p1; // pl_lb = 0
  else if (ml) j_l = __builtin_ctz(ml);
  else w.exact = false;
  const int lb = p1 - j_l, ub = p1 + j_r;
  // the largest value outside [lb, ub): the peak strip's own share ...
  float out = hb.z;
#pragma unroll
  for (int i = 0; i < kRowLags; ++i) {
    const int lag = p1 - kRowHalf + i;
    const bool counts = lag_owned(best, lag) && (lag < lb || lag >= ub);
    const float x = counts ? r[i] : -FLT_MAX;
    out = x > out ? x : out;
  }
  // ... and the other strips': their maximum, unless it lies on the slope of this lobe
#pragma unroll
  for (int s = 0; s < kStrips; ++s) {
    if (s == best) continue;
    const int p = __float_as_int(h[s].y) & ~kHeadNoMax;
    float v = h[s].x;
    if (p >= lb && p < ub) v = strip_outside_lobe(heads, rows, s, lb, ub);
    if (v > out) out = v;
  }
  // empty ranges stand for their first element: the left one is empty exactly when the peak sits at lag <= 2,
  // the right one when the lobe reaches lag 9726 (avg[9726..9727] = 0)
  const float sl = (w.peak_pos <= 2) ? h[0].w : out;
  const float sr = (ub >= kNLags) ? 0.f : out;
  const float side = sl > sr ? sl : sr;
  w.psr = w.peak_value / side;
  return w;
}

// ---- the strip's matched filter

// window-relative index of block element q of strip s
__device__ __forceinline__ int strip_sample(int s, int q) { return kStripPitch * s + kStripIn0 + q; }

// A capture as the kernels see it: the chunk handed to this run plus what the previous chunks left behind.
//   absolute sample a >= t0          -> iq[a - t0]            (a < t0 + n)
//   t0 - hist_len <= a < t0          -> hist[a - t0 + hist_len]
//   a < t0 - hist_len (or a < 0)     -> 0   ("zero history at stream start", set_history lib/pss_impl.cc:81)
struct Capture {
  const v2f *iq;    // this channel's chunk
  const v2f *hist;  // this channel's history (may be null when hist_len == 0)
  int64_t t0;       // absolute index of iq[0]
  int64_t n;        // samples in the chunk
  int64_t hist_len; // samples of history kept in front of t0
};
// Branch-free (the loads of a block must all be in flight together): out-of-range samples read the chunk's
// first sample and are zeroed afterwards.
__device__ __forceinline__ v2f ld_abs(const Capture &c, int64_t a)
{
  const int64_t i = a - c.t0, h = i + c.hist_len;
  const bool in_chunk = i >= 0 && i < c.n, in_hist = i < 0 && h >= 0;
  const v2f *p = in_hist ? c.hist + h : c.iq + (in_chunk ? i : 0);
  const v2f v = *p;
  v2f z = {0.f, 0.f};
  return (in_chunk || in_hist) ? v : z;
}

// the 1024 samples of strip s of the window starting at absolute position w0 (zero outside the window)
__device__ __forceinline__ void strip_load(const Capture &c, int64_t w0, int s, int lane, fft::c32 (&x)[16])
{
  const int i0 = strip_sample(s, lane);
  if (w0 >= c.t0 && w0 + kHalfFrame <= c.t0 + c.n) { // window inside the chunk (always, but for the first ones of a run)
    // The window as a buffer resource of 9600 samples: the block's samples in front of the window (negative
    // offsets, huge as unsigned) and behind it are out of range, and an out-of-range buffer load returns zero --
    // the window's zero pads without a compare, and one 32-bit offset per load instead of a 64-bit address.
    // (the window's address is wave-uniform, but the compiler cannot always prove it -- a loop counter it keeps in a
    // vector register is enough -- and a resource it takes for divergent is built lane by lane in a waterfall loop
    // around EVERY load: 14 instructions instead of one.  v_readfirstlane states the fact.)
    const uint64_t wa = (uint64_t)(c.iq + (w0 - c.t0));
    const uint32_t wa_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wa), wa_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(wa >> 32));
    const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)wa_hi << 32) | wa_lo), 0, kHalfFrame * (int)sizeof(v2f), 0x00020000);
    const int off = i0 * (int)sizeof(v2f);
#pragma unroll
    for (int R = 0; R < 16; ++R) {
      typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(win, off + 64 * R * (int)sizeof(v2f), 0, LT_LOAD_AUX);
      x[R] = __builtin_bit_cast(v2f, v);
    }
  } else {
    // window across the start of the chunk (the first windows of a lt_batch_push), or across its end: window
    // sample i lies in the chunk for i in [lo_c, hi_c), in the history for i in [lo_h, lo_c), nowhere else.
    // Wave-uniform 64-bit bases, 32-bit lane offsets.
    const int64_t d = c.t0 - w0; // window index of the chunk's first sample
    const int lo_c = (int)(d < -kHalfFrame ? -kHalfFrame : d > 2 * kHalfFrame ? 2 * kHalfFrame : d);
    const int64_t dh = d + c.n, dl = d - c.hist_len;
    const int hi_c = (int)(dh < -kHalfFrame ? -kHalfFrame : dh > 2 * kHalfFrame ? 2 * kHalfFrame : dh);
    const int lo_h = (int)(dl < -kHalfFrame ? -kHalfFrame : dl > 2 * kHalfFrame ? 2 * kHalfFrame : dl);
    const v2f *__restrict__ pc = c.iq - d;                                  // pc[i] = window sample i, from the chunk
    const v2f *__restrict__ ph = c.hist ? c.hist + (c.hist_len - d) : c.iq; // ph[i] = window sample i, from the history
#pragma unroll
    for (int R = 0; R < 16; ++R) {
      const int i = i0 + 64 * R;
      const bool in_win = i >= 0 && i < kHalfFrame;
      const bool in_c = in_win && i >= lo_c && i < hi_c, in_h = in_win && i >= lo_h && i < lo_c;
      const v2f *p = in_h ? ph + i : pc + (in_c ? i : (int)d); // (neither: the chunk's first sample, dropped below)
      const v2f v = *p;
      v2f z = {0.f, 0.f};
      x[R] = (in_c || in_h) ? v : z;
    }
  }
}

// in-place 1024-point transform of the wave's block (element lane + 64 R in x[R]); e = the wave's exchange area
template <bool kInv> __device__ __forceinline__ void strip_fft(int lane, fft::c32 (&x)[16], const fft::c32 *tw1,
                                                                const fft::c32 *tw2, fft::c32 *e)
{
  fft::fft1024w_stageA<kInv>(lane, x, tw1, e);
  __builtin_amdgcn_sched_barrier(0);
  fft::fft1024w_stageB<kInv>(lane, x, tw2, e, e);
  __builtin_amdgcn_sched_barrier(0);
  fft::fft1024w_stageC<kInv>(lane, x, e);
  __builtin_amdgcn_sched_barrier(0);
}

// |.|^2 of the strip's lags: met[r] = metric of lag c0 + lane + 64 r
__device__ __forceinline__ void strip_metric(const fft::c32 (&v)[16], float (&met)[kStripRegs])
{
#pragma unroll
  for (int r = 0; r < kStripRegs; ++r) {
    const fft::c32 y = v[r + 2];
    const fft::c32 sq = y * y;
    float m; // (as assembly: left to itself the compiler shuffles two lags' squares into pairs to add them packed)
    asm("v_add_f32 %0, %1, %2" : "=v"(m) : "v"(sq.x), "v"(sq.y));
    met[r] = m;
  }
}
// ... and the exponential average with its three roundings (srsLTE's three vector passes, oracle/lte_oracle.c:289-293).
// Lags outside [0, 9726) stay 0.
__device__ __forceinline__ void strip_ema_apply(int s, int lane, const float (&met)[kStripRegs], ema_t &ema)
{
  const bool valid0 = !(s == 0 && lane < kStripHalo);
  const int end13 = kNLags - (kStripPitch * (kStrips - 1) - kStripHalo) - 64 * 13; // strip 11: lanes below hold lags < 9726
  const bool valid13 = !(s == kStrips - 1 && lane >= end13);
  // (pair by pair, in the tuple's own registers: two packed products and a packed sum per two lags)
  typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int r = 0; r < kStripRegs; r += 2) {
    const f2 m2 = {met[r], met[r + 1]};
    const f2 old = {ema[r], ema[r + 1]};
    const f2 ne = m2 * 0.2f + old * (1.0f - 0.2f); // (no contraction: the build has -ffp-contract=off)
    ema[r] = ne.x;
    ema[r + 1] = ne.y;
  }
  ema[0] = valid0 ? ema[0] : 0.f;
  ema[kStripRegs - 1] = valid13 ? ema[kStripRegs - 1] : 0.f;
}
__device__ __forceinline__ void strip_ema_update(int s, int lane, const fft::c32 (&v)[16], ema_t &ema)
{
  float met[kStripRegs];
  strip_metric(v, met);
  strip_ema_apply(s, lane, met, ema);
}

// One window of one root of one strip: spectrum of the block (X) times the filter spectrum, back to the lag
// domain, then the average.  spec = the filter spectrum's bins lane + 64 R, tw = the inverse transform's
// first-stage twiddles: both read by the caller ahead of time.
// (the same up to the lags: v = the matched filter's output, element lane + 64 R in v[R])
__device__ __forceinline__ void strip_root_lags(int lane, fft::c32 (&v)[16], const v2f (&spec)[16], const fft::Twiddles &tw,
                                                const fft::c32 *tw2, fft::c32 *e)
{
#pragma unroll
  for (int R = 0; R < 16; R += 2) fft::cmulw2<false>(v[R], spec[R], v[R + 1], spec[R + 1]);
  fft::fft1024w_stageA<true>(lane, v, tw, e);
  __builtin_amdgcn_sched_barrier(0);
  fft::fft1024w_stageB<true>(lane, v, tw2, e, e);
  __builtin_amdgcn_sched_barrier(0);
  fft::fft1024w_stageC<true>(lane, v, e);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void strip_root_update(int s, int lane, fft::c32 (&v)[16], const v2f (&spec)[16], const fft::Twiddles &tw,
                                                  const fft::c32 *tw2, fft::c32 *e, ema_t &ema)
{
  strip_root_lags(lane, v, spec, tw, tw2, e);
  strip_ema_update(s, lane, v, ema);
}
__device__ __forceinline__ void strip_root_update(int s, int lane, fft::c32 (&v)[16], const v2f *spec_tab,
                                                  const fft::c32 *tw1, const fft::c32 *tw2, fft::c32 *e, ema_t &ema)
{
  v2f spec[16];
#pragma unroll
  for (int R = 0; R < 16; ++R) spec[R] = spec_tab[lane + 64 * R];
  fft::Twiddles tw;
  fft::stageA_twiddles(lane, tw1, tw);
  strip_root_update(s, lane, v, spec, tw, tw2, e, ema);
}

#ifdef LT_SCAN_PROF
// diagnostics build: the same with shader-clock stamps between the stages (tools/scan_bench.hip)
__device__ __forceinline__ void strip_root_update_prof(int s, int lane, fft::c32 (&v)[16], const v2f (&spec)[16], const fft::Twiddles &tw,
                                                       const fft::c32 *tw2, fft::c32 *e,
                                                       ema_t &ema, unsigned *pr, unsigned &tp)
{
#define LT_PS(i) { __builtin_amdgcn_sched_barrier(0); const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&pr[i], t_ - tp); tp = t_; __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
  for (int R = 0; R < 16; R += 2) fft::cmulw2<false>(v[R], spec[R], v[R + 1], spec[R + 1]);
  LT_PS(2) // filter spectrum
  fft::fft1024w_stageA<true>(lane, v, tw, e);
  LT_PS(3)
  fft::fft1024w_stageB<true>(lane, v, tw2, e, e);
  LT_PS(4)
  fft::fft1024w_stageC<true>(lane, v, e);
  LT_PS(5)
  strip_ema_update(s, lane, v, ema);
  LT_PS(6)
#undef LT_PS
}
#endif

// the strip's slice of conv_output_avg (natural lag order in HBM) <-> registers
__device__ __forceinline__ void strip_ema_load(const float *__restrict__ avg, int s, int lane, ema_t &ema)
{
  const int c0 = kStripPitch * s - kStripHalo;
  ema[14] = -FLT_MAX; // (the two spare registers of the tuple never win a maximum)
  ema[15] = -FLT_MAX;
#pragma unroll
  for (int r = 0; r < kStripRegs; ++r) {
    const int lag = c0 + lane + 64 * r;
    ema[r] = (lag >= 0 && lag < kNLags) ? avg[lag] : 0.f;
  }
}
__device__ __forceinline__ void strip_ema_store(float *__restrict__ avg, int s, int lane, const ema_t &ema)
{
  const int c0 = kStripPitch * s - kStripHalo;
#pragma unroll
  for (int r = 0; r < kStripRegs; ++r) {
    const int o = lane + 64 * r, lag = c0 + o;
    if (o >= kStripHalo && o < kStripHalo + kStripPitch && lag < kConvLen) avg[lag] = (lag < kNLags) ? ema[r] : 0.f;
  }
}

} // namespace lt
