// The CFO side of the tracked half-frames (lib/pss_impl.cc:199-204), run by the per-chain kernel once a chain's records
// are final: srslte_pss_cfo_compute on the uncorrected PSS symbol of every tracked half-frame -- eight records per wave
// at a time, 8 lanes each -- and then, in call order, the d_cfo_data ring with mean_cfo() (lib/pss_impl.cc:94-109).
// Neither feeds the PSS state machine; both used to be launches of their own (lt_cfo_kernel, lt_cfo_ring_kernel).
#pragma once
#include "lt_scan.h"

namespace lt {

// cfo_estimate_wave (lt_device.h) for a group of 8 lanes: the same partial sums in the same order -- lane k of the group
// holds the k-th of the eight interleaved partial sums (m mod 8) of BOTH halves of the symbol, an xor butterfly over the
// group combines each half's as ((p0+p1)+(p2+p3))+((p4+p5)+(p6+p7)) -- so eight records go through a wave at once.
// h: the root's 128 conjugated taps (LDS), cap / pss0: the capture and the absolute position of the PSS symbol's first
// sample.  Every lane of the group returns the estimate.
__device__ __forceinline__ float cfo_estimate_group8(const v2f *h, const Capture &cap, int64_t pss0, int lane)
{
  const int l8 = lane & 7;
  v2f rr[16]; // all reads first: one round trip
  const int64_t i0 = pss0 - cap.t0;
  if (i0 >= 0 && i0 + kSymbolSz <= cap.n) {
    // the symbol lies inside the chunk (always, but for a half-frame that straddles the start of a pushed chunk): one
    // 64-bit address per lane, the sixteen loads at constant offsets from it
    const v2f *__restrict__ p = cap.iq + i0 + l8;
#pragma unroll
    for (int j = 0; j < 16; ++j) rr[j] = p[64 * (j >> 3) + 8 * (j & 7)]; // j < 8: first half of the symbol, j >= 8: second half
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) rr[j] = ld_abs(cap, pss0 + 64 * (j >> 3) + l8 + 8 * (j & 7));
  }
  v2f hh[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) hh[j] = h[64 * (j >> 3) + l8 + 8 * (j & 7)];
  v2f y0 = {0.f, 0.f}, y1 = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const v2f p = cmul_plain(hh[j], rr[j]), q = cmul_plain(hh[8 + j], rr[8 + j]);
    y0.x = __fadd_rn(y0.x, p.x);
    y0.y = __fadd_rn(y0.y, p.y);
    y1.x = __fadd_rn(y1.x, q.x);
    y1.y = __fadd_rn(y1.y, q.y);
  }
  y0.x = __fadd_rn(y0.x, lane_xor<1>(y0.x)); y0.y = __fadd_rn(y0.y, lane_xor<1>(y0.y));
  y1.x = __fadd_rn(y1.x, lane_xor<1>(y1.x)); y1.y = __fadd_rn(y1.y, lane_xor<1>(y1.y));
  y0.x = __fadd_rn(y0.x, lane_xor<2>(y0.x)); y0.y = __fadd_rn(y0.y, lane_xor<2>(y0.y));
  y1.x = __fadd_rn(y1.x, lane_xor<2>(y1.x)); y1.y = __fadd_rn(y1.y, lane_xor<2>(y1.y));
  y0.x = __fadd_rn(y0.x, lane_xor<4>(y0.x)); y0.y = __fadd_rn(y0.y, lane_xor<4>(y0.y));
  y1.x = __fadd_rn(y1.x, lane_xor<4>(y1.x)); y1.y = __fadd_rn(y1.y, lane_xor<4>(y1.y));
  const v2f c = {y0.x, -y0.y};
  const v2f z = cmul_plain(c, y1);
  return (float)(atan2((double)z.y, (double)z.x) / 3.14159265358979323846);
}

// ---- d_cfo_data ring and mean_cfo() in call order, one wave.  The ring restarts where reset_score cleared d_cfo_i
// (lib/pss_impl.cc:145): the call that does so tags its half-frame tracking_lost, and d_cfo_i is 0 whenever a chain
// is not tracking.
//
// compute_moving_avg sums the first min(d_cfo_i, 200) ring entries in index order in double.  While the exponents of
// all entries seen since the ring restarted span at most 20 binades (no subnormals), every partial sum of any subset
// is a multiple of the smallest ulp below 2^53 of them: no addition rounds, so the sum is the same in any order -- and
// it can be carried along (add the new entry, drop the one it overwrites) instead of being formed anew for every call.
// Outside that window the sum runs in ring order (moving_avg_wave).
struct CfoRing {
  uint32_t cfo_i, emin, emax;
  bool narrow;
  double acc;
};

// ring: kMovingAvgSz floats of LDS owned by the wave
__device__ __forceinline__ void cfo_ring_begin(const ChainState *st, float *ring, CfoRing &R, int lane)
{
  for (int i = lane; i < kMovingAvgSz; i += 64) ring[i] = st->cfo_data[i];
  R.cfo_i = st->cfo_i;
  __builtin_amdgcn_wave_barrier();
  // the entries the previous runs left behind: exponent range and (if it is narrow) their exact sum
  const uint32_t npts = R.cfo_i < (uint32_t)kMovingAvgSz ? R.cfo_i : (uint32_t)kMovingAvgSz;
  double part = 0.0;
  uint32_t lmin = 255u, lmax = 0u;
  bool sub = false;
  for (uint32_t i = lane; i < npts; i += 64) {
    const float v = ring[i];
    const uint32_t bits = __float_as_uint(v) & 0x7fffffffu, e = bits >> 23;
    if (bits) {
      sub = sub || e == 0u;
      lmin = e < lmin ? e : lmin;
      lmax = e > lmax ? e : lmax;
    }
    part += (double)v;
  }
  R.emin = (uint32_t)wave_min_i((int)lmin);
  R.emax = (uint32_t)wave_max_i((int)lmax);
  R.narrow = !__ballot(sub) && (R.emax == 0u || (R.emax - R.emin <= 20u && R.emax < 255u));
  part += lane_xor_d<1>(part); part += lane_xor_d<2>(part); part += lane_xor_d<4>(part);
  part += lane_xor_d<8>(part); part += lane_xor_d<16>(part); part += lane_xor_d<32>(part);
  R.acc = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(part)), __builtin_amdgcn_readfirstlane(__double2loint(part)));
}

// `count` records in call order: get(i, flags, cfo) for i < count (flags with LT_F_SSS_RAN: the record has an estimate;
// with LT_F_TAG_LOST: the ring restarts in front of it), put(i, mean) = the record's mean_cfo().
template <typename Get, typename Put>
__device__ __forceinline__ void cfo_ring_records(float *ring, CfoRing &R, int count, int lane, Get get, Put put)
{
  uint32_t cfo_i = R.cfo_i, emin = R.emin, emax = R.emax;
  bool narrow = R.narrow;
  double acc = R.acc;
  for (int base = 0; base < count; base += 64) {
    const int i = base + lane;
    uint32_t flags = 0u;
    float cfo_v = 0.f;
    if (i < count) get(i, flags, cfo_v);
    const bool is_ran = (flags & LT_F_SSS_RAN) != 0;
    const float c = is_ran ? cfo_v : 0.f;
    const unsigned long long ran = __ballot(is_ran);
    unsigned long long lost = __ballot((flags & LT_F_TAG_LOST) != 0);
    if (!ran && !lost) continue;
    // tracking_lost tags in front of every estimate of these records (the half-frames a chain emits while it builds its
    // score, lib/pss_impl.cc:210-214, then tracking starts): the ring restarts, and the estimates behind go the fast way
    if (lost && !(ran & ((2ull << (63 - __builtin_clzll(lost))) - 1ull))) {
      cfo_i = 0;
      emin = 255u; emax = 0u; narrow = true; acc = 0.0;
      lost = 0;
      if (!ran) continue;
    }
    // ---- the 64 records at once.  While the exponent window holds for every entry up to the last of these
    // records, all the sums are exact whatever their order: record j's sum is the carried one plus the entries
    // of the records up to j minus the ring entries those overwrite -- two prefix sums across the lanes.
    if (narrow && !lost) {
      const uint32_t bits = __float_as_uint(c) & 0x7fffffffu, e = bits >> 23;
      const bool counts = is_ran && bits != 0u;
      const uint32_t bmin = (uint32_t)wave_min_i(counts ? (int)e : 255), bmax = (uint32_t)wave_max_i(counts ? (int)e : 0);
      const uint32_t nmin = bmin < emin ? bmin : emin, nmax = bmax > emax ? bmax : emax;
      const bool sub = __ballot(counts && e == 0u) != 0;
      if (!sub && (nmax == 0u || (nmax - nmin <= 20u && nmax < 255u))) {
        const uint32_t rank = (uint32_t)__builtin_popcountll(ran & ((1ull << lane) - 1ull));
        const uint32_t idx = cfo_i + rank; // this record's entry (if it has one)
        const uint32_t slot = idx % kMovingAvgSz;
        double add = is_ran ? (double)c : 0.0;
        double drop = (is_ran && idx >= (uint32_t)kMovingAvgSz) ? (double)ring[slot] : 0.0; // the entry it overwrites
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const double a2 = __hiloint2double(__shfl_up(__double2hiint(add), d), __shfl_up(__double2loint(add), d));
          const double d2 = __hiloint2double(__shfl_up(__double2hiint(drop), d), __shfl_up(__double2loint(drop), d));
          if (lane >= d) {
            add += a2;
            drop += d2;
          }
        }
        const double mine_acc = (acc + add) - drop;
        const uint32_t npts = idx + 1 < (uint32_t)kMovingAvgSz ? idx + 1 : (uint32_t)kMovingAvgSz;
        __builtin_amdgcn_wave_barrier();
        if (is_ran) {
          ring[slot] = c; // (64 records at most, 200 slots: no two of them share one)
          put(i, (float)(mine_acc / (double)npts));
        }
        __builtin_amdgcn_wave_barrier();
        const int last = 63 - __builtin_clzll(ran);
        acc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mine_acc), last), __builtin_amdgcn_readlane(__double2loint(mine_acc), last));
        cfo_i += (uint32_t)__builtin_popcountll(ran);
        emin = nmin;
        emax = nmax;
        continue;
      }
    }
    // ---- in call order, one record after the other
    unsigned long long events = ran | lost;
    float mine = 0.f;
    double my_acc = 0.0; // narrow mode: this lane's record's sum and count; the division waits until after the walk
    uint32_t my_n = 0;
    while (events) {
      const int j = __builtin_ctzll(events);
      events &= events - 1;
      if ((lost >> j) & 1) {
        cfo_i = 0;
        emin = 255u; emax = 0u; narrow = true; acc = 0.0;
      }
      if ((ran >> j) & 1) {
        const float cj = rlane(c, j);
        const uint32_t slot = cfo_i % kMovingAvgSz;
        const uint32_t npts = cfo_i + 1 < (uint32_t)kMovingAvgSz ? cfo_i + 1 : (uint32_t)kMovingAvgSz;
        const uint32_t bits = __float_as_uint(cj) & 0x7fffffffu, e = bits >> 23;
        if (bits) {
          if (e == 0u) narrow = false;
          emin = e < emin ? e : emin;
          emax = e > emax ? e : emax;
          if (emax - emin > 20u || emax >= 255u) narrow = false;
        }
        if (narrow) {
          if (cfo_i >= (uint32_t)kMovingAvgSz) acc -= (double)ring[slot]; // the entry this call overwrites
          acc += (double)cj;
          if (lane == j) {
            my_acc = acc;
            my_n = npts;
          }
        } else {
          const float mc = moving_avg_wave(ring, cfo_i + 1, slot, cj);
          if (lane == j) {
            mine = mc;
            my_n = 0;
          }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) ring[slot] = cj;
        __builtin_amdgcn_wave_barrier();
        cfo_i++;
      }
    }
    if (my_n) mine = (float)(my_acc / (double)my_n);
    if (i < count && (flags & LT_F_SSS_RAN)) put(i, mine);
  }
  R.cfo_i = cfo_i; R.emin = emin; R.emax = emax; R.narrow = narrow; R.acc = acc;
}

__device__ __forceinline__ void cfo_ring_end(ChainState *st, const float *ring, const CfoRing &R, int lane)
{
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < kMovingAvgSz; i += 64) st->cfo_data[i] = ring[i];
  if (lane == 0) st->cfo_i = R.cfo_i;
}

} // namespace lt
