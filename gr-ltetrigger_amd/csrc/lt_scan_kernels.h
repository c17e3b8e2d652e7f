// Scan rounds: srslte_pss_find_pss (lib/pss_impl.cc:165-167) of every chain, window after window, with the
// exponential average held in registers and nothing but 32-byte partial results per (window, strip) written
// back.  See lt_scan.h for the strip geometry.
//
// Where a chain will search next is known in advance as long as nothing changes its timing: an idle chain (not
// tracking, score 0) searches every 9600-sample window, misses and moves on by exactly 9600 samples
// (lib/pss_impl.cc:163-182, 217); a chain that builds its score keeps hitting at peak_pos 960 and searches every
// window, a tracked one every (track_every + 1)-th (lib/pss_impl.cc:163-172, 185-195).  That is what nearly all
// chains do nearly all the time, and the search is where the arithmetic is.  The work runs in rounds:
// lt_plan_kernel puts the chains of a channel into one job -- up to three window sequences, one per set of chains
// that will search the same windows --, lt_scan_kernel runs the next windows of every sequence speculatively, and
// lt_resolve_kernel executes general_work call by call on the results, for as long as the chain stays on the
// windows that were scanned.  A chain that leaves them -- a hit re-aligns it, tracking is lost -- gets a sequence
// of its own in the next round; its average is carried over exactly (ChainState: a valid state + the windows to
// run it through again, or all zeros after srslte_pss_reset).  A window the strips cannot resolve hands the chain
// to the per-chain kernel (lt_chain_kernel.h).
//
// Work decomposition.  A job's 12 strips go to four workgroups of three *groups*; a group is four waves that own
// one strip for the whole round: a forward wave, which fetches the strip's 1024 samples of every window of every
// sequence and runs the first two stages of their transform, and one wave per root, which runs the last stage out
// of the hand-over buffer, multiplies the spectrum by its root's filter spectrum (held in registers), transforms
// back, keeps its root's average and reduces it to the strip's partial result.
//   workgroup = 12 waves = 3 per SIMD; LDS: 12 exchange areas (102 KB) + 6 hand-over buffers (49 KB) + twiddles
//   (8.6 KB) = 163 224 of 163 840 bytes
#pragma once
#include "lt_scan.h"

// Ordering of the LDS hand-over.  A wave's LDS instructions execute in program order, so a counter written
// after the data (or read before it) needs no s_waitcnt -- only the compiler must keep the order.  A
// workgroup-scope fence would also wait for the wave's global loads and stores (vmcnt(0)): the forward wave's
// prefetch of the next windows and the root waves' partial-result stores, i.e. a full memory latency per window.
#define LT_LDS_ORDER __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")


namespace lt {

// One per channel and round, built by lt_plan_kernel from the chain states: the channel's chains grouped into up to
// three window sequences.  Chains that will search the same windows (normally all three: idle, on one grid) share
// a sequence and with it the forward transforms; a chain on windows of its own (it has re-aligned, it tracks)
// forms a sequence of its own, served by the same workgroups in its own root-wave slot.
struct ScanSeq {
  int64_t pos_a;      // prefix: windows pos_a + k * stride_a * 9600, k < n_a -- lag metric and average only
  int64_t pos_b;      // the round's windows: pos_b + k * stride_b * 9600, k < n_b
  int64_t pos_c;      // ... and, for a chain about to start tracking, pos_c + k * stride_c * 9600, k < n_c: the
                      // searches of the tracked chain, its average cleared in front of the first one
                      // (srslte_pss_reset where tracking starts, lib/pss_impl.cc:119-126)
  int32_t stride_a, n_a, stride_b, n_b, stride_c, n_c;
  int32_t roots;      // bit r: chain (channel, N_id_2 = r) follows this sequence
  int32_t zero;       // its chains start from an all-zero average
};
struct ScanJob {
  ScanSeq seq[3];     // seq[0] is the longest
  int32_t channel;
  int32_t n_seq;
  int32_t seq_of_root[3]; // -1: the chain is not served this round
  int32_t cached;         // the lag metric of every window of the job comes from the window cache (small batches): no
                          // forward wave, no transforms -- lt_metric_kernel has formed them all before the scan starts
  int32_t slot_base[3];   // ... chain r's windows in slots slot_base[r] + k, k counted through its sequence's three parts
  int32_t pad;
};
// (the job array is written by the kernel before: the compiler reads it with vector loads; its fields are
// wave-uniform all the same, and everything computed from them -- window positions, buffer resources -- should
// live in scalar registers)
__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uniform_i64(int64_t v)
{
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ ScanSeq uniform_seq(const ScanSeq &q)
{
  ScanSeq u;
  u.pos_a = uniform_i64(q.pos_a);
  u.pos_b = uniform_i64(q.pos_b);
  u.pos_c = uniform_i64(q.pos_c);
  u.stride_c = uniform_i32(q.stride_c);
  u.n_c = uniform_i32(q.n_c);
  u.stride_a = uniform_i32(q.stride_a);
  u.n_a = uniform_i32(q.n_a);
  u.stride_b = uniform_i32(q.stride_b);
  u.n_b = uniform_i32(q.n_b);
  u.roots = uniform_i32(q.roots);
  u.zero = uniform_i32(q.zero);
  return u;
}
__device__ __forceinline__ int seq_windows(const ScanSeq &q) { return q.n_a + q.n_b + q.n_c; }
__device__ __forceinline__ int64_t seq_window_pos(const ScanSeq &q, int k)
{
  if (k < q.n_a) return q.pos_a + (int64_t)k * q.stride_a * kHalfFrame;
  k -= q.n_a;
  if (k < q.n_b) return q.pos_b + (int64_t)k * q.stride_b * kHalfFrame;
  return q.pos_c + (int64_t)(k - q.n_b) * q.stride_c * kHalfFrame;
}

// A workgroup carries G strips (G groups of four waves).  G = 3 fills the chip with 64 channels (4 workgroups per
// channel, 12 waves per CU: one root wave per SIMD sits in a third slot and is slower for it); with
// fewer channels more, smaller workgroups put more CUs on a channel and leave no SIMD with three waves.
constexpr int scan_waves(int G) { return 4 * G; }
constexpr int scan_threads(int G) { return 64 * scan_waves(G); }
constexpr int scan_wg_per_job(int G) { return kStrips / G; }
#ifndef LT_SCAN_EARLY_READY
#define LT_SCAN_EARLY_READY 1 // a root wave reads its hand-over counters in front of the window's table reads (below)
#endif
#ifndef LT_POLL_SLEEP
#define LT_POLL_SLEEP 1 // s_sleep units (64 clocks) between two looks at a hand-over counter
#endif
constexpr int kScanMaxRound = 128;           // windows per job and round at most (one resolver lane each)
// ---- window cache (lt_chain_kernel.h): sizes of a request
constexpr int kCacheStripLags = 64 * kStripRegs;   // 896 floats per (slot, strip)
constexpr int kCacheMinSearches = 6;               // fewer searches ahead than this: not worth a pass of its own
constexpr int kCacheMaxPerChain = 256;
constexpr int kStreamBlock = kCacheStripLags;           // 896 stream elements per block = per (slot, strip)
constexpr int kStreamSlot = kStrips * kStreamBlock;     // 10752 per slot
constexpr int kStreamBack = 1024;                       // the stream starts this far in front of the chain's position:
                                                        // one hit moves the grid back by at most 960

struct ScanArgs {
  const v2f *iq;       // [n_channels][ch_stride]
  int64_t ch_stride;
  const v2f *hist;     // [n_channels][hist_stride] or null
  int64_t hist_stride;
  int64_t t0, n, hist_len;
  const ScanJob *jobs; // [jobs_cap]: long jobs from the front, short ones from the back
  const int *n_jobs;   // [2]: how many of each
  int jobs_cap;
  ChainState *state;   // [n_channels * 3]
  StripHead *heads;    // [n_channels * 3][kScanMaxRound][12]: what a strip says about a window (lt_scan.h, the deferred form) ...
  float *rows;         // [n_channels * 3][kScanMaxRound][12][32]: ... and the 32 lags around its maximum
  const DeviceTables *tab;
  unsigned long long *prof; // diagnostics build only (LT_SCAN_PROF): shader-clock sums per phase of one root wave
  const float *cache;       // window cache (slots of kStrips x 64 * kStripRegs floats), or null
  int skip_cached = 0;      // cached jobs are lt_scan_cached_kernel's
};

__device__ __forceinline__ void set_wave_priority(int p)
{
  if (p <= 0) __builtin_amdgcn_s_setprio(0);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(3);
}

template <int kScanGroups> __global__ __launch_bounds__(scan_threads(kScanGroups)) void lt_scan_kernel(ScanArgs a)
{
  constexpr int kScanWaves = scan_waves(kScanGroups), kScanThreads = scan_threads(kScanGroups), kScanWgPerJob = scan_wg_per_job(kScanGroups);
  static_assert(kStrips % kScanGroups == 0, "a job's strips divide evenly among its workgroups");
  __shared__ __attribute__((aligned(16))) v2f E[kScanWaves * fft::kScratchW]; // one exchange area per wave
  // a forward wave leaves its blocks (two transform stages done) here for the group's root waves, in turn: while
  // they read one, it fills the other
  __shared__ __attribute__((aligned(16))) v2f OUT[kScanGroups][2][fft::kE2Len];
  __shared__ __attribute__((aligned(16))) fft::c32 TW1[fft::kTw1Len];
  __shared__ __attribute__((aligned(16))) fft::c32 TW2[fft::kTw2Len];
  __shared__ int sync_ready[kScanGroups][3], sync_taken[kScanGroups][3];
#ifdef LT_SCAN_PROF
  __shared__ unsigned PROF[kScanWaves][8]; // shader-clock ticks per phase and wave (in LDS: the kernel has no scalar registers to spare)
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Workgroup -> (job, part).  Workgroups are dispatched to the 8 XCDs in turn (b & 7), and the workgroups of a job
  // should meet on one XCD: their strips overlap by 213 samples, the second read of those then comes from that XCD's
  // L2.  All (job, part) pairs in job order -- long jobs first: they sit at the front of the job array, the short
  // ones at its back -- are cut into 8 equal runs, one per XCD: every XCD gets the same number of workgroups
  // whatever the number of jobs, and at most 7 jobs are split between two XCDs.
  const int b = blockIdx.x;
  const int n_front = a.n_jobs[0], n_back = a.n_jobs[1];
  const int n_parts = (n_front + n_back) * kScanWgPerJob;
  const int per_xcd = (n_parts + 7) >> 3;
  const int part_i = (b & 7) * per_xcd + (b >> 3);
  if ((b >> 3) >= per_xcd || part_i >= n_parts) return;
  const int quarter = part_i % kScanWgPerJob;
  int job_i = part_i / kScanWgPerJob;
  if (job_i >= n_front) job_i = a.jobs_cap - 1 - (job_i - n_front);
  const ScanJob *jp = a.jobs + job_i; // (read piecewise: a local copy indexed by sequence would live in scratch)
  const int n_seq = uniform_i32(jp->n_seq), channel = uniform_i32(jp->channel);
  if (n_seq <= 0) return;
  const bool cached = a.cache != nullptr && uniform_i32(jp->cached) != 0;
  if (cached && a.skip_cached) return;
  // group g = w % 3 owns strip 3 quarter + g; role w / 3: 0 = forward, 1..3 = root 0..2 (the three forward
  // waves land on three different SIMDs)
  const int g = w % kScanGroups, role = w / kScanGroups;
  const int strip = quarter * kScanGroups + g;
  const int root = role - 1;
  const int my_seq = role ? uniform_i32(jp->seq_of_root[root]) : 0;
  const bool active = (role == 0 && !cached) || (role != 0 && my_seq >= 0);

  if (!cached) { // (a cached job has no transforms and no hand-over: nothing to set up, and no barrier -- its forward waves are gone)
    for (int i = tid; i < fft::kTw1Len; i += kScanThreads) TW1[i] = *reinterpret_cast<const fft::c32 *>(a.tab->fft_tw1[i]);
    if (tid < fft::kTw2Len) TW2[tid] = *reinterpret_cast<const fft::c32 *>(a.tab->fft_tw2[tid]);
    if (tid < kScanGroups * 3) (&sync_ready[0][0])[tid] = (&sync_taken[0][0])[tid] = 0;
#ifdef LT_SCAN_PROF
    if (tid < kScanWaves * 8) (&PROF[0][0])[tid] = 0u;
#endif
    __syncthreads(); // tables in place: the only workgroup barrier
  }
#ifdef LT_SCAN_DROP_LAST_STRIP
  // (experiment, tools/scan_bench.hip only: what the chip gains if a window needed 11 blocks instead of 12 -- strips without
  // halo -- with the workgroups' shape as it is: the fourth workgroup of a channel runs two groups)
  if (strip == kStrips - 1) return;
#endif
  fft::c32 *e = E + w * fft::kScratchW;

  Capture cap;
  cap.iq = a.iq + (int64_t)channel * a.ch_stride;
  cap.hist = a.hist ? a.hist + (int64_t)channel * a.hist_stride : nullptr;
  cap.t0 = a.t0;
  cap.n = a.n;
  cap.hist_len = a.hist ? a.hist_len : 0;

  // The two roles run separate loops, so that each gets a register allocation of its own: the forward wave
  // carries blocks of samples, a root wave one block and its average.  Hand-over per group and sequence through
  // two LDS counters (no workgroup barrier after the table load: the three groups of a workgroup drift apart,
  // which is wanted -- every SIMD holds one wave of each group, and waves in different phases of the transform
  // keep both the vector unit and the LDS busy):
  //   ready[g][q] = windows of sequence q whose first two transform stages the forward wave has left in OUT[g]
  //   taken[g][r] = windows root wave r has read from there (a root wave may be one window ahead of its sequence's others)
  // Hand-over number h (all sequences of the job counted together, in the forward wave's order) uses OUT[g][h & 1].
  if (role == 0 && !active) {
    // (the forward wave of a cached job: nothing to transform -- and no root wave that would ever release a buffer)
  } else if (role == 0) {
    // Blocks are fetched one hand-over of their kind ahead, each into the registers the block before has just
    // left (two register sets taking turns for sequence 0, so that no block is ever moved; one for the others'
    // blocks -- chains on windows of their own, 1 <-> 2 in turn).
    const ScanSeq s0 = uniform_seq(jp->seq[0]), s1 = uniform_seq(jp->seq[1]), s2 = uniform_seq(jp->seq[2]);
    const int n0 = seq_windows(s0);
    const int n1 = n_seq > 1 ? seq_windows(s1) : 0; // (n0 >= n1 >= n2)
    const int n2 = n_seq > 2 ? seq_windows(s2) : 0;
    const int r0 = s0.roots & 7, r1 = s1.roots & 7, r2 = s2.roots & 7;
    fft::c32 xa[16], xb[16]; // sequence 0: even / odd windows
    fft::c32 xr[16];         // the next block of sequence 1 or 2
    strip_load(cap, seq_window_pos(s0, 0), strip, lane, xa);
    if (n0 > 1) strip_load(cap, seq_window_pos(s0, 1), strip, lane, xb);
    if (n1 > 0) strip_load(cap, seq_window_pos(s1, 0), strip, lane, xr);
    // what sits in OUT[g][0 / 1]: the root waves it is for and the `taken` count of theirs that frees the buffer
    int held_roots0 = 0, held_roots1 = 0, held_need0 = 0, held_need1 = 0;
    int h = 0;
#ifdef LT_SCAN_PROF
    unsigned tp = (unsigned)__builtin_amdgcn_s_memtime();
#define LT_FSTAMP(i) { __builtin_amdgcn_sched_barrier(0); const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&PROF[w][i], t_ - tp); tp = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define LT_FSTAMP(i)
#endif
    auto hand_over = [&](int q, int k, int roots_q, int ln, fft::c32 (&x)[16]) {
      LT_FSTAMP(0) // loads issued
      fft::fft1024w_stageA<false>(ln, x, TW1, e);
      __builtin_amdgcn_sched_barrier(0);
      LT_FSTAMP(1) // first stage (and the wait for the block's samples)
      // the block that used this buffer two hand-overs ago has been taken by all its root waves
      const int hr = (h & 1) ? held_roots1 : held_roots0, hn = (h & 1) ? held_need1 : held_need0;
      bool waited = false;
#pragma unroll
      for (int r = 0; r < 3; ++r)
        if ((hr >> r) & 1)
          while (__hip_atomic_load(&sync_taken[g][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < hn) {
            waited = true;
            __builtin_amdgcn_s_sleep(LT_POLL_SLEEP);
          }
      // Issue priority (s_setprio).  Three waves share a SIMD and the arbiter serves the oldest first: left alone, the wave
      // in a SIMD's third slot gets what the other two leave, its strip falls behind and the workgroup waits for it at the
      // end.  A forward wave that finds its buffer taken already is what its root waves wait for: first in line; one
      // that had to wait has time to spare.
      if constexpr (kScanGroups == 3) {
        // (with three window sequences the forward wave transforms three blocks per window: always what the others wait for)
        if (waited && n_seq < 3) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(3);
      }
      LT_FSTAMP(2) // waiting for the buffer
      LT_LDS_ORDER;
      fft::fft1024w_stageB<false>(ln, x, TW2, e, OUT[g][h & 1]);
      LT_LDS_ORDER;
      LT_FSTAMP(3) // second stage
      if (lane == 0) __hip_atomic_store(&sync_ready[g][q], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (h & 1) {
        held_roots1 = roots_q;
        held_need1 = k + 1;
      } else {
        held_roots0 = roots_q;
        held_need0 = k + 1;
      }
      ++h;
    };
    auto window = [&](int k, fft::c32 (&x)[16]) {
      int ln = lane; // (opaque per window: keeps lane-dependent addresses from being hoisted out of the loop and spilled)
      asm volatile("" : "+v"(ln));
      hand_over(0, k, r0, ln, x);
      if (k + 2 < n0) strip_load(cap, seq_window_pos(s0, k + 2), strip, ln, x);
      if (k < n1) {
        hand_over(1, k, r1, ln, xr);
        if (k < n2) strip_load(cap, seq_window_pos(s2, k), strip, ln, xr);
        else if (k + 1 < n1) strip_load(cap, seq_window_pos(s1, k + 1), strip, ln, xr);
      }
      if (k < n2) {
        hand_over(2, k, r2, ln, xr);
        if (k + 1 < n1) strip_load(cap, seq_window_pos(s1, k + 1), strip, ln, xr);
      }
    };
    for (int k = 0; k < n0; k += 2) {
      window(k, xa);
      if (k + 1 < n0) window(k + 1, xb);
    }
#ifdef LT_SCAN_PROF
    if (a.prof && blockIdx.x == 8 && lane == 0) {
      for (int i = 0; i < 8; ++i) a.prof[w * 8 + i] = PROF[w][i];
      unsigned id;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
      a.prof[96 + w] = id;
    }
#endif
  } else if (active && cached) {
    // ---- a cached job's root wave: the windows' lag metrics come from the window cache (lt_metric_kernel has formed
    // them, strip by strip, with the search's own instructions), 14 loads per lane and window, the next window's on
    // their way while this one's average and peak search run.  No forward wave, no LDS.
    const ScanSeq sq = uniform_seq(jp->seq[my_seq]);
    const int n_w = seq_windows(sq);
    ema_t ema;
    const int chain = channel * 3 + root;
    ChainState *st = a.state + chain;
    const int sel = st->ema_sel;
    if (sq.zero) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ema[r] = r < kStripRegs ? 0.f : -FLT_MAX;
    } else {
      strip_ema_load(st->avg[sel], strip, lane, ema);
    }
    StripHead *heads = a.heads + (int64_t)chain * kScanMaxRound * kStrips + strip;
    float *rows = a.rows + ((int64_t)chain * kScanMaxRound * kStrips + strip) * kRowLags;
    const float *m = a.cache + (size_t)uniform_i32(jp->slot_base[root]) * (kStrips * 64 * kStripRegs) + (size_t)strip * (64 * kStripRegs) + lane;
    float met[kStripRegs], nxt[kStripRegs];
#pragma unroll
    for (int r = 0; r < kStripRegs; ++r) nxt[r] = m[64 * r];
    for (int k = 0; k < n_w; ++k) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
#pragma unroll
      for (int r = 0; r < kStripRegs; ++r) met[r] = nxt[r];
      if (k + 1 < n_w) {
        const float *mn = m + (size_t)(k + 1) * (kStrips * 64 * kStripRegs);
#pragma unroll
        for (int r = 0; r < kStripRegs; ++r) nxt[r] = mn[64 * r];
      }
      if (k == sq.n_a + sq.n_b && sq.n_c > 0) { // tracking starts here: srslte_pss_reset
        asm volatile("; the average cleared" ::: "memory");
#pragma unroll
        for (int r = 0; r < kStripRegs; ++r) ema[r] = 0.f;
      }
      strip_ema_apply(strip, ln, met, ema);
      if (k < sq.n_a) {
        if (k == sq.n_a - 1) strip_ema_store(st->avg[(sel + 1) % 3], strip, ln, ema);
      } else {
        StripHead hd;
        float row;
        int idx;
        strip_head_row(ema, strip, ln, hd, row, idx);
        if (idx >= 0) rows[(int64_t)(k - sq.n_a) * (kStrips * kRowLags) + idx] = row;
        if (lane == 0) heads[(int64_t)(k - sq.n_a) * kStrips] = hd;
      }
    }
    strip_ema_store(st->avg[(sel + 2) % 3], strip, lane, ema);
  } else if (active) {
    const ScanSeq sq = uniform_seq(jp->seq[my_seq]);
    const int n_w = seq_windows(sq);
    // hand-over number of this sequence's window k (the forward wave's order: per k, sequence 0, 1, 2 as far as
    // they reach)
    const int n1 = n_seq > 1 ? uniform_i32(jp->seq[1].n_a + jp->seq[1].n_b + jp->seq[1].n_c) : 0;
    const int n2 = n_seq > 2 ? uniform_i32(jp->seq[2].n_a + jp->seq[2].n_b + jp->seq[2].n_c) : 0;
    fft::c32 x[16]; // spectrum, then lags
    ema_t ema;
    const int chain = channel * 3 + root;
    ChainState *st = a.state + chain;
    const int sel = st->ema_sel;
    if (sq.zero) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ema[r] = r < kStripRegs ? 0.f : -FLT_MAX;
    } else {
      strip_ema_load(st->avg[sel], strip, lane, ema);
    }
    // the filter spectrum's 16 bins of this lane stay in registers for the whole job
    v2f spec[16];
#pragma unroll
    for (int R = 0; R < 16; ++R) {
      const cf32 hh = a.tab->pss_spec[root][lane + 64 * R];
      spec[R] = v2f{hh.re, hh.im};
    }
    StripHead *heads = a.heads + (int64_t)chain * kScanMaxRound * kStrips + strip;
    float *rows = a.rows + ((int64_t)chain * kScanMaxRound * kStrips + strip) * kRowLags;
#ifdef LT_SCAN_PROF
    unsigned tp = (unsigned)__builtin_amdgcn_s_memtime();
#define LT_PSTAMP(i) { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&PROF[w][i], t_ - tp); tp = t_; }
#else
#define LT_PSTAMP(i)
#endif
    // Two strips per workgroup, two waves per SIMD: the strips are dealt alike, but of a strip's three root waves one
    // shares its SIMD with the forward wave and two with each other, the second of them in the younger slot.  With all
    // roots on one window sequence (a channel without a cell: the common case of a band scan) a root wave that is behind
    // its strip's others moves up; with more sequences the forward wave is what matters, a root wave on its own has
    // nobody to measure itself against, and nothing is set (measured: 0.213 -> 0.199 ms per 64 windows of 40 channels
    // with one sequence; with two or three, priorities of any kind cost 7-18 %).
    const int r1 = (root + 1) % 3, r2 = (root + 2) % 3;
    const bool on1 = uniform_i32(jp->seq_of_root[r1]) >= 0, on2 = uniform_i32(jp->seq_of_root[r2]) >= 0;
    const bool prio2 = kScanGroups == 2 && n_seq == 1 && (on1 || on2);
    for (int k = 0; k < n_w; ++k) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
#if LT_SCAN_EARLY_READY
      // The hand-over counters first, the stage's twiddles behind them: a wave's LDS reads return in order, so counters read
      // behind the eight 16-byte table reads would wait for those, and a first look at `ready` taken only after the priorities
      // are set is a second trip through the LDS -- on every window of every root wave.  Read up front, all three words are back
      // while the tables are still on their way, and in the usual case (the forward wave is ahead) no poll follows.
      int rdy0 = __hip_atomic_load(&sync_ready[g][my_seq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      int p1v = 0, p2v = 0;
      if constexpr (kScanGroups == 3) {
        p1v = __hip_atomic_load(&sync_taken[(g + 1) % 3][root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        p2v = __hip_atomic_load(&sync_taken[(g + 2) % 3][root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __builtin_amdgcn_sched_barrier(0);
#endif
      fft::Twiddles tw; // (table reads in flight while the wave waits for the block)
      fft::stageA_twiddles(ln, TW1, tw);
      __builtin_amdgcn_sched_barrier(0);
      const int hk = k + (k < n1 ? k : n1) + (k < n2 ? k : n2) + (my_seq >= 1) + (my_seq >= 2 && k < n1);
      const fft::c32 *e_fwd = OUT[g][hk & 1];
      if constexpr (kScanGroups == 3) {
        // ... and the workgroup's three strips should finish together: a root wave measures itself against its root's
        // waves of the other two strips (windows taken so far) -- behind them it moves up, ahead of both it steps back
#if LT_SCAN_EARLY_READY
        const int p1 = uniform_i32(p1v), p2 = uniform_i32(p2v);
#else
        const int p1 = uniform_i32(__hip_atomic_load(&sync_taken[(g + 1) % 3][root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        const int p2 = uniform_i32(__hip_atomic_load(&sync_taken[(g + 2) % 3][root], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
#endif
        const int behind = (k < p1) + (k < p2), ahead = (k > p1) + (k > p2);
        set_wave_priority(1 + behind - (ahead == 2 ? 1 : 0));
      }
      if constexpr (kScanGroups == 2) {
        if (prio2) {
          const int p1 = uniform_i32(__hip_atomic_load(&sync_taken[g][r1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
          const int p2 = uniform_i32(__hip_atomic_load(&sync_taken[g][r2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
          set_wave_priority(((on1 && k < p1) || (on2 && k < p2)) ? 2 : 1);
        }
      }
#if LT_SCAN_EARLY_READY
      if (uniform_i32(rdy0) <= k)
#endif
      while (__hip_atomic_load(&sync_ready[g][my_seq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= k) __builtin_amdgcn_s_sleep(LT_POLL_SLEEP);
      LT_LDS_ORDER;
      LT_PSTAMP(0) // waiting for the forward wave
      fft::fft1024w_stageC<false>(ln, x, e_fwd); // x = the block's spectrum
      LT_LDS_ORDER;
      if (lane == 0) __hip_atomic_store(&sync_taken[g][root], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      LT_PSTAMP(1) // last forward stage
      if (k == sq.n_a + sq.n_b && sq.n_c > 0) { // tracking starts here: srslte_pss_reset
        asm volatile("; the average cleared" ::: "memory"); // (a branch taken once a round, not fourteen selects per window)
#pragma unroll
        for (int r = 0; r < kStripRegs; ++r) ema[r] = 0.f;
      }
#ifdef LT_SCAN_PROF
      strip_root_update_prof(strip, ln, x, spec, tw, TW2, e, ema, PROF[w], tp);
#else
      strip_root_update(strip, ln, x, spec, tw, TW2, e, ema);
#endif
      if (k < sq.n_a) {
        // prefix window: the chain has been through it already (its record exists); after the last one the
        // average is a state the chain can be restarted from
        if (k == sq.n_a - 1) strip_ema_store(st->avg[(sel + 1) % 3], strip, ln, ema);
      } else {
        // (the deferred form: the strip's maximum, the 64 lags around it, the maximum over the rest -- the lobe is walked by
        // the resolver, for the one strip of twelve that holds the window's peak)
        StripHead hd;
        float row;
        int idx;
        strip_head_row(ema, strip, ln, hd, row, idx);
        if (idx >= 0) rows[(int64_t)(k - sq.n_a) * (kStrips * kRowLags) + idx] = row;
        if (lane == 0) heads[(int64_t)(k - sq.n_a) * kStrips] = hd;
      }
      LT_PSTAMP(7) // resolve + store
    }
#ifdef LT_SCAN_PROF
    if (a.prof && blockIdx.x == 8 && lane == 0) {
      for (int i = 0; i < 8; ++i) a.prof[w * 8 + i] = PROF[w][i];
      unsigned id;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
      a.prof[96 + w] = id;
    }
#endif
    strip_ema_store(st->avg[(sel + 2) % 3], strip, lane, ema);
  }
}

// ---------------------------------------------------------------- cached jobs: a round's windows in segments
// A cached job (a batch of <= 12 channels: lt_metric_kernel has formed every window's lag metrics) leaves a root wave of
// lt_scan_kernel nothing but the average and the strip's head per window -- 330 dependent instructions, 0.8 us, one
// wave per (strip, root), 36 waves on the whole chip for a channel, window after window: config 2's round of 99 windows
// was 79 us of that.  A window's average is a recurrence over the windows before it, but a cheap one (14 loads and 7
// packed operations per lane against the 300 instructions of the head), so the round is cut into kSeg segments and
// wave j of a (strip, root) runs the average alone -- the same operations in the same order: the same bits -- through
// the windows in front of its segment and the whole work on its own.  Segments grow shorter towards the end so that all
// waves finish together: with `light` = a light window's cost over a full one's, b_j = n (1 - q^j) / (1 - q^S),
// q = 1 - light (measured on config 2, 99-window rounds, us per round: one wave 79; 4 / 8 / 16 / 32 segments 38 / 33 / 33 / 40,
// light = 0.08 ... 0.25 within 2 us of each other -- a wave has at most 64 loads in flight, four windows' worth, so a light
// window costs what a quarter of a trip to the L2 costs: profiles/r05_config2_segments.txt).  One-wave workgroups (nothing is shared but the cache lines, and a wave alone on its SIMD issues
// fastest); the S waves of a (strip, root) on one XCD, whose L2 then serves S - 1 of their S reads of a window.
struct CachedScanArgs {
  const ScanJob *jobs;
  const int *n_jobs;
  int jobs_cap;
  ChainState *state;
  StripHead *heads;
  float *rows;
  const float *cache;
  int n_seg;   // waves per (job, strip, root)
  float light; // cost of a window that only updates the average, in full windows
};
constexpr int kCachedAhead = 8; // windows whose loads are in flight
constexpr int kCachedSegMax = 32;

__global__ __launch_bounds__(64) void lt_scan_cached_kernel(CachedScanArgs a)
{
  constexpr int P = kCachedAhead;
  const int lane = threadIdx.x;
  // workgroup -> (unit = (job, strip, root), segment): a unit's segments on one XCD (workgroups go to the XCDs in turn)
  const int b = blockIdx.x, S = a.n_seg;
  const int unit = ((b >> 3) / S) * 8 + (b & 7), seg = (b >> 3) % S;
  const int n_front = a.n_jobs[0], n_back = a.n_jobs[1];
  int job_i = unit / (kStrips * 3);
  const int strip = (unit % (kStrips * 3)) / 3, root = unit % 3;
  if (job_i >= n_front + n_back) return;
  if (job_i >= n_front) job_i = a.jobs_cap - 1 - (job_i - n_front);
  const ScanJob *jp = a.jobs + job_i;
  if (uniform_i32(jp->n_seq) <= 0 || uniform_i32(jp->cached) == 0) return;
  const int my_seq = uniform_i32(jp->seq_of_root[root]);
  if (my_seq < 0) return;
  const ScanSeq sq = uniform_seq(jp->seq[my_seq]);
  const int n_w = seq_windows(sq);
  const int chain = uniform_i32(jp->channel) * 3 + root;
  ChainState *st = a.state + chain;
  const int sel = st->ema_sel;
  // this wave's segment [seg_start, seg_end)
  int seg_start, seg_end;
  {
    const float q = 1.f - a.light;
    float pw = 1.f, q0 = 1.f, q1 = 1.f;
    for (int i = 0; i < S; ++i) {
      if (i == seg) q0 = pw;
      pw *= q;
      if (i == seg) q1 = pw;
    }
    const float scale = (float)n_w / (1.f - pw);
    seg_start = seg == 0 ? 0 : (int)(scale * (1.f - q0) + 0.5f);
    seg_end = seg == S - 1 ? n_w : (int)(scale * (1.f - q1) + 0.5f);
    seg_start = uniform_i32(seg_start < n_w ? seg_start : n_w);
    seg_end = uniform_i32(seg_end < n_w ? seg_end : n_w);
  }
  if (seg_end <= seg_start && !(n_w == 0 && seg == 0)) return;
  // the average restarts where tracking starts (srslte_pss_reset): nothing in front of that window matters behind it
  const int k_reset = sq.n_c > 0 ? sq.n_a + sq.n_b : -1;
  const int k0 = (k_reset >= 0 && seg_start >= k_reset) ? k_reset : 0;
  ema_t ema;
  if (sq.zero || k0 > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) ema[r] = r < kStripRegs ? 0.f : -FLT_MAX;
  } else {
    strip_ema_load(st->avg[sel], strip, lane, ema);
  }
  StripHead *heads = a.heads + (int64_t)chain * kScanMaxRound * kStrips + strip;
  float *rows = a.rows + ((int64_t)chain * kScanMaxRound * kStrips + strip) * kRowLags;
  const float *m = a.cache + (size_t)uniform_i32(jp->slot_base[root]) * (kStrips * 64 * kStripRegs) + (size_t)strip * (64 * kStripRegs) + lane;
  float buf[P][kStripRegs];
#pragma unroll
  for (int p = 0; p < P; ++p)
    if (k0 + p < seg_end) {
      const float *mn = m + (size_t)(k0 + p) * (kStrips * 64 * kStripRegs);
#pragma unroll
      for (int r = 0; r < kStripRegs; ++r) buf[p][r] = mn[64 * r];
    }
  for (int k = k0; k < seg_end; k += P) {
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int kk = k + p;
      if (kk < seg_end) {
        float met[kStripRegs];
#pragma unroll
        for (int r = 0; r < kStripRegs; ++r) met[r] = buf[p][r];
        if (kk + P < seg_end) {
          const float *mn = m + (size_t)(kk + P) * (kStrips * 64 * kStripRegs);
#pragma unroll
          for (int r = 0; r < kStripRegs; ++r) buf[p][r] = mn[64 * r];
        }
        if (kk == k_reset) {
          asm volatile("; the average cleared" ::: "memory");
#pragma unroll
          for (int r = 0; r < kStripRegs; ++r) ema[r] = 0.f;
        }
        strip_ema_apply(strip, ln, met, ema);
        if (kk >= seg_start) {
          if (kk < sq.n_a) {
            if (kk == sq.n_a - 1) strip_ema_store(st->avg[(sel + 1) % 3], strip, ln, ema);
          } else {
            StripHead hd;
            float row;
            int idx;
            strip_head_row(ema, strip, ln, hd, row, idx);
            if (idx >= 0) rows[(int64_t)(kk - sq.n_a) * (kStrips * kRowLags) + idx] = row;
            if (lane == 0) heads[(int64_t)(kk - sq.n_a) * kStrips] = hd;
          }
        }
      }
    }
  }
  if (seg_end == n_w) strip_ema_store(st->avg[(sel + 2) % 3], strip, lane, ema);
}

// ---------------------------------------------------------------- plan: the jobs of the next round
// One thread per channel.  A chain can be served when its next search is at a position that is known now
// (always: the calls between two searches of a tracked chain neither search nor move it off its timing,
// lib/pss_impl.cc:170-172,185-186) and that search may start in this chunk.  Chains of a channel that will search the
// same windows from a plainly stored average share a job (one forward transform per window for all of them).
struct PlanArgs {
  ChainState *state;
  ScanJob *jobs;     // [jobs_cap]
  int *n_jobs;       // [2]: jobs allocated from the front (long) / from the back (short); zero when the round is planned
  int jobs_cap;
  int *rec_count;    // [n_channels*3]: zeroed in the first round
  int n_channels;
  int rec_cap;
  int64_t t1;        // absolute end of the chunk
  int flush;
  int enable;        // 0: no chain goes to the scan (lt_batch_set_mode 0)
  int round_windows; // windows per job in this round
  int first_round;
  int track_after;
  int track_every;
  int call_reach;    // kCallReach (lt_chain_kernel.h)
  // first launch of a run only (first_round): counters to clear behind n_jobs, the one to leave alone, positions to 0
  int n_counters, keep_counter, rebase;
  // window cache for the scan rounds (small batches): every job's windows get slots, lt_metric_kernel fills them
  // before the round's scan.  round_slots counts the slots handed out in this round, from 0 in every round (never past
  // slot_cap).
  int cached;
  int2 *slot_owner;   // [slot_cap]: (chain, window index in its sequence)
  int64_t *slot_pos;  // [slot_cap]: the window's position
  int *slot_next;
  int *round_slots;
  int slot_cap;
};

// floor(d / step) for 0 <= d < 2^52, step > 0: through a double-precision quotient and a correction by one (a 64-bit
// integer division is a hundred dependent instructions, and the plan runs on one thread per channel)
__device__ __forceinline__ int64_t div_floor_pos(int64_t d, int64_t step)
{
  int64_t q = (int64_t)((double)d / (double)step);
  if (q * step > d) --q;
  else if ((q + 1) * step <= d) ++q;
  return q;
}

// the job of channel c for the next round (null: none)
__device__ __forceinline__ ScanJob *plan_channel(const PlanArgs &a, int c)
{
  int n[3], stride[3], n_c[3], fresh[3];
  int64_t first[3], pos_c[3];
  bool plain[3]; // the average is a stored state as it stands (avg[ema_sel], or the zeros a stream starts from), and the windows end with the round
  const int64_t limit = a.t1 - (a.flush ? kHalfFrame : a.call_reach); // last position a call may start at
#pragma unroll
  for (int r = 0; r < 3; ++r) { // (unrolled: the three chains' state comes in together, not chain after chain)
    ChainState *st = a.state + c * 3 + r;
    if (a.first_round) {
      a.rec_count[c * 3 + r] = 0;
      st->run_done = 0;
      st->need_chain = 0;
      st->listed = 0;
      st->cache_n = 0;
    }
    st->scan_job = -1;
    n[r] = 0;
    n_c[r] = 0;
    pos_c[r] = 0;
    stride[r] = st->is_tracking ? a.track_every + 1 : 1;
    // the calls before the next search emit without searching: timer of them (lib/pss_impl.cc:170-172)
    first[r] = st->pos + (st->is_tracking ? (int64_t)st->timer * kHalfFrame : 0);
    fresh[r] = st->ema_zero == 2; // (lt_batch_reset: zeros that are no srslte_pss_reset)
    plain[r] = st->pre_n == 0 && (st->ema_zero == 0 || fresh[r]);
    if (!a.enable || st->run_done || st->need_chain || st->tracking_lost) continue;
    if (first[r] > limit) continue;
    int64_t m = div_floor_pos(limit - first[r], (int64_t)stride[r] * kHalfFrame) + 1;
    const int64_t room = (a.rec_cap - a.rec_count[c * 3 + r]) / stride[r]; // a window makes at most `stride` records
    if (m > room) m = room;
    if (m > a.round_windows) m = a.round_windows;
    // a chain that is building its score leaves these windows when tracking starts, track_after - score hits on
    // at the latest (lib/pss_impl.cc:119-126): nothing beyond that window is worth scanning
    if (!st->is_tracking && st->score > 0 && m > a.track_after - st->score) m = a.track_after - st->score;
    // Speculation is wasted on a chain that is about to leave its windows again.  One that has just re-aligned
    // (its average still holds the old peak beside the new one: the next search often misses and resets it,
    // lib/pss_impl.cc:134-151) gets two windows; one whose average has just been reset tends to fire at once and
    // to start tracking track_after windows later, so it gets no more than that.
    if (!st->is_tracking && st->pre_n > 0 && m > 2) m = 2;
    // ... and, if it does, to start tracking track_after windows later: no more than those here.  (ema_zero is set
    // = 1 by srslte_pss_reset only: a stream starts from avg[0] = zeros or, after lt_batch_reset, from the value 2.)
    const bool after_reset = !st->is_tracking && st->score == 0 && st->ema_zero == 1;
    if (after_reset && m > a.track_after) m = a.track_after;
    if (m > kScanMaxRound - st->pre_n) m = kScanMaxRound - st->pre_n;
    n[r] = m > 0 ? (int)m : 0;
    // A chain that is building its score on windows of its own keeps its position while it hits (the peak sits
    // where the last hit put it), starts tracking with the hit that takes the score to track_after, clears its
    // average there and searches every (track_every + 1)-th half-frame from then on: those searches ride along
    // in the same round.  (If a window misses, the resolver stops following the sequence there.)
    const int hits = a.track_after - st->score;
    if (!st->is_tracking && (st->score > 0 || after_reset) && st->pre_n == 0 && n[r] == hits && hits > 0) {
      const int sc = a.track_every + 1;
      const int64_t pc = first[r] + (int64_t)(hits - 1) * kHalfFrame + (int64_t)sc * kHalfFrame;
      if (pc <= limit) {
        int64_t mc = div_floor_pos(limit - pc, (int64_t)sc * kHalfFrame) + 1;
        const int64_t room_c = (a.rec_cap - a.rec_count[c * 3 + r] - hits - a.track_every) / sc;
        if (mc > room_c) mc = room_c;
        if (mc > a.round_windows - n[r]) mc = a.round_windows - n[r];
        if (mc > kScanMaxRound - n[r]) mc = kScanMaxRound - n[r];
        if (mc > 0) {
          n_c[r] = (int)mc;
          pos_c[r] = pc;
          plain[r] = false;
        }
      }
    }
  }
  // Group the chains into sequences: lead[r] = the first chain of r's group.  (Everything below is written with
  // constant indices after unrolling: a job assembled in a dynamically indexed local array would live in scratch.)
  int lead[3], len[3];
  ScanSeq q[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    lead[r] = n[r] > 0 ? r : -1;
#pragma unroll
    for (int p = 0; p < r; ++p)
      if (lead[r] == r && lead[p] == p && plain[p] && plain[r] && fresh[p] == fresh[r] && first[p] == first[r] && stride[p] == stride[r]) lead[r] = p;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const ChainState *st = a.state + c * 3 + r;
    q[r].pos_a = st->pre_pos; q[r].stride_a = st->pre_stride; q[r].n_a = st->pre_n;
    q[r].pos_b = first[r]; q[r].stride_b = stride[r]; q[r].n_b = n[r];
    q[r].pos_c = pos_c[r]; q[r].stride_c = a.track_every + 1; q[r].n_c = n_c[r];
    q[r].roots = 0;
    q[r].zero = st->ema_zero;
#pragma unroll
    for (int p = r; p < 3; ++p)
      if (lead[p] == r) {
        q[r].roots |= 1 << p;
        if (n[p] < q[r].n_b) q[r].n_b = n[p];
      }
    len[r] = lead[r] == r ? q[r].n_a + q[r].n_b + q[r].n_c : 0;
  }
  const int longest = len[0] > len[1] ? (len[0] > len[2] ? len[0] : len[2]) : (len[1] > len[2] ? len[1] : len[2]);
  if (longest == 0) return nullptr;
  // slot of a leader = the number of leaders that come before it, the longest sequence first (the forward wave
  // prefetches for it and loops over its windows)
  int slot[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    slot[r] = 0;
#pragma unroll
    for (int p = 0; p < 3; ++p)
      if (p != r && len[p] > 0 && (len[p] > len[r] || (len[p] == len[r] && p < r))) ++slot[r];
  }
  const bool is_long = longest > a.round_windows / 2;
  const int j = is_long ? atomicAdd(a.n_jobs, 1) : a.jobs_cap - 1 - atomicAdd(a.n_jobs + 1, 1);
  ScanJob *job = a.jobs + j;
  int n_seq = 0;
#pragma unroll
  for (int r = 0; r < 3; ++r)
    if (len[r] > 0) {
      job->seq[slot[r]] = q[r];
      ++n_seq;
    }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    int sq = -1;
#pragma unroll
    for (int p = 0; p < 3; ++p)
      if (lead[r] == p) sq = slot[p];
    job->seq_of_root[r] = sq;
    if (sq >= 0) a.state[c * 3 + r].scan_job = j;
  }
  for (int i = n_seq; i < 3; ++i) { // (unused slots: read by the scan kernel's forward wave, never followed)
    ScanSeq z = {};
    job->seq[i] = z;
  }
  job->channel = c;
  job->n_seq = n_seq;
  job->pad = 0;
  // window cache: slots for the windows of every chain the job serves, or none at all
  int total = 0, base_r[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    base_r[r] = total;
    if (lead[r] >= 0) total += len[lead[r]];
  }
  int base = -1;
  if (a.cached && total > 0) {
    // (from the round's own counter: a round's slots live until its scan, the next round starts at slot 0 again -- a long
    // capture does not run out of them.  The counter only ever counts slots that were handed out -- lt_metric_kernel takes
    // its share of the work from it, and a slot nobody owns has no owner to read --: a job that does not fit leaves it
    // alone and stays uncached; the slots in use are always the first ones.)
    int seen = __hip_atomic_load(a.round_slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (seen + total <= a.slot_cap) {
      const int was = atomicCAS(a.round_slots, seen, seen + total);
      if (was == seen) {
        base = seen;
        break;
      }
      seen = was;
    }
  }
  job->cached = base >= 0;
#pragma unroll
  for (int r = 0; r < 3; ++r) job->slot_base[r] = (base >= 0 && lead[r] >= 0) ? base + base_r[r] : -1;
  return job;
}

// The slots of a cached job: who owns each and which window it stands for (all threads of the workgroup that planned
// the job, after a barrier behind the plan).
__device__ __forceinline__ void fill_job_slots(const PlanArgs &a, const ScanJob *job, int tid, int nthreads)
{
  if (!job->cached) return;
  for (int r = 0; r < 3; ++r) {
    const int sq = job->seq_of_root[r];
    if (sq < 0) continue;
    const ScanSeq q = job->seq[sq];
    const int n = seq_windows(q), base = job->slot_base[r];
    for (int k = tid; k < n; k += nthreads) {
      a.slot_owner[base + k] = make_int2(job->channel * 3 + r, k);
      a.slot_pos[base + k] = seq_window_pos(q, k);
    }
  }
}

// lt_batch_reset: the chains back to their constructor state -- everything in front of the three average buffers; those are
// not cleared but declared zero (ema_zero = 2).  The members only the per-record kernels touch (the CP averages; the CFO ring
// is the per-chain kernel's) are parts bit 1, the rest bit 0 (with lt_batch_set_overlap the two are cleared on different streams).
__device__ __forceinline__ void reset_chain_states(ChainState *state, int n_chains, int parts, int tid, int nthreads)
{
  constexpr int n = (int)(offsetof(ChainState, avg) / 4);
  constexpr int i_norm = (int)(offsetof(ChainState, m_norm_avg) / 4), i_ext = (int)(offsetof(ChainState, m_ext_avg) / 4);
  constexpr int i_zero = (int)(offsetof(ChainState, ema_zero) / 4);
  static_assert(offsetof(ChainState, avg) % 4 == 0, "ChainState is cleared word by word");
  for (int c = 0; c < n_chains; ++c) {
    uint32_t *w = reinterpret_cast<uint32_t *>(state + c);
    for (int i = tid; i < n; i += nthreads) {
      const bool rec = i == i_norm || i == i_ext;
      if (parts & (rec ? 2 : 1)) w[i] = i == i_zero ? 2u : 0u;
    }
  }
}

// The run's first launch: one workgroup.  It first brings the run's counters to zero (the job counters of every round
// and this run's work-list counter; `keep` is the other run's, which its per-record kernels may still be using) and,
// for a run that is a capture of its own, the chains' positions; then it plans the first round.
__global__ __launch_bounds__(256) void lt_plan_kernel(PlanArgs a)
{
  if (a.first_round) {
    for (int i = threadIdx.x; i < a.n_counters; i += 256)
      if (i != a.keep_counter) a.n_jobs[i] = 0;
    if (a.rebase)
      for (int i = threadIdx.x; i < a.n_channels * 3; i += 256) a.state[i].pos = 0;
    // (one workgroup: what it has just stored is read back -- by other threads, and through atomics that execute in the
    // L2 -- by this CU only, whose L1 held none of those lines when the kernel started: the stores have to be in the L2,
    // which is what a workgroup-scope fence waits for.  The agent-scope fence that stood here also wrote the XCD's L2
    // back and invalidated it: 3 us of the run's first launch.)
    __threadfence_block();
    __syncthreads();
  }
  __shared__ ScanJob *planned[256];
  for (int c0 = 0; c0 < a.n_channels; c0 += 256) {
    const int c = c0 + threadIdx.x;
    planned[threadIdx.x] = c < a.n_channels ? plan_channel(a, c) : nullptr;
    if (!a.cached) continue;
    __threadfence_block();
    __syncthreads();
    for (int i = 0; i < 256 && c0 + i < a.n_channels; ++i)
      if (planned[i]) fill_job_slots(a, planned[i], threadIdx.x, 256);
    __syncthreads();
  }
}

// ---------------------------------------------------------------- resolve: general_work on the scanned windows
// One wave per chain.  Lane k combines the 12 partials of the job's window k (resolve_window); then the calls run
// in order, wave-uniform: the search result of window k goes through lib/pss_impl.cc:163-223 exactly as the
// per-chain kernel does it (score, tracking, alignment, one record per call), followed by the calls that do not
// search.  The chain stays in the job as long as its next search is the job's next window and its average is
// the one the scan carried on; it leaves it when a hit re-aligns it, when tracking starts or is lost
// (srslte_pss_reset: the average restarts from zero) or when nothing more fits into the chunk.
struct ResolveArgs {
  ChainState *state;
  const ScanJob *jobs;
  const StripHead *heads; // (ScanArgs)
  const float *rows;
  lt_record *recs;
  int rec_cap;
  int *rec_count;
  int64_t t1;
  int flush;
  int call_reach;
  float psr_threshold;
  int track_after;
  int track_every;
  PlanArgs next;      // the plan of the next round, made by the same launch (next.jobs / next.n_jobs: the other buffers)
  int plan_next;
  // the run's LAST resolve launch (plan_next == 0) with the window cache on: every chain's request for the cache -- what pass 0
  // of lt_chain_kernel would ask for -- is made here, from the state the rounds leave, so that the per-chain kernel runs once,
  // behind lt_metric_kernel (one launch fewer per run; levels 1, 2 and 4 -- level 3 keeps its host cue and its pass 0)
  int ask;            // 0 = no requests here
  int ask_stream;     // level 4: untracked chains ask for their lag-metric stream
  int slot_quota;
  int2 *slot_owner;
  int64_t *slot_pos;
  int *slot_next;     // (+3: chains that asked, diagnostics)
  int *launch_slots;
};

__device__ __forceinline__ void resolve_chain(const ResolveArgs &a, int chain, int lane)
{
  ChainState *st = a.state + chain;
  if (st->scan_job < 0) return;
  const int channel = chain / 3, n_id_2 = chain % 3;
  const ScanSeq job = a.jobs[st->scan_job].seq[a.jobs[st->scan_job].seq_of_root[n_id_2]]; // (the chain's window sequence)
  lt_record *recs = a.recs + (int64_t)chain * a.rec_cap;
  int64_t pos = st->pos;
  int is_tracking = st->is_tracking, score = st->score, timer = st->timer, tracking_lost = st->tracking_lost;
  int peak_pos = st->peak_pos;
  uint32_t psr_i = st->psr_i;
  float psr = st->psr, psr_max = st->psr_max;
  int nrec = a.rec_count[chain];
  const int call_base = st->n_calls - nrec;
  int n_emits = 0;
  const int64_t limit = a.t1 - (a.flush ? kHalfFrame : a.call_reach);
  auto put = [&](int64_t in_pos, int64_t out_pos, uint32_t flags, int used_peak, float pv) {
    if (lane == 0) {
      lt_record r;
      r.in_pos = in_pos; r.out_pos = out_pos; r.channel = channel; r.n_id_2 = n_id_2;
      r.call_idx = call_base + nrec; r.flags = flags; r.peak_pos = used_peak; r.score = score;
      r.psr = psr; r.peak_value = pv; r.cfo = 0.f; r.mean_cfo = 0.f;
      r.m_norm_avg = 0.f; r.m_ext_avg = 0.f; r.m0_value = 0.f;
      r.m1_value = 0.f; r.m0 = -1; r.m1 = -1; r.n_id_1 = -1; r.cell_id = -1;
      r.psr_max = psr_max; r.timer = timer;
      recs[nrec] = r;
    }
    nrec++;
  };
  // the calls of a tracked chain between two searches (lib/pss_impl.cc:170-172, 184-204 with frame_start = 0)
  auto run_timer_calls = [&]() {
    if (!(is_tracking && timer > 0)) return;
    int nb = timer; // one record per call, one lane each (timer <= track_every; more than 64 take several turns)
    if (pos > limit) nb = 0;
    else if ((limit - pos) / kHalfFrame + 1 < nb) nb = (int)((limit - pos) / kHalfFrame + 1);
    if (a.rec_cap - nrec < nb) nb = a.rec_cap - nrec;
    for (int idx = lane; idx < nb; idx += 64) {
      const int64_t p = pos + (int64_t)idx * kHalfFrame;
      lt_record r;
      r.in_pos = p; r.out_pos = p; r.channel = channel; r.n_id_2 = n_id_2;
      r.call_idx = call_base + nrec + idx; r.flags = LT_F_OVER | LT_F_EMITTED | LT_F_SSS_RAN | LT_F_TRACKING;
      r.peak_pos = kSlotLen; r.score = score;
      r.psr = psr; r.peak_value = 0.f; r.cfo = 0.f; r.mean_cfo = 0.f;
      r.m_norm_avg = 0.f; r.m_ext_avg = 0.f; r.m0_value = 0.f;
      r.m1_value = 0.f; r.m0 = -1; r.m1 = -1; r.n_id_1 = -1; r.cell_id = -1;
      r.psr_max = psr_max; r.timer = timer - 1 - idx;
      recs[nrec + idx] = r;
    }
    timer -= nb;
    nrec += nb;
    n_emits += nb;
    pos += (int64_t)nb * kHalfFrame;
  };

  int used = 0;          // windows of the job whose search has been executed
  bool ema_reset = false, stop_run = false, need_chain = false;
  run_timer_calls();
  // The round's windows, 64 at a time: every window's search result, one per lane, then the calls in order.
  const int n_bc = job.n_b + job.n_c;
  int track_k = -1; // the window whose hit started tracking (and cleared the average)
  for (int blk = 0; blk < n_bc; blk += 64) {
    const int nb = n_bc - blk < 64 ? n_bc - blk : 64;
    WindowResult wr;
    wr.peak_pos = 0; wr.peak_value = 0.f; wr.psr = 0.f; wr.exact = true;
    if (lane < nb) {
      const int64_t wi = ((int64_t)chain * kScanMaxRound + blk + lane) * kStrips;
      wr = resolve_window_deferred(a.heads + wi, a.rows + wi * kRowLags);
    }
    // ---- an idle chain (lib/pss_impl.cc:134-135: a miss with score 0 changes nothing but d_psr, the ring and the
    // position): all the misses up to the first window that fires, one lane each
    int k0 = 0;
    if (!ema_reset && !is_tracking && score == 0 && !tracking_lost && job.stride_b == 1 && blk < job.n_b && pos == job.pos_b + (int64_t)blk * kHalfFrame) {
      int fit = job.n_b - blk < nb ? job.n_b - blk : nb;
      if (pos > limit) fit = 0;
      else if ((limit - pos) / kHalfFrame + 1 < fit) fit = (int)((limit - pos) / kHalfFrame + 1);
      if (a.rec_cap - nrec < fit) fit = a.rec_cap - nrec;
      const bool valid = lane < fit;
      const bool fire = valid && (!wr.exact || wr.psr > a.psr_threshold);
      const unsigned long long fm = __ballot(fire);
      const int n_miss = fm ? __builtin_ctzll(fm) : fit;
      if (n_miss > 0) {
        const bool miss = lane < n_miss;
        float pm = miss ? wr.psr : -FLT_MAX; // d_psr_max as an inclusive prefix maximum
        if (!(pm == pm)) pm = -FLT_MAX;      // a NaN never raises it
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const float o = __shfl_up(pm, d);
          if (lane >= d) pm = fmaxf(pm, o);
        }
        const float my_max = fmaxf(psr_max, pm);
        if (miss) {
          lt_record r;
          r.in_pos = pos + (int64_t)lane * kHalfFrame; r.out_pos = -1; r.channel = channel; r.n_id_2 = n_id_2;
          r.call_idx = call_base + nrec + lane; r.flags = LT_F_SEARCHED; r.peak_pos = wr.peak_pos; r.score = 0;
          r.psr = wr.psr; r.peak_value = wr.peak_value; r.cfo = 0.f; r.mean_cfo = 0.f;
          r.m_norm_avg = 0.f; r.m_ext_avg = 0.f; r.m0_value = 0.f;
          r.m1_value = 0.f; r.m0 = -1; r.m1 = -1; r.n_id_1 = -1; r.cell_id = -1;
          r.psr_max = my_max; r.timer = a.track_every;
          recs[nrec + lane] = r;
          st->psr_data[(psr_i + (uint32_t)lane) % kMovingAvgSz] = wr.psr; // d_psr_data ring (lib/pss_impl.cc:166)
        }
        psr_max = rlane(my_max, n_miss - 1); // (wave-uniform lane: v_readlane, not a trip through the LDS)
        psr = rlane(wr.psr, n_miss - 1);
        peak_pos = __builtin_amdgcn_readlane(wr.peak_pos, n_miss - 1);
        timer = a.track_every;
        psr_i += (uint32_t)n_miss;
        nrec += n_miss;
        pos += (int64_t)n_miss * kHalfFrame;
        used = blk + n_miss;
        k0 = n_miss;
      }
    }
    bool off = false; // the chain has left the job's windows (or the run): nothing more of this round is its
    for (int k = k0; k < nb; ++k) {
      const int gk = blk + k;
      const bool in_c = gk >= job.n_b;
      const int64_t pk = in_c ? job.pos_c + (int64_t)(gk - job.n_b) * job.stride_c * kHalfFrame : job.pos_b + (int64_t)gk * job.stride_b * kHalfFrame;
      off = true;
      // The sequence's tracked part is entered through the hit that starts tracking, on its last window before:
      // the scan has cleared its average there, as srslte_pss_reset did the chain's.  Any other way in (tracking
      // started a window late, or -- track_every = 0, the strides are the same -- not at all) the two differ.
      const bool enter_c = in_c && gk == job.n_b;
      if (enter_c && !(ema_reset && track_k == gk - 1)) break;
      if ((ema_reset && !enter_c) || pos != pk || (is_tracking ? a.track_every + 1 : 1) != (in_c ? job.stride_c : job.stride_b)) break; // off the job's windows
      if (is_tracking && timer != 0) break; // (the calls before the search did not all fit)
      if (!(pos <= limit && nrec < a.rec_cap)) {
        stop_run = true;
        break;
      }
      if (!__builtin_amdgcn_readlane((int)wr.exact, k)) { // a lobe the strips could not close: the per-chain kernel searches this window
        need_chain = true;
        break;
      }
      // ---- a chain building its score (lib/pss_impl.cc:111-127, 184-214): hits that leave the peak where it is,
      // short of the one that starts tracking -- one lane each: the score goes up by one per window, every half-frame
      // is emitted with the tracking_lost tag (quirk q2).
      if (!is_tracking && !tracking_lost && !ema_reset && !in_c && score + 1 < a.track_after) {
        const int seg_end = job.n_b - blk < nb ? job.n_b - blk : nb;
        const int t = lane - k; // this lane's window, counted from k
        const int64_t my_pos = pos + (int64_t)t * kHalfFrame;
        const bool mine = lane >= k && lane < seg_end && score + t + 1 < a.track_after;
        const bool fits = mine && my_pos <= limit && my_pos + kHalfFrame <= a.t1 && nrec + t + 1 <= a.rec_cap;
        const bool hit = fits && wr.exact && wr.psr > a.psr_threshold && wr.peak_pos == kSlotLen;
        const unsigned long long bad = ~__ballot(hit) & (~0ull << k);
        const int m = (bad ? __builtin_ctzll(bad) : 64) - k; // windows k .. k + m - 1
        if (m > 1) {
          const bool on = lane >= k && lane < k + m;
          float pm = on ? wr.psr : -FLT_MAX; // d_psr_max as an inclusive prefix maximum
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) {
            const float o = __shfl_up(pm, d);
            if (lane >= d) pm = fmaxf(pm, o);
          }
          const float my_max = fmaxf(psr_max, pm);
          if (on) {
            lt_record r;
            r.in_pos = my_pos; r.out_pos = my_pos; r.channel = channel; r.n_id_2 = n_id_2;
            r.call_idx = call_base + nrec + t;
            r.flags = LT_F_SEARCHED | LT_F_OVER | LT_F_EMITTED | LT_F_TAG_LOST;
            r.peak_pos = kSlotLen; r.score = score + t + 1; r.psr = wr.psr; r.peak_value = wr.peak_value;
            r.cfo = 0.f; r.mean_cfo = 0.f; r.m_norm_avg = 0.f; r.m_ext_avg = 0.f; r.m0_value = 0.f; r.m1_value = 0.f;
            r.m0 = -1; r.m1 = -1; r.n_id_1 = -1; r.cell_id = -1; r.psr_max = my_max; r.timer = a.track_every;
            recs[nrec + t] = r;
            st->psr_data[(psr_i + (uint32_t)t) % kMovingAvgSz] = wr.psr;
          }
          psr_max = rlane(my_max, k + m - 1);
          psr = rlane(wr.psr, k + m - 1);
          peak_pos = kSlotLen;
          timer = a.track_every;
          score += m;
          psr_i += (uint32_t)m;
          nrec += m;
          n_emits += m;
          pos += (int64_t)m * kHalfFrame;
          used = blk + k + m;
          k += m - 1;
          off = false;
          continue;
        }
      }
      // ---- a tracked chain in its steady state (score at track_after, every search a hit that leaves the peak where
      // it is): the searches up to the first window that is anything else, one lane each -- the search's record and
      // those of the track_every calls behind it (lib/pss_impl.cc:163-204 with frame_start = 0).
      if (is_tracking && score == a.track_after && !tracking_lost && !ema_reset && !enter_c) {
        const int S = a.track_every + 1;                  // calls (and records) per window
        const int seg_end = in_c ? nb : (job.n_b - blk < nb ? job.n_b - blk : nb); // same part of the sequence as window k
        const int t = lane - k;                           // this lane's window, counted from k
        const int64_t my_pos = pos + (int64_t)t * S * kHalfFrame;
        const bool mine = lane >= k && lane < seg_end;
        const bool fits = mine && my_pos + (int64_t)a.track_every * kHalfFrame <= limit && my_pos + kHalfFrame <= a.t1 &&
                          nrec + (int64_t)(t + 1) * S <= a.rec_cap;
        const bool steady = fits && wr.exact && wr.psr > a.psr_threshold && wr.peak_pos == kSlotLen;
        const unsigned long long bad = ~__ballot(steady) & (~0ull << k);
        const int m = (bad ? __builtin_ctzll(bad) : 64) - k; // windows k .. k + m - 1
        if (m > 1) {
          const bool on = lane >= k && lane < k + m;
          float pm = on ? wr.psr : -FLT_MAX; // d_psr_max as an inclusive prefix maximum
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) {
            const float o = __shfl_up(pm, d);
            if (lane >= d) pm = fmaxf(pm, o);
          }
          const float my_max = fmaxf(psr_max, pm);
          if (on) {
            lt_record r;
            r.channel = channel; r.n_id_2 = n_id_2; r.score = score; r.psr = wr.psr; r.cfo = 0.f; r.mean_cfo = 0.f;
            r.m_norm_avg = 0.f; r.m_ext_avg = 0.f; r.m0_value = 0.f; r.m1_value = 0.f; r.m0 = -1; r.m1 = -1; r.n_id_1 = -1;
            r.cell_id = -1; r.psr_max = my_max; r.peak_pos = kSlotLen;
            const int base = nrec + t * S;
            r.in_pos = my_pos; r.out_pos = my_pos; r.call_idx = call_base + base;
            r.flags = LT_F_SEARCHED | LT_F_OVER | LT_F_EMITTED | LT_F_SSS_RAN | LT_F_TRACKING;
            r.peak_value = wr.peak_value; r.timer = a.track_every;
            recs[base] = r;
            r.flags = LT_F_OVER | LT_F_EMITTED | LT_F_SSS_RAN | LT_F_TRACKING;
            r.peak_value = 0.f;
            for (int i = 0; i < a.track_every; ++i) {
              r.in_pos = r.out_pos = my_pos + (int64_t)(1 + i) * kHalfFrame;
              r.call_idx = call_base + base + 1 + i;
              r.timer = a.track_every - 1 - i;
              recs[base + 1 + i] = r;
            }
            st->psr_data[(psr_i + (uint32_t)t) % kMovingAvgSz] = wr.psr;
          }
          psr_max = rlane(my_max, k + m - 1);
          psr = rlane(wr.psr, k + m - 1);
          peak_pos = kSlotLen;
          timer = 0;
          psr_i += (uint32_t)m;
          nrec += m * S;
          n_emits += m * S;
          pos += (int64_t)m * S * kHalfFrame;
          used = blk + k + m;
          k += m - 1;
          off = false;
          continue;
        }
      }
      used = blk + k + 1;
      if (enter_c) ema_reset = false; // (the window is the chain's: from here on the scan's average is its own again)
      // ---- lib/pss_impl.cc:163-172: the search
      uint32_t flags = LT_F_SEARCHED;
      timer = a.track_every;
      peak_pos = __builtin_amdgcn_readlane(wr.peak_pos, k);
      psr = rlane(wr.psr, k);
      const float peak_value = rlane(wr.peak_value, k);
      if (lane == 0) st->psr_data[psr_i % kMovingAvgSz] = psr;
      psr_i++;
      const int used_peak_pos = peak_pos;
      // ---- score, lib/pss_impl.cc:174-182 (incr_score :111-127, reset_score :129-152)
      const bool over = psr > a.psr_threshold;
      if (over) {
        if (!(is_tracking && score == a.track_after)) {
          score++;
          if (!is_tracking && score == a.track_after) {
            is_tracking = 1;
            ema_reset = true; // srslte_pss_reset
            track_k = gk;
          }
        }
        flags |= LT_F_OVER;
      } else if (score != 0) {
        score = 0;
        timer = 0;
        is_tracking = 0;
        ema_reset = true;
        psr_i = 0;
        tracking_lost = 1;
      }
      if (psr > psr_max) psr_max = psr;
      int64_t out_pos = -1, next_pos;
      if (over || tracking_lost) { // ---- lib/pss_impl.cc:184-214
        const int frame_start = peak_pos - kSlotLen;
        peak_pos = kSlotLen;
        const int64_t hf0 = pos + frame_start;
        if (hf0 + kHalfFrame > a.t1) { // (flush only) quirk q1 normalised: never read past the capture
          stop_run = true;
          break;
        }
        next_pos = hf0 + kHalfFrame;
        out_pos = hf0;
        flags |= LT_F_EMITTED;
        if (is_tracking) {
          flags |= LT_F_SSS_RAN;
        } else {
          flags |= LT_F_TAG_LOST;
          tracking_lost = 0;
        }
        n_emits++;
      } else {
        next_pos = pos + kHalfFrame; // lib/pss_impl.cc:217
      }
      if (is_tracking) flags |= LT_F_TRACKING;
      put(pos, out_pos, flags, used_peak_pos, peak_value);
      pos = next_pos;
      run_timer_calls();
      off = false;
    }
    if (off) break;
  }
  if (lane == 0) {
    st->pos = pos;
    st->is_tracking = is_tracking; st->score = score; st->timer = timer;
    st->tracking_lost = tracking_lost; st->peak_pos = peak_pos;
    st->psr_i = psr_i;
    st->n_calls = call_base + nrec; st->n_emits += n_emits;
    st->psr = psr; st->psr_max = psr_max;
    a.rec_count[chain] = nrec;
    if (stop_run) st->run_done = 1;
    if (need_chain) st->need_chain = 1;
    // ---- where the average stands now
    const int sel = st->ema_sel;
    if (ema_reset) { // zeros, whatever the scan went on to compute
      st->ema_zero = 1;
      st->pre_n = 0;
    } else if (used == n_bc) { // all of the round's windows: the scan's final state is the chain's
      st->ema_sel = (sel + 2) % 3;
      st->ema_zero = 0;
      st->pre_n = 0;
    } else if (used > job.n_b) { // into the tracked part: zeros run through its first windows
      st->ema_zero = 1;
      st->pre_pos = job.pos_c;
      st->pre_stride = job.stride_c;
      st->pre_n = used - job.n_b;
    } else if (used > 0) { // the first `used` windows only: from the state behind the prefix, run through them again
      if (job.n_a > 0) {
        st->ema_sel = (sel + 1) % 3;
        st->ema_zero = 0;
      }
      st->pre_pos = job.pos_b;
      st->pre_stride = job.stride_b;
      st->pre_n = used;
    } // (used == 0: nothing consumed, the chain's average is described as before)
  }
}

// What pass 0 of lt_chain_kernel asks of the window cache (lt_chain_kernel.h: chain_run, "pass 0 of two"), by the chain's wave
// of the last resolve launch: a chain with kCacheMinSearches or more searches ahead of it -- all on the grid it stands on,
// unless a hit moves it -- lists its coming windows (every window from pos for an idle or acquiring chain, every
// (track_every + 1)-th for a tracked one) or, at level 4 and untracked, its lag-metric stream; the slots are its own
// (chain * slot_quota + k), the work list's order is whatever the atomics make it.  Same arithmetic as pass 0: same requests.
__device__ __forceinline__ void chain_request_cache(const ResolveArgs &a, int chain, int lane)
{
  ChainState *st = a.state + chain;
  // (lane 0 reads what a lane 0 has written -- this wave's a moment ago, or an earlier launch's -- and hands it round)
  int done = 0, tracking = 0, timer = 0, pre_n = 0, nrec = 0;
  int64_t pos = 0, pre_pos = 0;
  if (lane == 0) {
    done = st->run_done; tracking = st->is_tracking; timer = st->timer; pre_n = st->pre_n; nrec = a.rec_count[chain];
    pos = st->pos; pre_pos = st->pre_pos;
  }
  done = uniform_i32(done); tracking = uniform_i32(tracking); timer = uniform_i32(timer); pre_n = uniform_i32(pre_n);
  nrec = uniform_i32(nrec); pos = uniform_i64(pos); pre_pos = uniform_i64(pre_pos);
  if (done) return;
  const bool callable = (a.flush ? pos + kHalfFrame <= a.t1 : pos + a.call_reach <= a.t1) && nrec < a.rec_cap;
  if (!callable) return;
  const int stride = tracking ? a.track_every + 1 : 1;
  const int64_t first = pos + (tracking ? (int64_t)timer * kHalfFrame : 0);
  const int64_t last_start = a.flush ? a.t1 - kHalfFrame : a.t1 - a.call_reach; // last position a call may start at
  int64_t ahead = first <= last_start ? (last_start - first) / ((int64_t)stride * kHalfFrame) + 1 : 0;
  if (ahead > kCacheMaxPerChain) ahead = kCacheMaxPerChain;
  if (a.ask_stream && ahead >= kCacheMinSearches && !tracking) {
    int64_t o = first - kStreamBack;
    if (pre_n > 0 && pre_pos < o) o = pre_pos;
    int64_t want = (last_start + kNLags - o + kStreamSlot - 1) / kStreamSlot;
    if (want > kCacheMaxPerChain) want = kCacheMaxPerChain;
    const int got = a.slot_quota < (int)want ? a.slot_quota : (int)want;
    if (got >= 2) {
      int wi = 0;
      if (lane == 0) wi = atomicAdd(a.slot_next, got);
      wi = uniform_i32(wi);
      for (int i = lane; i < got; i += 64) {
        a.slot_owner[wi + i] = make_int2(chain, -1 - i); // (second member negative: a slot of a stream)
        a.slot_pos[wi + i] = o + (int64_t)i * kStreamSlot;
      }
      if (lane == 0) {
        atomicAdd(a.launch_slots, got);
        atomicAdd(a.slot_next + 3, 1); // diagnostics
        st->cache_pos = o; st->cache_base = chain * a.slot_quota; st->cache_n = got; st->cache_stride = 0; // (0: a stream)
      }
      return;
    }
    ahead = 0; // (no room: nothing to ask for)
  }
  if (ahead >= kCacheMinSearches) {
    const int got = a.slot_quota < (int)ahead ? a.slot_quota : (int)ahead;
    if (got >= kCacheMinSearches) {
      int wi = 0;
      if (lane == 0) wi = atomicAdd(a.slot_next, got);
      wi = uniform_i32(wi);
      for (int i = lane; i < got; i += 64) {
        a.slot_owner[wi + i] = make_int2(chain, i);
        a.slot_pos[wi + i] = first + (int64_t)i * stride * kHalfFrame;
      }
      if (lane == 0) {
        atomicAdd(a.launch_slots, got);
        atomicAdd(a.slot_next + 3, 1); // diagnostics
        st->cache_pos = first; st->cache_base = chain * a.slot_quota; st->cache_n = got; st->cache_stride = stride;
      }
    }
  }
}

// One workgroup per channel, one wave per chain; when the three chains are through, the channel's jobs of the
// next round are planned by the same launch.
__global__ __launch_bounds__(192) void lt_resolve_kernel(ResolveArgs a)
{
  const int c = blockIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  resolve_chain(a, c * 3 + w, lane);
  if (!a.plan_next) {
    if (a.ask) chain_request_cache(a, c * 3 + w, lane);
    return;
  }
  __threadfence_block();
  __syncthreads();
  __shared__ ScanJob *planned;
  if (threadIdx.x == 0) planned = plan_channel(a.next, c);
  if (!a.next.cached) return;
  __threadfence_block();
  __syncthreads();
  if (planned) fill_job_slots(a.next, planned, threadIdx.x, 192);
}

} // namespace lt
