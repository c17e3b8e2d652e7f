// Grid scan: srslte_pss_find_pss (lib/pss_impl.cc:165-167) of every idle chain, window after window, with
// the exponential average held in registers and nothing but 32-byte partial results per (window, strip)
// written back.  See lt_scan.h for the strip geometry.
//
// A chain is idle when it is neither tracking nor building a score: every general_work call then searches its
// 9600-sample window, misses and moves on by exactly 9600 samples (lib/pss_impl.cc:163-182, 217), so the
// windows a chain will search are known in advance as long as it keeps missing.  That is what nearly all
// chains do nearly all the time, and the search is where the arithmetic is.  The scan runs those windows
// for all such chains at once, speculatively: lt_resolve_kernel then walks each chain's windows in call order,
// writes the records of the misses and stops at the first window that fires (PSR over the threshold) or that
// the strips could not resolve; from that window on the chain belongs to the per-chain kernel (lt_chain.h),
// which executes general_work call by call.
//
// Work decomposition.  A *job* is a capture channel together with the idle chains (roots) of it that share
// one window alignment -- normally all three.  A job's 12 strips go to four workgroups of three *groups*;
// a group is four waves that own one strip for good: a forward wave, which fetches the strip's 1024 samples of
// every window and transforms them once, and one wave per root, which multiplies that spectrum by its root's
// filter spectrum, transforms back and keeps its root's average.  The hand-over costs no LDS of its own: the
// forward wave stops after the second exchange of its transform, and each root wave runs the last radix-4 stage
// itself, straight out of the forward wave's exchange area.
//   workgroup = 12 waves = 3 per SIMD; LDS: 12 exchange areas (102 KB) + filter spectra (24 KB) + twiddles (17 KB)
#pragma once
#include "lt_scan.h"

// Ordering of the LDS hand-over.  A wave's LDS instructions execute in program order, so a counter written
// after the data (or read before it) needs no s_waitcnt -- only the compiler must keep the order.  A
// workgroup-scope fence would also wait for the wave's global loads and stores (vmcnt(0)): the forward wave's
// prefetch of the next windows and the root waves' partial-result stores, i.e. a full memory latency per window.
#define LT_LDS_ORDER __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")

#ifndef LT_SCAN_STAGGER
#define LT_SCAN_STAGGER 0 // s_sleep units (64 clocks) by which group g starts later than group g - 1
#endif

namespace lt {

// One per (channel, alignment): built by lt_plan_kernel from the chain states.
struct ScanJob {
  int64_t pos0;      // absolute position of the job's window 0
  int32_t channel;
  int32_t n_windows; // windows the idle chains may search in this run
  int32_t roots;     // bit r: chain (channel, N_id_2 = r) is part of the job
  int32_t rec0[3];   // per root: record index of window 0
};

constexpr int kScanWaves = 12;
constexpr int kScanThreads = 64 * kScanWaves;
constexpr int kScanGroups = 3;              // strips per workgroup
constexpr int kScanWgPerJob = kStrips / kScanGroups;

struct ScanArgs {
  const v2f *iq;       // [n_channels][ch_stride]
  int64_t ch_stride;
  const v2f *hist;     // [n_channels][hist_stride] or null
  int64_t hist_stride;
  int64_t t0, n, hist_len;
  const ScanJob *jobs;
  const int *n_jobs;
  ChainState *state;   // [n_channels * 3]
  StripPartial *parts; // [n_channels * 3][parts_windows][12]
  int64_t parts_windows;
  int w_begin, w_end;  // this segment: windows [w_begin, w_end) of every job
  const DeviceTables *tab;
  unsigned long long *prof; // diagnostics build only (LT_SCAN_PROF): shader-clock sums per phase of one root wave
};

__global__ __launch_bounds__(kScanThreads) void lt_scan_kernel(ScanArgs a)
{
  __shared__ v2f E[kScanWaves * fft::kScratchW]; // one exchange area per wave
  __shared__ v2f SPEC[3][fft::kN];               // filter spectra / 1024, natural order
  __shared__ fft::c32x2 TW1[16 * 64];
  __shared__ fft::c32x2 TW2[16 * 4];
  __shared__ int sync_ready[kScanGroups], sync_taken[kScanGroups];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroups b, b + 8, b + 16, b + 24 (one XCD under round-robin dispatch) share a job: its strips overlap
  // by 213 samples, so the second read of those comes from that XCD's L2
  const int b = blockIdx.x;
  const int slot = b >> 3;
  const int job_i = (slot / kScanWgPerJob) * 8 + (b & 7);
  const int quarter = slot % kScanWgPerJob;
  if (job_i >= *a.n_jobs) return;
  const ScanJob job = a.jobs[job_i];
  const int w_end = a.w_end < job.n_windows ? a.w_end : job.n_windows;
  if (a.w_begin >= w_end) return;
  // group g = w % 3 owns strip 3 quarter + g; role w / 3: 0 = forward, 1..3 = root 0..2 (the three forward
  // waves land on three different SIMDs)
  const int g = w % kScanGroups, role = w / kScanGroups;
  const int strip = quarter * kScanGroups + g;
  const int root = role - 1;
  // a root leaves the job for good once the resolver has seen it fire (scan_live cleared)
  int live_mask = 0;
#pragma unroll
  for (int r = 0; r < 3; ++r)
    if ((job.roots >> r) & 1)
      if (a.state[job.channel * 3 + r].scan_live) live_mask |= 1 << r;
  if (!live_mask) return;
  const bool active = role == 0 || ((live_mask >> root) & 1);

  for (int i = tid; i < 16 * 64; i += kScanThreads) TW1[i] = *reinterpret_cast<const fft::c32x2 *>(a.tab->fft_tw1[i]);
  if (tid < 64) TW2[tid] = *reinterpret_cast<const fft::c32x2 *>(a.tab->fft_tw2[tid]);
  for (int i = tid; i < 3 * fft::kN; i += kScanThreads) {
    const cf32 h = a.tab->pss_spec[i >> 10][i & 1023];
    SPEC[i >> 10][i & 1023] = v2f{h.re, h.im};
  }
  if (tid < kScanGroups) sync_ready[tid] = sync_taken[tid] = 0;
  __syncthreads(); // tables in place: the only workgroup barrier
  fft::c32 *e = E + w * fft::kScratchW;
  const fft::c32 *e_fwd = E + g * fft::kScratchW; // the group's forward wave is wave g

  Capture cap;
  cap.iq = a.iq + (int64_t)job.channel * a.ch_stride;
  cap.hist = a.hist ? a.hist + (int64_t)job.channel * a.hist_stride : nullptr;
  cap.t0 = a.t0;
  cap.n = a.n;
  cap.hist_len = a.hist ? a.hist_len : 0;

  // The two roles run separate loops, so that each gets a register allocation of its own: the forward wave
  // carries two blocks of samples, a root wave one block and its average.  Hand-over per group through two LDS
  // counters (no workgroup barrier after the table load: the three groups of a workgroup drift apart, which is
  // wanted -- every SIMD holds one wave of each group, and waves in different phases of the transform keep
  // both the vector unit and the LDS busy):
  //   ready[g] = windows whose first two transform stages sit in the forward wave's exchange area (1 at a time)
  //   taken[g] = sum over the root waves of the windows they have read from it
  const int n_roots = __builtin_popcount(live_mask);
  const int n_w = w_end - a.w_begin;
#if LT_SCAN_STAGGER
  for (int i = 0; i < g; ++i) __builtin_amdgcn_s_sleep(LT_SCAN_STAGGER);
#endif
  if (role == 0) {
    fft::c32 x[16];  // the block being transformed
    fft::c32 xn[16]; // the next window's block, in flight
    strip_load(cap, job.pos0 + (int64_t)a.w_begin * kHalfFrame, strip, lane, x);
    if (n_w > 1) strip_load(cap, job.pos0 + (int64_t)(a.w_begin + 1) * kHalfFrame, strip, lane, xn);
    for (int k = 0; k < n_w; ++k) {
      int ln = lane; // (opaque per iteration: keeps lane-dependent addresses from being hoisted out of the loop and spilled)
      asm volatile("" : "+v"(ln));
      // the root waves have taken window k - 1: the exchange area is free
      while (__hip_atomic_load(&sync_taken[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < n_roots * k) __builtin_amdgcn_s_sleep(1);
      LT_LDS_ORDER;
      strip_fft_ab(ln, x, TW1, TW2, e);
      LT_LDS_ORDER;
      if (lane == 0) __hip_atomic_store(&sync_ready[g], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
      for (int R = 0; R < 16; ++R) x[R] = xn[R];
      if (k + 2 < n_w) strip_load(cap, job.pos0 + (int64_t)(a.w_begin + k + 2) * kHalfFrame, strip, ln, xn);
    }
  } else if (active) {
#ifdef LT_ROOT_PRIO
    __builtin_amdgcn_s_setprio(LT_ROOT_PRIO);
#endif
    fft::c32 x[16]; // spectrum, then lags
    float ema[kStripRegs];
    ChainState *st = a.state + job.channel * 3 + root;
    strip_ema_load(st->avg[st->ema_sel], strip, lane, ema);
    StripPartial *parts = a.parts + ((int64_t)(job.channel * 3 + root) * a.parts_windows + a.w_begin) * kStrips + strip;
#ifdef LT_SCAN_PROF
    unsigned long long pr[8] = {0}, tp = __builtin_amdgcn_s_memtime();
#define LT_PSTAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pr[i] += t_ - tp; tp = t_; }
#else
#define LT_PSTAMP(i)
#endif
    for (int k = 0; k < n_w; ++k) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      while (__hip_atomic_load(&sync_ready[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= k) __builtin_amdgcn_s_sleep(1);
      LT_LDS_ORDER;
      LT_PSTAMP(0) // waiting for the forward wave
      fft::fft1024w_stageC<false>(ln, x, e_fwd); // x = the block's spectrum
      LT_LDS_ORDER;
      if (lane == 0) __hip_atomic_fetch_add(&sync_taken[g], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      LT_PSTAMP(1) // last forward stage
#ifdef LT_SCAN_PROF
      strip_root_update_prof(strip, ln, x, SPEC[root], TW1, TW2, e, ema, pr, tp);
#else
      strip_root_update(strip, ln, x, SPEC[root], TW1, TW2, e, ema);
#endif
      StripPartial p;
#ifndef LT_ABL_NO_RESOLVE
      strip_resolve(ema, strip, ln, p);
#else
      p.m1 = ema[3]; p.p1 = 0; p.s = ema[5]; p.lb = p.ub = 0; p.a0 = 0.f; p.flags = 0; p.pad = 0;
#endif
      if (lane == 0) parts[(int64_t)k * kStrips] = p;
      LT_PSTAMP(7) // resolve + store
    }
#ifdef LT_SCAN_PROF
    if (a.prof && blockIdx.x == 8 && w == 7 && lane == 0)
      for (int i = 0; i < 8; ++i) a.prof[i] = pr[i];
#endif
    strip_ema_store(st->avg[st->ema_sel ^ 1], strip, lane, ema);
  }
}

// ---------------------------------------------------------------- plan: which chains does the scan serve?
// One thread per channel.  A chain is eligible when it is idle (not tracking, score 0, no tracking_lost pending:
// every call searches, and a miss moves on by exactly 9600 samples, lib/pss_impl.cc:163-182,217) and at
// least one window fits into the chunk.  Chains of a channel at the same stream position share a job.
struct PlanArgs {
  ChainState *state;
  ScanJob *jobs;
  int *n_jobs;       // zeroed before the launch
  int *rec_count;    // [n_channels*3]: zeroed here
  int n_channels;
  int rec_cap;
  int64_t t1;        // absolute end of the chunk
  int flush;
  int enable;        // 0: no chain goes to the scan (lt_batch_set_mode 0)
  int max_windows;   // room in the partial-result array
  int call_reach;    // kCallReach (lt_chain_kernel.h)
};

__global__ void lt_plan_kernel(PlanArgs a)
{
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.n_channels) return;
  int nwin[3];
  int64_t pos[3];
  for (int r = 0; r < 3; ++r) {
    ChainState *st = a.state + c * 3 + r;
    a.rec_count[c * 3 + r] = 0;
    st->scan_job = -1;
    st->scan_live = 0;
    st->scan_done = 0;
    st->scan_seg0 = 0;
    pos[r] = st->pos;
    const bool idle = !st->is_tracking && st->score == 0 && !st->tracking_lost;
    int64_t n = 0;
    if (idle && a.enable) {
      // calls that may start in this chunk (the per-chain kernel applies the same rule)
      const int64_t room = a.t1 - pos[r] - (a.flush ? kHalfFrame : a.call_reach);
      n = room >= 0 ? room / kHalfFrame + 1 : 0;
      if (n > a.rec_cap) n = a.rec_cap;
      if (n > a.max_windows) n = a.max_windows;
    }
    nwin[r] = (int)n;
  }
  for (int r = 0; r < 3; ++r) {
    if (nwin[r] <= 0) continue;
    bool leader = true;
    for (int q = 0; q < r; ++q) leader = leader && !(nwin[q] > 0 && pos[q] == pos[r]);
    if (!leader) continue;
    const int j = atomicAdd(a.n_jobs, 1);
    ScanJob job;
    job.pos0 = pos[r];
    job.channel = c;
    job.n_windows = nwin[r];
    job.roots = 0;
    for (int q = r; q < 3; ++q)
      if (nwin[q] > 0 && pos[q] == pos[r]) {
        job.roots |= 1 << q;
        a.state[c * 3 + q].scan_job = j;
        a.state[c * 3 + q].scan_live = 1;
      }
    job.rec0[0] = job.rec0[1] = job.rec0[2] = 0;
    a.jobs[j] = job;
  }
}

// ---------------------------------------------------------------- resolve: partials -> records, in call order
// One wave per chain, one lane per window (64 at a time): every lane combines the 12 partials of its window
// (resolve_window); the first window that fires -- PSR over the threshold -- or that the strips could not
// resolve ends the chain's time in the scan.  The windows before it are misses: general_work searched, did
// not emit and consumed 9600 samples (lib/pss_impl.cc:163-182, 217); their records are written here.
struct ResolveArgs {
  ChainState *state;
  const ScanJob *jobs;
  const StripPartial *parts;
  int64_t parts_windows;
  lt_record *recs;
  int rec_cap;
  int *rec_count;
  int w_begin, w_end; // the scan segment just run
  float psr_threshold;
  int track_every;
};

__global__ __launch_bounds__(64) void lt_resolve_kernel(ResolveArgs a)
{
  const int chain = blockIdx.x, lane = threadIdx.x;
  ChainState *st = a.state + chain;
  if (!st->scan_live) return;
  const ScanJob job = a.jobs[st->scan_job];
  const int w_end = a.w_end < job.n_windows ? a.w_end : job.n_windows;
  if (a.w_begin >= w_end) return;
  const StripPartial *parts = a.parts + (int64_t)chain * a.parts_windows * kStrips;
  lt_record *recs = a.recs + (int64_t)chain * a.rec_cap;
  float psr_max = st->psr_max, last_psr = st->psr;
  int last_peak = st->peak_pos;
  uint32_t psr_i = st->psr_i;
  const float m_norm_avg = st->m_norm_avg, m_ext_avg = st->m_ext_avg;
  const int call_base = st->n_calls - a.w_begin; // window k of the job is call call_base + k (records start at 0 each run)
  int fired_at = -1;
  for (int base = a.w_begin; base < w_end; base += 64) {
    const int k = base + lane;
    const bool valid = k < w_end;
    WindowResult r;
    r.peak_pos = 0; r.peak_value = 0.f; r.psr = 0.f; r.exact = true;
    if (valid) r = resolve_window(parts + (int64_t)k * kStrips);
    const bool fire = valid && (!r.exact || r.psr > a.psr_threshold);
    const unsigned long long fm = __ballot(fire);
    const int first = fm ? __builtin_ctzll(fm) : 64;
    const bool miss = valid && lane < first;
    // d_psr_max over the calls so far (if (psr > psr_max) psr_max = psr), as an inclusive prefix maximum
    float pm = miss ? r.psr : -FLT_MAX;
    if (!(pm == pm)) pm = -FLT_MAX; // a NaN never raises the maximum
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float o = __shfl_up(pm, d);
      if (lane >= d) pm = fmaxf(pm, o);
    }
    const float my_max = fmaxf(psr_max, pm);
    if (miss) {
      lt_record rec;
      rec.in_pos = job.pos0 + (int64_t)k * kHalfFrame; rec.out_pos = -1; rec.channel = chain / 3; rec.n_id_2 = chain % 3;
      rec.call_idx = call_base + k; rec.flags = LT_F_SEARCHED; rec.peak_pos = r.peak_pos; rec.score = 0;
      rec.psr = r.psr; rec.peak_value = r.peak_value; rec.cfo = 0.f; rec.mean_cfo = 0.f;
      rec.m_norm_avg = m_norm_avg; rec.m_ext_avg = m_ext_avg; rec.m0_value = 0.f;
      rec.m1_value = 0.f; rec.m0 = -1; rec.m1 = -1; rec.n_id_1 = -1; rec.cell_id = -1;
      rec.psr_max = my_max; rec.timer = a.track_every;
      recs[k] = rec;
      st->psr_data[(psr_i + (uint32_t)(k - base)) % kMovingAvgSz] = r.psr; // d_psr_data ring (lib/pss_impl.cc:166)
    }
    const int n_miss = first < 64 ? first : (w_end - base < 64 ? w_end - base : 64);
    if (n_miss > 0) {
      psr_max = __shfl(my_max, n_miss - 1);
      last_psr = __shfl(r.psr, n_miss - 1);
      last_peak = __shfl(r.peak_pos, n_miss - 1);
      psr_i += (uint32_t)n_miss;
    }
    if (first < 64) {
      fired_at = base + first;
      break;
    }
  }
  if (lane == 0) {
    const int done = fired_at >= 0 ? fired_at : w_end; // windows of the job turned into records so far
    const int n_new = done - a.w_begin;
    st->pos = job.pos0 + (int64_t)done * kHalfFrame;
    st->n_calls += n_new;
    st->psr_i = psr_i;
    st->psr_max = psr_max;
    if (n_new > 0) {
      st->psr = last_psr;
      st->peak_pos = last_peak;
      st->timer = a.track_every;
    }
    st->scan_done = done;
    a.rec_count[chain] = done;
    if (fired_at >= 0) {
      st->scan_live = 0;          // the per-chain kernel takes over at window `fired_at` ...
      st->scan_seg0 = a.w_begin;  // ... from the average as it was before this segment (avg[ema_sel])
    } else {
      st->ema_sel ^= 1;           // the scan has left the average after window w_end - 1 in the other buffer
    }
  }
}

} // namespace lt
