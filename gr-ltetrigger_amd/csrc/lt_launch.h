// Host-visible launch interface of lt_kernels.hip (keeps device types out of the C-ABI layer).
#pragma once
#include <vector>
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/ltetrigger_hip.h"
#include "lt_tables.h"

namespace lt {

// One pass of the batched path over a chunk (lt_batch_run / lt_batch_push): plan, scan + resolve per segment,
// per-chain kernel, per-record kernels, history.
struct RunLaunch {
  const void *iq;     // device, [n_channels][ch_stride] cf32: the chunk
  int64_t ch_stride;
  int64_t t0;         // absolute stream position of iq[.][0]
  int64_t n;          // samples in the chunk
  const void *hist = nullptr; // device, [n_channels][hist_stride] cf32: the hist_len samples in front of t0
  int64_t hist_stride = 0;
  int64_t hist_len = 0;
  void *hist_new = nullptr;   // device: where the history for the next chunk goes (hist_len_new samples per channel)
  int64_t hist_len_new = 0;
  int flush = 1;      // the stream ends with this chunk
  int n_channels = 0;
  void *state = nullptr;      // device, ChainState[n_channels*3]
  lt_record *recs = nullptr;  // device, [n_channels*3][rec_cap]
  int rec_cap = 0;
  int *rec_count = nullptr;   // device, [n_channels*3]
  void *out_hf = nullptr;     // device or null, [n_channels*3][out_cap][9600] cf32
  int out_cap = 0;
  const DeviceTables *tab = nullptr; // device
  float psr_threshold = 0.f;
  int track_after = 16;
  int track_every = 8;
  int scan = 1;               // 0: every chain goes through the per-chain kernel
  void *jobs = nullptr;       // device, 2 x ScanJob[n_channels*3]
  int *n_jobs = nullptr;      // device, scan_counter_ints() counters
  void *parts = nullptr;      // device, scan_parts_bytes_per_chain() per chain
  int seg_first = 2;          // windows per job in the first scan round (cells present from the start fire right away)
  int seg_len = 0;            // windows per job in the later rounds (<= scan_max_round()); 0 = the chunk's windows split evenly over as few rounds as possible
  int scan_groups = 0;        // strips per scan workgroup: 1, 2 or 3; 0 = chosen from n_channels and n_cus
  int cached_segments = 8;    // cached jobs (batches of <= cached_scan_max_channels()): waves a round's windows of one (strip, root) are dealt to; 0 = one wave in lt_scan_kernel
  float cached_light = 0.25f; // ... and what a window that only updates the average costs such a wave, in windows with a search
  int n_cus = 0;              // compute units of the device (0: 256)
  int extra_rounds = 0;       // rounds beyond what an idle chain needs (what the rounds leave over goes through the window cache)
  void *cp_scratch = nullptr; // device, [n_channels*3][rec_cap] x per_record_scratch_bytes(): CP metrics, then the two work lists
  hipEvent_t ev_scan_done = nullptr;  // optional: recorded after the last scan segment / after the per-chain kernel
  hipEvent_t ev_chain_done = nullptr; // (required when the per-record kernels go to a stream of their own)
  int rebase = 0;             // the run is a capture of its own: the chains' positions restart at 0
  void *cache = nullptr;      // device, slot_cap x cache_slot_bytes(): the window cache (lag metrics formed in bulk for the chains the
                              // scan rounds leave to the per-chain kernel); null: no cache, the per-chain kernel runs once
  void *slot_owner = nullptr; // device, slot_cap x int2
  void *slot_pos = nullptr;   // device, slot_cap x int64
  int cache_level = 2;        // 0 = no cache, 1 = the per-chain kernel's two passes, 2 = also the scan rounds of small batches
  int slot_cap = 0;
  int slot_quota = 0;         // slots per chain for the per-chain kernel's requests (slot_quota * n_channels * 3 <= slot_cap)
  int records_delay_us = 0;   // diagnostics (lt_batch_debug_hold_records): the per-record kernels start this much later
  int *wait_hint = nullptr;   // device view of two words in pinned host memory: the number of the last run in which a chain had
                              // enough searches ahead of it to ask for the window cache, and of the last run the per-chain kernel ran in
  int run_id = 0;             // this run's number (from 1)
  int stream = 0;             // window cache level 4: untracked chains ask for their lag-metric stream (lt_chain_kernel.h)
  int ask_early = 0;          // with the rounds on and the cache there: the chains' cache requests are made by the run's last resolve
                              // launch, the per-chain kernel runs once behind lt_metric_kernel (no pass 0 launch)
  int two_pass = 1;           // 0: the per-chain kernel in one pass without the cache (no lt_metric_kernel, no second launch):
                              // no chain asked in the latest run the caller knows in full.  Timing only, never a record
  int *n_launches = nullptr;  // host: out, the number of kernel launches the run enqueued
  int list_slot = 0;          // which of the two work lists (and of their counters behind the job counters) this run uses (0 / 1)
};

// records_stream: null = everything on `stream`; else the per-record kernels (and the half-frame capture) are enqueued
// there, behind R.ev_chain_done -- the caller orders that stream against whatever reads the records
hipError_t launch_run(const RunLaunch &R, hipStream_t stream, hipStream_t records_stream = nullptr);
// lt_batch_reset: chain state back to the constructor's; parts: 1 = all but the members only the per-record kernels
// touch, 2 = those, 3 = both
hipError_t launch_reset(void *state, int n_chains, int parts, int *rec_count, hipStream_t stream); // rec_count: [n_chains] to clear, or null
int call_reach();
int scan_job_bytes();
int scan_parts_bytes_per_chain();
int scan_max_round();
int cached_scan_max_channels(); // batches up to this size scan from the window cache (levels >= 2)
int scan_max_rounds();
int scan_counter_ints();
int per_record_scratch_bytes();
int cache_slot_bytes();
int cache_counter_index(); // d_njobs[this] = slots asked for; [+1] searches served from the cache, [+2] searches through the
                           // transforms in pass 1, [+3] chains that waited for the cache, [+4] hits that moved a waiting chain off its grid,
                           // [+5] / [+6] windows an average was run through again, from the cache / through the transforms
                           // (the 7 values of lt_batch_cache_stats, include/ltetrigger_hip.h)
int cache_max_per_chain();
size_t chain_state_bytes();
size_t chain_state_avg_offset();

// res: device struct {int peak_pos; float psr; float peak_value;}
hipError_t launch_find(const void *win, float *avg, const DeviceTables *tab, int n_id_2, void *res,
                       hipStream_t stream);
hipError_t launch_cfo_compute(const void *sym, const DeviceTables *tab, int n_id_2, float *cfo,
                              hipStream_t stream);
hipError_t launch_cfo_correct(const void *in, void *out, int n, float freq, const DeviceTables *tab,
                              hipStream_t stream);
// sync_state: device struct {float m_norm_avg, m_ext_avg;}
hipError_t launch_detect_cp(const void *in, int peak_pos, void *sync_state, int *cp_out,
                            hipStream_t stream);
// res: device struct {int m0, m1; float m0_value, m1_value;}
hipError_t launch_sss_m0m1(const void *sym, const DeviceTables *tab, int n_id_2, void *res,
                           hipStream_t stream);
// tap_table: device copy of decim_tap_table(D, taps, ntaps) for the rates that have one (decim_has_tap_table), else null
hipError_t launch_decim(const void *x, void *y, int n_channels, int64_t n_in, int D, const float *taps,
                        int ntaps, const float *tap_table, hipStream_t stream);
std::vector<float> decim_tap_table(int D, const float *taps, int ntaps);
bool decim_has_tap_table(int D, int ntaps);

} // namespace lt
