// Host-visible launch interface of lt_kernels.hip (keeps device types out of the C-ABI layer).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/ltetrigger_hip.h"
#include "lt_tables.h"

namespace lt {

struct ChainLaunch {
  const void *iq;     // device, [n_channels][ch_stride] cf32
  int64_t n_samples;
  int64_t ch_stride;
  int n_channels;
  void *state;        // device, ChainState[n_channels*3]
  lt_record *recs;    // device, [n_channels*3][rec_cap]
  int rec_cap;
  int *rec_count;     // device, [n_channels*3]
  void *out_hf;       // device or null, [n_channels*3][out_cap][9600] cf32
  int out_cap;
  const DeviceTables *tab; // device
  float psr_threshold;
  int track_after;
  int track_every;
  float *metric = nullptr;        // device, [3][n_channels][metric_ch_stride]: bulk + replay mode when set
  int64_t metric_ch_stride = 0;
  int64_t metric_root_stride = 0;
  float *edge = nullptr;          // device, [3][n_channels][max_windows][256]: grid edge lags (bulk + replay mode)
  int64_t edge_ch_stride = 0;
  int64_t edge_root_stride = 0;
  hipEvent_t ev_metric_done = nullptr; // optional: recorded after the lag-metric kernel / after the edge kernel
  hipEvent_t ev_edge_done = nullptr;
  hipEvent_t ev_replay_done = nullptr; // ... / after the replay kernel (the per-record post-pass kernels follow)
  void *cp_scratch = nullptr;     // device, [n_channels*3][rec_cap] x 16 B: per-record CP metrics (bulk + replay mode)
  int n_cus = 256;      // compute units of the device (grid of the persistent metric kernel)
  int ablate = 0;       // diagnostics only
  void *prof = nullptr; // diagnostics only: 8 x uint64 on the device
};

hipError_t launch_chain(const ChainLaunch &L, hipStream_t stream);
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
hipError_t launch_decim(const void *x, void *y, int n_channels, int64_t n_in, int D, const float *taps,
                        int ntaps, hipStream_t stream);

} // namespace lt
