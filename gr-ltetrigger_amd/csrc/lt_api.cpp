// C-ABI of the gfx950 LTE cell-search path (include/ltetrigger_hip.h).
// Host code only: handle management, staging copies, stream/event plumbing, error mapping.
// There is deliberately no CPU implementation of any compute entry point here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/ltetrigger_hip.h"
#include "lt_launch.h"
#include "lt_tables.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                            \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return fail(LT_ERROR, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,     \
                  __LINE__);                                                                     \
  } while (0)

// Constant tables, uploaded once per device and shared by every handle on it.
struct DeviceCtx {
  lt::DeviceTables *d_tab = nullptr;
  int n_cus = 0;
};
std::mutex g_ctx_mu;
std::map<int, DeviceCtx> g_ctx;

int use_device(int device)
{
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(LT_ERROR, "no HIP device available (%s); this library has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if (device < 0 || device >= n) return fail(LT_ERROR_INVALID_INPUTS, "device %d out of range [0,%d)", device, n);
  HIP_TRY(hipSetDevice(device));
  return LT_OK;
}

int get_tables(int device, const lt::DeviceTables **out)
{
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  DeviceCtx &c = g_ctx[device];
  if (!c.d_tab) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(LT_ERROR, "device %d is %s; this library only carries gfx950 code", device,
                  prop.gcnArchName);
    auto *host = new (std::nothrow) lt::DeviceTables;
    if (!host) return fail(LT_ERROR, "out of host memory");
    lt::build_device_tables(host);
    lt::DeviceTables *d = nullptr;
    hipError_t e = hipMalloc(&d, sizeof *host);
    if (e == hipSuccess) e = hipMemcpy(d, host, sizeof *host, hipMemcpyHostToDevice);
    delete host;
    if (e != hipSuccess) return fail(LT_ERROR, "uploading tables failed: %s", hipGetErrorString(e));
    c.d_tab = d;
    c.n_cus = prop.multiProcessorCount;
  }
  *out = c.d_tab;
  return LT_OK;
}

struct FindResultHost {
  int peak_pos;
  float psr;
  float peak_value;
};
struct SyncStateHost {
  float m_norm_avg, m_ext_avg;
};
struct M0M1Host {
  int m0, m1;
  float m0_value, m1_value;
};

} // namespace

struct lt_pss {
  int device = 0;
  int n_id_2 = 10; // invalid until lt_pss_set_N_id_2, like srslte_pss_init
  hipStream_t stream = nullptr;
  const lt::DeviceTables *tab = nullptr;
  float *d_avg = nullptr;
  lt_cf32 *d_win = nullptr;
  FindResultHost *d_res = nullptr;
  float *d_cfo = nullptr;
  std::atomic<float> peak_value{0.f}; // (read by lt_pss_get_peak_value from any thread)
};

struct lt_cfo {
  int device = 0;
  int nsamples = 0;
  float last_freq = 0.f;
  hipStream_t stream = nullptr;
  const lt::DeviceTables *tab = nullptr;
  lt_cf32 *d_buf = nullptr;
};

struct lt_sync {
  int device = 0;
  int n_id_2 = 10;
  // (the three below are read by the plain getters from any thread while the handle is busy on its own)
  std::atomic<int> cp{LT_CP_NORM};
  std::atomic<int> cp_len{9};
  std::atomic<int> n_id_1{1000}; // invalid
  hipStream_t stream = nullptr;
  const lt::DeviceTables *tab = nullptr;
  lt_cf32 *d_in = nullptr;
  SyncStateHost *d_state = nullptr;
  int *d_cp = nullptr;
  M0M1Host *d_res = nullptr;
};

struct lt_batch {
  int device = 0;
  int n_channels = 0;
  int64_t max_samples = 0;
  lt_chain_params params{};
  hipStream_t stream = nullptr;    // own stream
  hipStream_t last_stream = nullptr;
  const lt::DeviceTables *tab = nullptr;
  lt_cf32 *d_iq = nullptr;         // staging for host input, allocated on first use
  void *d_state = nullptr;
  lt_record *d_recs = nullptr;
  int *d_count = nullptr;
  int rec_cap = 0;
  lt_cf32 *d_hf = nullptr;
  int64_t hf_cap = 0;
  void *d_jobs = nullptr;    // grid scan: ScanJob[n_channels*3]
  int *d_njobs = nullptr;
  void *d_parts = nullptr;   // grid scan: the partial results of a round, per chain
  void *d_cache = nullptr;   // window cache: cache_slots x cache_slot_bytes()
  void *d_slot_owner = nullptr;
  void *d_slot_pos = nullptr;
  int cache_slots = 0;
  int cache_quota = 0;       // ... of which every chain owns this many for the per-chain kernel's requests
  int use_cache = 4;         // lt_batch_set_window_cache (4 = lag-metric streams for untracked chains; deterministic since round 5)
  void *d_cp = nullptr;      // per-record scratch (CP metrics, work list), [n_channels*3][rec_cap] x per_record_scratch_bytes()
  lt_cf32 *d_hist[2] = {nullptr, nullptr}; // streaming: the samples in front of the next chunk, [n_channels][hist_cap]
  int hist_sel = 0;
  int64_t hist_len = 0;      // valid samples in d_hist[hist_sel] (its tail)
  int64_t hist_cap = 0;
  int64_t t0 = 0;            // absolute stream position of the next chunk
  int mode = 1;              // 0 per-chain kernel only, 1 grid scan + per-chain kernel
  int seg_first = 2, seg_len = 0;
  int scan_groups = 0;       // strips per scan workgroup (0: chosen per run from the channel count)
  int n_cus = 0;
  bool pending_reset = false; // lt_batch_reset: applied on the stream of the next run, in front of it
  // events of the last kPhaseRing runs: [run % kPhaseRing][0..3] = start, scan done, per-chain kernel done, end
  static constexpr int kPhaseRing = 8;
  hipEvent_t ev[kPhaseRing][4] = {};
  uint64_t n_runs = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr; // = ev[last run][0], [3]
  int phase_timing = 2;            // lt_batch_set_phase_timing: 2 = events between the stages of a run, 1 = around it, 0 = none
  bool ev_phased[kPhaseRing] = {}; // the run in this slot recorded the stage events
  bool ev_timed = false;           // the last run recorded its first and last event
  int last_launches = 0;           // kernel launches the last run enqueued (lt_batch_last_launch_count)
  bool ran = false;
  // lt_batch_collect's landing area: pinned (a copy into pageable memory goes through the runtime's own staging buffer, a
  // second copy and ~100 us per call: what a 10 ms chunk of a stream pays once per chunk); the vectors only if pinning fails
  lt_record *h_recs_pin = nullptr;
  int *h_count_pin = nullptr;
  std::vector<lt_record> h_recs;
  std::vector<int> h_count;
  // lt_batch_set_overlap: the per-record kernels of a run on a stream of their own, beside the next run's scan rounds.
  // Two sets of records take turns; rec_done[i] = "the per-record kernels that worked on set i are through".
  int overlap = 0;
  hipStream_t side = nullptr;
  lt_record *d_recs_b = nullptr;
  int *d_count_b = nullptr;
  hipEvent_t rec_done[2] = {nullptr, nullptr};
  bool rec_done_valid[2] = {false, false};
  bool side_busy = false;          // something may still be running on `side`
  int records_delay_us = 0;        // lt_batch_debug_hold_records
  lt_record *cur_recs = nullptr;   // the set the last run wrote
  int *cur_count = nullptr;
  uint64_t set_run[2] = {0, 0};    // 1 + the number of the run whose records set 0 / 1 holds (0: none)
  hipStream_t copy = nullptr;      // lt_batch_collect_previous: its copies, beside the run in flight
  // Two passes of the per-chain kernel (around lt_metric_kernel) only while chains want the window cache: the kernel writes
  // the number of the last run in which one did, and of the last run it ran in, into two words of pinned host memory; a
  // run launched when the latest run known in full had no such chain launches the kernel once, without the cache (two
  // launches fewer: 10 us of a 64-channel step).  Timing only: the records are the same either way.
  int *h_wait_hint = nullptr;
  int *d_wait_hint = nullptr;
};

struct lt_decim {
  int device = 0;
  int decimation = 1;
  std::vector<float> taps;
  float *d_taps = nullptr;
  float *d_tap_table = nullptr; // the taps in the order lt_decim_kernel<D, NTAPS> walks them (the reference's rates)
  hipStream_t stream = nullptr;
  lt_cf32 *d_x = nullptr, *d_y = nullptr;
  size_t x_cap = 0, y_cap = 0;
};

extern "C" {

const char *lt_version(void) { return "ltetrigger_hip 0.1 (gfx950)"; }
const char *lt_last_error(void) { return g_err; }

int lt_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ------------------------------------------------------------------ pss
int lt_pss_init(lt_pss **out, int device, int frame_size, int fft_size)
{
  if (!out) return fail(LT_ERROR_INVALID_INPUTS, "null handle pointer");
  *out = nullptr;
  if (frame_size != LT_HALF_FRAME_LEN || fft_size != LT_SYMBOL_SZ)
    return fail(LT_ERROR_INVALID_INPUTS, "only frame_size=9600, fft_size=128 (1.92 Msps) are supported");
  int rc = use_device(device);
  if (rc) return rc;
  lt_pss *q = new (std::nothrow) lt_pss;
  if (!q) return fail(LT_ERROR, "out of host memory");
  q->device = device;
  rc = get_tables(device, &q->tab);
  if (rc) { delete q; return rc; }
  hipError_t e = hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&q->d_avg, sizeof(float) * LT_CONV_LEN);
  if (e == hipSuccess) e = hipMalloc(&q->d_win, sizeof(lt_cf32) * LT_HALF_FRAME_LEN);
  if (e == hipSuccess) e = hipMalloc(&q->d_res, sizeof(FindResultHost));
  if (e == hipSuccess) e = hipMalloc(&q->d_cfo, sizeof(float));
  if (e == hipSuccess) e = hipMemset(q->d_avg, 0, sizeof(float) * LT_CONV_LEN);
  if (e != hipSuccess) {
    lt_pss_free(q);
    return fail(LT_ERROR, "lt_pss_init: %s", hipGetErrorString(e));
  }
  *out = q;
  return LT_OK;
}

int lt_pss_set_N_id_2(lt_pss *q, int N_id_2)
{
  if (!q || N_id_2 < 0 || N_id_2 > 2) return fail(LT_ERROR_INVALID_INPUTS, "invalid N_id_2 %d", N_id_2);
  q->n_id_2 = N_id_2;
  return LT_OK;
}

void lt_pss_free(lt_pss *q)
{
  if (!q) return;
  (void)hipSetDevice(q->device);
  if (q->d_avg) (void)hipFree(q->d_avg);
  if (q->d_win) (void)hipFree(q->d_win);
  if (q->d_res) (void)hipFree(q->d_res);
  if (q->d_cfo) (void)hipFree(q->d_cfo);
  if (q->stream) (void)hipStreamDestroy(q->stream);
  delete q;
}

int lt_pss_reset(lt_pss *q)
{
  if (!q) return fail(LT_ERROR_INVALID_INPUTS, "null handle");
  HIP_TRY(hipSetDevice(q->device));
  HIP_TRY(hipMemsetAsync(q->d_avg, 0, sizeof(float) * LT_CONV_LEN, q->stream));
  HIP_TRY(hipStreamSynchronize(q->stream));
  return LT_OK;
}

int lt_pss_find_pss(lt_pss *q, const lt_cf32 *in, float *psr)
{
  if (!q || !in) return fail(LT_ERROR_INVALID_INPUTS, "null argument");
  if (q->n_id_2 < 0 || q->n_id_2 > 2) return fail(LT_ERROR, "Error finding PSS peak, Must set N_id_2 first");
  HIP_TRY(hipSetDevice(q->device));
  HIP_TRY(hipMemcpyAsync(q->d_win, in, sizeof(lt_cf32) * LT_HALF_FRAME_LEN, hipMemcpyHostToDevice, q->stream));
  HIP_TRY(lt::launch_find(q->d_win, q->d_avg, q->tab, q->n_id_2, q->d_res, q->stream));
  FindResultHost r;
  HIP_TRY(hipMemcpyAsync(&r, q->d_res, sizeof r, hipMemcpyDeviceToHost, q->stream));
  HIP_TRY(hipStreamSynchronize(q->stream));
  q->peak_value = r.peak_value;
  if (psr) *psr = r.psr;
  return r.peak_pos;
}

int lt_pss_get_peak_value(lt_pss *q, float *v)
{
  if (!q || !v) return fail(LT_ERROR_INVALID_INPUTS, "null argument");
  *v = q->peak_value;
  return LT_OK;
}

int lt_pss_get_avg(lt_pss *q, float *avg)
{
  if (!q || !avg) return fail(LT_ERROR_INVALID_INPUTS, "null argument");
  HIP_TRY(hipSetDevice(q->device));
  HIP_TRY(hipMemcpyAsync(avg, q->d_avg, sizeof(float) * LT_CONV_LEN, hipMemcpyDeviceToHost, q->stream));
  HIP_TRY(hipStreamSynchronize(q->stream));
  avg[LT_CONV_LEN] = 0.f; // the 9729th entry of conv_output_avg is never written
  return LT_OK;
}

int lt_pss_cfo_compute(lt_pss *q, const lt_cf32 *pss_recv, float *cfo)
{
  if (!q || !pss_recv || !cfo) return fail(LT_ERROR_INVALID_INPUTS, "null argument");
  if (q->n_id_2 < 0 || q->n_id_2 > 2) return fail(LT_ERROR, "Must set N_id_2 first");
  HIP_TRY(hipSetDevice(q->device));
  HIP_TRY(hipMemcpyAsync(q->d_win, pss_recv, sizeof(lt_cf32) * LT_SYMBOL_SZ, hipMemcpyHostToDevice, q->stream));
  HIP_TRY(lt::launch_cfo_compute(q->d_win, q->tab, q->n_id_2, q->d_cfo, q->stream));
  HIP_TRY(hipMemcpyAsync(cfo, q->d_cfo, sizeof(float), hipMemcpyDeviceToHost, q->stream));
  HIP_TRY(hipStreamSynchronize(q->stream));
  return LT_OK;
}

// ------------------------------------------------------------------ cfo
int lt_cfo_init(lt_cfo **out, int device, int nsamples)
{
  if (!out || nsamples <= 0) return fail(LT_ERROR_INVALID_INPUTS, "invalid argument");
  *out = nullptr;
  int rc = use_device(device);
  if (rc) return rc;
  lt_cfo *h = new (std::nothrow) lt_cfo;
  if (!h) return fail(LT_ERROR, "out of host memory");
  h->device = device;
  h->nsamples = nsamples;
  rc = get_tables(device, &h->tab);
  if (rc) { delete h; return rc; }
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&h->d_buf, sizeof(lt_cf32) * (size_t)nsamples);
  if (e != hipSuccess) {
    lt_cfo_free(h);
    return fail(LT_ERROR, "lt_cfo_init: %s", hipGetErrorString(e));
  }
  *out = h;
  return LT_OK;
}

void lt_cfo_free(lt_cfo *h)
{
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->d_buf) (void)hipFree(h->d_buf);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int lt_cfo_set_last_freq(lt_cfo *h, float freq)
{
  if (!h) return fail(LT_ERROR_INVALID_INPUTS, "null handle");
  h->last_freq = freq; // the phase table is recomputed per call; kept for API parity only
  return LT_OK;
}

int lt_cfo_correct(lt_cfo *h, const lt_cf32 *in, lt_cf32 *out, float freq)
{
  if (!h || !in || !out) return fail(LT_ERROR_INVALID_INPUTS, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  const size_t bytes = sizeof(lt_cf32) * (size_t)h->nsamples;
  HIP_TRY(hipMemcpyAsync(h->d_buf, in, bytes, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(lt::launch_cfo_correct(h->d_buf, h->d_buf, h->nsamples, freq, h->tab, h->stream));
  HIP_TRY(hipMemcpyAsync(out, h->d_buf, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->last_freq = freq;
  return LT_OK;
}

// ------------------------------------------------------------------ sync / sss
int lt_sync_init(lt_sync **out, int device, int frame_size, int max_offset, int fft_size)
{
  if (!out) return fail(LT_ERROR_INVALID_INPUTS, "null handle pointer");
  *out = nullptr;
  (void)max_offset; // "set arbitrarily, does it matter?" lib/sss_impl.h:47 -- unused by the path
  if (frame_size != LT_HALF_FRAME_LEN || fft_size != LT_SYMBOL_SZ)
    return fail(LT_ERROR_INVALID_INPUTS, "only frame_size=9600, fft_size=128 (1.92 Msps) are supported");
  int rc = use_device(device);
  if (rc) return rc;
  lt_sync *q = new (std::nothrow) lt_sync;
  if (!q) return fail(LT_ERROR, "out of host memory");
  q->device = device;
  rc = get_tables(device, &q->tab);
  if (rc) { delete q; return rc; }
  hipError_t e = hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&q->d_in, sizeof(lt_cf32) * LT_HALF_FRAME_LEN);
  if (e == hipSuccess) e = hipMalloc(&q->d_state, sizeof(SyncStateHost));
  if (e == hipSuccess) e = hipMalloc(&q->d_cp, sizeof(int));
  if (e == hipSuccess) e = hipMalloc(&q->d_res, sizeof(M0M1Host));
  if (e == hipSuccess) e = hipMemset(q->d_state, 0, sizeof(SyncStateHost));
  if (e != hipSuccess) {
    lt_sync_free(q);
    return fail(LT_ERROR, "lt_sync_init: %s", hipGetErrorString(e));
  }
  *out = q;
  return LT_OK;
}

int lt_sync_set_N_id_2(lt_sync *q, int N_id_2)
{
  if (!q || N_id_2 < 0 || N_id_2 > 2) return fail(LT_ERROR_INVALID_INPUTS, "invalid N_id_2 %d", N_id_2);
  q->n_id_2 = N_id_2;
  return LT_OK;
}

void lt_sync_free(lt_sync *q)
{
  if (!q) return;
  (void)hipSetDevice(q->device);
  if (q->d_in) (void)hipFree(q->d_in);
  if (q->d_state) (void)hipFree(q->d_state);
  if (q->d_cp) (void)hipFree(q->d_cp);
  if (q->d_res) (void)hipFree(q->d_res);
  if (q->stream) (void)hipStreamDestroy(q->stream);
  delete q;
}

int lt_sync_reset(lt_sync *q)
{
  if (!q) return fail(LT_ERROR_INVALID_INPUTS, "null handle");
  HIP_TRY(hipSetDevice(q->device));
  HIP_TRY(hipMemsetAsync(q->d_state, 0, sizeof(SyncStateHost), q->stream));
  HIP_TRY(hipStreamSynchronize(q->stream));
  return LT_OK;
}

int lt_sync_detect_cp(lt_sync *q, const lt_cf32 *in, int peak_pos)
{
  if (!q || !in || peak_pos < 0 || peak_pos > LT_HALF_FRAME_LEN)
    return fail(LT_ERROR_INVALID_INPUTS, "invalid argument");
  HIP_TRY(hipSetDevice(q->device));
  if (peak_pos > 0)
    HIP_TRY(hipMemcpyAsync(q->d_in, in, sizeof(lt_cf32) * (size_t)peak_pos, hipMemcpyHostToDevice, q->stream));
  HIP_TRY(lt::launch_detect_cp(q->d_in, peak_pos, q->d_state, q->d_cp, q->stream));
  int cp = 0;
  HIP_TRY(hipMemcpyAsync(&cp, q->d_cp, sizeof cp, hipMemcpyDeviceToHost, q->stream));
  HIP_TRY(hipStreamSynchronize(q->stream));
  return cp;
}

int lt_sync_set_cp(lt_sync *q, int cp)
{
  if (!q || (cp != LT_CP_NORM && cp != LT_CP_EXT)) return fail(LT_ERROR_INVALID_INPUTS, "invalid cp");
  q->cp = cp;
  q->cp_len = (cp == LT_CP_NORM) ? 9 : 32; // SRSLTE_CP_LEN_NORM(1,128) / SRSLTE_CP_LEN_EXT(128)
  return LT_OK;
}

int lt_sync_get_cp_len(lt_sync *q) { return q ? q->cp_len.load(std::memory_order_relaxed) : fail(LT_ERROR_INVALID_INPUTS, "null handle"); }

int lt_sync_get_cp_metrics(lt_sync *q, float *n, float *e)
{
  if (!q) return fail(LT_ERROR_INVALID_INPUTS, "null handle");
  HIP_TRY(hipSetDevice(q->device));
  SyncStateHost s;
  HIP_TRY(hipMemcpyAsync(&s, q->d_state, sizeof s, hipMemcpyDeviceToHost, q->stream));
  HIP_TRY(hipStreamSynchronize(q->stream));
  if (n) *n = s.m_norm_avg;
  if (e) *e = s.m_ext_avg;
  return LT_OK;
}

int lt_sss_m0m1_partial(lt_sync *q, const lt_cf32 *input, uint32_t *m0, float *m0_value, uint32_t *m1,
                        float *m1_value)
{
  if (!q || !input || !m0 || !m1) return fail(LT_ERROR_INVALID_INPUTS, "null argument");
  if (q->n_id_2 < 0 || q->n_id_2 > 2) return fail(LT_ERROR, "Must set N_id_2 first");
  HIP_TRY(hipSetDevice(q->device));
  HIP_TRY(hipMemcpyAsync(q->d_in, input, sizeof(lt_cf32) * LT_SYMBOL_SZ, hipMemcpyHostToDevice, q->stream));
  HIP_TRY(lt::launch_sss_m0m1(q->d_in, q->tab, q->n_id_2, q->d_res, q->stream));
  M0M1Host r;
  HIP_TRY(hipMemcpyAsync(&r, q->d_res, sizeof r, hipMemcpyDeviceToHost, q->stream));
  HIP_TRY(hipStreamSynchronize(q->stream));
  *m0 = (uint32_t)r.m0;
  *m1 = (uint32_t)r.m1;
  if (m0_value) *m0_value = r.m0_value;
  if (m1_value) *m1_value = r.m1_value;
  return LT_OK;
}

int lt_sss_N_id_1(lt_sync *q, uint32_t m0, uint32_t m1)
{
  if (!q) return fail(LT_ERROR_INVALID_INPUTS, "null handle");
  // 36.211 Table 6.11.2.1-1 through the zero-defaulted 30x30 table (quirk q5); pure integer
  // lookup, identical to the device copy the batched kernel uses.
  static lt::DeviceTables *host_tab = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    host_tab = new lt::DeviceTables;
    lt::build_device_tables(host_tab);
  });
  int n = -1;
  if (m1 > m0) {
    if (m0 < 30u && m1 - 1u < 30u) n = host_tab->nid1_table[m0][m1 - 1u];
  } else {
    if (m1 < 30u && m0 - 1u < 30u) n = host_tab->nid1_table[m1][m0 - 1u];
  }
  if (n >= 0) q->n_id_1 = n;
  return n;
}

int lt_sync_get_cell_id(lt_sync *q)
{
  if (!q) return fail(LT_ERROR_INVALID_INPUTS, "null handle");
  const int n_id_1 = q->n_id_1.load(std::memory_order_relaxed);
  if (q->n_id_2 < 0 || q->n_id_2 > 2 || n_id_1 < 0 || n_id_1 > 167) return -1;
  return n_id_1 * 3 + q->n_id_2;
}

// ------------------------------------------------------------------ batch
namespace {
// samples a chain may need in front of a chunk: a call that could not start in the previous chunk lies at most
// call_reach - 1 samples before its end, and its half-frame may begin one slot earlier still (lib/pss_impl.cc:185-193)
int64_t history_capacity() { return ((int64_t)lt::call_reach() + LT_SLOT_LEN + 255) / 256 * 256; }
} // namespace

int64_t lt_batch_record_capacity(int64_t n_samples)
{
  return (n_samples + history_capacity()) / (LT_HALF_FRAME_LEN - LT_SLOT_LEN) + 2;
}

namespace {
// Window cache: every chain has slots of its own (43 KB each: one window's lag metric, or 10752 elements of a chain's lag-metric
// stream) -- as many as the windows of a run of max_samples, 256 at most -- so that what a chain gets out of the cache never
// depends on what the other chains of the launch asked for, or on the order they asked in (with one pool handed
// out first come first served, which chains of a 150-chain batch got their stream at level 4 was decided by a race).  1.7 GB
// for 64 channels x 1 s; a small batch scans from the cache and gets three times the windows (its rounds list windows ahead
// of chains that may leave them early).  Beyond kCacheBudget the quota shrinks (what lies beyond a chain's slots goes through
// the transforms, for every chain alike).  Allocated by lt_batch_create and by lt_batch_set_window_cache(1 ... 4), freed by
// lt_batch_set_window_cache(0).  Failure is not an error: the batch then runs without a cache.
constexpr int64_t kCacheBudget = 48ll << 30;
void free_window_cache(lt_batch *b)
{
  if (b->d_cache) (void)hipFree(b->d_cache);
  if (b->d_slot_owner) (void)hipFree(b->d_slot_owner);
  if (b->d_slot_pos) (void)hipFree(b->d_slot_pos);
  b->d_cache = nullptr;
  b->d_slot_owner = nullptr;
  b->d_slot_pos = nullptr;
  b->cache_slots = 0;
  b->cache_quota = 0;
}
void alloc_window_cache(lt_batch *b)
{
  if (b->d_cache) return;
  const int64_t nchains = (int64_t)b->n_channels * 3;
  const int64_t windows = b->max_samples / LT_HALF_FRAME_LEN + 2;
  int64_t quota = std::min<int64_t>(windows, lt::cache_max_per_chain());
  quota = std::min<int64_t>(quota, kCacheBudget / lt::cache_slot_bytes() / nchains);
  // (batches that scan from the cache: a round's windows of every chain have slots at once -- at most scan_max_round() per chain;
  // the per-chain part's requests come after the rounds and use the same slots again)
  const int64_t per_chain = b->n_channels <= lt::cached_scan_max_channels() ? std::max<int64_t>(lt::scan_max_round(), quota) : quota;
  if (quota < 1 || nchains * per_chain > 0x7fffffff / 16) return;
  b->cache_slots = (int)(nchains * per_chain);
  b->cache_quota = (int)quota;
  hipError_t ce = hipMalloc(&b->d_cache, (size_t)b->cache_slots * (size_t)lt::cache_slot_bytes());
  if (ce == hipSuccess) ce = hipMalloc(&b->d_slot_owner, (size_t)b->cache_slots * 2 * sizeof(int));
  if (ce == hipSuccess) ce = hipMalloc(&b->d_slot_pos, (size_t)b->cache_slots * sizeof(int64_t));
  if (ce != hipSuccess) {
    (void)hipGetLastError();
    free_window_cache(b);
  }
}
} // namespace

int lt_batch_create(lt_batch **out, int device, int n_channels, int64_t max_samples,
                    const lt_chain_params *params)
{
  if (!out || n_channels <= 0 || max_samples < LT_HALF_FRAME_LEN || !params)
    return fail(LT_ERROR_INVALID_INPUTS, "invalid argument");
  *out = nullptr;
  if (params->track_after <= 0 || params->track_every < 0)
    return fail(LT_ERROR_INVALID_INPUTS, "invalid chain parameters");
  if (lt_batch_record_capacity(max_samples) > 4 * 65535)
    return fail(LT_ERROR_INVALID_INPUTS, "max_samples %lld too large for one run (limit: 37 minutes at 1.92 Msps)", (long long)max_samples);
  int rc = use_device(device);
  if (rc) return rc;
  lt_batch *b = new (std::nothrow) lt_batch;
  if (!b) return fail(LT_ERROR, "out of host memory");
  b->device = device;
  b->n_channels = n_channels;
  b->max_samples = max_samples;
  b->params = *params;
  b->rec_cap = (int)lt_batch_record_capacity(max_samples);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) b->n_cus = prop.multiProcessorCount;
    else (void)hipGetLastError();
  }
  b->hist_cap = history_capacity();
  rc = get_tables(device, &b->tab);
  if (rc) { delete b; return rc; }
  // Everything a run needs is allocated here, so that no run can fail -- or change its execution strategy -- for
  // want of memory later.
  const size_t nchains = (size_t)n_channels * 3;
  hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&b->d_state, lt::chain_state_bytes() * nchains);
  if (e == hipSuccess) e = hipMalloc(&b->d_recs, sizeof(lt_record) * nchains * (size_t)b->rec_cap);
  if (e == hipSuccess) e = hipMalloc(&b->d_count, sizeof(int) * nchains);
  if (e == hipSuccess) e = hipMalloc(&b->d_jobs, (size_t)lt::scan_job_bytes() * nchains * 2);
  if (e == hipSuccess) e = hipMalloc(&b->d_njobs, sizeof(int) * (size_t)lt::scan_counter_ints());
  if (e == hipSuccess) e = hipMalloc(&b->d_parts, (size_t)lt::scan_parts_bytes_per_chain() * nchains);
  if (e == hipSuccess) e = hipMalloc(&b->d_cp, (size_t)lt::per_record_scratch_bytes() * nchains * (size_t)b->rec_cap);
  if (e == hipSuccess) alloc_window_cache(b); // (an accelerator, not a necessity: without the memory for it the per-chain kernel runs in one pass)
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipMalloc(&b->d_hist[i], sizeof(lt_cf32) * (size_t)n_channels * (size_t)b->hist_cap);
  if (e == hipSuccess) e = hipMemset(b->d_state, 0, lt::chain_state_bytes() * nchains);
  if (e == hipSuccess) e = hipMemset(b->d_count, 0, sizeof(int) * nchains);
  for (int i = 0; i < lt_batch::kPhaseRing && e == hipSuccess; ++i)
    for (int j = 0; j < 4 && e == hipSuccess; ++j) e = hipEventCreate(&b->ev[i][j]);
  if (e == hipSuccess) { // (a convenience: without the word every run has its two passes)
    if (hipHostMalloc(reinterpret_cast<void **>(&b->h_wait_hint), 2 * sizeof(int), hipHostMallocMapped) == hipSuccess &&
        hipHostGetDevicePointer(reinterpret_cast<void **>(&b->d_wait_hint), b->h_wait_hint, 0) == hipSuccess) {
      b->h_wait_hint[0] = b->h_wait_hint[1] = 0;
    } else {
      (void)hipGetLastError();
      if (b->h_wait_hint) (void)hipHostFree(b->h_wait_hint);
  if (b->h_recs_pin) (void)hipHostFree(b->h_recs_pin);
  if (b->h_count_pin) (void)hipHostFree(b->h_count_pin);
      b->h_wait_hint = nullptr;
      b->d_wait_hint = nullptr;
    }
  }
  b->ev0 = b->ev[0][0];
  b->ev1 = b->ev[0][3];
  b->cur_recs = b->d_recs;
  b->cur_count = b->d_count;
  if (e != hipSuccess) {
    (void)hipGetLastError();
    lt_batch_free(b);
    return fail(LT_ERROR, "lt_batch_create: %s", hipGetErrorString(e));
  }
  *out = b;
  return LT_OK;
}

void lt_batch_free(lt_batch *b)
{
  if (!b) return;
  (void)hipSetDevice(b->device);
  if (b->last_stream) (void)hipStreamSynchronize(b->last_stream);
  if (b->side) (void)hipStreamSynchronize(b->side);
  void *bufs[] = {b->d_iq, b->d_state, b->d_recs, b->d_count, b->d_hf, b->d_jobs, b->d_njobs, b->d_parts, b->d_cp, b->d_hist[0], b->d_hist[1], b->d_cache, b->d_slot_owner, b->d_slot_pos,
                  b->d_recs_b, b->d_count_b};
  for (void *p : bufs)
    if (p) (void)hipFree(p);
  for (auto &set : b->ev)
    for (hipEvent_t e : set)
      if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : b->rec_done)
    if (e) (void)hipEventDestroy(e);
  if (b->h_wait_hint) (void)hipHostFree(b->h_wait_hint);
  if (b->copy) (void)hipStreamDestroy(b->copy);
  if (b->side) (void)hipStreamDestroy(b->side);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
}

int lt_batch_reset(lt_batch *b)
{
  if (!b) return fail(LT_ERROR_INVALID_INPUTS, "null handle");
  // The state is cleared on the stream of the next run, right in front of it: a memset queued here, on whatever
  // stream ran last, would not be ordered against a run on another stream.
  b->pending_reset = true;
  b->t0 = 0;
  b->hist_len = 0;
  return LT_OK;
}

int lt_batch_set_psr_threshold(lt_batch *b, float threshold)
{
  if (!b) return fail(LT_ERROR_INVALID_INPUTS, "null handle");
  b->params.psr_threshold = threshold;
  return LT_OK;
}

int lt_batch_enable_halfframes(lt_batch *b, int64_t max_per_chain)
{
  if (!b || max_per_chain < 0) return fail(LT_ERROR_INVALID_INPUTS, "invalid argument");
  if (max_per_chain > 0 && b->rec_cap > 65535)
    return fail(LT_ERROR_INVALID_INPUTS, "half-frame capture needs max_samples below %lld (one grid row per record slot: at most 65535; this batch has %d)",
                (long long)65535 * (LT_HALF_FRAME_LEN - LT_SLOT_LEN), b->rec_cap);
  HIP_TRY(hipSetDevice(b->device));
  if (b->last_stream) HIP_TRY(hipStreamSynchronize(b->last_stream));
  if (b->side_busy) HIP_TRY(hipStreamSynchronize(b->side));
  if (b->d_hf) {
    HIP_TRY(hipFree(b->d_hf));
    b->d_hf = nullptr;
  }
  b->hf_cap = 0;
  if (max_per_chain > 0) {
    const size_t bytes = sizeof(lt_cf32) * LT_HALF_FRAME_LEN * (size_t)max_per_chain * (size_t)b->n_channels * 3;
    HIP_TRY(hipMalloc(&b->d_hf, bytes));
    b->hf_cap = max_per_chain;
  }
  return LT_OK;
}

namespace {
// One chunk through the path.  restart: the chunk is a capture of its own (positions restart at 0, no history)
// with the chain state left by the previous run -- lt_batch_run; otherwise it continues the stream -- lt_batch_push.
int batch_chunk(lt_batch *b, const lt_cf32 *iq, int iq_on_device, int64_t n_samples, bool restart, int flush, void *hip_stream)
{
  if (!b || !iq) return fail(LT_ERROR_INVALID_INPUTS, "null argument");
  if (n_samples < 0 || n_samples > b->max_samples)
    return fail(LT_ERROR_INVALID_INPUTS, "n_samples %lld exceeds max_samples %lld", (long long)n_samples,
                (long long)b->max_samples);
  HIP_TRY(hipSetDevice(b->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : b->stream;
  // runs on different streams are ordered by the host here: state, records and history are shared
  if (b->last_stream && b->last_stream != s) HIP_TRY(hipStreamSynchronize(b->last_stream));
  const size_t nchains = (size_t)b->n_channels * 3;
  // Overlap: for captures of their own handed over on the device (a streamed chunk's per-record kernels read the
  // history buffer the next chunk rewrites; a host buffer is staged through one device buffer).
  const bool ov = b->overlap && restart && iq_on_device;
  const int set = ov ? (int)(b->n_runs & 1) : 0;
  if (b->side_busy && !ov) { // back to one stream: let them finish
    HIP_TRY(hipStreamSynchronize(b->side));
    b->side_busy = false;
  }
  if (ov && b->rec_done_valid[set]) HIP_TRY(hipStreamWaitEvent(s, b->rec_done[set], 0)); // this set's last user, two runs ago
  if (b->pending_reset) {
    int *count = set ? b->d_count_b : b->d_count; // (this set's record counts: cleared by the same launch)
    if (ov) { // the per-record kernels' members on their stream, behind the previous run's
      HIP_TRY(lt::launch_reset(b->d_state, (int)nchains, 1, count, s));
      HIP_TRY(lt::launch_reset(b->d_state, (int)nchains, 2, nullptr, b->side));
      b->side_busy = true;
    } else {
      // (measured and dropped, round 5: the run's first launch -- lt_plan_kernel, a single workgroup -- clearing the state
      // itself: one workgroup takes 32 us for the 83 K words 192 workgroups clear in 4.8)
      HIP_TRY(lt::launch_reset(b->d_state, (int)nchains, 3, count, s));
    }
    b->pending_reset = false;
  }
  if (restart) {
    b->t0 = 0;
    b->hist_len = 0;
  }
  const lt_cf32 *d_iq = iq;
  if (!iq_on_device) {
    const size_t bytes = sizeof(lt_cf32) * (size_t)b->n_channels * (size_t)n_samples;
    if (!b->d_iq)
      HIP_TRY(hipMalloc(&b->d_iq, sizeof(lt_cf32) * (size_t)b->n_channels * (size_t)b->max_samples));
    if (bytes) HIP_TRY(hipMemcpyAsync(b->d_iq, iq, bytes, hipMemcpyHostToDevice, s));
    d_iq = b->d_iq;
  }
  lt::RunLaunch R;
  R.iq = d_iq;
  R.ch_stride = n_samples;
  R.t0 = b->t0;
  R.n = n_samples;
  R.hist = b->hist_len > 0 ? b->d_hist[b->hist_sel] : nullptr;
  R.hist_stride = b->hist_cap;
  R.hist_len = b->hist_len;
  const int64_t hist_new = restart ? 0 : std::min(b->hist_cap, b->hist_len + n_samples);
  R.hist_new = hist_new > 0 ? b->d_hist[b->hist_sel ^ 1] : nullptr;
  R.hist_len_new = hist_new;
  R.flush = flush;
  R.n_channels = b->n_channels;
  R.state = b->d_state;
  R.recs = set ? b->d_recs_b : b->d_recs;
  R.rec_cap = b->rec_cap;
  R.rec_count = set ? b->d_count_b : b->d_count;
  R.list_slot = set;
  R.rebase = restart ? 1 : 0;
  R.out_hf = b->d_hf;
  R.out_cap = (int)b->hf_cap;
  R.tab = b->tab;
  R.psr_threshold = b->params.psr_threshold;
  R.track_after = b->params.track_after;
  R.track_every = b->params.track_every;
  R.scan = b->mode;
  R.jobs = b->d_jobs;
  R.n_jobs = b->d_njobs;
  R.parts = b->d_parts;
  R.seg_first = b->seg_first;
  R.seg_len = b->seg_len;
  R.scan_groups = b->scan_groups;
  R.n_cus = b->n_cus;
  R.cp_scratch = b->d_cp;
  R.cache = b->use_cache ? b->d_cache : nullptr;
  R.cache_level = b->use_cache > 2 ? 2 : b->use_cache;
  R.stream = b->use_cache == 4;
  R.slot_owner = b->d_slot_owner;
  R.slot_pos = b->d_slot_pos;
  R.slot_cap = b->cache_slots;
  R.slot_quota = b->cache_quota;
  R.records_delay_us = b->records_delay_us;
  R.wait_hint = b->d_wait_hint;
  R.run_id = (int)((b->n_runs + 1) & 0x3fffffff);
  R.n_launches = &b->last_launches;
  // h_wait_hint[0] = the last run in which a chain asked, [1] = the last run whose per-chain kernel is known to have run:
  // everything up to run [1] - 1 is through (stream order), so "nobody asked in the latest run known in full" is
  // [0] <= [1] - 2.  The host may be any number of runs ahead of the device; until the first report (asked == 0) a run has
  // one pass.  (Level 3 only: the other levels do not look at the words.)
  {
    const int asked = b->h_wait_hint ? reinterpret_cast<volatile int *>(b->h_wait_hint)[0] : 0;
    const int done = b->h_wait_hint ? reinterpret_cast<volatile int *>(b->h_wait_hint)[1] : 0;
    // level 3: one pass until a chain has asked, then two passes while chains keep asking (the host's position relative to
    // the device picks the path: timing only at this level, whose two paths give the same bits); levels 1, 2 and 4: two passes
    // always -- at level 4 a search out of a stream and one through the transforms differ in the low bits of PSR, so which of
    // the two a run takes is decided by the chains' state on the device alone (pass 0), never by the host's cue
    R.two_pass = b->use_cache != 3 || !b->h_wait_hint || (asked > 0 && asked > done - 2);
    R.ask_early = b->use_cache != 3 && b->use_cache != 0; // (level 3's second pass hangs on pass 0's report to the host)
    if (b->n_runs == 0x3fffffff && b->h_wait_hint) b->h_wait_hint[0] = b->h_wait_hint[1] = 0; // (the run numbers wrap)
  }
  const int slot = (int)(b->n_runs % lt_batch::kPhaseRing);
  // (an event between two launches costs the stream ~5 us: the stage events are there for lt_batch_phase_ms only --
  // and the one behind the per-chain kernel for the per-record kernels' stream in overlap mode)
  R.ev_scan_done = b->phase_timing >= 2 ? b->ev[slot][1] : nullptr;
  R.ev_chain_done = (b->phase_timing >= 2 || ov) ? b->ev[slot][2] : nullptr;
  b->ev_phased[slot] = b->phase_timing >= 2;
  b->ev_timed = b->phase_timing >= 1;
  b->ev0 = b->ev[slot][0];
  b->ev1 = b->ev[slot][3];
  b->n_runs++;
  b->set_run[set] = b->n_runs;
  if (b->ev_timed) HIP_TRY(hipEventRecord(b->ev0, s));
  HIP_TRY(lt::launch_run(R, s, ov ? b->side : nullptr));
  if (b->ev_timed) HIP_TRY(hipEventRecord(b->ev1, ov ? b->side : s));
  if (ov) {
    HIP_TRY(hipEventRecord(b->rec_done[set], b->side));
    b->rec_done_valid[set] = true;
    b->side_busy = true;
  } else if (b->overlap && b->rec_done[set]) {
    // A run that is not overlapped itself (host buffer, lt_batch_push) between overlapped ones: its records are complete
    // when its stream is through -- the event says so to lt_batch_collect_previous (and to the overlapped run two runs on
    // that writes this set again), which would otherwise wait on whatever an older run left in rec_done[set].
    HIP_TRY(hipEventRecord(b->rec_done[set], s));
    b->rec_done_valid[set] = true;
  } else {
    b->rec_done_valid[set] = false;
  }
  b->cur_recs = R.recs;
  b->cur_count = R.rec_count;
  if (!restart) {
    b->t0 += n_samples;
    if (hist_new > 0) b->hist_sel ^= 1;
    b->hist_len = hist_new;
  }
  b->last_stream = s;
  b->ran = true;
  return LT_OK;
}
} // namespace

int lt_batch_run(lt_batch *b, const lt_cf32 *iq, int iq_on_device, int64_t n_samples, void *hip_stream)
{
  return batch_chunk(b, iq, iq_on_device, n_samples, true, 1, hip_stream);
}

int lt_batch_push(lt_batch *b, const lt_cf32 *iq, int iq_on_device, int64_t n_samples, int end_of_stream, void *hip_stream)
{
  return batch_chunk(b, iq, iq_on_device, n_samples, false, end_of_stream ? 1 : 0, hip_stream);
}

int lt_batch_phase_ms(lt_batch *b, int runs_back, float ms[4])
{
  if (!b || !ms) return fail(LT_ERROR_INVALID_INPUTS, "lt_batch_phase_ms: null argument");
  if (runs_back < 0 || runs_back >= lt_batch::kPhaseRing || (uint64_t)runs_back >= b->n_runs)
    return fail(LT_ERROR_INVALID_INPUTS, "lt_batch_phase_ms: run %d back is not on record (kept: the last %d of %llu runs)",
                runs_back, lt_batch::kPhaseRing, (unsigned long long)b->n_runs);
  int rc = use_device(b->device);
  if (rc) return rc;
  const int slot = (int)((b->n_runs - 1 - (uint64_t)runs_back) % lt_batch::kPhaseRing);
  if (!b->ev_phased[slot])
    return fail(LT_ERROR, "lt_batch_phase_ms: that run was made with the stage events off (lt_batch_set_phase_timing)");
  hipEvent_t *e = b->ev[slot];
  HIP_TRY(hipEventSynchronize(e[3]));
  ms[1] = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms[0], e[0], e[1]));
  HIP_TRY(hipEventElapsedTime(&ms[2], e[1], e[2]));
  HIP_TRY(hipEventElapsedTime(&ms[3], e[2], e[3]));
  return LT_OK;
}

int lt_batch_last_launch_count(lt_batch *b, int *n_launches)
{
  if (!b || !n_launches) return fail(LT_ERROR_INVALID_INPUTS, "null argument");
  if (!b->ran) return fail(LT_ERROR, "no run yet");
  *n_launches = b->last_launches;
  return LT_OK;
}

int lt_batch_last_kernel_ms(lt_batch *b, float *ms)
{
  if (!b || !ms) return fail(LT_ERROR_INVALID_INPUTS, "null argument");
  if (!b->ran) return fail(LT_ERROR, "no run yet");
  if (!b->ev_timed) return fail(LT_ERROR, "lt_batch_last_kernel_ms: the last run was made without events (lt_batch_set_phase_timing 0)");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipEventSynchronize(b->ev1));
  HIP_TRY(hipEventElapsedTime(ms, b->ev0, b->ev1));
  return LT_OK;
}

namespace {
// -> where the device's records and counts land on the host (pinned if it can be had)
void collect_area(lt_batch *b, lt_record **recs, int **count)
{
  const size_t nchains = (size_t)b->n_channels * 3;
  if (!b->h_recs_pin && b->h_recs.empty()) {
    if (hipHostMalloc(reinterpret_cast<void **>(&b->h_recs_pin), sizeof(lt_record) * nchains * (size_t)b->rec_cap, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&b->h_count_pin), sizeof(int) * nchains, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      if (b->h_recs_pin) (void)hipHostFree(b->h_recs_pin);
      if (b->h_count_pin) (void)hipHostFree(b->h_count_pin);
      b->h_recs_pin = nullptr;
      b->h_count_pin = nullptr;
    }
  }
  if (b->h_recs_pin) {
    *recs = b->h_recs_pin;
    *count = b->h_count_pin;
    return;
  }
  b->h_count.resize(nchains);
  b->h_recs.resize(nchains * (size_t)b->rec_cap);
  *recs = b->h_recs.data();
  *count = b->h_count.data();
}
} // namespace

int lt_batch_collect(lt_batch *b, lt_record *recs, int64_t cap_per_chain, int64_t *counts)
{
  if (!b || !counts || (cap_per_chain > 0 && !recs)) return fail(LT_ERROR_INVALID_INPUTS, "null argument");
  if (!b->ran) return fail(LT_ERROR, "no run to collect");
  HIP_TRY(hipSetDevice(b->device));
  hipStream_t s = b->last_stream;
  if (b->side_busy) { // the last run's per-record kernels
    HIP_TRY(hipStreamSynchronize(b->side));
    b->side_busy = false;
  }
  const size_t nchains = (size_t)b->n_channels * 3;
  lt_record *hr = nullptr;
  int *hc = nullptr;
  collect_area(b, &hr, &hc);
  HIP_TRY(hipMemcpyAsync(hc, b->cur_count, sizeof(int) * nchains, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(hr, b->cur_recs, sizeof(lt_record) * nchains * (size_t)b->rec_cap, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (size_t c = 0; c < nchains; ++c) {
    int64_t n = hc[c];
    counts[c] = n;
    if (n > cap_per_chain) n = cap_per_chain;
    if (n > 0) memcpy(recs + c * (size_t)cap_per_chain, hr + c * (size_t)b->rec_cap, sizeof(lt_record) * (size_t)n);
  }
  return LT_OK;
}

int lt_batch_collect_previous(lt_batch *b, lt_record *recs, int64_t cap_per_chain, int64_t *counts)
{
  if (!b || !counts || (cap_per_chain > 0 && !recs)) return fail(LT_ERROR_INVALID_INPUTS, "null argument");
  if (b->n_runs < 2) return fail(LT_ERROR, "lt_batch_collect_previous: fewer than two runs so far");
  // the run before the last one wrote the other set if and only if both were overlapped runs
  const int set = b->cur_recs == b->d_recs ? 1 : 0;
  if (!b->d_recs_b || b->set_run[set] != b->n_runs - 1 || !b->rec_done_valid[set])
    return fail(LT_ERROR, "lt_batch_collect_previous: the last run has overwritten the records of the one before "
                          "(two sets of records exist only while lt_batch_set_overlap is on and the last run was an overlapped one)");
  HIP_TRY(hipSetDevice(b->device));
  if (!b->copy) HIP_TRY(hipStreamCreateWithFlags(&b->copy, hipStreamNonBlocking));
  // that run's per-record kernels are through when its event is (the last run's may still be under way)
  HIP_TRY(hipStreamWaitEvent(b->copy, b->rec_done[set], 0));
  const size_t nchains = (size_t)b->n_channels * 3;
  lt_record *hr = nullptr;
  int *hc = nullptr;
  collect_area(b, &hr, &hc);
  HIP_TRY(hipMemcpyAsync(hc, set ? b->d_count_b : b->d_count, sizeof(int) * nchains, hipMemcpyDeviceToHost, b->copy));
  HIP_TRY(hipMemcpyAsync(hr, set ? b->d_recs_b : b->d_recs, sizeof(lt_record) * nchains * (size_t)b->rec_cap, hipMemcpyDeviceToHost, b->copy));
  HIP_TRY(hipStreamSynchronize(b->copy));
  for (size_t c = 0; c < nchains; ++c) {
    int64_t n = hc[c];
    counts[c] = n;
    if (n > cap_per_chain) n = cap_per_chain;
    if (n > 0) memcpy(recs + c * (size_t)cap_per_chain, hr + c * (size_t)b->rec_cap, sizeof(lt_record) * (size_t)n);
  }
  return LT_OK;
}

int lt_batch_get_halfframes(lt_batch *b, int chain, lt_cf32 *out, int64_t cap, int64_t *count)
{
  if (!b || !count || chain < 0 || chain >= b->n_channels * 3) return fail(LT_ERROR_INVALID_INPUTS, "invalid argument");
  if (!b->d_hf || !b->ran) return fail(LT_ERROR, "half-frame capture not enabled or no run yet");
  HIP_TRY(hipSetDevice(b->device));
  hipStream_t s = b->last_stream;
  HIP_TRY(hipStreamSynchronize(s));
  if (b->side_busy) {
    HIP_TRY(hipStreamSynchronize(b->side));
    b->side_busy = false;
  }
  // number of emitted half-frames = records with LT_F_EMITTED in the last run
  int nrec = 0;
  HIP_TRY(hipMemcpy(&nrec, b->cur_count + chain, sizeof nrec, hipMemcpyDeviceToHost));
  std::vector<lt_record> r((size_t)nrec);
  if (nrec)
    HIP_TRY(hipMemcpy(r.data(), b->cur_recs + (size_t)chain * b->rec_cap, sizeof(lt_record) * (size_t)nrec,
                      hipMemcpyDeviceToHost));
  int64_t n = 0;
  for (const lt_record &x : r)
    if (x.flags & LT_F_EMITTED) n++;
  if (n > b->hf_cap) n = b->hf_cap;
  *count = n;
  if (n > cap) n = cap;
  if (n > 0 && out)
    HIP_TRY(hipMemcpy(out, b->d_hf + (size_t)chain * (size_t)b->hf_cap * LT_HALF_FRAME_LEN,
                      sizeof(lt_cf32) * LT_HALF_FRAME_LEN * (size_t)n, hipMemcpyDeviceToHost));
  return LT_OK;
}

// 1 (default) = grid scan of the idle chains + per-chain kernel for the others; 0 = per-chain kernel for every
// chain.  Same arithmetic on the same strips either way: identical records.
int lt_batch_set_mode(lt_batch *b, int mode)
{
  if (!b || (mode != 0 && mode != 1)) return fail(LT_ERROR_INVALID_INPUTS, "mode must be 0 or 1");
  b->mode = mode;
  return LT_OK;
}

int lt_batch_get_mode(lt_batch *b) { return b ? b->mode : fail(LT_ERROR_INVALID_INPUTS, "null handle"); }

int lt_batch_set_scan_groups(lt_batch *b, int strips_per_workgroup)
{
  if (!b || strips_per_workgroup < 0 || strips_per_workgroup > 3)
    return fail(LT_ERROR_INVALID_INPUTS, "strips per scan workgroup: 0 (automatic), 1, 2 or 3");
  b->scan_groups = strips_per_workgroup;
  return LT_OK;
}

int lt_batch_set_overlap(lt_batch *b, int on)
{
  if (!b || (on != 0 && on != 1)) return fail(LT_ERROR_INVALID_INPUTS, "overlap must be 0 or 1");
  HIP_TRY(hipSetDevice(b->device));
  if (on && !b->side) {
    const size_t nchains = (size_t)b->n_channels * 3;
    // lowest priority: the scan rounds' workgroups go first wherever both streams have work, the per-record kernels
    // take what the resolvers and the ends of the rounds leave free
    int prio_low = 0, prio_high = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&b->side, hipStreamNonBlocking, prio_low);
    if (e == hipSuccess) e = hipMalloc(&b->d_recs_b, sizeof(lt_record) * nchains * (size_t)b->rec_cap);
    if (e == hipSuccess) e = hipMalloc(&b->d_count_b, sizeof(int) * nchains);
    if (e == hipSuccess) e = hipMemset(b->d_count_b, 0, sizeof(int) * nchains);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&b->rec_done[i], hipEventDisableTiming);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail(LT_ERROR, "lt_batch_set_overlap: %s", hipGetErrorString(e));
    }
  }
  b->overlap = on;
  return LT_OK;
}

int lt_batch_cache_stats(lt_batch *b, int64_t stats[7])
{
  if (!b || !stats) return fail(LT_ERROR_INVALID_INPUTS, "null argument");
  if (!b->ran) return fail(LT_ERROR, "no run yet");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipStreamSynchronize(b->last_stream));
  int v[7];
  HIP_TRY(hipMemcpy(v, b->d_njobs + lt::cache_counter_index(), sizeof v, hipMemcpyDeviceToHost));
  for (int i = 0; i < 7; ++i) stats[i] = v[i];
  return LT_OK;
}

int lt_batch_set_window_cache(lt_batch *b, int on)
{
  if (!b || on < 0 || on > 4) return fail(LT_ERROR_INVALID_INPUTS, "window cache must be 0 ... 4");
  HIP_TRY(hipSetDevice(b->device));
  if ((on == 0) != (b->d_cache == nullptr)) { // the cache's memory goes with level 0 and comes back with 1 / 2
    if (b->last_stream) HIP_TRY(hipStreamSynchronize(b->last_stream));
    if (on == 0) free_window_cache(b);
    else alloc_window_cache(b);
  }
  b->use_cache = on;
  return LT_OK;
}

int lt_batch_set_phase_timing(lt_batch *b, int level)
{
  if (!b || level < 0 || level > 2) return fail(LT_ERROR_INVALID_INPUTS, "phase timing must be 0, 1 or 2");
  b->phase_timing = level;
  return LT_OK;
}

int lt_batch_debug_hold_records(lt_batch *b, int microseconds)
{
  if (!b || microseconds < 0 || microseconds > 100000) return fail(LT_ERROR_INVALID_INPUTS, "hold must be within [0, 100000] us");
  b->records_delay_us = microseconds;
  return LT_OK;
}

int lt_batch_set_scan_segments(lt_batch *b, int first, int later)
{
  if (!b || first < 1 || later < 0 || first > lt::scan_max_round() || later > lt::scan_max_round())
    return fail(LT_ERROR_INVALID_INPUTS, "windows per scan round must be in [1, %d] (later: 0 = automatic)", lt::scan_max_round());
  b->seg_first = first;
  b->seg_len = later;
  return LT_OK;
}

// ------------------------------------------------------------------ decimator
int lt_decim_create(lt_decim **out, int device, int decimation)
{
  if (!out || decimation < 2 || decimation > 64) return fail(LT_ERROR_INVALID_INPUTS, "decimation must be in [2,64]");
  *out = nullptr;
  int rc = use_device(device);
  if (rc) return rc;
  lt_decim *d = new (std::nothrow) lt_decim;
  if (!d) return fail(LT_ERROR, "out of host memory");
  d->device = device;
  d->decimation = decimation;
  d->taps = lt::decim_design(decimation);
  hipError_t e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&d->d_taps, sizeof(float) * d->taps.size());
  if (e == hipSuccess) e = hipMemcpy(d->d_taps, d->taps.data(), sizeof(float) * d->taps.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess && lt::decim_has_tap_table(decimation, (int)d->taps.size())) {
    const std::vector<float> tt = lt::decim_tap_table(decimation, d->taps.data(), (int)d->taps.size());
    e = hipMalloc(&d->d_tap_table, sizeof(float) * tt.size());
    if (e == hipSuccess) e = hipMemcpy(d->d_tap_table, tt.data(), sizeof(float) * tt.size(), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    lt_decim_free(d);
    return fail(LT_ERROR, "lt_decim_create: %s", hipGetErrorString(e));
  }
  *out = d;
  return LT_OK;
}

void lt_decim_free(lt_decim *d)
{
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->d_taps) (void)hipFree(d->d_taps);
  if (d->d_tap_table) (void)hipFree(d->d_tap_table);
  if (d->d_x) (void)hipFree(d->d_x);
  if (d->d_y) (void)hipFree(d->d_y);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
}

int lt_decim_ntaps(lt_decim *d) { return d ? (int)d->taps.size() : fail(LT_ERROR_INVALID_INPUTS, "null handle"); }

int lt_decim_get_taps(lt_decim *d, float *taps)
{
  if (!d || !taps) return fail(LT_ERROR_INVALID_INPUTS, "null argument");
  memcpy(taps, d->taps.data(), sizeof(float) * d->taps.size());
  return LT_OK;
}

int lt_decim_run(lt_decim *d, const lt_cf32 *x, lt_cf32 *y, int n_channels, int64_t n_in, int on_device,
                 void *hip_stream)
{
  if (!d || !x || !y || n_channels <= 0 || n_in < 0) return fail(LT_ERROR_INVALID_INPUTS, "invalid argument");
  HIP_TRY(hipSetDevice(d->device));
  const int64_t n_out = n_in / d->decimation;
  if (on_device) {
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : d->stream;
    HIP_TRY(lt::launch_decim(x, y, n_channels, n_in, d->decimation, d->d_taps, (int)d->taps.size(), d->d_tap_table, s));
    return LT_OK;
  }
  const size_t xb = sizeof(lt_cf32) * (size_t)n_channels * (size_t)n_in;
  const size_t yb = sizeof(lt_cf32) * (size_t)n_channels * (size_t)n_out;
  if (xb > d->x_cap) {
    if (d->d_x) HIP_TRY(hipFree(d->d_x));
    d->d_x = nullptr; d->x_cap = 0;
    HIP_TRY(hipMalloc(&d->d_x, xb));
    d->x_cap = xb;
  }
  if (yb > d->y_cap) {
    if (d->d_y) HIP_TRY(hipFree(d->d_y));
    d->d_y = nullptr; d->y_cap = 0;
    HIP_TRY(hipMalloc(&d->d_y, yb ? yb : 8));
    d->y_cap = yb;
  }
  if (xb) HIP_TRY(hipMemcpyAsync(d->d_x, x, xb, hipMemcpyHostToDevice, d->stream));
  HIP_TRY(lt::launch_decim(d->d_x, d->d_y, n_channels, n_in, d->decimation, d->d_taps, (int)d->taps.size(), d->d_tap_table, d->stream));
  if (yb) HIP_TRY(hipMemcpyAsync(y, d->d_y, yb, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  return LT_OK;
}

} // extern "C"
