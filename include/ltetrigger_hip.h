/*
 * ltetrigger_hip.h -- C-ABI of the MI355X (gfx950) LTE cell-search hot path.
 *
 * This is the drop-in boundary for NTIA/gr-ltetrigger's PSS/SSS path: every entry
 * point replaces one srsLTE call that the reference's GNU Radio blocks make
 * (citations "lib/..." are into the reference tree).  The reference pokes srsLTE
 * structs directly (d_cfo.last_freq, sync->cp_len, sync->m0 ...: lib/pss_impl.cc:148,
 * lib/sss_impl.cc:110-122), so a link-time swap of libsrslte is impossible; the
 * block sources are instead written against these opaque handles.
 *
 * Conventions
 *  - plain C, no exceptions, no exit(): int return 0 = LT_OK, -1 = LT_ERROR,
 *    -2 = LT_ERROR_INVALID_INPUTS (srsLTE's SRSLTE_SUCCESS / _ERROR / _ERROR_INVALID_INPUTS);
 *    lt_last_error() returns a thread-local description of the last failure.
 *  - lt_cf32 is cf_t: interleaved float32 I/Q, 8 bytes (io signatures, lib/pss_impl.cc:61-63).
 *  - `const lt_cf32 *` arguments of the per-call API are HOST pointers that only need to
 *    stay valid during the call (GNU Radio owns input_items/output_items for the call
 *    only); the library stages them to the device itself and returns synchronously.
 *  - every handle owns its device buffers and HIP stream; distinct handles may be used
 *    from distinct threads concurrently (GNU Radio runs one thread per block); one handle
 *    must not be used from two threads at once -- except for the plain getters
 *    lt_pss_get_peak_value, lt_sync_get_cp_len and lt_sync_get_cell_id, which read host-side
 *    fields and may be polled from another thread while the handle is busy (the reference's
 *    getters are called cross-thread without a lock, lib/pss_impl.h:95-100).
 *  - there is NO CPU fallback: every compute entry point fails with LT_ERROR when no
 *    gfx950 device is usable.
 */
#ifndef LTETRIGGER_HIP_H
#define LTETRIGGER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LT_OK 0
#define LT_ERROR (-1)
#define LT_ERROR_INVALID_INPUTS (-2)

#define LT_SLOT_LEN 960          /* lib/pss_impl.h:52 */
#define LT_HALF_FRAME_LEN 9600   /* lib/pss_impl.h:53 */
#define LT_SYMBOL_SZ 128         /* lib/pss_impl.h:55 */
#define LT_CONV_LEN 9728         /* frame_size + fft_size lags computed per search */
#define LT_NLAGS 9726            /* lags that enter the EMA / arg-max */
#define LT_MOVING_AVG_SZ 200     /* lib/pss_impl.h:31 */
#define LT_CP_NORM 0
#define LT_CP_EXT 1

typedef struct { float re, im; } lt_cf32;

typedef struct lt_pss lt_pss;     /* replaces srslte_pss_t  (lib/pss_impl.h:66) */
typedef struct lt_cfo lt_cfo;     /* replaces srslte_cfo_t  (lib/pss_impl.h:73) */
typedef struct lt_sync lt_sync;   /* replaces srslte_sync_t (lib/sss_impl.h:42) */
typedef struct lt_batch lt_batch; /* batched pss->sss chains for many channels (no reference analogue) */
typedef struct lt_decim lt_decim; /* decimating FIR in front of the path */

/* ------------------------------------------------------------------ library */
const char *lt_version(void);
const char *lt_last_error(void);
int lt_device_count(void); /* number of usable HIP devices, 0 when none */

/* ------------------------------------------------------------------ srslte_pss_* */
/* srslte_pss_init(&d_pss, half_frame_length)            lib/pss_impl.cc:72
 * frame_size must be 9600 and fft_size 128 (srslte_use_standard_symbol_size(true), :69). */
int lt_pss_init(lt_pss **q, int device, int frame_size, int fft_size);
/* srslte_pss_set_N_id_2(&d_pss, N_id_2)                 lib/pss_impl.cc:75 */
int lt_pss_set_N_id_2(lt_pss *q, int N_id_2);
/* srslte_pss_free(&d_pss)                               lib/pss_impl.cc:90 */
void lt_pss_free(lt_pss *q);
/* srslte_pss_reset(&d_pss): zero the EMA of the lag metric   lib/pss_impl.cc:125,142 */
int lt_pss_reset(lt_pss *q);
/* srslte_pss_find_pss(&d_pss, in, &d_psr)               lib/pss_impl.cc:165-167
 * in: 9600 samples.  Returns peak_pos in [0, 9725] (index one past the PSS symbol) or <0. */
int lt_pss_find_pss(lt_pss *q, const lt_cf32 *in, float *psr);
/* q->peak_value after the last find (srsLTE struct field) */
int lt_pss_get_peak_value(lt_pss *q, float *peak_value);
/* q->conv_output_avg: 9729 floats (test/diagnostic access to the EMA vector) */
int lt_pss_get_avg(lt_pss *q, float *avg9729);
/* srslte_pss_cfo_compute(&d_pss, &out[pss_start])       lib/pss_impl.cc:199-201
 * pss_recv: 128 samples.  CFO in units of the 15 kHz subcarrier spacing. */
int lt_pss_cfo_compute(lt_pss *q, const lt_cf32 *pss_recv, float *cfo);

/* ------------------------------------------------------------------ srslte_cfo_* */
/* srslte_cfo_init(&d_cfo, half_frame_length)            lib/pss_impl.cc:78 */
int lt_cfo_init(lt_cfo **h, int device, int nsamples);
/* srslte_cfo_free(&d_cfo)                               lib/pss_impl.cc:91 */
void lt_cfo_free(lt_cfo *h);
/* d_cfo.last_freq = 0                                   lib/pss_impl.cc:148 (struct poke) */
int lt_cfo_set_last_freq(lt_cfo *h, float freq);
/* srslte_cfo_correct(&d_cfo, out, out, -mean_cfo()/symbol_sz)   lib/pss_impl.cc:204
 * out[n] = in[n] * exp(j 2 pi freq n) through the 4096-entry phase table; in may equal out. */
int lt_cfo_correct(lt_cfo *h, const lt_cf32 *in, lt_cf32 *out, float freq);

/* ------------------------------------------------------------------ srslte_sync_* / srslte_sss_* */
/* srslte_sync_init(&d_sync, half_frame_length, max_offset, symbol_sz)   lib/sss_impl.cc:63 */
int lt_sync_init(lt_sync **q, int device, int frame_size, int max_offset, int fft_size);
/* srslte_sync_set_N_id_2 + srslte_sss_set_N_id_2        lib/sss_impl.cc:66,69 */
int lt_sync_set_N_id_2(lt_sync *q, int N_id_2);
/* srslte_sync_free(&d_sync)                             lib/sss_impl.cc:80 */
void lt_sync_free(lt_sync *q);
/* srslte_sync_reset(&d_sync): CP-metric averages to 0   lib/sss_impl.cc:94 */
int lt_sync_reset(lt_sync *q);
/* srslte_sync_detect_cp(sync, in, slot_length)          lib/sss_impl.cc:104-106
 * in: at least peak_pos samples.  Returns LT_CP_NORM / LT_CP_EXT or <0. */
int lt_sync_detect_cp(lt_sync *q, const lt_cf32 *in, int peak_pos);
/* srslte_sync_set_cp(sync, cp)                          lib/sss_impl.cc:108 */
int lt_sync_set_cp(lt_sync *q, int cp);
/* sync->cp_len (9 or 32 at 1.92 Msps)                   lib/sss_impl.cc:110 */
int lt_sync_get_cp_len(lt_sync *q);
/* q->M_norm_avg / q->M_ext_avg (diagnostic) */
int lt_sync_get_cp_metrics(lt_sync *q, float *m_norm_avg, float *m_ext_avg);
/* srslte_sss_m0m1_partial(&sync->sss, &in[sss_idx], 1, NULL, &m0,&m0_value,&m1,&m1_value)
 *                                                       lib/sss_impl.cc:112-116; input: 128 samples */
int lt_sss_m0m1_partial(lt_sync *q, const lt_cf32 *input, uint32_t *m0, float *m0_value,
                        uint32_t *m1, float *m1_value);
/* srslte_sss_N_id_1(&sync->sss, m0, m1): N_id_1 or -1   lib/sss_impl.cc:118; also stores sync->N_id_1 */
int lt_sss_N_id_1(lt_sync *q, uint32_t m0, uint32_t m1);
/* srslte_sync_get_cell_id(sync) = 3*N_id_1 + N_id_2     lib/sss_impl.cc:124 */
int lt_sync_get_cell_id(lt_sync *q);

/* ------------------------------------------------------------------ batched chains */
/* One record per general_work() call of one pss->sss chain (lib/pss_impl.cc:155-223 followed by
 * lib/sss_impl.cc:83-156 on the emitted half-frame).  Stream tags become flag bits. */
#define LT_F_SEARCHED (1u << 0)  /* find_pss ran (lib/pss_impl.cc:163-169) */
#define LT_F_OVER (1u << 1)      /* d_psr > d_psr_threshold (:174) */
#define LT_F_TRACKING (1u << 2)  /* d_tracking after the score update */
#define LT_F_EMITTED (1u << 3)   /* a 9600-sample half-frame was produced (:184-195) */
#define LT_F_TAG_LOST (1u << 4)  /* "tracking_lost" tag on it (:210-214) */
#define LT_F_TAG_CELL (1u << 5)  /* "cell_id" + "cp_type" tags added (lib/sss_impl.cc:141-150) */
#define LT_F_CP_EXT (1u << 6)    /* cp_type == PMT_F */
#define LT_F_SSS_RAN (1u << 7)   /* CP detect + m0/m1 ran on this half-frame */

typedef struct {
  int64_t in_pos;   /* absolute index of in[0] (first unconsumed sample) at this call */
  int64_t out_pos;  /* absolute index of the first emitted sample; -1 if nothing emitted */
  int32_t channel;
  int32_t n_id_2;
  int32_t call_idx;
  uint32_t flags;
  int32_t peak_pos; /* d_peak_pos used by this call */
  int32_t score;    /* d_tracking.score after this call */
  float psr;        /* d_psr after this call */
  float peak_value; /* EMA metric at the peak; 0 when not searched */
  float cfo;        /* srslte_pss_cfo_compute (tracking calls only) */
  float mean_cfo;   /* mean_cfo() used for the correction */
  float m_norm_avg;
  float m_ext_avg;
  float m0_value;
  float m1_value;
  int32_t m0;
  int32_t m1;
  int32_t n_id_1;   /* -1: srslte_sss_N_id_1 failed or sss did not run */
  int32_t cell_id;  /* -1: no cell_id tag */
  float psr_max;    /* d_psr_max (max_psr() getter, include/ltetrigger/pss.h:72) */
  int32_t timer;    /* d_tracking.timer after this call */
} lt_record;

typedef struct {
  float psr_threshold;  /* pss::make psr_threshold, include/ltetrigger/pss.h:66-69 */
  int32_t track_after;  /* default 16 */
  int32_t track_every;  /* default 8 */
} lt_chain_params;

/* n_channels independent capture channels, three chains (N_id_2 = 0,1,2) each, as wired by
 * python/downlink_trigger_c.py:27-45.  max_samples bounds one lt_batch_run. */
int lt_batch_create(lt_batch **b, int device, int n_channels, int64_t max_samples,
                    const lt_chain_params *params);
void lt_batch_free(lt_batch *b);
/* All chains back to their constructor state. */
int lt_batch_reset(lt_batch *b);
/* Execution strategy: 1 (default) = the searches of all idle chains run as one chip-wide grid scan (exponential
 * averages in registers, strip-wise peak search), and a per-chain kernel executes general_work call by call
 * for the chains that fire, build a score or track; 0 = the per-chain kernel serves every chain.  Same
 * arithmetic on the same strips either way: identical records.  Everything either mode needs is allocated by
 * lt_batch_create; a mode never changes behind the caller's back. */
int lt_batch_set_mode(lt_batch *b, int mode);
int lt_batch_get_mode(lt_batch *b);
/* Window cache, levels 0 to 4 (default 4).  Level 1: what the scan rounds leave over -- chains that keep firing and losing a weak
 * cell, every chain in mode 0 -- the per-chain kernel takes in two passes: the first lists the windows every such chain
 * has ahead of it on its present timing, lt_metric_kernel forms their lag metrics for all chains at once, chip-wide,
 * and the second pass runs general_work on them (a window off the list goes through the transforms there and then).
 * Level 2: in batches of up to twelve channels -- whose scan rounds are bound by the instruction stream of one root
 * wave per CU, not by the chip -- the rounds' windows are listed too and the scan takes their lag metrics from the cache
 * (no transforms in the round itself).  Level 3 = level 2, with the per-chain kernel's two passes (and lt_metric_kernel
 * between them) launched only while chains ask for the cache: the kernel tells the host, through two words of pinned memory,
 * in which run a chain last had enough searches ahead of it to ask; a run enqueued when the latest run the host knows in
 * full had none -- and every run before the first such report -- launches the kernel once, without the cache (two launches
 * fewer per run; a capture that turns marginal pays its first runs' searches through the transforms until the report is in).
 * 0 = every search through the transforms.  Levels 0 to 3 give the same records bit for bit.
 * Level 4 = level 3 with lag-metric STREAMS for the chains that are not tracking: at marginal SNR most hits move a chain's
 * timing, and every window after such a hit is off the list (at -4 dB two searches in three went through the transforms
 * at level 3).  An interior lag of a window (127 <= lag < 9600) depends on the absolute sample position only, so for such
 * a chain lt_metric_kernel forms |sum_k h[k] x[p - k]|^2 for EVERY position p from the chain's position to the end of the
 * chunk (the same overlap-save blocks, laid on the stream instead of on windows), and a search on whatever timing reads
 * its interior lags out of the stream and forms the 253 edge lags (whose sums stop at the window's first or last sample)
 * directly.  Tracking chains keep their windows.  The stream's blocks round differently from a window's strips: level 4
 * matches the oracle under the same bar as every level (integer, index and flag fields equal, everything downstream of
 * the lag metric bit-exact, PSR / peak value within 1e-4) but NOT the bits of levels 0 to 3.  It is deterministic: which
 * of a run's searches come out of a stream is decided on the device from the chains' state alone (the per-chain kernel's
 * first pass; both passes are always enqueued at this level -- when no chain asks, lt_metric_kernel and the second pass are
 * two launches that end at once), and every chain has cache slots of its own, so the same capture through the same
 * settings gives the same bytes every time (tests/test_gpu_stream.py).  Across SETTINGS that change which searches the scan
 * rounds leave to the per-chain kernel (execution mode, round lengths, workgroup shape) the low bits of PSR may differ at
 * this level; levels 0 to 3 are bit-identical across those too.  THE DEFAULT (since round 5).
 * Footprint: every chain owns min(windows of max_samples, 256) slots of 43 008 bytes (1.7 GB for 64 channels x 1 s; at most
 * 48 GB: beyond that the per-chain share shrinks and what lies behind a chain's slots goes through the transforms);
 * batches of up to twelve channels at least 128 slots per chain (a scan round's windows).  Allocated by lt_batch_create, freed by level 0 and allocated
 * again by 1 ... 4 (if the memory cannot be had the batch runs as at level 0).  The slots of a scan round (levels >= 2, up to
 * twelve channels) are free again after the round: a capture of any length scans from the cache. */
int lt_batch_set_window_cache(lt_batch *b, int level);
/* Diagnostics of the last run: stats[0] = windows asked for, [1] = searches of the second pass served from the cache,
 * [2] = searches of the second pass that went through the transforms, [3] = chains that waited for the cache,
 * [4] = hits that moved such a chain off the timing its windows were listed on, [5] / [6] = windows a chain's average
 * had to be run through again (left over from a scan round), from the cache / through the transforms. */
int lt_batch_cache_stats(lt_batch *b, int64_t stats[7]);
/* Overlap (default 0): with 1, the per-record kernels of an lt_batch_run on a device buffer (CFO, CP, SSS: they feed
 * nothing but the records) are enqueued on a stream of the batch's own and run beside the scan rounds of the next
 * lt_batch_run; two sets of records take turns.  The run's stream then does NOT wait for them: the records are
 * complete once lt_batch_collect (or lt_batch_get_halfframes / lt_batch_last_kernel_ms) has returned, and the sample
 * buffer of a run must stay untouched until then.  lt_batch_push and runs from host memory are not overlapped.
 * Same records either way. */
int lt_batch_set_overlap(lt_batch *b, int on);
/* Diagnostics: hold every run's per-record kernels back by `microseconds` (0 = off, at most 100000) on the stream they
 * are enqueued on.  With overlap on, this opens the window in which a run's per-record kernels and the next run's
 * launches are in flight together as wide as a test wants it.  Records are the same with any hold. */
int lt_batch_debug_hold_records(lt_batch *b, int microseconds);
/* The grid scan runs in rounds of windows; a chain that fires leaves the scan at the end of its round.
 * first = windows in the first round (default 2: cells present from the start of a capture fire at once),
 * later = windows in every following round, at most 128 (default 0: the chunk's windows split evenly over as few
 * rounds as they need). */
int lt_batch_set_scan_segments(lt_batch *b, int first, int later);
/* Shape of the scan's workgroups: 1, 2 or 3 of a window's 12 strips per workgroup (12, 6 or 4 workgroups per channel).
 * 0 (default) chooses per run from the channel count and the device's CU count: three strips fill an MI355X with 64
 * channels, fewer strips put more CUs on a channel when there are fewer channels.  Timing only, never a record. */
int lt_batch_set_scan_groups(lt_batch *b, int strips_per_workgroup);
/* downlink_trigger_c.set_psr_threshold fan-out (python/downlink_trigger_c.py:63-69) */
int lt_batch_set_psr_threshold(lt_batch *b, float threshold);
/* Run every chain over iq = [n_channels][n_samples] cf32 (channel-major, contiguous), a capture of its own:
 * positions restart at sample 0 of `iq`, with the chain state left by the previous run unless
 * lt_batch_reset is called.
 * iq_on_device != 0: iq is a device pointer on the batch's device and the work is enqueued on
 * `hip_stream` (a hipStream_t; NULL = the batch's own stream) without host synchronisation.
 * iq_on_device == 0: iq is a host pointer, staged through the batch's device buffer.
 * A chain stops when fewer than 9600 new samples remain or an emission would read past the
 * capture (reference quirk q1 normalised). */
int lt_batch_run(lt_batch *b, const lt_cf32 *iq, int iq_on_device, int64_t n_samples,
                 void *hip_stream);
/* The same for a stream fed in chunks of any size (an SDR ring): the chunk continues where the previous one
 * ended -- set_history(9600) / consume_each of lib/pss_impl.cc:81,160,195: every chain resumes at its own
 * position, the library keeps the samples a chain may still need from earlier chunks, and a call starts only
 * once everything it could touch has arrived (the next call's 9600-sample window and the half-frame a peak
 * anywhere in it would emit: 18365 samples).  Records carry absolute stream positions; the cuts do not show in
 * them.  end_of_stream != 0: no more samples follow -- the chains run on under lt_batch_run's rule.
 * lt_batch_reset starts a new stream. */
int lt_batch_push(lt_batch *b, const lt_cf32 *iq, int iq_on_device, int64_t n_samples, int end_of_stream,
                  void *hip_stream);
/* Wait for the last run; copy records out: recs = [n_channels*3][cap_per_chain],
 * counts = [n_channels*3]; chain index = channel*3 + N_id_2. */
int lt_batch_collect(lt_batch *b, lt_record *recs, int64_t cap_per_chain, int64_t *counts);
/* With overlap on, the records of the run BEFORE the last one (the two sets of records take turns): waits for that
 * run's per-record kernels only and copies on a stream of its own, so that  run(k + 1); collect_previous() -> run k's
 * records  keeps a run in flight all the time and still hands every run's records out.  Fails (LT_ERROR) when the last
 * run has overwritten the records of the one before: the last run was not an overlapped one (a host buffer, a
 * lt_batch_push, overlap off), or overlap was off when the run before it was made. */
int lt_batch_collect_previous(lt_batch *b, lt_record *recs, int64_t cap_per_chain, int64_t *counts);
/* Upper bound on records per chain and run for chunks of n_samples (a call consumes >= 8640 samples; a chunk
 * may also finish calls left over from the one before). */
int64_t lt_batch_record_capacity(int64_t n_samples);
/* Optional capture of the emitted (CFO-corrected) half-frames = the pss block's output stream
 * (lib/pss_impl.cc:188-193,204).  max_per_chain = 0 disables.  Refused (LT_ERROR_INVALID_INPUTS) for a batch
 * whose max_samples gives more than 65535 record slots per chain (about 9.4 minutes at 1.92 Msps).  Read back
 * with lt_batch_get_halfframes: out = [max_per_chain][9600] for one chain. */
int lt_batch_enable_halfframes(lt_batch *b, int64_t max_per_chain);
int lt_batch_get_halfframes(lt_batch *b, int chain, lt_cf32 *out, int64_t cap, int64_t *count);
/* Device time of the chain kernel of the last run, in milliseconds (HIP events on the run's stream). */
int lt_batch_last_kernel_ms(lt_batch *b, float *ms);
/* Kernel launches the last lt_batch_run / lt_batch_push enqueued (state reset and copies not counted): what a chunk of a
 * stream costs beside its bytes.  No reference counterpart (diagnostics). */
int lt_batch_last_launch_count(lt_batch *b, int *n_launches);
/* The same, per stage, of the run `runs_back` runs ago (0 = the last one; the last 8 are kept):
 * ms[0] = plan + grid scan + resolve (all segments), ms[1] = 0, ms[2] = per-chain kernel, ms[3] = the
 * per-record kernels (CP length, SSS) and the history copy. */
int lt_batch_phase_ms(lt_batch *b, int runs_back, float ms[4]);
/* HIP events of a run: level 2 (default) = around the run and between its stages (what lt_batch_phase_ms reads),
 * 1 = around the run only (lt_batch_last_kernel_ms works, lt_batch_phase_ms fails for such a run), 0 = none (both
 * fail).  Every event costs the run's stream about 5 us of idle time. */
int lt_batch_set_phase_timing(lt_batch *b, int level);

/* ------------------------------------------------------------------ decimating FIR */
/* gnuradio.filter.rational_resampler_ccc(interpolation=1, decimation=N) with its default taps
 * (examples/cell_search_file.py:56-57, python/qa_downlink_trigger_c.py:109-112). */
int lt_decim_create(lt_decim **d, int device, int decimation);
void lt_decim_free(lt_decim *d);
int lt_decim_ntaps(lt_decim *d);
int lt_decim_get_taps(lt_decim *d, float *taps);
/* y[c][k] = sum_j taps[j] * x[c][k*N - j], k < n_in/N; x = [n_channels][n_in], y = [n_channels][n_in/N].
 * on_device != 0: both are device pointers, enqueued on hip_stream. */
int lt_decim_run(lt_decim *d, const lt_cf32 *x, lt_cf32 *y, int n_channels, int64_t n_in,
                 int on_device, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* LTETRIGGER_HIP_H */
