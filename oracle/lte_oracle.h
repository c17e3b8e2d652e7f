/*
 * oracle/lte_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * A plain-C restatement of the algorithm the reference (NTIA/gr-ltetrigger)
 * executes on its LTE cell-search hot path.  The reference owns only the
 * stream state machine (lib/pss_impl.cc, lib/sss_impl.cc); every piece of
 * arithmetic lives in srsLTE tags/release_18_06_1 (README.rst:41), an
 * un-vendored system dependency that is absent from /root/reference.  The
 * arithmetic is therefore restated from 3GPP TS 36.211 and from the published
 * srsLTE algorithm, and anchored on the reference's own
 * call sites, which every function below cites.
 *
 * PARITY STATUS: the final cell fields (cell_id, cp, and -- through
 * oracle/mib_oracle.py -- nof_prb / nof_tx_ports / phich) are pinned by the
 * four bundled fixtures and the dictionaries asserted in
 * python/qa_downlink_trigger_c.py:93-98,128-133,163-168,198-203.
 * Intermediate quantities (PSR, peak value, CFO, m0/m1 values, CP metric) are
 * PARITY UNPINNED: no reference test holds a golden value for them and the
 * reference cannot be built or imported here.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this library.  The product (gr-ltetrigger_amd/) never does.
 */
#ifndef LTE_ORACLE_H
#define LTE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } orc_cf; /* cf_t: interleaved float32 I/Q (lib/pss_impl.cc:62) */

enum {
  ORC_SLOT_LEN = 960,         /* lib/pss_impl.h:52 */
  ORC_HALF_FRAME_LEN = 9600,  /* lib/pss_impl.h:53 */
  ORC_SYMBOL_SZ = 128,        /* lib/pss_impl.h:55 */
  ORC_CONV_LEN = 9728,        /* frame_size + fft_size: FFT length of srslte_conv_fft_cc */
  ORC_NLAGS = 9726,           /* conv_output_len-1 lags enter the EMA / arg-max */
  ORC_MOVING_AVG_SZ = 200,    /* lib/pss_impl.h:31 */
  ORC_CEXPTAB_SIZE = 4096,    /* SRSLTE_CFO_CEXPTAB_SIZE */
  ORC_SSS_N = 31
};

/* One record per general_work() call of one pss->sss chain.  Same layout as
 * lt_record in include/ltetrigger_hip.h (the tests assert the sizes agree). */
enum {
  ORC_F_SEARCHED = 1u << 0,  /* srslte_pss_find_pss ran in this call (lib/pss_impl.cc:163-169) */
  ORC_F_OVER = 1u << 1,      /* d_psr > d_psr_threshold (lib/pss_impl.cc:174) */
  ORC_F_TRACKING = 1u << 2,  /* d_tracking after the score update */
  ORC_F_EMITTED = 1u << 3,   /* a 9600-sample half-frame was produced (lib/pss_impl.cc:184-195) */
  ORC_F_TAG_LOST = 1u << 4,  /* "tracking_lost" tag on it (lib/pss_impl.cc:210-214) */
  ORC_F_TAG_CELL = 1u << 5,  /* sss added cell_id + cp_type tags (lib/sss_impl.cc:141-150) */
  ORC_F_CP_EXT = 1u << 6,    /* cp_type == PMT_F (extended) */
  ORC_F_SSS_RAN = 1u << 7    /* sss ran CP detect + m0/m1 on this half-frame */
};

typedef struct {
  int64_t in_pos;   /* absolute index of in[0] = first unconsumed sample at this call */
  int64_t out_pos;  /* absolute index of the first emitted sample (in_pos+frame_start); -1 if none */
  int32_t channel;
  int32_t n_id_2;
  int32_t call_idx;
  uint32_t flags;
  int32_t peak_pos; /* d_peak_pos used by this call (find_pss result, or the carried 960) */
  int32_t score;    /* d_tracking.score after this call */
  float psr;        /* d_psr after this call */
  float peak_value; /* EMA metric at the peak; 0 when not searched */
  float cfo;        /* srslte_pss_cfo_compute, tracking calls only */
  float mean_cfo;   /* mean_cfo() used for the correction */
  float m_norm_avg; /* CP metrics after this call */
  float m_ext_avg;
  float m0_value;
  float m1_value;
  int32_t m0;
  int32_t m1;
  int32_t n_id_1;   /* -1 when srslte_sss_N_id_1 failed or sss did not run */
  int32_t cell_id;  /* -1 when no cell_id tag */
  float psr_max;    /* d_psr_max after this call */
  int32_t timer;    /* d_tracking.timer after this call */
} orc_record;

/* ---- sequence generators (36.211 6.11.1 / 6.11.2) ---- */
void orc_pss_zc(int n_id_2, orc_cf d[62]);
void orc_pss_taps(int n_id_2, orc_cf h[128]);
void orc_sss_mseq(int8_t s[31], int8_t c[31], int8_t z[31]);
void orc_sss_m0m1_of(int n_id_1, int *m0, int *m1);
void orc_sss_nid1_table(int32_t table[30][30]);
void orc_cexptab(orc_cf tab[ORC_CEXPTAB_SIZE]);

/* ---- FFT-9728 used by the convolution (exposed so tests can check it) ---- */
void orc_fft9728(const orc_cf *in, orc_cf *out, int inverse);

/* ---- srslte_pss_* (call sites lib/pss_impl.cc:72,75,125,142,165,200) ---- */
typedef struct orc_pss orc_pss_t;
orc_pss_t *orc_pss_create(int n_id_2);
void orc_pss_destroy(orc_pss_t *q);
void orc_pss_reset(orc_pss_t *q);
int orc_pss_find_pss(orc_pss_t *q, const orc_cf *in9600, float *psr);
float orc_pss_peak_value(const orc_pss_t *q);
const float *orc_pss_avg(const orc_pss_t *q); /* 9729 floats */
float orc_pss_cfo_compute(const orc_pss_t *q, const orc_cf *sym128);

/* ---- srslte_cfo_correct (lib/pss_impl.cc:204) ---- */
void orc_cfo_correct(const orc_cf *in, orc_cf *out, int n, float freq);

/* ---- srslte_sync_* / srslte_sss_* (call sites lib/sss_impl.cc:63-69,94,104-124) ---- */
typedef struct orc_sync orc_sync_t;
orc_sync_t *orc_sync_create(int n_id_2);
void orc_sync_destroy(orc_sync_t *q);
void orc_sync_reset(orc_sync_t *q);
int orc_sync_detect_cp(orc_sync_t *q, const orc_cf *in, int peak_pos); /* 0 normal, 1 extended */
void orc_sync_cp_metrics(const orc_sync_t *q, float *m_norm_avg, float *m_ext_avg);
void orc_sss_m0m1(const orc_sync_t *q, const orc_cf *sym128, int *m0, float *m0_value, int *m1,
                  float *m1_value);
int orc_sss_n_id_1(const orc_sync_t *q, unsigned m0, unsigned m1);

/* ---- the pss -> sss chain of one N_id_2 over a whole capture (lib/pss_impl.cc:155-223,
 *      lib/sss_impl.cc:83-156, python/downlink_trigger_c.py:43-45) ---- */
typedef struct {
  float psr_threshold;
  int32_t track_after; /* include/ltetrigger/pss.h:68 default 16 */
  int32_t track_every; /* include/ltetrigger/pss.h:69 default 8 */
} orc_chain_params;

/* Runs until fewer than 9600 new samples remain or an emission would read past
 * the end (quirk q1 normalised).  Returns the number of records.
 * out_hf (nullable): emitted half-frames, [out_cap][9600]; *n_out = count. */
int64_t orc_chain_run(int channel, int n_id_2, const orc_cf *x, int64_t n,
                      const orc_chain_params *p, orc_record *recs, int64_t rec_cap,
                      orc_cf *out_hf, int64_t out_cap, int64_t *n_out);

/* All 3 chains of n_channels channels, x = [n_channels][n]; recs = [n_channels*3][rec_cap],
 * n_recs[n_channels*3].  nthreads<=1: serial. */
void orc_batch_run(const orc_cf *x, int n_channels, int64_t n, const orc_chain_params *p,
                   orc_record *recs, int64_t rec_cap, int64_t *n_recs, int nthreads);

/* ---- decimating FIR in front of the path (examples/cell_search_file.py:56-57:
 *      gnuradio.filter.rational_resampler_ccc(1, N) with its default taps) ---- */
int orc_decim_ntaps(int decim);
void orc_decim_design(int decim, float *taps);
int64_t orc_decimate(const orc_cf *x, int64_t n, int decim, const float *taps, int ntaps,
                     orc_cf *y);

int orc_sizeof_record(void);

#ifdef __cplusplus
}
#endif
#endif
