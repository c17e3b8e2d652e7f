/*
 * oracle/lte_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * See lte_oracle.h for the status statement.  Build with -ffp-contract=off so
 * that every float operation rounds exactly once, as written.
 *
 * Citations "lib/..." and "python/..." are into /root/reference (the reference's
 * own state machine and call sites).  Citations "[srsLTE 18.06.1 ...]" name the
 * third-party function whose published algorithm is restated (SURVEY.md
 * Appendix A); that code is not available in this tree.
 */
#include "lte_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define NFFT ORC_CONV_LEN /* 9728 = 19 * 512 */
#define N1 19
#define N2 512

static inline orc_cf cf(float re, float im) { orc_cf r = {re, im}; return r; }
static inline orc_cf cmul(orc_cf a, orc_cf b)
{
  return cf(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}

/* ------------------------------------------------------------------------- */
/* Sequence generators                                                        */
/* ------------------------------------------------------------------------- */

/* 36.211 6.11.1.1: Zadoff-Chu roots 25/29/34 for N_id_2 = 0/1/2
 * [srsLTE 18.06.1 srslte_pss_generate]; SURVEY A.1. */
static void pss_zc_d(int n_id_2, double re[62], double im[62])
{
  static const int root[3] = {25, 29, 34};
  const int u = root[n_id_2];
  for (int n = 0; n < 62; n++) {
    /* -pi*u*n(n+1)/63 for n<31, -pi*u*(n+1)(n+2)/63 for n>=31; reduce the
     * integer product mod 126 so the argument stays small and exact. */
    long prod = (n < 31) ? (long)u * n * (n + 1) : (long)u * (n + 1) * (n + 2);
    double ph = -M_PI * (double)(prod % 126) / 63.0;
    re[n] = cos(ph);
    im[n] = sin(ph);
  }
}

void orc_pss_zc(int n_id_2, orc_cf d[62])
{
  double re[62], im[62];
  pss_zc_d(n_id_2, re, im);
  for (int n = 0; n < 62; n++) d[n] = cf((float)re[n], (float)im[n]);
}

/* Time-domain matched filter [srsLTE 18.06.1 srslte_pss_init_N_id_2]:
 * d[0..30] -> bins -31..-1, d[31..61] -> bins +1..+31, DC empty (mirror+dc DFT
 * plan); h[m] = conj(IDFT128(X))[m] / sqrt(128) / 62.  SURVEY A.1. */
void orc_pss_taps(int n_id_2, orc_cf h[128])
{
  double dre[62], dim[62];
  pss_zc_d(n_id_2, dre, dim);
  const double scale = 1.0 / sqrt(128.0) / 62.0;
  for (int m = 0; m < 128; m++) {
    double sr = 0.0, si = 0.0;
    for (int k = 0; k < 62; k++) {
      int bin = (k < 31) ? (k - 31) : (k - 30);
      int e = ((bin * m) % 128 + 128) % 128;
      double ph = 2.0 * M_PI * (double)e / 128.0;
      double c = cos(ph), s = sin(ph);
      sr += dre[k] * c - dim[k] * s;
      si += dre[k] * s + dim[k] * c;
    }
    h[m] = cf((float)(sr * scale), (float)(-si * scale));
  }
}

/* 36.211 6.11.2.1 m-sequences, values 1-2x [srsLTE 18.06.1 generate_s/c/z_tilde]. */
void orc_sss_mseq(int8_t s[31], int8_t c[31], int8_t z[31])
{
  int x[31];
  memset(x, 0, sizeof x); x[4] = 1;
  for (int i = 0; i < 26; i++) x[i + 5] = (x[i + 2] + x[i]) % 2;
  for (int i = 0; i < 31; i++) s[i] = (int8_t)(1 - 2 * x[i]);
  memset(x, 0, sizeof x); x[4] = 1;
  for (int i = 0; i < 26; i++) x[i + 5] = (x[i + 3] + x[i]) % 2;
  for (int i = 0; i < 31; i++) c[i] = (int8_t)(1 - 2 * x[i]);
  memset(x, 0, sizeof x); x[4] = 1;
  for (int i = 0; i < 26; i++) x[i + 5] = (x[i + 4] + x[i + 2] + x[i + 1] + x[i]) % 2;
  for (int i = 0; i < 31; i++) z[i] = (int8_t)(1 - 2 * x[i]);
}

/* 36.211 Table 6.11.2.1-1 closed form [srsLTE 18.06.1 generate_m0m1]. */
void orc_sss_m0m1_of(int n_id_1, int *m0, int *m1)
{
  int q_prime = n_id_1 / 30;
  int q = (n_id_1 + q_prime * (q_prime + 1) / 2) / 30;
  int m_prime = n_id_1 + q * (q + 1) / 2;
  *m0 = m_prime % 31;
  *m1 = (*m0 + m_prime / 31 + 1) % 31;
}

/* Zero-initialised 30x30 table, T[m0][m1-1] = N_id_1
 * [srsLTE 18.06.1 generate_N_id_1_table]; SURVEY A.5, quirk q5. */
void orc_sss_nid1_table(int32_t table[30][30])
{
  memset(table, 0, sizeof(int32_t) * 900);
  for (int n = 0; n < 168; n++) {
    int m0, m1;
    orc_sss_m0m1_of(n, &m0, &m1);
    table[m0][m1 - 1] = n;
  }
}

/* [srsLTE 18.06.1 srslte_cexptab_init]: tab[i] = exp(j*2*pi*i/size). */
void orc_cexptab(orc_cf tab[ORC_CEXPTAB_SIZE])
{
  for (int i = 0; i < ORC_CEXPTAB_SIZE; i++) {
    double ph = 2.0 * M_PI * (double)i / (double)ORC_CEXPTAB_SIZE;
    tab[i] = cf((float)cos(ph), (float)sin(ph));
  }
}

/* ------------------------------------------------------------------------- */
/* FFT of length 9728 = 19 x 512 (single precision, like FFTW3F in srsLTE)    */
/* ------------------------------------------------------------------------- */

static struct {
  int ready;
  orc_cf w19[N1];        /* exp(-j2pi i/19) */
  orc_cf wN[N1][N2];     /* exp(-j2pi n2*k1/9728) */
  orc_cf w512[N2 / 2];   /* exp(-j2pi i/512) */
  uint16_t brev[N2];
} g_fft;

static void fft_init(void)
{
  if (g_fft.ready) return;
  for (int i = 0; i < N1; i++) {
    double ph = -2.0 * M_PI * i / (double)N1;
    g_fft.w19[i] = cf((float)cos(ph), (float)sin(ph));
  }
  for (int k1 = 0; k1 < N1; k1++)
    for (int n2 = 0; n2 < N2; n2++) {
      double ph = -2.0 * M_PI * (double)(n2 * k1) / (double)NFFT;
      g_fft.wN[k1][n2] = cf((float)cos(ph), (float)sin(ph));
    }
  for (int i = 0; i < N2 / 2; i++) {
    double ph = -2.0 * M_PI * i / (double)N2;
    g_fft.w512[i] = cf((float)cos(ph), (float)sin(ph));
  }
  for (int i = 0; i < N2; i++) {
    int r = 0;
    for (int b = 0; b < 9; b++)
      if (i & (1 << b)) r |= 1 << (8 - b);
    g_fft.brev[i] = (uint16_t)r;
  }
  g_fft.ready = 1;
}

static void fft512_inplace(orc_cf *a)
{
  for (int i = 0; i < N2; i++) {
    int j = g_fft.brev[i];
    if (j > i) { orc_cf t = a[i]; a[i] = a[j]; a[j] = t; }
  }
  for (int len = 2; len <= N2; len <<= 1) {
    int half = len >> 1, step = N2 / len;
    for (int i = 0; i < N2; i += len)
      for (int k = 0; k < half; k++) {
        orc_cf w = g_fft.w512[k * step];
        orc_cf u = a[i + k], v = cmul(a[i + k + half], w);
        a[i + k] = cf(u.re + v.re, u.im + v.im);
        a[i + k + half] = cf(u.re - v.re, u.im - v.im);
      }
  }
}

/* Forward: X[k] = sum x[n] exp(-j2pi nk/N).  Inverse: unnormalised, like FFTW. */
void orc_fft9728(const orc_cf *in, orc_cf *out, int inverse)
{
  fft_init();
  static _Thread_local orc_cf A[N1][N2];
  const float sgn = inverse ? -1.0f : 1.0f; /* inverse = conj(fwd(conj(x))) */
  for (int k1 = 0; k1 < N1; k1++) {
    orc_cf *row = A[k1];
    for (int n2 = 0; n2 < N2; n2++) row[n2] = cf(0.f, 0.f);
    for (int n1 = 0; n1 < N1; n1++) {
      const orc_cf w = g_fft.w19[(n1 * k1) % N1];
      const orc_cf *x = in + N2 * n1;
      for (int n2 = 0; n2 < N2; n2++) {
        float xr = x[n2].re, xi = sgn * x[n2].im;
        row[n2].re += xr * w.re - xi * w.im;
        row[n2].im += xr * w.im + xi * w.re;
      }
    }
    for (int n2 = 0; n2 < N2; n2++) row[n2] = cmul(row[n2], g_fft.wN[k1][n2]);
    fft512_inplace(row);
    for (int k2 = 0; k2 < N2; k2++) out[k1 + N1 * k2] = cf(row[k2].re, sgn * row[k2].im);
  }
}

/* ------------------------------------------------------------------------- */
/* srslte_pss_*                                                               */
/* ------------------------------------------------------------------------- */

struct orc_pss {
  int n_id_2;
  float ema_alpha;
  orc_cf h[128];
  orc_cf *filt_freq; /* NFFT: DFT of zero-padded h, normalised 1/sqrt(N) */
  orc_cf *tmp_in, *tmp_fft, *conv;
  float *abs2;
  float *avg; /* fft_size + frame_size + 1 = 9729 */
  float peak_value;
};

/* [srsLTE 18.06.1 srslte_pss_init + srslte_pss_set_N_id_2 + srslte_conv_fft_cc_init];
 * call sites lib/pss_impl.cc:72,75.  SURVEY A.1. */
orc_pss_t *orc_pss_create(int n_id_2)
{
  if (n_id_2 < 0 || n_id_2 > 2) return NULL;
  fft_init();
  orc_pss_t *q = (orc_pss_t *)calloc(1, sizeof *q);
  if (!q) return NULL;
  q->n_id_2 = n_id_2;
  q->ema_alpha = 0.2f;
  q->filt_freq = (orc_cf *)calloc(NFFT, sizeof(orc_cf));
  q->tmp_in = (orc_cf *)calloc(NFFT, sizeof(orc_cf));
  q->tmp_fft = (orc_cf *)calloc(NFFT, sizeof(orc_cf));
  q->conv = (orc_cf *)calloc(NFFT, sizeof(orc_cf));
  q->abs2 = (float *)calloc(NFFT + 1, sizeof(float));
  q->avg = (float *)calloc(NFFT + 1, sizeof(float));
  orc_pss_taps(n_id_2, q->h);
  memcpy(q->tmp_in, q->h, sizeof q->h);
  orc_fft9728(q->tmp_in, q->filt_freq, 0);
  const float norm = 1.0f / sqrtf((float)NFFT); /* filter_plan norm=true */
  for (int i = 0; i < NFFT; i++) { q->filt_freq[i].re *= norm; q->filt_freq[i].im *= norm; }
  memset(q->tmp_in, 0, NFFT * sizeof(orc_cf));
  return q;
}

void orc_pss_destroy(orc_pss_t *q)
{
  if (!q) return;
  free(q->filt_freq); free(q->tmp_in); free(q->tmp_fft); free(q->conv); free(q->abs2);
  free(q->avg); free(q);
}

/* [srsLTE 18.06.1 srslte_pss_reset]: zero the EMA vector; lib/pss_impl.cc:125,142. */
void orc_pss_reset(orc_pss_t *q) { memset(q->avg, 0, (NFFT + 1) * sizeof(float)); }

float orc_pss_peak_value(const orc_pss_t *q) { return q->peak_value; }
const float *orc_pss_avg(const orc_pss_t *q) { return q->avg; }

/* [srsLTE 18.06.1 srslte_vec_max_fi]: first strict maximum, start value -FLT_MAX. */
static uint32_t vec_max_fi(const float *x, int len)
{
  float m = -FLT_MAX;
  uint32_t p = 0;
  for (int i = 0; i < len; i++)
    if (x[i] > m) { m = x[i]; p = (uint32_t)i; }
  return p;
}

/* [srsLTE 18.06.1 srslte_pss_find_pss, CONVOLUTION_FFT + SRSLTE_PSS_RETURN_PSR];
 * call site lib/pss_impl.cc:165-167.  SURVEY A.2. */
int orc_pss_find_pss(orc_pss_t *q, const orc_cf *in, float *psr)
{
  /* srslte_conv_fft_cc_run_opt: FFT(in, 1/sqrt N) * filter -> unnormalised IFFT;
   * returns output_len-1 = 9727. */
  memcpy(q->tmp_in, in, ORC_HALF_FRAME_LEN * sizeof(orc_cf)); /* tail [9600,9728) stays 0 */
  orc_fft9728(q->tmp_in, q->tmp_fft, 0);
  const float norm = 1.0f / sqrtf((float)NFFT);
  for (int i = 0; i < NFFT; i++) {
    orc_cf v = cf(q->tmp_fft[i].re * norm, q->tmp_fft[i].im * norm);
    q->tmp_fft[i] = cmul(v, q->filt_freq[i]);
  }
  orc_fft9728(q->tmp_fft, q->conv, 1);
  const int conv_output_len = NFFT - 1; /* 9727 */
  const int L = conv_output_len - 1;    /* 9726 */

  for (int i = 0; i < L; i++) /* srslte_vec_abs_square_cf */
    q->abs2[i] = q->conv[i].re * q->conv[i].re + q->conv[i].im * q->conv[i].im;
  /* EMA, alpha = 0.2: three separate vector passes in srsLTE => three roundings */
  const float a = q->ema_alpha, b = 1 - q->ema_alpha;
  for (int i = 0; i < L; i++) q->abs2[i] = q->abs2[i] * a;
  for (int i = 0; i < L; i++) q->avg[i] = q->avg[i] * b;
  for (int i = 0; i < L; i++) q->avg[i] = q->abs2[i] + q->avg[i];

  const float *avg = q->avg;
  int peak = (int)vec_max_fi(avg, L);
  q->peak_value = avg[peak];

  /* main-lobe walk; avg[9726..9728] are never written and stay 0 */
  int pl_ub = peak + 1;
  while (avg[pl_ub + 1] <= avg[pl_ub] && pl_ub < conv_output_len) pl_ub++;
  int pl_lb;
  if (peak > 2) {
    pl_lb = peak - 1;
    while (avg[pl_lb - 1] <= avg[pl_lb] && pl_lb > 1) pl_lb--;
  } else {
    pl_lb = 0;
  }
  int sl_distance_right = conv_output_len - 1 - pl_ub;
  if (sl_distance_right < 0) sl_distance_right = 0;
  int sl_distance_left = pl_lb;
  int sl_right = pl_ub + (int)vec_max_fi(&avg[pl_ub], sl_distance_right);
  int sl_left = (int)vec_max_fi(avg, sl_distance_left);
  float side = avg[sl_left] > avg[sl_right] ? avg[sl_left] : avg[sl_right]; /* SRSLTE_MAX */
  if (psr) *psr = avg[peak] / side;
  return peak;
}

/* [srsLTE 18.06.1 srslte_pss_cfo_compute]; call site lib/pss_impl.cc:199-201. SURVEY A.3. */
float orc_pss_cfo_compute(const orc_pss_t *q, const orc_cf *r)
{
  /* srslte_vec_dot_prod_ccc (no conjugate) is a SIMD kernel in srsLTE: lane-wise partial sums
   * followed by a horizontal add.  Restated with eight interleaved partial sums per half
   * (m mod 8), combined as ((p0+p1)+(p2+p3))+((p4+p5)+(p6+p7)); the kernels use the same order. */
  orc_cf y[2];
  for (int half = 0; half < 2; half++) {
    orc_cf p[8];
    for (int l = 0; l < 8; l++) p[l] = cf(0.f, 0.f);
    for (int j = 0; j < 8; j++)
      for (int l = 0; l < 8; l++) {
        const int m = 64 * half + 8 * j + l;
        orc_cf t = cmul(q->h[m], r[m]);
        p[l].re += t.re; p[l].im += t.im;
      }
    y[half] = cf(((p[0].re + p[1].re) + (p[2].re + p[3].re)) + ((p[4].re + p[5].re) + (p[6].re + p[7].re)),
                 ((p[0].im + p[1].im) + (p[2].im + p[3].im)) + ((p[4].im + p[5].im) + (p[6].im + p[7].im)));
  }
  const orc_cf y0 = y[0], y1 = y[1];
  /* carg(conjf(y0)*y1)/M_PI: float product, double atan2 */
  orc_cf z = cmul(cf(y0.re, -y0.im), y1);
  return (float)(atan2((double)z.im, (double)z.re) / M_PI);
}

/* [srsLTE 18.06.1 srslte_cfo_correct + srslte_cexptab_gen]; call site lib/pss_impl.cc:204.
 * out[i] = tab[idx(i)] * in[i], idx = floor(phase mod 4096), phase = i * (freq*4096).
 * NORMALISED (quirk q6): srsLTE accumulates `phase += phase_inc` in a
 * float recurrence; here the phase of sample i is formed exactly in double, which
 * differs from the recurrence only where its accumulated rounding crosses an
 * integer boundary (one table step = 1.5e-3 rad).  Product and oracle use the same
 * rule and the same fused form of the complex product. */
static orc_cf g_cexptab[ORC_CEXPTAB_SIZE];
static int g_cexptab_ready;
static void cexptab_init(void)
{
  if (!g_cexptab_ready) { orc_cexptab(g_cexptab); g_cexptab_ready = 1; }
}

void orc_cfo_correct(const orc_cf *in, orc_cf *out, int n, float freq)
{
  cexptab_init();
  const orc_cf *tab = g_cexptab;
  const float phase_inc = freq * (float)ORC_CEXPTAB_SIZE;
  for (int i = 0; i < n; i++) {
    double p = (double)phase_inc * (double)i;
    p -= 4096.0 * floor(p / 4096.0);
    uint32_t idx = (uint32_t)p;
    if (idx > 4095u) idx = 4095u;
    const orc_cf e = tab[idx], v = in[i];
    out[i] = cf(fmaf(e.re, v.re, -(e.im * v.im)), fmaf(e.re, v.im, e.im * v.re));
  }
}

/* ------------------------------------------------------------------------- */
/* srslte_sync_* / srslte_sss_*                                                */
/* ------------------------------------------------------------------------- */

struct orc_sync {
  int n_id_2;
  float m_norm_avg, m_ext_avg;
  float c0[31], c1[31];
  float s[31][31];
  float z1[31][31];
  int32_t table[30][30];
};

/* [srsLTE 18.06.1 srslte_sync_init, srslte_sync_set_N_id_2, srslte_sss_set_N_id_2 ->
 * generate_sss_all_tables]; call sites lib/sss_impl.cc:63-69.  SURVEY A.5. */
orc_sync_t *orc_sync_create(int n_id_2)
{
  if (n_id_2 < 0 || n_id_2 > 2) return NULL;
  orc_sync_t *q = (orc_sync_t *)calloc(1, sizeof *q);
  if (!q) return NULL;
  q->n_id_2 = n_id_2;
  int8_t s[31], c[31], z[31];
  orc_sss_mseq(s, c, z);
  for (int i = 0; i < 31; i++) {
    q->c0[i] = c[(i + n_id_2) % 31];
    q->c1[i] = c[(i + n_id_2 + 3) % 31];
  }
  for (int m = 0; m < 31; m++)
    for (int i = 0; i < 31; i++) {
      q->s[m][i] = s[(i + m) % 31];
      q->z1[m][i] = z[(i + (m % 8)) % 31];
    }
  orc_sss_nid1_table(q->table);
  return q;
}

void orc_sync_destroy(orc_sync_t *q) { free(q); }

/* [srsLTE 18.06.1 srslte_sync_reset]: CP averages to 0; call site lib/sss_impl.cc:94. */
void orc_sync_reset(orc_sync_t *q) { q->m_norm_avg = 0.f; q->m_ext_avg = 0.f; }

void orc_sync_cp_metrics(const orc_sync_t *q, float *n, float *e)
{
  if (n) *n = q->m_norm_avg;
  if (e) *e = q->m_ext_avg;
}

/* [srsLTE 18.06.1 srslte_sync_detect_cp]; call site lib/sss_impl.cc:104-106. SURVEY A.4. */
int orc_sync_detect_cp(orc_sync_t *q, const orc_cf *in, int peak_pos)
{
  const int fft = ORC_SYMBOL_SZ, cp_norm = 9, cp_ext = 32;
  const float alpha = 0.1f; /* CP_EMA_ALPHA */
  int nsym = peak_pos / (fft + cp_ext);
  if (nsym > 3) nsym = 3;
  if (nsym <= 0) return 0;
  float R[2] = {0, 0}, C[2] = {0, 0}, M[2] = {0, 0};
  const int cpl[2] = {cp_norm, cp_ext};
  for (int hyp = 0; hyp < 2; hyp++) {
    const int L = cpl[hyp];
    const orc_cf *p = in + peak_pos - nsym * (fft + L);
    for (int s = 0; s < nsym; s++) {
      /* srslte_vec_dot_prod_conj_ccc / srslte_vec_avg_power_cf are SIMD kernels: four interleaved
       * partial sums (i mod 4) combined as (p0+p1)+(p2+p3); the kernels use the same order */
      float pr[4] = {0, 0, 0, 0}, pp[4] = {0, 0, 0, 0};
      for (int i = 0; i < L; i++) { /* dot_prod_conj_ccc(tail, cp): Re(tail * conj(cp)) */
        orc_cf a = p[fft + i], b = p[i];
        pr[i & 3] += a.re * b.re + a.im * b.im;
        pp[i & 3] += b.re * b.re + b.im * b.im;
      }
      const float dre = (pr[0] + pr[1]) + (pr[2] + pr[3]);
      const float pw = (pp[0] + pp[1]) + (pp[2] + pp[3]);
      R[hyp] += dre;
      C[hyp] += (float)L * (pw / (float)L); /* cp_len * srslte_vec_avg_power_cf */
      p += fft + L;
    }
    if (C[hyp] > 0) M[hyp] = R[hyp] / C[hyp];
  }
  q->m_norm_avg = alpha * (M[0] / (float)nsym) + (1 - alpha) * q->m_norm_avg; /* SRSLTE_VEC_EMA */
  q->m_ext_avg = alpha * (M[1] / (float)nsym) + (1 - alpha) * q->m_ext_avg;
  if (q->m_norm_avg > q->m_ext_avg) return 0;
  if (q->m_norm_avg < q->m_ext_avg) return 1;
  return (R[0] > R[1]) ? 0 : 1;
}

/* [srsLTE 18.06.1 srslte_sss_m0m1_partial(M=1, ce=NULL) -> extract_pair_sss +
 * corr_all_sz_partial]; call site lib/sss_impl.cc:112-116.  SURVEY A.5. */
void orc_sss_m0m1(const orc_sync_t *q, const orc_cf *sym, int *m0, float *m0_value, int *m1,
                  float *m1_value)
{
  /* 128-point forward DFT in single precision (srsLTE: FFTW3F plan, mirror + dc): iterative
   * radix-2 decimation-in-time FFT, bit-reversed input order, butterflies
   *   t = w * b = (b.re w.re - b.im w.im, b.re w.im + b.im w.re);  a' = a + t;  b' = a - t
   * with w = W128^(k * 128/len) from a float table.  Centred output with DC removed: 62 bins
   * -31..-1, +1..+31.  The kernels evaluate the same butterflies (one pair of points per lane). */
  static float twc[64], tws[64];
  static int tw_ready;
  if (!tw_ready) {
    for (int e = 0; e < 64; e++) {
      double ph = -2.0 * M_PI * (double)e / 128.0;
      twc[e] = (float)cos(ph);
      tws[e] = (float)sin(ph);
    }
    tw_ready = 1;
  }
  orc_cf a[128];
  for (int i = 0; i < 128; i++) {
    int r = 0;
    for (int b = 0; b < 7; b++)
      if (i & (1 << b)) r |= 1 << (6 - b);
    a[i] = sym[r];
  }
  for (int len = 2; len <= 128; len <<= 1) {
    const int half = len >> 1, step = 128 / len;
    for (int i = 0; i < 128; i += len)
      for (int k = 0; k < half; k++) {
        const float wr = twc[k * step], wi = tws[k * step];
        const orc_cf u = a[i + k], v = a[i + k + half];
        const float tr = v.re * wr - v.im * wi, ti = v.re * wi + v.im * wr;
        a[i + k] = cf(u.re + tr, u.im + ti);
        a[i + k + half] = cf(u.re - tr, u.im - ti);
      }
  }
  orc_cf Y[62];
  for (int k = 0; k < 62; k++) {
    int bin = (k < 31) ? (k - 31) : (k - 30);
    Y[k] = a[(bin + 128) & 127];
  }
  orc_cf y0[31], y1[31];
  for (int i = 0; i < 31; i++) {
    y0[i] = cf(Y[2 * i].re * q->c0[i], Y[2 * i].im * q->c0[i]);
    y1[i] = cf(Y[2 * i + 1].re * q->c1[i], Y[2 * i + 1].im * q->c1[i]);
  }
  /* corr_all_sz_partial (M = 1): srslte_vec_dot_prod_cfc is SIMD -> four interleaved partial sums */
  float corr[31];
  for (int m = 0; m < 31; m++) {
    float pr[4] = {0, 0, 0, 0}, pi[4] = {0, 0, 0, 0};
    for (int i = 0; i < 31; i++) { pr[i & 3] += y0[i].re * q->s[m][i]; pi[i & 3] += y0[i].im * q->s[m][i]; }
    const float re = (pr[0] + pr[1]) + (pr[2] + pr[3]), im = (pi[0] + pi[1]) + (pi[2] + pi[3]);
    corr[m] = re * re + im * im;
  }
  *m0 = (int)vec_max_fi(corr, 31);
  if (m0_value) *m0_value = corr[*m0];
  for (int i = 0; i < 31; i++) {
    y1[i].re *= q->z1[*m0][i];
    y1[i].im *= q->z1[*m0][i];
  }
  for (int m = 0; m < 31; m++) {
    float pr[4] = {0, 0, 0, 0}, pi[4] = {0, 0, 0, 0};
    for (int i = 0; i < 31; i++) { pr[i & 3] += y1[i].re * q->s[m][i]; pi[i & 3] += y1[i].im * q->s[m][i]; }
    const float re = (pr[0] + pr[1]) + (pr[2] + pr[3]), im = (pi[0] + pi[1]) + (pi[2] + pi[3]);
    corr[m] = re * re + im * im;
  }
  *m1 = (int)vec_max_fi(corr, 31);
  if (m1_value) *m1_value = corr[*m1];
}

/* [srsLTE 18.06.1 srslte_sss_N_id_1]; call site lib/sss_impl.cc:118. Unsigned wrap kept. */
int orc_sss_n_id_1(const orc_sync_t *q, unsigned m0, unsigned m1)
{
  int n = -1;
  if (m1 > m0) {
    if (m0 < 30 && m1 - 1 < 30) n = q->table[m0][m1 - 1];
  } else {
    if (m1 < 30 && m0 - 1 < 30) n = q->table[m1][m0 - 1];
  }
  return n;
}

/* ------------------------------------------------------------------------- */
/* The chain: pss_impl::general_work -> sss_impl::work, call by call           */
/* ------------------------------------------------------------------------- */

static float moving_avg(const float *data, size_t npts) /* lib/pss_impl.cc:94-109 */
{
  if (!npts) return 0.0f;
  double acc = 0.0;
  if (npts > ORC_MOVING_AVG_SZ) npts = ORC_MOVING_AVG_SZ;
  for (size_t i = 0; i < npts; i++) acc += data[i];
  return (float)(acc / (double)npts);
}

static inline orc_cf sample_at(const orc_cf *x, int64_t n, int64_t i)
{
  /* "Zero history at stream start" (set_history, lib/pss_impl.cc:81) */
  if (i < 0 || i >= n) return cf(0.f, 0.f);
  return x[i];
}

int64_t orc_chain_run(int channel, int n_id_2, const orc_cf *x, int64_t n,
                      const orc_chain_params *p, orc_record *recs, int64_t rec_cap,
                      orc_cf *out_hf, int64_t out_cap, int64_t *n_out)
{
  orc_pss_t *pss = orc_pss_create(n_id_2);
  orc_sync_t *sync = orc_sync_create(n_id_2);
  const int HF = ORC_HALF_FRAME_LEN, SLOT = ORC_SLOT_LEN, SYM = ORC_SYMBOL_SZ;

  /* pss_impl state, lib/pss_impl.h:41-81 */
  int is_tracking = 0, score = 0, timer = 0, tracking_lost = 0;
  float psr_data[ORC_MOVING_AVG_SZ] = {0}, cfo_data[ORC_MOVING_AVG_SZ] = {0};
  size_t psr_i = 0, cfo_i = 0;
  float psr = 0.f, psr_max = 0.f;
  (void)psr_data; /* the ring is kept (lib/pss_impl.cc:166) but only mean_psr(), which nothing here calls, reads it */
  int peak_pos = 0;

  orc_cf *win = (orc_cf *)malloc(sizeof(orc_cf) * HF);
  orc_cf *out = (orc_cf *)malloc(sizeof(orc_cf) * HF);
  int64_t pos = 0, nrec = 0, nemit = 0;

  while (pos + HF <= n && nrec < rec_cap) {
    orc_record r;
    memset(&r, 0, sizeof r);
    r.in_pos = pos; r.out_pos = -1; r.channel = channel; r.n_id_2 = n_id_2;
    r.call_idx = (int32_t)nrec; r.n_id_1 = -1; r.cell_id = -1; r.m0 = -1; r.m1 = -1;

    /* lib/pss_impl.cc:163-172 */
    if (!is_tracking || timer == 0) {
      timer = p->track_every;
      memcpy(win, x + pos, sizeof(orc_cf) * HF);
      peak_pos = orc_pss_find_pss(pss, win, &psr);
      psr_data[psr_i++ % ORC_MOVING_AVG_SZ] = psr;
      r.flags |= ORC_F_SEARCHED;
      r.peak_value = orc_pss_peak_value(pss);
    } else {
      timer--;
    }
    r.peak_pos = peak_pos;

    /* lib/pss_impl.cc:174-182; incr_score :111-127, reset_score :129-152 */
    const int over = psr > p->psr_threshold;
    if (over) {
      if (!(is_tracking && score == p->track_after)) {
        score++;
        if (!is_tracking && score == p->track_after) {
          is_tracking = 1;
          orc_pss_reset(pss);
        }
      }
    } else if (score != 0) {
      score = 0;
      timer = 0;
      is_tracking = 0;
      orc_pss_reset(pss);
      psr_i = 0; /* partial memsets (quirk q4) have no observable effect */
      cfo_i = 0;
      tracking_lost = 1;
    }
    if (psr > psr_max) psr_max = psr;
    if (over) r.flags |= ORC_F_OVER;

    int64_t next_pos;
    if (over || tracking_lost) { /* lib/pss_impl.cc:184-214 */
      const int frame_start = peak_pos - SLOT;
      peak_pos = SLOT;
      const int64_t hf0 = pos + frame_start;
      if (hf0 + HF > n) break; /* quirk q1 normalised: never read past the capture */
      for (int i = 0; i < HF; i++) out[i] = sample_at(x, n, hf0 + i);
      next_pos = hf0 + HF;
      r.flags |= ORC_F_EMITTED;
      r.out_pos = hf0;
      int tag_lost = 0;
      if (is_tracking) {
        const float cfo = orc_pss_cfo_compute(pss, &out[SLOT - SYM]);
        cfo_data[cfo_i++ % ORC_MOVING_AVG_SZ] = cfo;
        const float mcfo = moving_avg(cfo_data, cfo_i);
        orc_cfo_correct(out, out, HF, -mcfo / (float)SYM);
        r.cfo = cfo;
        r.mean_cfo = mcfo;
        /* srslte_pss_chest (lib/pss_impl.cc:206-209): result never consumed, dropped */
      } else {
        tag_lost = 1;
        tracking_lost = 0;
        r.flags |= ORC_F_TAG_LOST;
      }

      /* ---- sss_impl::work on the emitted half-frame, lib/sss_impl.cc:83-156 ---- */
      if (tag_lost) {
        orc_sync_reset(sync);
      } else {
        const int cp_ext = orc_sync_detect_cp(sync, out, SLOT);
        const int cp_len = cp_ext ? 32 : 9; /* srslte_sync_set_cp */
        const int sss_idx = SLOT - 2 * SYM - cp_len;
        int m0, m1; float v0, v1;
        orc_sss_m0m1(sync, &out[sss_idx], &m0, &v0, &m1, &v1);
        r.flags |= ORC_F_SSS_RAN;
        r.m0 = m0; r.m1 = m1; r.m0_value = v0; r.m1_value = v1;
        if (cp_ext) r.flags |= ORC_F_CP_EXT;
        const int nid1 = orc_sss_n_id_1(sync, (unsigned)m0, (unsigned)m1);
        r.n_id_1 = nid1;
        if (nid1 >= 0) {
          r.cell_id = 3 * nid1 + n_id_2; /* srslte_sync_get_cell_id */
          r.flags |= ORC_F_TAG_CELL;
        }
      }
      if (out_hf && nemit < out_cap) memcpy(out_hf + nemit * HF, out, sizeof(orc_cf) * HF);
      nemit++;
    } else {
      next_pos = pos + HF; /* lib/pss_impl.cc:217 */
    }

    if (is_tracking) r.flags |= ORC_F_TRACKING;
    r.score = score; r.psr = psr; r.psr_max = psr_max; r.timer = timer;
    orc_sync_cp_metrics(sync, &r.m_norm_avg, &r.m_ext_avg);
    recs[nrec++] = r;
    pos = next_pos;
  }
  if (n_out) *n_out = nemit;
  free(win); free(out);
  orc_pss_destroy(pss);
  orc_sync_destroy(sync);
  return nrec;
}

void orc_batch_run(const orc_cf *x, int n_channels, int64_t n, const orc_chain_params *p,
                   orc_record *recs, int64_t rec_cap, int64_t *n_recs, int nthreads)
{
  fft_init();
  cexptab_init();
  const int nchains = n_channels * 3;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int c = 0; c < nchains; c++) {
    const int ch = c / 3, root = c % 3;
    n_recs[c] = orc_chain_run(ch, root, x + (int64_t)ch * n, n, p, recs + (int64_t)c * rec_cap,
                              rec_cap, NULL, 0, NULL);
  }
}

/* ------------------------------------------------------------------------- */
/* Decimating FIR: gnuradio.filter.rational_resampler_ccc(1, N) defaults        */
/* [GNU Radio 3.7 rational_resampler.py design_filter(1, N, 0.4) ->             */
/*  firdes.low_pass(1, 1, 0.45/N, 0.1/N, WIN_KAISER, 7.0)]                      */
/* ------------------------------------------------------------------------- */

static double izero(double x) /* [GNU Radio fft::window Izero] */
{
  double sum = 1, u = 1, halfx = x / 2.0;
  int n = 1;
  do {
    double t = halfx / (double)n;
    n++;
    t *= t;
    u *= t;
    sum += u;
  } while (u >= 1e-21 * sum);
  return sum;
}

int orc_decim_ntaps(int decim)
{
  const double beta = 7.0, tw = 0.1 / (double)decim;
  const double atten = beta / 0.1102 + 8.7; /* fft::window::max_attenuation(WIN_KAISER) */
  int ntaps = (int)(atten * 1.0 / (22.0 * tw));
  if ((ntaps & 1) == 0) ntaps++;
  return ntaps;
}

void orc_decim_design(int decim, float *taps)
{
  const int ntaps = orc_decim_ntaps(decim);
  const double beta = 7.0, cutoff = 0.45 / (double)decim;
  const int M = (ntaps - 1) / 2;
  const double fwT0 = 2 * M_PI * cutoff;
  const double ibeta = 1.0 / izero(beta), inm1 = 1.0 / (double)(ntaps - 1);
  double *w = (double *)malloc(sizeof(double) * ntaps);
  for (int i = 0; i < ntaps; i++) {
    double t = 2 * i * inm1 - 1;
    w[i] = (double)(float)(izero(beta * sqrt(1.0 - t * t)) * ibeta); /* window is float */
  }
  for (int nn = -M; nn <= M; nn++) {
    if (nn == 0) taps[nn + M] = (float)(fwT0 / M_PI * w[nn + M]);
    else taps[nn + M] = (float)(sin(nn * fwT0) / (nn * M_PI) * w[nn + M]);
  }
  double fmax = taps[M];
  for (int nn = 1; nn <= M; nn++) fmax += 2 * taps[nn + M];
  const double gain = 1.0 / fmax;
  for (int i = 0; i < ntaps; i++) taps[i] = (float)(taps[i] * gain);
  free(w);
}

/* y[k] = sum_j taps[j] * x[k*decim - j], x[<0] = 0 (history ntaps-1 of zeros).
 * The sum runs in tap order as a chain of fused multiply-adds, acc = fma(taps[j], x, acc) -- one rounding per tap.
 * GNU Radio's fir_filter hands this sum to VOLK, whose kernel (and with it the association and the use of FMA) is
 * chosen per machine: the reference's own rounding is not defined beyond "a dot product in float", no fixture pins
 * a decimated sample, and the cell fields the fixtures do pin (tests/test_oracle_fixtures.py) do not depend on it.
 * Tap order + FMA is what AVX2/FMA VOLK kernels and the GPU's v_fma / v_pk_fma both execute natively. */
int64_t orc_decimate(const orc_cf *x, int64_t n, int decim, const float *taps, int ntaps,
                     orc_cf *y)
{
  const int64_t nout = n / decim;
  for (int64_t k = 0; k < nout; k++) {
    float re = 0.f, im = 0.f;
    const int64_t base = k * decim;
    for (int j = 0; j < ntaps; j++) {
      const int64_t i = base - j;
      if (i < 0) break;
      re = fmaf(taps[j], x[i].re, re);
      im = fmaf(taps[j], x[i].im, im);
    }
    y[k] = cf(re, im);
  }
  return nout;
}

int orc_sizeof_record(void) { return (int)sizeof(orc_record); }
