// Constant-table generation for the gfx950 cell-search kernels (host side, double precision,
// rounded once to float).  See lt_tables.h for the sources.
#include "lt_tables.h"

#include <cmath>
#include <cstring>

namespace lt {

namespace {

constexpr double kPi = 3.14159265358979323846;

// 36.211 6.11.1.1: d_u(n), roots u = 25, 29, 34.
void zadoff_chu(int n_id_2, double re[62], double im[62])
{
  static const int roots[3] = {25, 29, 34};
  const long u = roots[n_id_2];
  for (int n = 0; n < 62; ++n) {
    const long k = (n < 31) ? u * n * (n + 1) : u * (n + 1) * (n + 2);
    const double ph = -kPi * (double)(k % 126) / 63.0; // exp(-j pi k / 63) has period 126 in k
    re[n] = std::cos(ph);
    im[n] = std::sin(ph);
  }
}

// Matched filter taps: the 62 ZC values sit on bins -31..-1, +1..+31 of a 128-point grid
// (DC empty); the filter is the conjugate of the unnormalised inverse DFT, scaled by
// 1/sqrt(128)/62.
void pss_filter(int n_id_2, cf32 h[128])
{
  double dre[62], dim[62];
  zadoff_chu(n_id_2, dre, dim);
  const double scale = 1.0 / std::sqrt(128.0) / 62.0;
  for (int m = 0; m < 128; ++m) {
    double sr = 0.0, si = 0.0;
    for (int k = 0; k < 62; ++k) {
      const int bin = (k < 31) ? (k - 31) : (k - 30);
      const int e = ((bin * m) % 128 + 128) % 128;
      const double ph = 2.0 * kPi * (double)e / 128.0;
      const double c = std::cos(ph), s = std::sin(ph);
      sr += dre[k] * c - dim[k] * s;
      si += dre[k] * s + dim[k] * c;
    }
    h[m].re = (float)(sr * scale);
    h[m].im = (float)(-si * scale);
  }
}

// 36.211 6.11.2.1: length-31 m-sequences from x(0..4) = 0,0,0,0,1; output 1 - 2x.
void m_sequence(const int taps[], int ntaps, float out[31])
{
  int x[31] = {0};
  x[4] = 1;
  for (int i = 0; i + 5 < 31; ++i) {
    int acc = 0;
    for (int t = 0; t < ntaps; ++t) acc += x[i + taps[t]];
    x[i + 5] = acc % 2;
  }
  for (int i = 0; i < 31; ++i) out[i] = (float)(1 - 2 * x[i]);
}

void pair_entry(float e[2], float re, float im)
{
  e[0] = re;
  e[1] = im;
}

double bessel_i0(double x)
{
  double sum = 1, u = 1;
  const double halfx = x / 2.0;
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

} // namespace

void build_device_tables(DeviceTables *t)
{
  std::memset(t, 0, sizeof *t);
  for (int r = 0; r < 3; ++r) pss_filter(r, t->pss_taps[r]);
  for (int r = 0; r < 3; ++r)
    for (int m = 0; m < 128; ++m) t->pss_taps_pad[r][m] = t->pss_taps[r][m]; // (the rest stays zero)

  for (int i = 0; i < kCexpTabSize; ++i) {
    const double ph = 2.0 * kPi * (double)i / (double)kCexpTabSize;
    t->cexptab[i].re = (float)std::cos(ph);
    t->cexptab[i].im = (float)std::sin(ph);
  }
  for (int e = 0; e < 128; ++e) { // forward DFT twiddles exp(-j 2 pi e / 128)
    const double ph = -2.0 * kPi * (double)e / 128.0;
    t->dft128_cos[e] = std::cos(ph);
    t->dft128_sin[e] = std::sin(ph);
  }

  for (int k1 = 1; k1 < 16; ++k1) // (k1 = 0, t = 0: the factor 1, not stored)
    for (int l = 0; l < 64; ++l) {
      const double ph = -2.0 * kPi * (double)(l * k1) / 1024.0;
      pair_entry(t->fft_tw1[fft::tw1_index(k1, l)], (float)std::cos(ph), (float)std::sin(ph));
    }
  for (int tt = 1; tt < 16; ++tt)
    for (int c = 0; c < 4; ++c) {
      const double ph = -2.0 * kPi * (double)(c * tt) / 64.0;
      pair_entry(t->fft_tw2[fft::tw2_index(tt, c)], (float)std::cos(ph), (float)std::sin(ph));
    }
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 1024; ++k) {
      double sr = 0.0, si = 0.0;
      for (int m = 0; m < 128; ++m) {
        const double ph = -2.0 * kPi * (double)((k * m) % 1024) / 1024.0;
        const double c = std::cos(ph), sn = std::sin(ph);
        const double hr = t->pss_taps[r][m].re, hi = t->pss_taps[r][m].im;
        sr += hr * c - hi * sn;
        si += hr * sn + hi * c;
      }
      t->pss_spec[r][k].re = (float)(sr / 1024.0);
      t->pss_spec[r][k].im = (float)(si / 1024.0);
    }

  float s[31], c[31], z[31];
  static const int ts[] = {2, 0}, tc[] = {3, 0}, tz[] = {4, 2, 1, 0};
  m_sequence(ts, 2, s);
  m_sequence(tc, 2, c);
  m_sequence(tz, 4, z);
  for (int i = 0; i < 31; ++i) {
    t->sss_s[i] = s[i];
    t->sss_z[i] = z[i];
    if (s[i] < 0) t->sss_s_mask |= 1u << i;
    if (z[i] < 0) t->sss_z_mask |= 1u << i;
    for (int r = 0; r < 3; ++r) {
      t->sss_c0[r][i] = c[(i + r) % 31];
      t->sss_c1[r][i] = c[(i + r + 3) % 31];
      if (t->sss_c0[r][i] < 0) t->sss_c0_mask[r] |= 1u << i;
      if (t->sss_c1[r][i] < 0) t->sss_c1_mask[r] |= 1u << i;
    }
  }
  // 36.211 Table 6.11.2.1-1 in closed form; T[m0][m1-1] = N_id_1, everything else stays 0
  // (the zero default is observable: SURVEY 8(a) quirk q5).
  for (int n = 0; n < 168; ++n) {
    const int qp = n / 30;
    const int q = (n + qp * (qp + 1) / 2) / 30;
    const int mp = n + q * (q + 1) / 2;
    const int m0 = mp % 31;
    const int m1 = (m0 + mp / 31 + 1) % 31;
    t->nid1_table[m0][m1 - 1] = n;
  }
}

// GNU Radio 3.7 rational_resampler design_filter(1, N, 0.4):
// firdes.low_pass(1, 1, 0.45/N, 0.1/N, WIN_KAISER, beta=7).
int decim_ntaps(int decimation)
{
  const double beta = 7.0, tw = 0.1 / (double)decimation;
  const double atten = beta / 0.1102 + 8.7;
  int ntaps = (int)(atten * 1.0 / (22.0 * tw));
  if ((ntaps & 1) == 0) ntaps++;
  return ntaps;
}

std::vector<float> decim_design(int decimation)
{
  const int ntaps = decim_ntaps(decimation);
  std::vector<float> taps(ntaps);
  const double beta = 7.0, cutoff = 0.45 / (double)decimation;
  const int M = (ntaps - 1) / 2;
  const double fwT0 = 2 * kPi * cutoff;
  const double ibeta = 1.0 / bessel_i0(beta), inm1 = 1.0 / (double)(ntaps - 1);
  std::vector<double> w(ntaps);
  for (int i = 0; i < ntaps; ++i) {
    const double t = 2 * i * inm1 - 1;
    w[i] = (double)(float)(bessel_i0(beta * std::sqrt(1.0 - t * t)) * ibeta);
  }
  for (int n = -M; n <= M; ++n) {
    if (n == 0) taps[n + M] = (float)(fwT0 / kPi * w[n + M]);
    else taps[n + M] = (float)(std::sin(n * fwT0) / (n * kPi) * w[n + M]);
  }
  double fmax = taps[M];
  for (int n = 1; n <= M; ++n) fmax += 2 * taps[n + M];
  const double gain = 1.0 / fmax;
  for (int i = 0; i < ntaps; ++i) taps[i] = (float)(taps[i] * gain);
  return taps;
}

} // namespace lt
