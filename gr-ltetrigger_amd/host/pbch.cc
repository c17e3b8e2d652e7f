#include "pbch.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace lt {
namespace pbch {
namespace {

using cd = std::complex<double>;
constexpr int kFft = 128;
constexpr int kNre = 72;
constexpr double kPi = 3.14159265358979323846;

// 36.211 7.2: length-31 Gold sequence, Nc = 1600.
std::vector<uint8_t> gold_sequence(uint32_t c_init, int n)
{
  const int nc = 1600;
  std::vector<uint8_t> x1(nc + n + 31, 0), x2(nc + n + 31, 0), c(n);
  x1[0] = 1;
  for (int i = 0; i < 31; ++i) x2[i] = (c_init >> i) & 1u;
  for (int i = 0; i < nc + n; ++i) {
    x1[i + 31] = x1[i + 3] ^ x1[i];
    x2[i + 31] = x2[i + 3] ^ x2[i + 2] ^ x2[i + 1] ^ x2[i];
  }
  for (int i = 0; i < n; ++i) c[i] = x1[nc + i] ^ x2[nc + i];
  return c;
}

void fft128(const std::complex<float> *in, cd out[kFft])
{
  cd a[kFft];
  for (int i = 0; i < kFft; ++i) {
    int r = 0;
    for (int b = 0; b < 7; ++b)
      if (i & (1 << b)) r |= 1 << (6 - b);
    a[r] = cd(in[i].real(), in[i].imag());
  }
  for (int len = 2; len <= kFft; len <<= 1) {
    const double ang = -2.0 * kPi / len;
    for (int i = 0; i < kFft; i += len)
      for (int k = 0; k < len / 2; ++k) {
        const cd w(std::cos(ang * k), std::sin(ang * k));
        const cd u = a[i + k], v = a[i + k + len / 2] * w;
        a[i + k] = u + v;
        a[i + k + len / 2] = u - v;
      }
  }
  for (int i = 0; i < kFft; ++i) out[i] = a[i];
}

int symbol_start(int l, bool cp_ext)
{
  if (cp_ext) return l * (kFft + 32) + 32;
  return l == 0 ? 10 : 138 + (l - 1) * 137 + 9;
}

// 72 centre subcarriers, k = 0..71 <-> bins -36..-1, +1..+36
void ofdm_symbol(const std::complex<float> *slot, int l, bool cp_ext, cd y[kNre])
{
  cd Y[kFft];
  fft128(slot + symbol_start(l, cp_ext), Y);
  for (int k = 0; k < 36; ++k) {
    y[k] = Y[kFft - 36 + k];
    y[36 + k] = Y[1 + k];
  }
}

// CRS subcarriers of `port` in symbol l of slot ns (36.211 6.10.1.2); false if none
bool crs_positions(int cell_id, int port, int ns, int l, bool cp_ext, int pos[12])
{
  const int nsymb = cp_ext ? 6 : 7;
  int v;
  if (port < 2) {
    if (l == 0) v = port == 0 ? 0 : 3;
    else if (l == nsymb - 3) v = port == 0 ? 3 : 0;
    else return false;
  } else {
    if (l != 1) return false;
    v = port == 2 ? 3 * (ns % 2) : 3 + 3 * (ns % 2);
  }
  for (int m = 0; m < 12; ++m) pos[m] = 6 * m + (v + cell_id % 6) % 6;
  return true;
}

void crs_values(int cell_id, int ns, int l, bool cp_ext, cd r[12])
{
  const uint32_t n_cp = cp_ext ? 0 : 1;
  const uint32_t c_init = (1u << 10) * (7u * (ns + 1) + l + 1) * (2u * cell_id + 1) + 2u * cell_id + n_cp;
  const std::vector<uint8_t> c = gold_sequence(c_init, 2 * 220);
  for (int m = 0; m < 12; ++m) {
    const int mp = m + 110 - 6;
    r[m] = cd(1 - 2 * c[2 * mp], 1 - 2 * c[2 * mp + 1]) / std::sqrt(2.0);
  }
}

// LS estimate on slot 1's CRS of `port`, linear interpolation over the 72 subcarriers
void channel_estimate(const std::complex<float> *sf, int cell_id, int port, bool cp_ext, cd h[kNre])
{
  const std::complex<float> *slot1 = sf + 960;
  const int nsymb = cp_ext ? 6 : 7;
  std::vector<std::pair<int, cd>> pilots;
  for (int l = 0; l < nsymb; ++l) {
    int pos[12];
    if (!crs_positions(cell_id, port, 1, l, cp_ext, pos)) continue;
    cd y[kNre], r[12];
    ofdm_symbol(slot1, l, cp_ext, y);
    crs_values(cell_id, 1, l, cp_ext, r);
    for (int m = 0; m < 12; ++m) pilots.emplace_back(pos[m], y[pos[m]] * std::conj(r[m]));
  }
  std::stable_sort(pilots.begin(), pilots.end(),
                   [](const std::pair<int, cd> &a, const std::pair<int, cd> &b) { return a.first < b.first; });
  size_t j = 0;
  for (int k = 0; k < kNre; ++k) {
    if (k <= pilots.front().first) { h[k] = pilots.front().second; continue; }
    if (k >= pilots.back().first) { h[k] = pilots.back().second; continue; }
    while (j + 1 < pilots.size() && pilots[j + 1].first <= k) ++j;
    const int k0 = pilots[j].first, k1 = pilots[j + 1].first;
    if (k1 == k0) { h[k] = pilots[j].second; continue; }
    const double t = double(k - k0) / double(k1 - k0);
    h[k] = pilots[j].second * (1.0 - t) + pilots[j + 1].second * t;
  }
}

const int kPerm[32] = {1, 17, 9, 25, 5, 21, 13, 29, 3, 19, 11, 27, 7, 23, 15, 31,
                       0, 16, 8, 24, 4, 20, 12, 28, 2, 18, 10, 26, 6, 22, 14, 30};

// 36.212 5.1.4.2: circular-buffer order of the 120 coded bits (index = stream*40 + position)
void rate_match_pattern(int pattern[120])
{
  const int D = 40, C = 32, R = 2, nd = R * C - D;
  int n = 0;
  for (int s = 0; s < 3; ++s)
    for (int c = 0; c < C; ++c)
      for (int r = 0; r < R; ++r) {
        const int idx = r * C + kPerm[c] - nd;
        if (idx >= 0) pattern[n++] = s * D + idx;
      }
}

// tail-biting Viterbi, K = 7, generators 133/171/165 (octal); soft[t][i] > 0 means bit 0
void viterbi40(const double soft[40][3], uint8_t bits[40])
{
  static const int polys[3] = {0133, 0171, 0165};
  const int S = 64, passes = 3, T = 40 * passes;
  static int8_t outs[64][2][3];
  static bool init = false;
  if (!init) {
    for (int s = 0; s < S; ++s)
      for (int b = 0; b < 2; ++b) {
        const int reg = (b << 6) | s;
        for (int i = 0; i < 3; ++i) outs[s][b][i] = (int8_t)(__builtin_popcount(reg & polys[i]) & 1);
      }
    init = true;
  }
  std::vector<double> metric(S, 0.0), next(S);
  std::vector<uint8_t> decision((size_t)T * S);
  std::vector<uint8_t> prev((size_t)T * S);
  for (int t = 0; t < T; ++t) {
    const double *l = soft[t % 40];
    std::fill(next.begin(), next.end(), -std::numeric_limits<double>::infinity());
    for (int s = 0; s < S; ++s)
      for (int b = 0; b < 2; ++b) {
        double m = metric[s];
        for (int i = 0; i < 3; ++i) m += outs[s][b][i] ? -l[i] : l[i];
        const int ns = (s >> 1) | (b << 5);
        if (m > next[ns]) {
          next[ns] = m;
          decision[(size_t)t * S + ns] = (uint8_t)b;
          prev[(size_t)t * S + ns] = (uint8_t)s;
        }
      }
    const double mx = *std::max_element(next.begin(), next.end());
    for (int s = 0; s < S; ++s) metric[s] = next[s] - mx;
  }
  int s = (int)(std::max_element(metric.begin(), metric.end()) - metric.begin());
  std::vector<uint8_t> all(T);
  for (int t = T - 1; t >= 0; --t) {
    all[t] = decision[(size_t)t * S + s];
    s = prev[(size_t)t * S + s];
  }
  for (int i = 0; i < 40; ++i) bits[i] = all[40 + i];
}

uint16_t crc16(const uint8_t *bits, int n)
{
  uint16_t reg = 0;
  for (int i = 0; i < n; ++i) {
    const int fb = ((reg >> 15) & 1) ^ bits[i];
    reg = (uint16_t)(reg << 1);
    if (fb) reg ^= 0x1021;
  }
  return reg;
}

} // namespace

bool decode_mib(const std::complex<float> *sf, int cell_id, bool cp_ext, int nof_ports_hint, Mib *out)
{
  const std::complex<float> *slot1 = sf + 960;
  // PBCH resource elements of slot 1, symbols 0..3, CRS positions of all four ports excluded
  std::vector<cd> ys;
  std::vector<int> ks;
  for (int l = 0; l < 4; ++l) {
    cd y[kNre];
    ofdm_symbol(slot1, l, cp_ext, y);
    const bool has_crs = (l == 0 || l == 1 || (cp_ext && l == 3));
    for (int k = 0; k < kNre; ++k) {
      if (has_crs && (k % 3) == (cell_id % 3)) continue;
      ys.push_back(y[k]);
      ks.push_back(k);
    }
  }
  const int nre = (int)ys.size(), nbits = 2 * nre;
  const std::vector<uint8_t> scr = gold_sequence((uint32_t)cell_id, 1920);
  int pattern[120];
  rate_match_pattern(pattern);

  cd hport[4][kNre];
  bool have[4] = {false, false, false, false};
  static const int port_counts[3] = {1, 2, 4};
  static const uint16_t masks[3] = {0x0000, 0xFFFF, 0x5555};
  for (int pc = 0; pc < 3; ++pc) {
    const int nports = port_counts[pc];
    if (nof_ports_hint != 0 && nof_ports_hint != nports) continue;
    for (int p = 0; p < nports; ++p)
      if (!have[p]) {
        channel_estimate(sf, cell_id, p, cp_ext, hport[p]);
        have[p] = true;
      }
    std::vector<cd> x(nre);
    if (nports == 1) {
      for (int i = 0; i < nre; ++i) {
        const cd h = hport[0][ks[i]];
        x[i] = ys[i] * std::conj(h) / std::max(std::norm(h), 1e-12);
      }
    } else { // SFBC (2 ports) / SFBC+FSTD (4 ports), 36.211 6.3.4.3
      for (int i = 0; i + 1 < nre; i += 2) {
        int pa = 0, pb = 1;
        if (nports == 4 && ((i / 2) % 2) == 1) { pa = 1; pb = 3; }
        else if (nports == 4) { pa = 0; pb = 2; }
        const cd h0 = 0.5 * (hport[pa][ks[i]] + hport[pa][ks[i + 1]]);
        const cd h1 = 0.5 * (hport[pb][ks[i]] + hport[pb][ks[i + 1]]);
        const double den = std::max(std::norm(h0) + std::norm(h1), 1e-12);
        const cd r0 = ys[i], r1 = ys[i + 1];
        x[i] = (std::conj(h0) * r0 + h1 * std::conj(r1)) / den * std::sqrt(2.0);
        x[i + 1] = (-h1 * std::conj(r0) + std::conj(h0) * r1) / den * std::sqrt(2.0);
      }
    }
    for (int frame_idx = 0; frame_idx < 4; ++frame_idx) {
      double acc[120] = {0};
      for (int k = 0; k < nbits; ++k) {
        const double s = (k & 1) ? x[k >> 1].imag() : x[k >> 1].real();
        const double d = scr[(size_t)nbits * frame_idx + k] ? -s : s;
        acc[pattern[(nbits * frame_idx + k) % 120]] += d; // position in the 40 ms rate-matched stream
      }
      double soft[40][3];
      for (int t = 0; t < 40; ++t)
        for (int i = 0; i < 3; ++i) soft[t][i] = acc[i * 40 + t];
      uint8_t bits[40];
      viterbi40(soft, bits);
      uint16_t rx = 0;
      for (int i = 24; i < 40; ++i) rx = (uint16_t)((rx << 1) | bits[i]);
      bool any = false;
      for (int i = 0; i < 24; ++i) any = any || bits[i];
      if (!any || (uint16_t)(crc16(bits, 24) ^ masks[pc]) != rx) continue;
      const int bw = bits[0] << 2 | bits[1] << 1 | bits[2];
      static const int bw_table[6] = {6, 15, 25, 50, 75, 100};
      if (bw >= 6) continue;
      out->nof_prb = bw_table[bw];
      out->nof_ports = nports;
      out->phich_length_ext = bits[3];
      out->phich_resources = bits[4] << 1 | bits[5];
      int sfn = 0;
      for (int i = 6; i < 14; ++i) sfn = (sfn << 1) | bits[i];
      out->sfn = sfn << 2;
      out->frame_idx = frame_idx;
      std::memcpy(out->payload, bits, 24);
      return true;
    }
  }
  return false;
}

} // namespace pbch
} // namespace lt
