// Host-side generation of the constant tables the kernels read (uploaded once per device).
// Sources: 3GPP TS 36.211 6.11.1 (PSS), 6.11.2 (SSS); the srsLTE 18.06.1 table layout the
// reference relies on; GNU Radio 3.7 firdes for the decimator.
#pragma once
#include <cstdint>
#include <vector>

#include "lt_fft.h" // table geometry

namespace lt {

struct cf32 { float re, im; };

constexpr int kSlotLen = 960;
constexpr int kHalfFrame = 9600;
constexpr int kSymbolSz = 128;
constexpr int kConvLen = 9728;
constexpr int kNLags = 9726;
constexpr int kMovingAvgSz = 200;
constexpr int kCexpTabSize = 4096;

// Everything the kernels need, in one POD that is copied to the device verbatim.
struct DeviceTables {
  cf32 pss_taps[3][128];      // h[m] = conj(x_pss[m]) / sqrt(128) / 62, per N_id_2
  cf32 cexptab[kCexpTabSize]; // exp(j 2 pi i / 4096)
  double dft128_cos[128];     // cos(2 pi e / 128)
  double dft128_sin[128];     // sin(2 pi e / 128)
  float sss_c0[3][32];        // c0[n] = c~[(n + N_id_2) % 31], per N_id_2 (padded to 32)
  float sss_c1[3][32];        // c1[n] = c~[(n + N_id_2 + 3) % 31]
  float sss_s[32];            // s~[n]
  float sss_z[32];            // z~[n]
  int32_t nid1_table[30][30]; // T[m0][m1-1] = N_id_1, zero elsewhere
  // the same m-sequences as sign masks: bit i set <=> element i is -1
  uint32_t sss_s_mask, sss_z_mask, sss_c0_mask[3], sss_c1_mask[3];
  // overlap-save matched filter (csrc/lt_fft.h): the 1024-point spectrum of the zero-padded filter divided by
  // 1024, natural order (element k = lane + 64 reg is what lane `lane` holds in register `reg`), and the
  // transform's twiddles W1024^(l k1) and W64^(c t), (re, im), in the paired layouts the kernels read 16 bytes at a
  // time (fft::tw1_index, fft::tw2_index; the padding entries of fft_tw2 are zero)
  cf32 pss_spec[3][1024];
  float fft_tw1[fft::kTw1Len][2];
  float fft_tw2[fft::kTw2Len][2];
  // the taps once more with zeros behind them: the stream kernel's edge sums read 6 x 22 = 132 of them, wave-uniformly
  // (scalar loads), csrc/lt_chain_kernel.h
  cf32 pss_taps_pad[3][136];
};

void build_device_tables(DeviceTables *t);

int decim_ntaps(int decimation);
std::vector<float> decim_design(int decimation);

} // namespace lt
