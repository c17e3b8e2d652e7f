// CPU emulation of the FFT-1024 of csrc/lt_fft.h: the 64 lanes are run one after the other inside each
// stage (the stages are separated by LDS exchanges on the device).
// Reads 1024 complex64 from stdin; writes the forward transform in natural order, then the
// unnormalised inverse transform of that result (exercises "output layout == input layout").
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../gr-ltetrigger_amd/csrc/lt_fft.h"

using namespace lt::fft;

// 64 lanes x 16 values, two exchanges
template <bool kInv> static void transform_w(const std::vector<c32> &x, std::vector<c32> &X, const std::vector<c32> &tw1,
                                             const std::vector<c32> &tw2)
{
  std::vector<c32> e1(kScratchW), e2(kScratchW); // the device aliases them; lane-serial emulation cannot
  static c32 regs[64][16];
  for (int l = 0; l < 64; ++l) {
    for (int r = 0; r < 16; ++r) regs[l][r] = x[l + 64 * r];
    fft1024w_stageA<kInv>(l, regs[l], tw1.data(), e1.data());
  }
  for (int l = 0; l < 64; ++l) fft1024w_stageB<kInv>(l, regs[l], tw2.data(), e1.data(), e2.data());
  for (int l = 0; l < 64; ++l) fft1024w_stageC<kInv>(l, regs[l], e2.data());
  X.assign(kN, c32{0, 0});
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 16; ++r) X[l + 64 * r] = regs[l][r];
}

int main()
{
  std::vector<c32> x(kN);
  if (fread(x.data(), sizeof(c32), kN, stdin) != (size_t)kN) return 1;
  std::vector<c32> X, Y;
  std::vector<c32> tw1(kTw1Len), tw2(kTw2Len);
  for (int k1 = 1; k1 < 16; ++k1)
    for (int l = 0; l < 64; ++l) {
      const double ph = -2.0 * M_PI * (double)(l * k1) / 1024.0;
      tw1[tw1_index(k1, l)] = c32{(float)cos(ph), (float)sin(ph)};
    }
  for (int t = 1; t < 16; ++t)
    for (int c = 0; c < 4; ++c) {
      const double ph = -2.0 * M_PI * (double)(c * t) / 64.0;
      tw2[tw2_index(t, c)] = c32{(float)cos(ph), (float)sin(ph)};
    }
  transform_w<false>(x, X, tw1, tw2);
  fwrite(X.data(), sizeof(c32), kN, stdout);
  transform_w<true>(X, Y, tw1, tw2); // unnormalised inverse of the forward result = 1024 * x
  fwrite(Y.data(), sizeof(c32), kN, stdout);
  return 0;
}
