// CPU emulation of the half-wave FFT-1024 of csrc/lt_fft.h: the 32 lanes are run one after the
// other inside each stage (the stages are separated by one LDS exchange on the device).
// Reads 1024 complex64 from stdin; writes the forward transform in natural order, then the
// unnormalised inverse transform of that result (exercises "output layout == input layout").
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../gr-ltetrigger_amd/csrc/lt_fft.h"

using namespace lt::fft;

template <bool kInv> static void transform(const std::vector<c32> &x, std::vector<c32> &X, const std::vector<c32> &tw)
{
  std::vector<c32> e(kScratch);
  static c32 regs[32][32];
  for (int l = 0; l < 32; ++l) {
    for (int r = 0; r < 32; ++r) regs[l][r] = x[l + 32 * r];
    fft1024_stage1<kInv>(l, regs[l], tw.data(), e.data());
  }
  for (int l = 0; l < 32; ++l) fft1024_stage2<kInv>(l, regs[l], e.data());
  X.assign(kN, c32{0, 0});
  for (int l = 0; l < 32; ++l)
    for (int r = 0; r < 32; ++r) X[l + 32 * r] = regs[l][r];
}

int main()
{
  std::vector<c32> x(kN), tw(32 * 32);
  if (fread(x.data(), sizeof(c32), kN, stdin) != (size_t)kN) return 1;
  for (int r = 0; r < 32; ++r)
    for (int l = 0; l < 32; ++l) {
      const double ph = -2.0 * M_PI * (double)((l * r) % 1024) / 1024.0;
      tw[r * 32 + l] = c32{(float)cos(ph), (float)sin(ph)};
    }
  std::vector<c32> X, Y;
  transform<false>(x, X, tw);
  fwrite(X.data(), sizeof(c32), kN, stdout);
  transform<true>(X, Y, tw); // unnormalised inverse of the forward result = 1024 * x
  fwrite(Y.data(), sizeof(c32), kN, stdout);
  return 0;
}
