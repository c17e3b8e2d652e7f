// CPU emulation of the wave-level FFT-1024 of csrc/lt_fft.h: the 64 lanes of the wave are run one
// after the other inside each phase (phases are separated by LDS exchanges on the device).
// Reads 1024 complex64 from stdin, writes the forward transform (natural order) and the result of
// forward -> scatter_natural/gather_strided -> forward again (a second transform through exchange 3).
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../gr-ltetrigger_amd/csrc/lt_fft.h"

using namespace lt::fft;

static void forward(const std::vector<c32> &x, std::vector<c32> &X, const std::vector<c32> &tw)
{
  std::vector<c32> e1(kScratch), e2(kScratch);
  static c32 regs[64][16];
  for (int l = 0; l < 64; ++l) {
    for (int n1 = 0; n1 < 16; ++n1) regs[l][n1] = x[64 * n1 + l];
    phase_a(l, regs[l], tw.data(), e1.data());
  }
  for (int l = 0; l < 64; ++l) phase_b(l, regs[l], tw.data(), e1.data(), e2.data());
  for (int l = 0; l < 64; ++l) phase_c(l, regs[l], e2.data());
  X.assign(kN, c32{0, 0});
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 16; ++r) X[out_index(l, r)] = regs[l][r];
}

int main()
{
  std::vector<c32> x(kN), tw(16 * 64);
  if (fread(x.data(), sizeof(c32), kN, stdin) != (size_t)kN) return 1;
  for (int k1 = 0; k1 < 16; ++k1)
    for (int l = 0; l < 64; ++l) {
      const double ph = -2.0 * M_PI * (double)((l * k1) % 1024) / 1024.0;
      tw[k1 * 64 + l] = c32{(float)cos(ph), (float)sin(ph)};
    }
  std::vector<c32> X, Y;
  forward(x, X, tw);
  fwrite(X.data(), sizeof(c32), kN, stdout);
  // exchange 3 round trip: registers in out_index layout -> strided layout, then transform again
  std::vector<c32> e3(kScratch), x2(kN);
  static c32 regs[64][16];
  for (int l = 0; l < 64; ++l) {
    for (int r = 0; r < 16; ++r) regs[l][r] = X[out_index(l, r)];
    scatter_natural(l, regs[l], e3.data());
  }
  for (int l = 0; l < 64; ++l) {
    gather_strided(l, regs[l], e3.data());
    for (int n1 = 0; n1 < 16; ++n1) x2[64 * n1 + l] = regs[l][n1];
  }
  forward(x2, Y, tw);
  fwrite(Y.data(), sizeof(c32), kN, stdout);
  return 0;
}
