// Wavefront-level 1024-point complex FFT used by the overlap-save matched filter.
//
// One wave64 transforms one 1024-sample block; lane l owns 16 values at every stage and the three
// stages (1024 = 16 x 16 x 4) exchange data through a wave-private LDS scratch area:
//
//   phase A  lane l = m:          x[64 n1 + m], n1 = 0..15  --radix-16-->  A[k1] * W1024^(m k1)
//   phase B  lane l = 16 m2 + k1: A[k1][4 m1 + m2], m1 = 0..15 --radix-16--> B[q1] * W64^(m2 q1)
//   phase C  lane l = 16 g + k1:  B[k1][q1][m2], q1 = 4g + j   --radix-4-->  X[k1 + 16 q1 + 256 q2]
//
// so after phase C lane l holds X[k] for k = (l & 15) + 16 (4 (l >> 4) + j) + 256 q2, j, q2 = 0..3,
// in register 4 j + q2.  The code is plain C++ over a 2-float vector type and an explicit "LDS"
// pointer so that tests/fft_host_check.cpp can run it lane by lane on the CPU.
#pragma once

#if defined(__HIP__) // compiling in HIP mode (not merely "with hipcc")
#define LT_HD __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define LT_HD inline
#endif

namespace lt {
namespace fft {

typedef float c32 __attribute__((ext_vector_type(2)));

constexpr int kN = 1024;
constexpr int kRowA = 66;         // row stride of exchange 1 / 2 (64 + 2: conflict-free ds_read_b64)
constexpr int kScratch = 1280;    // c32 entries of wave-private scratch (exchange 3 pads 16 per 64)

LT_HD c32 cmul(c32 a, c32 b)
{
  c32 axx = {a.x, a.x}, ayy = {-a.y, a.y}, byx = {b.y, b.x};
  return __builtin_elementwise_fma(ayy, byx, axx * b);
}
LT_HD c32 mul_mj(c32 a) { c32 r = {a.y, -a.x}; return r; } // a * (-j)
LT_HD c32 conj(c32 a) { c32 r = {a.x, -a.y}; return r; }

// forward radix-4 butterfly (W4 = -j): outputs y0..y3 for inputs a, b, c, d
LT_HD void radix4(c32 a, c32 b, c32 c, c32 d, c32 &y0, c32 &y1, c32 &y2, c32 &y3)
{
  const c32 s0 = a + c, s1 = a - c, s2 = b + d, s3 = mul_mj(b - d);
  y0 = s0 + s2;
  y2 = s0 - s2;
  y1 = s1 + s3;
  y3 = s1 - s3;
}

// 16-point forward DFT in registers, natural order in and out (16 = 4 x 4, n = 4 n1 + n2,
// k = k1 + 4 k2).
LT_HD void fft16(c32 v[16])
{
  c32 t[16]; // t[4 k1 + n2]
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) radix4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2], t[n2], t[4 + n2], t[8 + n2], t[12 + n2]);
  // twiddles W16^(n2 k1)
  const c32 w1 = {0.92387953251128674f, -0.38268343236508977f};
  const c32 w2 = {0.70710678118654752f, -0.70710678118654752f};
  const c32 w3 = {0.38268343236508977f, -0.92387953251128674f};
  const c32 w6 = {-0.70710678118654752f, -0.70710678118654752f};
  const c32 w9 = {-0.92387953251128674f, 0.38268343236508977f};
  t[4 * 1 + 1] = cmul(t[4 * 1 + 1], w1);
  t[4 * 1 + 2] = cmul(t[4 * 1 + 2], w2);
  t[4 * 1 + 3] = cmul(t[4 * 1 + 3], w3);
  t[4 * 2 + 1] = cmul(t[4 * 2 + 1], w2);
  t[4 * 2 + 2] = mul_mj(t[4 * 2 + 2]); // W16^4 = -j
  t[4 * 2 + 3] = cmul(t[4 * 2 + 3], w6);
  t[4 * 3 + 1] = cmul(t[4 * 3 + 1], w3);
  t[4 * 3 + 2] = cmul(t[4 * 3 + 2], w6);
  t[4 * 3 + 3] = cmul(t[4 * 3 + 3], w9);
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1)
    radix4(t[4 * k1 + 0], t[4 * k1 + 1], t[4 * k1 + 2], t[4 * k1 + 3], v[k1], v[k1 + 4], v[k1 + 8], v[k1 + 12]);
}

// Twiddle table tw[k1 * 64 + l] = W1024^(l k1), k1 = 0..15, l = 0..63 (c32, 8 KB).
// index of the natural-order element held in register r = 4 j + q2 of lane l after phase C
LT_HD int out_index(int l, int r) { return (l & 15) + 16 * (4 * (l >> 4) + (r >> 2)) + 256 * (r & 3); }
// padded position of natural index k in exchange 3 (16 spare entries per 64: conflict-free both ways)
LT_HD int pad3(int k) { return k + 16 * (k >> 6); }

// ---- the phases.  Between two phases every lane's LDS writes must be visible to the whole wave.
LT_HD void phase_a(int l, c32 v[16], const c32 *tw, c32 *lds)
{
  fft16(v);
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) {
    if (k1) v[k1] = cmul(v[k1], tw[k1 * 64 + l]);
    lds[k1 * kRowA + l] = v[k1];
  }
}

LT_HD void phase_b(int l, c32 v[16], const c32 *tw, c32 *lds_in, c32 *lds_out)
{
  const int k1 = l & 15, m2 = l >> 4;
#pragma unroll
  for (int m1 = 0; m1 < 16; ++m1) v[m1] = lds_in[k1 * kRowA + 4 * m1 + m2];
  fft16(v);
#pragma unroll
  for (int q1 = 0; q1 < 16; ++q1) {
    if (q1) v[q1] = cmul(v[q1], tw[q1 * 64 + 16 * m2]); // W64^(m2 q1) = W1024^(16 m2 q1)
    lds_out[k1 * kRowA + 4 * q1 + m2] = v[q1];
  }
}

LT_HD void phase_c(int l, c32 v[16], const c32 *lds_in)
{
  const int k1 = l & 15, g = l >> 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const c32 *p = lds_in + k1 * kRowA + 4 * (4 * g + j);
    radix4(p[0], p[1], p[2], p[3], v[4 * j + 0], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
  }
}

// natural-order registers (out_index layout) -> phase-A input layout (x[64 n1 + l]) via exchange 3
LT_HD void scatter_natural(int l, const c32 v[16], c32 *lds)
{
#pragma unroll
  for (int r = 0; r < 16; ++r) lds[pad3(out_index(l, r))] = v[r];
}
LT_HD void gather_strided(int l, c32 v[16], const c32 *lds)
{
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) v[n1] = lds[pad3(64 * n1 + l)];
}

} // namespace fft
} // namespace lt
