// 1024-point complex FFT of the overlap-save matched filter, one transform per wave64: 64 lanes x 16 values.
// Plain C++ over a 2-float vector type and an explicit "LDS" pointer, so that
// tests/hostcheck/fft_host_check.cpp can run it lane by lane on the CPU.
#pragma once

#if defined(__HIP__) // compiling in HIP mode (not merely "with hipcc")
#define LT_HD __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define LT_HD inline
#endif

// Scheduling fence (device code only): pins the order "reads issued, butterflies, twiddle products and writes".
#if defined(__HIP_DEVICE_COMPILE__)
#define LT_FFT_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define LT_FFT_FENCE ((void)0)
#endif

namespace lt {
namespace fft {

typedef float c32 __attribute__((ext_vector_type(2)));

constexpr int kN = 1024;

// Complex products written so that every sign lives in a constant operand of a packed FMA (hipcc
// turns a negated *variable* operand into v_xor + v_mov, but folds constants for free).
// cmul<false>(a, b) = a * b,  cmul<true>(a, b) = a * conj(b).
template <bool kConjB> LT_HD c32 cmul(c32 a, c32 b)
{
  const c32 axx = {a.x, a.x}, ayy = {a.y, a.y}, byx = {b.y, b.x};
  const c32 p = axx * b;   // (ax bx, ax by)
  const c32 q = ayy * byx; // (ay by, ay bx)
  if (kConjB) {
    const c32 s = {1.f, -1.f};
    return __builtin_elementwise_fma(p, s, q); // (ax bx + ay by, -ax by + ay bx)
  }
  const c32 s = {-1.f, 1.f};
  return __builtin_elementwise_fma(q, s, p); // (ax bx - ay by, ax by + ay bx)
}
template <bool kInv> LT_HD c32 cw(c32 w) { return kInv ? c32{w.x, -w.y} : w; } // compile-time twiddle
// a * w for a compile-time w: both w and j w fold into constant operands, 2 packed ops
LT_HD c32 cmulc(c32 a, c32 w)
{
  const c32 axx = {a.x, a.x}, ayy = {a.y, a.y}, jw = {-w.y, w.x};
  return __builtin_elementwise_fma(axx, w, ayy * jw); // (ax wx - ay wy, ax wy + ay wx)
}
// a * w (kConjB: a * conj(w)) for a twiddle read from a table: two packed operations -- the operand swaps, the
// broadcast of a.x / a.y and the one negation are operand modifiers of v_pk_mul_f32 / v_pk_fma_f32 (written as
// inline assembly: the compiler turns a negated *variable* operand into v_xor + v_mov and a swap into v_mov's).
//   p = (ay wy, ay wx);   a w = (ax wx - p.x, ax wy + p.y);   a conj(w) = (ax wx + p.x, -(ax wy) + p.y)
template <bool kConjB> LT_HD c32 cmulw(c32 a, c32 w)
{
#if defined(__HIP_DEVICE_COMPILE__)
  c32 p, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(w));
  if (kConjB)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(p));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(p));
  return r;
#else
  const float px = a.y * w.y, py = a.y * w.x;
  if (kConjB) return c32{__builtin_fmaf(a.x, w.x, px), __builtin_fmaf(a.x, -w.y, py)};
  return c32{__builtin_fmaf(a.x, w.x, -px), __builtin_fmaf(a.x, w.y, py)};
#endif
}

// Two such products at once, the two v_pk_mul first: a v_pk_fma that takes the product written by the instruction
// right before it as its addend costs a wait state (the compiler pads the pair with an s_nop) and, in a wave that is
// alone on its SIMD, the full latency of the multiply.
template <bool kConjB> LT_HD void cmulw2(c32 &a0, c32 w0, c32 &a1, c32 w1)
{
#if defined(__HIP_DEVICE_COMPILE__)
  c32 p0, p1;
  if (kConjB)
    asm("v_pk_mul_f32 %[p0], %[a0], %[w0] op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %[p1], %[a1], %[w1] op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %[a0], %[a0], %[w0], %[p0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %[a1], %[a1], %[w1], %[p1] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
        : [a0] "+v"(a0), [a1] "+v"(a1), [p0] "=&v"(p0), [p1] "=&v"(p1)
        : [w0] "v"(w0), [w1] "v"(w1));
  else
    asm("v_pk_mul_f32 %[p0], %[a0], %[w0] op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %[p1], %[a1], %[w1] op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %[a0], %[a0], %[w0], %[p0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n\t"
        "v_pk_fma_f32 %[a1], %[a1], %[w1], %[p1] op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
        : [a0] "+v"(a0), [a1] "+v"(a1), [p0] "=&v"(p0), [p1] "=&v"(p1)
        : [w0] "v"(w0), [w1] "v"(w1));
#else
  a0 = cmulw<kConjB>(a0, w0);
  a1 = cmulw<kConjB>(a1, w1);
#endif
}

// radix-4 butterfly, forward (W4 = -j) or inverse (W4 = +j)
template <bool kInv> LT_HD void radix4(c32 a, c32 b, c32 c, c32 d, c32 &y0, c32 &y1, c32 &y2, c32 &y3)
{
  const c32 s0 = a + c, s1 = a - c, s2 = b + d, t = b - d;
  const c32 tyx = {t.y, t.x};
  const c32 mj = {1.f, -1.f}, pj = {-1.f, 1.f}; // t * (-j) = (t.y, -t.x);  t * (+j) = (-t.y, t.x)
  y0 = s0 + s2;
  y2 = s0 - s2;
  y1 = __builtin_elementwise_fma(tyx, kInv ? pj : mj, s1);
  y3 = __builtin_elementwise_fma(tyx, kInv ? mj : pj, s1);
}

// 16-point DFT of v[0], v[S], ..., v[15 S] in place, natural order in and out
// (16 = 4 x 4, n = 4 n1 + n2, k = k1 + 4 k2).
template <int S, bool kInv> LT_HD void fft16(c32 *v)
{
  c32 t[16]; // t[4 k1 + n2]
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2)
    radix4<kInv>(v[S * n2], v[S * (4 + n2)], v[S * (8 + n2)], v[S * (12 + n2)], t[n2], t[4 + n2], t[8 + n2], t[12 + n2]);
  const c32 w1 = cw<kInv>(c32{0.92387953251128674f, -0.38268343236508977f});
  const c32 w2 = cw<kInv>(c32{0.70710678118654752f, -0.70710678118654752f});
  const c32 w3 = cw<kInv>(c32{0.38268343236508977f, -0.92387953251128674f});
  const c32 w4 = cw<kInv>(c32{0.f, -1.f});
  const c32 w6 = cw<kInv>(c32{-0.70710678118654752f, -0.70710678118654752f});
  const c32 w9 = cw<kInv>(c32{-0.92387953251128674f, 0.38268343236508977f});
  // (three at a time: the products a.y * (j w) first, then the fused a.x * w + product -- a packed FMA whose addend
  // was written by the instruction right before it costs a wait state)
  const int idx[9] = {5, 6, 7, 9, 10, 11, 13, 14, 15};
  const c32 ww[9] = {w1, w2, w3, w2, w4, w6, w3, w6, w9};
#pragma unroll
  for (int g = 0; g < 9; g += 3) {
    c32 q[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const c32 a = t[idx[g + i]], ayy = {a.y, a.y}, jw = {-ww[g + i].y, ww[g + i].x};
      q[i] = ayy * jw;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const c32 a = t[idx[g + i]], axx = {a.x, a.x};
      t[idx[g + i]] = __builtin_elementwise_fma(axx, ww[g + i], q[i]);
    }
  }
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1)
    radix4<kInv>(t[4 * k1 + 0], t[4 * k1 + 1], t[4 * k1 + 2], t[4 * k1 + 3], v[S * k1], v[S * (k1 + 4)],
                 v[S * (k1 + 8)], v[S * (k1 + 12)]);
}

// 64 lanes transform one 1024-sample block, 16 values per lane (1024 = 16 x 16 x 4).  Element index
// i = lane + 64 * reg on input AND output, so the inverse transform (same code, conjugated twiddles) can be
// applied to the product with the filter spectrum directly.
//   stage A  radix-16 over reg:  a[k1] = sum_r x[l + 64 r] W16^(r k1),  times W1024^(l k1)
//   exchange 1                   E1[k1][l]  ->  lane (k1, c) = 4 k1 + c reads E1[k1][c + 4 s], s = 0..15
//   stage B  radix-16 over s:    b[t] = sum_s E1[k1][c + 4 s] W16^(s t),  times W64^(c t)
//   exchange 2                   E2[c][k1 + 16 t]  ->  lane l reads E2[0..3][l + 64 i], i = 0..3
//   stage C  radix-4 over c:     X[(l + 64 i) + 256 u] = sum_c E2[c][l + 64 i] W4^(c u)   (reg i + 4 u)
// LDS layouts, chosen by what the reads cost (MI355X_MICROARCH.md, LDS: ds_read_b64 and ds_read_b128 deliver 256 bytes
// per clock, the two-address forms the compiler merges separate 8-byte reads into only 128):
//   E1[16][68]: within a row the element l = c + 4 s sits at 8 (s / 2) + 2 c + (s % 2), so that lane (k1, c) reads its
//     sixteen s as eight 16-byte pairs.  A 16-lane group of the 8-byte stores writes 16 consecutive entries of one row:
//     conflict-free whatever the row stride.  The 16-byte reads of a group come from four rows (k1 = 0, 3, 5, 6 and the
//     like), 16 banks each; at stride 68 two of the four overlap on 8 banks (one extra LDS cycle per read).  Stride 72
//     would have none, and 6 KB more LDS per workgroup than the scan kernel has.
//   E2: what a lane reads together sits together: two halves {c = 0, 1} and {c = 2, 3} of 256 16-byte entries
//     (entry p = k1 + 16 t holds the two c of its half), so that stage C is eight 16-byte reads of consecutive
//     entries by consecutive lanes; the halves start 16 banks apart (8 entries of padding), which spreads the 8-byte
//     stores of a 16-lane group -- four k1 times four c -- over all 32 banks.
//   Twiddles the same way: tw1 holds W1024^(l k1) for k1 = 2 j + 1, 2 j + 2 side by side (entry j * 64 + l), tw2 holds
//     W64^(c t) for t = 2 j + 1, 2 j + 2 side by side in rows of 9 entries per c (the ninth is padding: rows on banks of
//     their own): eight 16-byte reads per stage instead of fifteen 8-byte ones.
// E1 and E2 may be the same memory: a wave has read all of E1 before it writes E2 (program order).
struct alignas(16) c32x2 {
  c32 a, b;
};
constexpr int kRowA = 68;                  // E1[16][68]
constexpr int kE2Half = 2 * 256 + 8;       // c32 entries from one half of E2 to the other
constexpr int kE2Len = 2 * kE2Half;        // c32 entries of an E2 (a hand-over buffer of the scan kernel)
constexpr int kScratchW = 16 * kRowA;      // c32 entries of scratch per transform (8704 B; an E2 fits too)
static_assert(kE2Len <= kScratchW, "exchange 2 must fit into the exchange area");
constexpr int kTw1Len = 16 * 64;           // c32 entries of the tables
constexpr int kTw2Len = 4 * 9 * 2;
// (index 0 of either table is the factor 1 and is not stored: entry j holds the indices 2 j + 1 and 2 j + 2)
LT_HD int tw1_index(int k1, int l) { return (((k1 - 1) >> 1) * 64 + l) * 2 + ((k1 - 1) & 1); } // W1024^(l k1), 1 <= k1 < 16, l < 64
LT_HD int tw2_index(int t, int c) { return (c * 9 + ((t - 1) >> 1)) * 2 + ((t - 1) & 1); }      // W64^(c t), 1 <= t < 16, c < 4
LT_HD int e2_index(int c, int p) { return (c >> 1) * kE2Half + 2 * p + (c & 1); }
// place of element l = c + 4 s in a row of E1: s = 2 j and 2 j + 1 side by side, the four c of a j next to each other
LT_HD int e1_pos(int l) { return ((l >> 3) << 3) | ((l & 3) << 1) | ((l >> 2) & 1); }
// A stage's twiddles are read before its butterflies (15 values, 30 registers): a wave's LDS reads return in
// 100-250 cycles while other waves keep the LDS busy, and read group by group after the butterflies that
// latency was exposed four times per stage (more than half of the stage's time).
struct Twiddles { c32 w[15]; };
LT_HD void stageA_twiddles(int l, const c32 *tw1, Twiddles &t)
{
  const c32x2 *p = reinterpret_cast<const c32x2 *>(tw1) + l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const c32x2 v = p[j * 64];
    t.w[2 * j] = v.a;                 // k1 = 2 j + 1
    if (j < 7) t.w[2 * j + 1] = v.b;  // k1 = 2 j + 2
  }
}
LT_HD void stageB_twiddles(int l, const c32 *tw2, Twiddles &t)
{
  const c32x2 *p = reinterpret_cast<const c32x2 *>(tw2) + (l & 3) * 9;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const c32x2 v = p[j];
    t.w[2 * j] = v.a;
    if (j < 7) t.w[2 * j + 1] = v.b;
  }
}
template <bool kInv> LT_HD void fft1024w_stageA(int l, c32 v[16], const Twiddles &t, c32 *e)
{
  fft16<1, kInv>(v);
  LT_FFT_FENCE;
  c32 *out = e + e1_pos(l);
  out[0] = v[0];
  v[1] = cmulw<kInv>(v[1], t.w[0]);
  out[kRowA] = v[1];
#pragma unroll
  for (int k1 = 2; k1 < 16; k1 += 2) {
    cmulw2<kInv>(v[k1], t.w[k1 - 1], v[k1 + 1], t.w[k1]);
    out[k1 * kRowA] = v[k1];
    out[(k1 + 1) * kRowA] = v[k1 + 1];
  }
}
template <bool kInv> LT_HD void fft1024w_stageB(int l, c32 v[16], const c32 *tw2, const c32 *e1, c32 *e2)
{
  const int k1 = l >> 2, c = l & 3;
  const c32x2 *in = reinterpret_cast<const c32x2 *>(e1 + k1 * kRowA + 2 * c);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const c32x2 p = in[4 * j]; // s = 2 j, 2 j + 1
    v[2 * j] = p.a;
    v[2 * j + 1] = p.b;
  }
  Twiddles t;
  stageB_twiddles(l, tw2, t);
  LT_FFT_FENCE;
  fft16<1, kInv>(v);
  LT_FFT_FENCE;
  c32 *out = e2 + e2_index(c, k1);
  out[0] = v[0];
  v[1] = cmulw<kInv>(v[1], t.w[0]);
  out[2 * 16] = v[1]; // entry p = k1 + 16 k
#pragma unroll
  for (int k = 2; k < 16; k += 2) {
    cmulw2<kInv>(v[k], t.w[k - 1], v[k + 1], t.w[k]);
    out[2 * 16 * k] = v[k];
    out[2 * 16 * (k + 1)] = v[k + 1];
  }
}
// (stage A reading its own twiddles: for callers with nothing to overlap the reads with)
template <bool kInv> LT_HD void fft1024w_stageA(int l, c32 v[16], const c32 *tw1, c32 *e)
{
  Twiddles t;
  stageA_twiddles(l, tw1, t);
  LT_FFT_FENCE;
  fft1024w_stageA<kInv>(l, v, t, e);
}
template <bool kInv> LT_HD void fft1024w_stageC(int l, c32 v[16], const c32 *e2)
{
  const c32x2 *lo = reinterpret_cast<const c32x2 *>(e2) + l, *hi = reinterpret_cast<const c32x2 *>(e2 + kE2Half) + l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const c32x2 a = lo[64 * i], b = hi[64 * i];
    radix4<kInv>(a.a, a.b, b.a, b.b, v[i], v[i + 4], v[i + 8], v[i + 12]);
  }
}

} // namespace fft
} // namespace lt
