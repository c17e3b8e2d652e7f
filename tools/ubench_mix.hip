// Micro-benchmark: do LDS exchanges of one wave overlap with packed VALU work of the other waves on the same SIMD?
// 12 waves per CU (3 per SIMD); `n_lds` of every 3 consecutive-SIMD-slot waves run an LDS exchange loop, the
// rest a v_pk_fma loop.  Compare with each kind alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(768) void k_mix(float *out, int iters_valu, int iters_lds, int lds_mask, int mode)
{
  __shared__ v2f E[12 * 16 * 68];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool do_lds = (lds_mask >> w) & 1;
  float acc = 0.f;
  if (do_lds) {
    v2f *e = E + w * (16 * 68);
    v2f v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = v2f{(float)l, (float)i};
    for (int it = 0; it < iters_lds; ++it) {
      if (mode == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) e[i * 68 + l] = v[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = e[(l >> 2) * 68 + (l & 3) + 4 * i];
      } else { // explicit single b64 ops (no read2 / write2 merging)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("ds_write_b64 %0, %1" :: "v"((unsigned)(size_t)(e + i * 68 + l)), "v"(v[i]) : "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("ds_read_b64 %0, %1" : "=v"(v[i]) : "v"((unsigned)(size_t)(e + (l >> 2) * 68 + (l & 3) + 4 * i)) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += v[i].x + v[i].y;
  } else {
    v2f a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = v2f{1.0f + i, 0.5f};
    v2f b = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
    for (int it = 0; it < iters_valu; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
  float *out;
  hipMalloc(&out, 256 * 768 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](const char *name, int iv, int il, int mask, int mode) {
    hipLaunchKernelGGL(k_mix, dim3(256), dim3(768), 0, 0, out, iv, il, mask, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mix, dim3(256), dim3(768), 0, 0, out, iv, il, mask, mode);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s %.3f ms\n", name, ms);
  };
  // waves 0..11; SIMD of wave w = w % 4 (cyclic).  mask 0x111 = waves 0,4,8 (one SIMD); 0x249 = waves 0,3,6,9 (one per SIMD... spread)
  for (int mode = 0; mode < 2; ++mode) {
    printf("mode %d (%s)\n", mode, mode ? "asm single b64" : "compiler-merged");
    run("12 waves VALU only (4096 x 32 pk_fma)", 4096, 0, 0x000, mode);
    run("12 waves LDS only (2048 exchanges)", 0, 2048, 0xFFF, mode);
    run("4 waves LDS (0,3,6,9) x2048, 8 waves idle-ish", 0, 2048, 0x249, mode);
    run("4 waves LDS x2048 + 8 waves VALU 4096x32", 4096, 2048, 0x249, mode);
    run("8 waves VALU 4096x32 alone", 4096, 0, 0x1000, mode);
    run("4 waves LDS x4096 + 8 waves VALU 4096x32", 4096, 4096, 0x249, mode);
    run("6 waves LDS x2048 + 6 waves VALU 4096x32", 4096, 2048, 0x555, mode);
  }
  return 0;
}
