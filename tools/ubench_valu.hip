// Micro-benchmark: VALU issue rates of plain and packed fp32 ops on gfx950, by waves per SIMD.
// Informs the FFT kernels (csrc/lt_fft.h use v_pk_* everywhere): is a packed op one issue slot or two?
// Build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_valu tools/ubench_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x

template <int kOp> __global__ void k_valu(float *out, int iters, unsigned long long *clk)
{
  v2f a[8];
  const float s = (float)threadIdx.x * 1e-9f;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = v2f{1.0f + s + i, 0.5f + s};
  v2f b = {1.0000001f, 0.9999999f}, c = {1e-9f, -1e-9f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (kOp == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
        if (kOp == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (kOp == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (kOp == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (kOp == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
        if (kOp == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
        if (kOp == 6) { // packed op with op_sel swizzle (as the complex multiplies use)
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(b), "v"(c));
        }
        if (kOp == 7) { // two plain FMAs on the halves (what a non-packed build would issue)
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(b.y), "v"(c.y));
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

// LDS exchange pattern: 16 ds_write_b64 + 16 ds_read_b64 per round (strided), as one FFT exchange
__global__ void k_lds(float *out, int iters, unsigned long long *clk)
{
  extern __shared__ v2f E[];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  v2f *e = E + w * (16 * 68);
  v2f v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = v2f{(float)l, (float)i};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) e[i * 68 + l] = v[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = e[(l >> 2) * 68 + (l & 3) + 4 * i];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main()
{
  float *out;
  unsigned long long *clk;
  CK(hipMalloc(&out, 256 * 1024 * sizeof(float)));
  CK(hipMalloc(&clk, 8));
  const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_add_f32", "v_mul_f32", "v_pk_fma_f32 op_sel", "2x v_fma_f32"};
  const int iters = 4096;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int op = 0; op < 8; ++op) {
    for (int wps : {1, 2, 3, 4, 8}) {
      const int threads = 256 * wps > 1024 ? 1024 : 256 * wps; // waves per SIMD = wps (one or more blocks per CU)
      const int blocks_per_cu = (256 * wps) / threads;
      const int grid = 256 * blocks_per_cu;
      auto launch = [&]() {
        switch (op) {
        case 0: hipLaunchKernelGGL(k_valu<0>, dim3(grid), dim3(threads), 0, 0, out, iters, clk); break;
        case 1: hipLaunchKernelGGL(k_valu<1>, dim3(grid), dim3(threads), 0, 0, out, iters, clk); break;
        case 2: hipLaunchKernelGGL(k_valu<2>, dim3(grid), dim3(threads), 0, 0, out, iters, clk); break;
        case 3: hipLaunchKernelGGL(k_valu<3>, dim3(grid), dim3(threads), 0, 0, out, iters, clk); break;
        case 4: hipLaunchKernelGGL(k_valu<4>, dim3(grid), dim3(threads), 0, 0, out, iters, clk); break;
        case 5: hipLaunchKernelGGL(k_valu<5>, dim3(grid), dim3(threads), 0, 0, out, iters, clk); break;
        case 6: hipLaunchKernelGGL(k_valu<6>, dim3(grid), dim3(threads), 0, 0, out, iters, clk); break;
        default: hipLaunchKernelGGL(k_valu<7>, dim3(grid), dim3(threads), 0, 0, out, iters, clk); break;
        }
      };
      launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long c;
      CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
      const double n_instr = (double)iters * 32 * (op == 7 ? 2 : 1); // per wave
      // s_memtime ticks at 100 MHz-class constant rate on some parts; report both
      printf("%-22s waves/SIMD %d: %.3f ms  memtime-ticks/instr/wave %.2f  ns/instr/SIMD %.3f (= %.2f cyc @2.4GHz)\n", names[op], wps, ms,
             (double)c / n_instr, ms * 1e6 / (n_instr * wps), ms * 1e6 / (n_instr * wps) * 2.4);
    }
  }
  for (int waves : {4, 8, 12, 16}) {
    const size_t lds = (size_t)waves * 16 * 68 * sizeof(v2f);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_lds, dim3(256), dim3(64 * waves), lds, 0, out, 2048, clk);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_lds, dim3(256), dim3(64 * waves), lds, 0, out, 2048, clk);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("LDS exchange (16 wr b64 + 16 rd b64) waves/CU %d: %.3f ms -> %.1f ns per exchange per CU (= %.0f cyc @2.4GHz)\n", waves, ms,
           ms * 1e6 / (2048.0 * waves), ms * 1e6 / (2048.0 * waves) * 2.4);
  }
  return 0;
}
