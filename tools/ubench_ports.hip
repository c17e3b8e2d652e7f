// Micro-benchmark: how the waves of one workgroup share a SIMD's vector issue on gfx950.
// Each wave of a workgroup runs the same chain of independent v_add_f32 / v_fma_f32 and reports its own elapsed
// s_memtime ticks together with the SIMD and wave slot it ran on (HW_REG_HW_ID).  With 12 waves per workgroup
// and one workgroup per CU (the scan kernel's shape) a SIMD holds three waves: do all three slow down equally,
// or do two of them share an issue port while the third keeps one to itself?
// A second pass gives every wave a weight (amount of work) to see what a light wave costs its neighbours.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_ports tools/ubench_ports.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct WaveRep { unsigned long long ticks; unsigned hw_id; unsigned pad; };

template <int kOp> __global__ void k_ports(float *out, int iters, const int *weight, WaveRep *rep, int prio_from = 99)
{
  extern __shared__ float lds_pad[]; // (sized to keep one workgroup per CU)
  float a[8];
  const float s = (float)threadIdx.x * 1e-9f;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = 1.0f + s + i;
  const float b = 1.0000001f, c = 1e-9f;
  const int w = threadIdx.x >> 6;
  const int n = iters * weight[w];
  if (w >= prio_from) __builtin_amdgcn_s_setprio(2);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (kOp == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (kOp == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 5) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    rep[w].ticks = t1 - t0;
    rep[w].hw_id = id;
  }
  if (threadIdx.x == 0) lds_pad[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int kOp> static int run(const char *name, int waves, const std::vector<int> &weight, float *out, int *dw, WaveRep *drep, int prio_from = 99)
{
  const int iters = 2000;
  CK(hipMemcpy(dw, weight.data(), sizeof(int) * waves, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void *)k_ports<kOp>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  k_ports<kOp><<<256, waves * 64, 100 * 1024>>>(out, 10, dw, drep, prio_from);
  CK(hipEventRecord(e0));
  k_ports<kOp><<<256, waves * 64, 100 * 1024>>>(out, iters, dw, drep, prio_from);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<WaveRep> rep(waves);
  CK(hipMemcpy(rep.data(), drep, sizeof(WaveRep) * waves, hipMemcpyDeviceToHost));
  printf("%s, %d waves per workgroup, one workgroup per CU: %.3f ms\n", name, waves, ms);
  for (int w = 0; w < waves; ++w) {
    const unsigned id = rep[w].hw_id;
    const double per = weight[w] ? (double)rep[w].ticks / ((double)iters * weight[w] * 32.0) : 0.0;
    printf("  wave %2d  weight %d  simd %u  slot %2u  cu %2u  ticks %9llu  (%.2f ticks per instruction)\n", w, weight[w], (id >> 4) & 3,
           id & 15, (id >> 8) & 15, rep[w].ticks, per);
  }
  return 0;
}

int main()
{
  float *out;
  int *dw;
  WaveRep *drep;
  CK(hipMalloc(&out, 256 * 1024 * sizeof(float)));
  CK(hipMalloc(&dw, 64 * sizeof(int)));
  CK(hipMalloc(&drep, 64 * sizeof(WaveRep)));
  for (int waves : {4, 8, 12, 16}) {
    std::vector<int> one(waves, 1);
    if (run<0>("v_add_f32, equal work", waves, one, out, dw, drep)) return 1;
  }
  {
    std::vector<int> one(12, 1);
    if (run<1>("v_fma_f32, equal work", 12, one, out, dw, drep)) return 1;
    // the scan kernel's shape: waves 0..2 light (forward), 3..11 heavy (root)
    std::vector<int> scan = {1, 1, 1, 3, 3, 3, 3, 3, 3, 3, 3, 3};
    if (run<0>("v_add_f32, waves 0..2 a third of the work", 12, scan, out, dw, drep)) return 1;
    // light waves last
    std::vector<int> scan2 = {3, 3, 3, 3, 3, 3, 3, 3, 3, 1, 1, 1};
    if (run<0>("v_add_f32, waves 9..11 a third of the work", 12, scan2, out, dw, drep)) return 1;
    // which waves does a SIMD's third wave compete with?  (weight 0 = the wave does nothing)
    std::vector<int> no1 = {1, 1, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1};
    if (run<0>("v_add_f32, second wave of every SIMD idle", 12, no1, out, dw, drep)) return 1;
    std::vector<int> no0 = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1};
    if (run<0>("v_add_f32, first wave of every SIMD idle", 12, no0, out, dw, drep)) return 1;
    // does s_setprio move the penalty?  the third wave of every SIMD at raised priority
    if (run<0>("v_add_f32, equal work, waves 8..11 at s_setprio 2", 12, one, out, dw, drep, 8)) return 1;
  }
  return 0;
}
