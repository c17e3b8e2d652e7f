// Stand-alone timing of lt_scan_kernel (csrc/lt_scan_kernels.h) on random data: 64 jobs x 64 windows (one scan round),
// all three roots live.  Also checks the DPP whole-wave rotates the strip resolver relies on.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o build/scan_bench tools/scan_bench.hip csrc/lt_tables.cpp
#include "../gr-ltetrigger_amd/csrc/lt_scan_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace lt;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_rot(float *out)
{
  const float v = (float)threadIdx.x;
  out[threadIdx.x] = wave_next(v);
  out[64 + threadIdx.x] = wave_prev(v);
}

int main(int argc, char **argv)
{
  const int n_ch = argc > 1 ? atoi(argv[1]) : 64;
  int n_win = argc > 2 ? atoi(argv[2]) : kScanMaxRound;
  if (n_win > kScanMaxRound) n_win = kScanMaxRound;
  const int64_t n = (int64_t)n_win * kHalfFrame;
  {
    float *d;
    CK(hipMalloc(&d, 128 * sizeof(float)));
    hipLaunchKernelGGL(k_rot, dim3(1), dim3(64), 0, 0, d);
    float h[128];
    CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    printf("wave_next: lane0=%g lane1=%g lane63=%g   wave_prev: lane0=%g lane1=%g lane63=%g\n", h[0], h[1], h[63], h[64], h[65], h[127]);
    bool ok = true;
    for (int l = 0; l < 64; ++l) ok = ok && h[l] == (float)((l + 1) & 63) && h[64 + l] == (float)((l + 63) & 63);
    printf("whole-wave rotates %s\n", ok ? "OK" : "WRONG");
  }
  DeviceTables *htab = new DeviceTables, *dtab;
  build_device_tables(htab);
  CK(hipMalloc(&dtab, sizeof(DeviceTables)));
  CK(hipMemcpy(dtab, htab, sizeof(DeviceTables), hipMemcpyHostToDevice));
  std::vector<float> hiq((size_t)n_ch * n * 2);
  unsigned s = 12345;
  for (auto &v : hiq) {
    s = s * 1664525u + 1013904223u;
    v = ((int)(s >> 8) & 0xFFFF) / 32768.0f - 1.0f;
  }
  v2f *diq;
  CK(hipMalloc(&diq, hiq.size() * sizeof(float)));
  CK(hipMemcpy(diq, hiq.data(), hiq.size() * sizeof(float), hipMemcpyHostToDevice));
  // roots mask, then optionally the number of sequences the roots are spread over (1: all on one grid; 2/3:
  // root r >= 1 rides in a sequence of its own, offset by 1000 (r) samples)
  const int roots = argc > 3 ? atoi(argv[3]) : 7;
  const int n_seq = argc > 4 ? atoi(argv[4]) : 1;
  std::vector<ScanJob> jobs(n_ch);
  for (int c = 0; c < n_ch; ++c) {
    ScanJob &j = jobs[c];
    memset(&j, 0, sizeof(j));
    j.channel = c;
    j.n_seq = n_seq;
    for (int q = 0; q < n_seq; ++q) {
      j.seq[q].pos_a = 0; j.seq[q].stride_a = 1; j.seq[q].n_a = 0;
      j.seq[q].pos_b = 1000 * q; j.seq[q].stride_b = 1; j.seq[q].n_b = n_win;
      j.seq[q].zero = 1;
    }
    for (int r = 0; r < 3; ++r) {
      const int q = r < n_seq ? r : 0;
      j.seq_of_root[r] = (roots >> r) & 1 ? q : -1;
      if ((roots >> r) & 1) j.seq[q].roots |= 1 << r;
    }
  }
  ScanJob *djobs;
  int *dnj;
  CK(hipMalloc(&djobs, sizeof(ScanJob) * n_ch));
  CK(hipMemcpy(djobs, jobs.data(), sizeof(ScanJob) * n_ch, hipMemcpyHostToDevice));
  CK(hipMalloc(&dnj, 2 * sizeof(int)));
  const int njj[2] = {n_ch, 0};
  CK(hipMemcpy(dnj, njj, sizeof njj, hipMemcpyHostToDevice));
  ChainState *dst;
  CK(hipMalloc(&dst, sizeof(ChainState) * n_ch * 3));
  CK(hipMemset(dst, 0, sizeof(ChainState) * n_ch * 3));
  {
    std::vector<ChainState> hst(n_ch * 3);
    memset(hst.data(), 0, sizeof(ChainState) * hst.size());
    CK(hipMemcpy(dst, hst.data(), sizeof(ChainState) * hst.size(), hipMemcpyHostToDevice));
  }
  const size_t n_parts = (size_t)n_ch * 3 * kScanMaxRound * kStrips;
  StripHead *dheads;
  float *drows;
  CK(hipMalloc(&dheads, sizeof(StripHead) * n_parts));
  CK(hipMalloc(&drows, sizeof(float) * kRowLags * n_parts));
  CK(hipMemset(dheads, 0, sizeof(StripHead) * n_parts));
  CK(hipMemset(drows, 0, sizeof(float) * kRowLags * n_parts));
  ScanArgs a;
  a.iq = diq; a.ch_stride = n; a.hist = nullptr; a.hist_stride = 0; a.t0 = 0; a.n = n; a.hist_len = 0;
  a.jobs = djobs; a.n_jobs = dnj; a.state = dst; a.heads = dheads; a.rows = drows; a.tab = dtab; a.jobs_cap = n_ch;
  unsigned long long *dprof;
  CK(hipMalloc(&dprof, 8 * 128));
  CK(hipMemset(dprof, 0, 8 * 128));
  a.prof = dprof;
  a.cache = nullptr;
  const int G = argc > 5 ? atoi(argv[5]) : 3; // strips per workgroup
  const int grid = ((n_ch * scan_wg_per_job(G) + 7) / 8) * 8;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int n_iter = argc > 6 ? atoi(argv[6]) : 4; // launches (the GPU's clock settles after ~20 ms of them)
  for (int it = 0; it < n_iter; ++it) {
    CK(hipEventRecord(e0));
    if (G == 1) hipLaunchKernelGGL(lt_scan_kernel<1>, dim3(grid), dim3(scan_threads(1)), 0, 0, a);
    else if (G == 2) hipLaunchKernelGGL(lt_scan_kernel<2>, dim3(grid), dim3(scan_threads(2)), 0, 0, a);
    else hipLaunchKernelGGL(lt_scan_kernel<3>, dim3(grid), dim3(scan_threads(3)), 0, 0, a);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (it < 4 || it + 2 >= n_iter) printf("scan %d ch x %d windows: %.3f ms  (%.1f Gsamp/s, %.3f of 8 TB/s at 8 B/sample)\n", n_ch, n_win, ms,
           (double)n_ch * n / ms * 1e-6, (double)n_ch * n * 8 / (ms * 1e-3) / 8e12);
  }
#ifdef LT_SCAN_PROF
  {
    unsigned long long hpr[128];
    CK(hipMemcpy(hpr, dprof, 8 * 128, hipMemcpyDeviceToHost));
    const char *nm[8] = {"wait fwd", "fwd C", "x spec", "inv A", "inv B", "inv C", "mag+ema", "resolve"};
    printf("  clocks per window (s_memtime ticks) of the root waves of workgroup 8:\n  wave simd slot");
    for (int i = 0; i < 8; ++i) printf(" %8s", nm[i]);
    printf("    total\n");
    for (int w = 0; w < 3; ++w) {
      const unsigned id = (unsigned)hpr[96 + w];
      printf("  forward wave %d simd %u slot %u: loads+moves %.0f, stage A (+ wait for samples) %.0f, wait for buffer %.0f, stage B %.0f\n", w,
             (id >> 4) & 3, id & 15, (double)hpr[w * 8] / n_win, (double)hpr[w * 8 + 1] / n_win, (double)hpr[w * 8 + 2] / n_win, (double)hpr[w * 8 + 3] / n_win);
    }
    for (int w = 3; w < 12; ++w) {
      unsigned long long tot = 0;
      const unsigned id = (unsigned)hpr[96 + w];
      printf("  %4d %4u %4u", w, (id >> 4) & 3, id & 15);
      for (int i = 0; i < 8; ++i) {
        tot += hpr[w * 8 + i];
        printf(" %8.0f", (double)hpr[w * 8 + i] / n_win);
      }
      printf(" %8.0f\n", (double)tot / n_win);
    }
  }
#endif
  {
    // checksum of what the strips leave for the resolver (heads and rows), and of the resolver's results on them (computed on
    // the host with the kernel's own arithmetic order: peak position and the bits of the side-lobe maximum)
    std::vector<StripHead> heads(n_parts);
    std::vector<float> rows(n_parts * kRowLags);
    CK(hipMemcpy(heads.data(), dheads, sizeof(StripHead) * n_parts, hipMemcpyDeviceToHost));
    CK(hipMemcpy(rows.data(), drows, sizeof(float) * rows.size(), hipMemcpyDeviceToHost));
    unsigned long long sum = 0, rsum = 0;
    for (int c = 0; c < n_ch * 3; ++c)
      if ((roots >> (c % 3)) & 1)
        for (int k = 0; k < n_win; ++k)
          for (int st = 0; st < kStrips; ++st) {
            const size_t i = ((size_t)c * kScanMaxRound + k) * kStrips + st;
            const StripHead &q = heads[i];
            unsigned m, f, a0;
            memcpy(&m, &q.m1, 4);
            memcpy(&f, &q.far, 4);
            memcpy(&a0, &q.a0, 4);
            sum = sum * 1000003ull + m + 3ull * f + 7ull * (unsigned)q.p1f + 5ull * a0;
            for (int j = 0; j < kRowLags; ++j) {
              unsigned v;
              memcpy(&v, &rows[i * kRowLags + j], 4);
              rsum = rsum * 1000003ull + v;
            }
          }
    printf("checksum of the heads (roots %d, %d sequences): %016llx   of the rows: %016llx\n", roots, n_seq, sum, rsum);
    for (int i = 0; i < 24; ++i)
      printf("head[%d]: m1=%g p1=%d far=%g a0=%g  row[14..18]=%g %g %g %g %g\n", i, heads[i].m1, heads[i].p1f, heads[i].far, heads[i].a0, rows[(size_t)i * kRowLags + kRowHalf - 2],
             rows[(size_t)i * kRowLags + kRowHalf - 1], rows[(size_t)i * kRowLags + kRowHalf], rows[(size_t)i * kRowLags + kRowHalf + 1], rows[(size_t)i * kRowLags + kRowHalf + 2]);
  }
  return 0;
}
