// Why does k_ldlt_cols take 12.6-14 us in a few launches of the pipelined bench (next to pyr_tower_kernel) and 19-20 us alone?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o ldlt_neighbours ldlt_neighbours.hip && ./ldlt_neighbours
// Times the n = 120 solve (a) alone, back to back, (b) next to a background kernel on a second stream: ALU spinners on every CU,
// memory streamers on every CU, a few workgroups only; and reads the shader clock against the constant 100 MHz counter in each
// setting (a dependent FP64 chain of known length).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define NO_PROFILE 1
#include "../../multi_orbslam3_amd/csrc/ldlt_mfma.hpp"

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void k_spin_alu(volatile int* stop, double* sink, int fp64) {
  double a = threadIdx.x * 1e-3 + 1.0; float f = (float)a; int it = 0;
  while (it < (1 << 16)) {      // ~10-20 ms, no polling of host memory (that starves the command processor's packet fetch)
    if (fp64) { for (int i = 0; i < 64; i++) a = __builtin_fma(a, 1.0000001, 1e-9); }
    else { for (int i = 0; i < 64; i++) f = __builtin_fmaf(f, 1.0000001f, 1e-9f); }
    it++;
  }
  if (a == 123.0 || f == 7.f) sink[0] = a + f;
}

__global__ void k_spin_mem(volatile int* stop, const float4* src, float4* dst, size_t n) {
  int it = 0; float4 acc = {0, 0, 0, 0};
  while (it < 40) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = src[i]; acc.x += v.x; acc.y += v.w; }
    it++;
  }
  if (acc.x == 1.2345f) dst[0] = acc;
}

__global__ void k_clock(long long* out) {       // 4096 dependent FP64 fma: ~8.5 shader cycles each
  double a = 1.0 + threadIdx.x * 1e-9;
  const long long r0 = wall_clock64(), c0 = clock64();
  for (int i = 0; i < 4096; i++) a = __builtin_fma(a, 1.0000001, 1e-9);
  const long long c1 = clock64(), r1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = r1 - r0; out[1] = c1 - c0; out[2] = (long long)a; }
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const int n = 120;
  std::mt19937_64 rng(12345);
  std::normal_distribution<double> N01(0.0, 1.0);
  std::vector<double> M((size_t)n * n), S((size_t)n * n), b(n);
  for (auto& v : M) v = N01(rng);
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      double s = 0;
      for (int k = 0; k < n; k++) s += M[(size_t)i * n + k] * M[(size_t)j * n + k];
      if (i == j) s += 0.05 * n;
      S[(size_t)i * n + j] = S[(size_t)j * n + i] = s;
    }
  for (auto& v : b) v = N01(rng);
  const ldltm::Geo g = ldltm::make_geo(n);
  std::vector<double> im(ldltm::tile_image_doubles(n), 0.0);
  for (int r = 0; r < n; r++) for (int c = 0; c < n; c++) { const int pos = ldltm::tile_image_pos(r, c); if (pos >= 0) im[pos] = S[(size_t)r * n + c]; }
  for (int r = 0; r < n; r++) ldltm::image_put_rhs(im.data(), n, r, b[r]);
  double *dS, *dx, *dw, *sink; int* dok;
  CK(hipMalloc(&dS, im.size() * 8)); CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&dok, 4)); CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&dw, ldltm::wglob_doubles(g) * 8));
  CK(hipMemcpy(dS, im.data(), im.size() * 8, hipMemcpyHostToDevice));
  CK(ldltm::launch_image_pad(n, dS, 0));
  int* stop; CK(hipHostMalloc(&stop, 4, hipHostMallocMapped)); *stop = 0;
  long long* clk; CK(hipHostMalloc(&clk, 64, hipHostMallocMapped));
  const size_t nmem = (size_t)64 << 20;   // 1 GiB of float4
  float4 *msrc, *mdst; CK(hipMalloc(&msrc, nmem * 16)); CK(hipMalloc(&mdst, 64)); CK(hipMemset(msrc, 0, nmem * 16));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  auto time_ldlt = [&](const char* what) {
    const int reps = 200;
    for (int i = 0; i < 20; i++) CK(ldltm::launch(n, dS, dx, dok, dw, s1));
    CK(hipEventRecord(e0, s1));
    for (int i = 0; i < reps; i++) CK(ldltm::launch(n, dS, dx, dok, dw, s1));
    CK(hipEventRecord(e1, s1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, s1, clk);
    CK(hipStreamSynchronize(s1));
    printf("%-52s %6.2f us per solve | 4096 dependent fma: %lld ticks of 100 MHz = %.1f us, clock64 delta %lld\n", what, 1000.0 * ms / reps, clk[0], clk[0] / 100.0, clk[1]);
  };
  auto with_bg = [&](const char* what, auto launch_bg) {
    *stop = 0;
    launch_bg();
    time_ldlt(what);
    *stop = 1;
    CK(hipStreamSynchronize(s2));
  };
  time_ldlt("alone");
  time_ldlt("alone (again)");
  for (int wg : {8, 64, 256, 512, 1024})
    for (int fp64 = 0; fp64 < 2; fp64++) {
      char buf[128]; snprintf(buf, sizeof buf, "next to %d x 256 threads spinning on %s fma", wg, fp64 ? "FP64" : "FP32");
      with_bg(buf, [&] { hipLaunchKernelGGL(k_spin_alu, dim3(wg), dim3(256), 0, s2, stop, sink, fp64); });
    }
  for (int wg : {64, 256, 1024}) {
    char buf[128]; snprintf(buf, sizeof buf, "next to %d x 256 threads streaming 1 GiB", wg);
    with_bg(buf, [&] { hipLaunchKernelGGL(k_spin_mem, dim3(wg), dim3(256), 0, s2, stop, msrc, mdst, nmem); });
  }
  time_ldlt("alone (after)");
  return 0;
}
