// Micro-benchmark: dependent-issue latency of FP64 FMA / rcp / LDS round trip on one wavefront (cycles from s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k_lat(double* out, long long* cyc, double seed, int lanes_active) {
  __shared__ double lds[256];
  const int t = threadIdx.x;
  if (t >= lanes_active) return;
  double a = seed + t * 1e-9, b = 1.0000001, c = 1e-9;
  long long t0, t1;
  // 1) dependent FMA chain
  t0 = clock64();
#pragma unroll
  for (int i = 0; i < 256; i++) { a = __builtin_fma(a, b, c); asm volatile("" : "+v"(a)); }
  t1 = clock64();
  if (t == 0) cyc[0] = t1 - t0;
  // 2) 4 independent FMA chains
  double a0 = a, a1 = a + 1, a2 = a + 2, a3 = a + 3;
  t0 = clock64();
#pragma unroll
  for (int i = 0; i < 64; i++) { a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c); asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
  t1 = clock64();
  if (t == 0) cyc[1] = t1 - t0;
  a = a0 + a1 + a2 + a3;
  // 3) dependent rcp chain
  t0 = clock64();
#pragma unroll
  for (int i = 0; i < 64; i++) a = __builtin_amdgcn_rcp(a);
  asm volatile("" :: "v"(a));
  t1 = clock64();
  if (t == 0) cyc[2] = t1 - t0;
  // 4) dependent LDS round trips (write then read neighbour)
  t0 = clock64();
#pragma unroll
  for (int i = 0; i < 32; i++) { lds[t] = a; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); a = lds[(t + 1) & 63] + 1.0; }
  asm volatile("" :: "v"(a));
  t1 = clock64();
  if (t == 0) cyc[3] = t1 - t0;
  // 5) dependent mul+add (unfused) chain
  t0 = clock64();
#pragma unroll
  for (int i = 0; i < 128; i++) { double m = a * b; asm volatile("" : "+v"(m)); a = m + c; }
  asm volatile("" :: "v"(a));
  t1 = clock64();
  if (t == 0) cyc[4] = t1 - t0;
  // 6) rcp accuracy: max relative error of raw rcp and of one Newton step
  double worst0 = 0, worst1 = 0;
  for (int i = 0; i < 2000; i++) {
    const double d = 1.0 + (i * 64 + t) * (1.0 / 128000.0) * 3.0;
    const double r0 = __builtin_amdgcn_rcp(d);
    const double r1 = __builtin_fma(r0, __builtin_fma(-d, r0, 1.0), r0);
    const double ex = 1.0 / d;
    worst0 = fmax(worst0, fabs(r0 - ex) / ex);
    worst1 = fmax(worst1, fabs(r1 - ex) / ex);
  }
  out[t] = a; out[64 + t] = worst0; out[128 + t] = worst1;
}
// FP64 issue rate of ONE SIMD as a function of the wavefronts resident on it: a workgroup of nw wavefronts (wavefront w runs on
// SIMD w mod 4, tools/micro/wave_simd_map), every wavefront runs 4 independent FMA chains (1024 FMAs) / 4 independent unfused
// mul + add chains (1024 instructions) between two workgroup barriers; cycles are wavefront 0's clock.  The spec sheet's vector FP64
// rate (78.6 TF = 16 lanes per clock per SIMD) is one wave64 instruction every 4 cycles per SIMD; a lone wavefront gets 8.
__global__ void k_issue(double* out, long long* cyc, double seed) {
  const int t = threadIdx.x;
  double b = 1.0000001, c = 1e-9;
  double a0 = seed + t * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  __syncthreads();
  long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < 256; i++) {
    a0 = __builtin_fma(a0, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c);
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
  }
  __syncthreads();
  long long t1 = clock64();
  if (t == 0) cyc[0] = t1 - t0;
  // ONE dependent chain per wavefront (what a serial solve looks like)
  __syncthreads();
  t0 = clock64();
#pragma unroll
  for (int i = 0; i < 512; i++) { a0 = __builtin_fma(a0, b, c); asm volatile("" : "+v"(a0)); }
  __syncthreads();
  t1 = clock64();
  if (t == 0) cyc[2] = t1 - t0;
  __syncthreads();
  t0 = clock64();
#pragma unroll
  for (int i = 0; i < 128; i++) {
    double m0 = a0 * b, m1 = a1 * b, m2 = a2 * b, m3 = a3 * b;
    asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3));
    a0 = m0 + c; a1 = m1 + c; a2 = m2 + c; a3 = m3 + c;
  }
  asm volatile("" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3));
  __syncthreads();
  t1 = clock64();
  if (t == 0) cyc[1] = t1 - t0;
  out[t] = a0 + a1 + a2 + a3;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 192 * 8); hipMalloc(&cyc, 8 * 8);
  for (int lanes : {64, 3}) {
    hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, out, cyc, 1.5, lanes);
    hipDeviceSynchronize();
    long long h[8]; double ho[192];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
    double w0 = 0, w1 = 0; for (int i = 0; i < lanes; i++) { w0 = fmax(w0, ho[64 + i]); w1 = fmax(w1, ho[128 + i]); }
    printf("lanes %d: dep FMA %.1f cyc/op | 4 indep chains %.1f cyc/op | dep rcp %.1f | LDS write->read %.1f | dep mul+add pair %.1f | rcp rel err raw %.3g, 1 Newton %.3g\n",
           lanes, h[0] / 256.0, h[1] / 256.0, h[2] / 64.0, h[3] / 32.0, h[4] / 128.0, w0, w1);
  }
  double* out2; hipMalloc(&out2, 1024 * 8);
  for (int nw : {1, 2, 4, 8, 12, 16}) {
    hipLaunchKernelGGL(k_issue, dim3(1), dim3(64 * nw), 0, 0, out2, cyc, 1.5);
    hipDeviceSynchronize();
    long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double per_simd = nw <= 4 ? 1.0 : nw / 4.0;      // wavefronts sharing the busiest SIMD
    printf("%2d wavefronts in the workgroup (%.0f per SIMD): 4 independent FMA chains %.2f cycles per instruction per wavefront, %.2f per SIMD | "
           "4 independent unfused mul/add chains %.2f per wavefront, %.2f per SIMD | one dependent FMA chain %.2f per wavefront, %.2f per SIMD\n",
           nw, per_simd, h[0] / 1024.0, h[0] / 1024.0 / per_simd, h[1] / 1024.0, h[1] / 1024.0 / per_simd, h[2] / 512.0, h[2] / 512.0 / per_simd);
  }
  return 0;
}
