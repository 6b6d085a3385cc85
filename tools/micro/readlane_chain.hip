// How long is one column of the back-substitution's dependent chain (v_readlane x2 -> v_fma_f64 -> v_readlane ...)?
//   hipcc --offload-arch=gfx950 -O3 -o readlane_chain readlane_chain.hip && ./readlane_chain
// Variants: the chain alone; with the per-lane mask (v_cndmask x2); with a second wavefront spinning on an LDS word on the same SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 2; } } while (0)

__device__ __forceinline__ double rdlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ __launch_bounds__(512) void k_chain(const double* w, double* out, long long* cyc, int reps) {
  __shared__ int s_flag;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_flag = 0;
  __syncthreads();
  if (wv != 0) {
    if (MODE == 2 && wv == 4) {       // same SIMD as wavefront 0
      while (__hip_atomic_load(&s_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {}
    }
    if (MODE == 3) {                  // everybody spins
      while (__hip_atomic_load(&s_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {}
    }
    return;
  }
  double wr[64];
#pragma unroll
  for (int c = 0; c < 64; c++) wr[c] = w[c * 64 + lane];
  double y = w[lane] + 1.0;
  const long long t0 = clock64();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int c = 63; c >= 0; c--) {
      const double m = MODE >= 1 ? (lane < c ? wr[c] : 0.0) : wr[c];
      y -= m * rdlane(y, c);
    }
  }
  const long long t1 = clock64();
  if (lane == 0) { cyc[0] = t1 - t0; __hip_atomic_store(&s_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  out[lane] = y;
}

int main() {
  double *w, *out; long long* cyc;
  CK(hipMalloc(&w, 64 * 64 * 8)); CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 8));
  CK(hipMemset(w, 0, 64 * 64 * 8));
  const int reps = 50;
  for (int mode = 0; mode < 4; mode++) {
    for (int it = 0; it < 2; it++) {
      if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(512), 0, 0, w, out, cyc, reps);
      if (mode == 1) hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(512), 0, 0, w, out, cyc, reps);
      if (mode == 2) hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(512), 0, 0, w, out, cyc, reps);
      if (mode == 3) hipLaunchKernelGGL(k_chain<3>, dim3(1), dim3(512), 0, 0, w, out, cyc, reps);
      CK(hipDeviceSynchronize());
    }
    long long c;
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("mode %d (%s): %.1f clock64 ticks per column\n", mode, mode == 0 ? "chain" : mode == 1 ? "chain + mask" : mode == 2 ? "chain + mask, wavefront 4 spins on LDS" : "chain + mask, 7 wavefronts spin on LDS",
           (double)c / (reps * 64.0));
  }
  return 0;
}
