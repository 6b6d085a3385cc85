// Which piece of the diagonal pivot loop of csrc/ldlt_mfma.hpp costs what (one wavefront, compile-time variants).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); exit(2); } } while (0)
__device__ __forceinline__ d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double rdlane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <int NEWTON>
__device__ __forceinline__ double rcpn(double d) {
  double x = __builtin_amdgcn_rcp(d);
#pragma unroll
  for (int i = 0; i < NEWTON; i++) x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
  return x;
}
// bits: 1 asm operand copy, 2 finite check, 4 runtime pivot-count branch, 8 LDS publish, 16 factor-row accumulate, 32 one Newton step
template <int V>
__global__ __launch_bounds__(64) void k_piv(double* out, long long* cyc, int npiv) {
  __shared__ double s_pub[16 * 64];
  __shared__ int s_cnt;
  const int lane = threadIdx.x, lr = lane >> 4, lc = lane & 15;
  d4 C, Wc = {0, 0, 0, 0};
  for (int g = 0; g < 4; g++) C[g] = (lr + 4 * g == lc) ? 4.0 + lc : 0.01 * (lc + lr + 4 * g);
  bool good = true;
  long long t0 = clock64();
  for (int rep = 0; rep < 4; rep++) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      if (!(V & 4) || j < npiv) {
        const int g = j >> 2, q = j & 3;
        const double d = rdlane(C[g], q * 16 + j);
        if (V & 2) { if (d == 0.0 || !(fabs(d) < INFINITY)) good = false; }
        const double r = (V & 32) ? rcpn<1>(d) : rcpn<2>(d);
        double u = C[g];
        if (V & 1) asm volatile("" : "+v"(u));
        if (V & 8) {
          s_pub[j * 64 + lane] = (lane == ((q + 1) & 3) * 16) ? r : u;
          asm volatile("" ::: "memory");
          __hip_atomic_store(&s_cnt, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const double rm = (lr == q) ? -r : 0.0;
        const double wc = u * rm;
        if (V & 16) Wc[g] -= wc;
        C = mfma(u, wc, C);
      }
    }
    for (int g = 0; g < 4; g++) C[g] += ((lr + 4 * g == lc) ? 4.0 + lc : 0.01 * (lc + lr + 4 * g)) + 1e-30 * (Wc[g] + good);
  }
  long long t1 = clock64();
  if (lane == 0) cyc[0] = t1 - t0;
  out[lane] = C[0] + C[1] + C[2] + C[3];
}
template <int V>
static void run(double* out, long long* cyc) {
  long long c = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_piv<V>, dim3(1), dim3(64), 0, 0, out, cyc, 16);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  }
  printf("variant %2d (%s%s%s%s%s%s): %.1f cycles per pivot\n", V, V & 1 ? "copy " : "", V & 2 ? "check " : "", V & 4 ? "branch " : "",
         V & 8 ? "publish " : "", V & 16 ? "rowacc " : "", V & 32 ? "newton1 " : "", (double)c / 64);
}
int main() {
  double* out; long long* cyc;
  CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 8));
  run<0>(out, cyc); run<1>(out, cyc); run<2>(out, cyc); run<4>(out, cyc); run<8>(out, cyc); run<16>(out, cyc); run<32>(out, cyc);
  run<3>(out, cyc); run<7>(out, cyc); run<15>(out, cyc); run<31>(out, cyc); run<63>(out, cyc); run<30>(out, cyc); run<62>(out, cyc);
  return 0;
}
