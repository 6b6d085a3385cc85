// Cost of the two-pivots-per-instruction loop of k_ldlt_cols (csrc/ldlt_mfma.hpp) on one wavefront, alone and next to
// seven wavefronts that poll the published counter.  bits: 1 LDS publish, 2 pollers, 4 one Newton step, 8 no det form
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
__device__ __forceinline__ double row_even_to_odd(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]);
}
template <int V>
__global__ __launch_bounds__(512) void k_pair(double* out, long long* cyc, int sleep) {
  __shared__ double s_pub[8 * 64];
  __shared__ int s_cnt;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = lane >> 4, lc = lane & 15;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  if (wv > 0) {
    if (V & 2) {
      while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < 1000) {
        if (sleep) __builtin_amdgcn_s_sleep(1);
      }
    }
    return;
  }
  d4 C, Wc = {0, 0, 0, 0};
  for (int g = 0; g < 4; g++) C[g] = (lr + 4 * g == lc) ? 4.0 + lc : 0.01 * (lc + lr + 4 * g);
  bool bad = false;
  long long t0 = clock64();
  for (int rep = 0; rep < 4; rep++) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int q0 = 2 * h, p0 = 4 * g + q0, p1 = p0 + 1;
        double u = C[g];
        asm volatile("" : "+v"(u));
        const double c00 = rdlane(u, q0 * 16 + p0), c01 = rdlane(u, q0 * 16 + p1), c11 = rdlane(u, (q0 + 1) * 16 + p1);
        double r0, r1, nl10;
        if (V & 8) {
          r0 = (V & 4) ? rcpn<1>(c00) : rcpn<2>(c00);
          nl10 = -(c01 * r0);
          const double d1 = __builtin_fma(nl10, c01, c11);
          r1 = (V & 4) ? rcpn<1>(d1) : rcpn<2>(d1);
          bad |= d1 == 0.0;
        } else {
          const double det = __builtin_fma(c00, c11, -(c01 * c01));
          r0 = (V & 4) ? rcpn<1>(c00) : rcpn<2>(c00);
          const double rdet = (V & 4) ? rcpn<1>(det) : rcpn<2>(det);
          r1 = c00 * rdet;
          nl10 = -(c01 * r0);
          bad |= det == 0.0;
        }
        const double u0b = row_even_to_odd(u);
        const double u1 = __builtin_fma(nl10, u0b, u);
        const bool in0 = lr == q0, in1 = lr == q0 + 1;
        const double bv = in1 ? u1 : u;
        const double av = in0 ? u * -r0 : in1 ? u1 * -r1 : 0.0;
        if (V & 1) {
          s_pub[(2 * g + h) * 64 + lane] = av;
          asm volatile("" ::: "memory");
          __hip_atomic_store(&s_cnt, p1 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        Wc[g] += av;
        C = mfma(av, bv, C);
      }
    }
    for (int g = 0; g < 4; g++) C[g] += ((lr + 4 * g == lc) ? 4.0 + lc : 0.01 * (lc + lr + 4 * g)) + 1e-30 * (Wc[g] + bad);
  }
  long long t1 = clock64();
  if (lane == 0) cyc[0] = t1 - t0;
  out[lane] = C[0] + C[1] + C[2] + C[3];
  __hip_atomic_store(&s_cnt, 1000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int V>
static void run(double* out, long long* cyc, int sleep) {
  long long c = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_pair<V>, dim3(1), dim3(512), 0, 0, out, cyc, sleep);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  }
  printf("variant %2d (%s%s%s%s%s): %.1f cycles per pair\n", V, V & 1 ? "publish " : "", V & 2 ? "pollers " : "", V & 4 ? "newton1 " : "",
         V & 8 ? "serial-rcp " : "", sleep ? "sleep" : "", (double)c / 32);
}
int main() {
  double* out; long long* cyc;
  CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 8));
  run<0>(out, cyc, 0); run<1>(out, cyc, 0); run<2>(out, cyc, 0); run<3>(out, cyc, 0); run<3>(out, cyc, 1); run<4>(out, cyc, 0); run<8>(out, cyc, 0);
  run<7>(out, cyc, 0); run<7>(out, cyc, 1);
  return 0;
}
