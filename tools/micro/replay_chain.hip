// The replay chain of k_ldlt_cols (fast path: operands already in registers) on one wavefront: cycles per pair.
// variant bits: 1 = with the D update per group of two pairs, 2 = second wavefront on the same SIMD runs the same chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); exit(2); } } while (0)
__device__ __forceinline__ d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double rdlane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double row_even_to_odd(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]);
}
template <int V>
__global__ __launch_bounds__(512) void k_replay(double* out, long long* cyc) {
  __shared__ double s_a[8 * 64];
  __shared__ double s_r[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = lane >> 4, lc = lane & 15;
  if (wv == 0) { for (int i = 0; i < 8; i++) s_a[i * 64 + lane] = (lr == (2 * (i & 1)) || lr == 2 * (i & 1) + 1) ? 1e-3 * (lane + i) : 0.0; if (lane < 16) s_r[lane] = 0.5 + lane; }
  __syncthreads();
  const bool active = wv == 0 || ((V & 2) && wv == 4);
  if (!active) return;
  d4 X, D = {0, 0, 0, 0}, Rc = {0, 0, 0, 0}, nW = {0, 0, 0, 0};
  for (int g = 0; g < 4; g++) X[g] = 0.01 * (lc + lr + 4 * g) + 1.0;
  double a8[8], rs[4];
  for (int i = 0; i < 8; i++) a8[i] = s_a[i * 64 + lane];
  for (int g = 0; g < 4; g++) rs[g] = s_r[lr + 4 * g];
  long long t0 = clock64();
  for (int rep = 0; rep < 4; rep++) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int q0 = 2 * h, p1 = 4 * g + q0 + 1;
        const double a = a8[2 * g + h];
        double xg = X[g];
        asm volatile("" : "+v"(xg));
        const double x0b = row_even_to_odd(xg);
        const double nl10 = rdlane(a, q0 * 16 + p1);
        const double rrow = (lr == q0 + 1) ? __builtin_fma(nl10, x0b, xg) : xg;
        Rc[g] = (h == 0 || lr >= 2) ? rrow : Rc[g];
        X = mfma(a, rrow, X);
      }
      if (V & 1) { nW[g] = Rc[g] * -rs[g]; D = mfma(Rc[g], nW[g], D); }
    }
  }
  long long t1 = clock64();
  if (lane == 0 && wv == 0) cyc[0] = t1 - t0;
  out[threadIdx.x & 127] = X[0] + X[1] + X[2] + X[3] + D[0] + D[1] + D[2] + D[3] + nW[0] + Rc[1];
}
template <int V>
static void run(double* out, long long* cyc) {
  long long c = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_replay<V>, dim3(1), dim3(512), 0, 0, out, cyc);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  }
  printf("variant %d (%s%s): %.1f cycles per pair\n", V, V & 1 ? "D-update " : "", V & 2 ? "simd-mate " : "", (double)c / 32);
}
int main() {
  double* out; long long* cyc;
  CK(hipMalloc(&out, 128 * 8)); CK(hipMalloc(&cyc, 8));
  run<0>(out, cyc); run<1>(out, cyc); run<2>(out, cyc); run<3>(out, cyc);
  return 0;
}
