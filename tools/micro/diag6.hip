// Micro-benchmark: the 6x6 diagonal-block step of the block LDL^T (exchange through LDS, right-looking LDL^T,
// explicit inverse of the unit factor, y = X r, stores) executed by one wavefront; cycles per repetition.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
  x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
  return x;
}
template <int STAGE>
__global__ void k_diag(const double* in, double* out, long long* cyc, int reps, int lanes) {
#pragma clang fp contract(fast)
  __shared__ __attribute__((aligned(16))) double Ajj[36], Fjj[32], Lg[36], zz[8], rr[8];
  const int t = threadIdx.x;
  if (t >= lanes) return;
  const int pr_d = t % 3;
  double a[12];
  for (int q = 0; q < 12; q++) a[q] = in[12 * pr_d + q];
  if (t < 6) rr[t] = 1.0 + t;
  long long t0 = clock64();
  double sink = 0;
  for (int rep = 0; rep < reps; rep++) {
    for (int q = 0; q < 12; q++) Ajj[12 * pr_d + q] = a[q] + sink * 1e-30;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double A[6][6], dinv[6], y[6], rj[6];
#pragma unroll
    for (int q = 0; q < 6; q++)
#pragma unroll
      for (int c = 0; c <= q; c++) A[q][c] = Ajj[6 * q + c];
#pragma unroll
    for (int c = 0; c < 6; c++) rj[c] = rr[c];
    if (STAGE >= 1) {
#pragma unroll
      for (int c = 0; c < 6; c++) {
        const double d = A[c][c];
        const double id = fast_rcp(d);
        dinv[c] = id;
        double W[6];
#pragma unroll
        for (int q = c + 1; q < 6; q++) { W[q] = A[q][c]; A[q][c] = W[q] * id; }
#pragma unroll
        for (int q = c + 1; q < 6; q++)
#pragma unroll
          for (int r = c + 1; r <= q; r++) A[q][r] -= A[q][c] * W[r];
      }
    } else {
      for (int c = 0; c < 6; c++) dinv[c] = A[c][c];
    }
    double X[6][6];
    if (STAGE >= 2) {
#pragma unroll
      for (int c = 0; c < 6; c++) {
#pragma unroll
        for (int q = c + 1; q < 6; q++) {
          double v = -A[q][c];
#pragma unroll
          for (int m = c + 1; m < q; m++) v -= A[q][m] * X[m][c];
          X[q][c] = v;
        }
      }
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double v = rj[c];
#pragma unroll
        for (int m = 0; m < c; m++) v += X[c][m] * rj[m];
        y[c] = v;
      }
    } else {
      for (int c = 0; c < 6; c++) { y[c] = rj[c] + A[5][c]; for (int q = 0; q < 6; q++) X[q][c] = A[q > c ? q : c][q > c ? c : q]; }
    }
    if (STAGE >= 3) {
#pragma unroll
      for (int pr = 0; pr < 3; pr++)
        if (pr_d == pr) {
#pragma unroll
          for (int q = 0; q < 2; q++)
#pragma unroll
            for (int c = 0; c < 6; c++)
              Lg[6 * (2 * pr + q) + c] = c < 2 * pr + q ? A[2 * pr + q][c] : (c > 2 * pr + q ? X[c][2 * pr + q] : 1.0);
        }
      if (pr_d == 0) {
#pragma unroll
        for (int c = 0; c < 6; c++) { zz[c] = y[c] * dinv[c]; Fjj[15 + c] = dinv[c]; Fjj[21 + c] = y[c]; }
      } else if (pr_d == 1) {
#pragma unroll
        for (int c = 1; c < 6; c++)
#pragma unroll
          for (int m = 0; m < c; m++) Fjj[c * (c - 1) / 2 + m] = X[c][m];
      }
    }
    sink += y[5] + dinv[5] + X[5][0] + A[5][4];
  }
  long long t1 = clock64();
  if (t == 0) cyc[0] = t1 - t0;
  out[t] = sink + Lg[t % 36] + Fjj[t % 28] + zz[t % 6];
}
template <int STAGE>
void run(const char* name, int lanes) {
  double h[36]; for (int q = 0; q < 6; q++) for (int c = 0; c < 6; c++) h[6 * q + c] = (q == c ? 10.0 + q : 1.0 / (1 + q + c));
  double *in, *out; long long* cyc;
  hipMalloc(&in, sizeof(h)); hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int reps = 200;
  hipLaunchKernelGGL(k_diag<STAGE>, dim3(1), dim3(64), 0, 0, in, out, cyc, reps, lanes);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-40s lanes %2d: %.0f cycles per repetition\n", name, lanes, (double)c / reps);
}
int main() {
  for (int lanes : {3, 63}) {
    run<0>("exchange only", lanes);
    run<1>("exchange + pivots", lanes);
    run<2>("exchange + pivots + X + y", lanes);
    run<3>("exchange + pivots + X + y + stores", lanes);
  }
  return 0;
}
