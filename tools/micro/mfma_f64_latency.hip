// Latency / issue numbers behind csrc/ldlt_mfma.hpp (one wavefront unless stated, cycles from clock64()).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o mfma_f64_latency mfma_f64_latency.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); exit(2); } } while (0)

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double rdlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rcp2(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
  x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
  return x;
}

// mode 0: 64 dependent MFMAs; 1: 64 x 2 independent accumulators; 2: 64 x 4 independent;
// 3: dependent MFMA -> readlane -> MFMA (operand from the result); 4: the pivot step of the LDL^T (16 pivots x 4 reps)
__global__ __launch_bounds__(512) void k_lat(int mode, double* out, long long* cyc, int spinners) {
  __shared__ int s_flag;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_flag = 0;
  __syncthreads();
  if (wv != 0) {
    if (spinners) {
      while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0) { if (spinners == 1) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(8); }
    }
    return;
  }
  const int lr = lane >> 4, lc = lane & 15;
  d4 C, E, C2 = {1, 2, 3, 4}, C3 = {2, 3, 4, 5};
  for (int g = 0; g < 4; g++) { C[g] = (lr + 4 * g == lc) ? 4.0 + lc : 0.01 * (lc + lr + 4 * g); E[g] = (lr + 4 * g == lc) ? 1.0 : 0.0; }
  double a = 1e-3 * lane, b = 1e-3 * (lane ^ 5);
  long long t0 = clock64();
  if (mode == 0) {
#pragma unroll
    for (int i = 0; i < 64; i++) C = mfma(a, b, C);
  } else if (mode == 1) {
#pragma unroll
    for (int i = 0; i < 64; i++) { C = mfma(a, b, C); E = mfma(a, b, E); }
  } else if (mode == 2) {
#pragma unroll
    for (int i = 0; i < 64; i++) { C = mfma(a, b, C); E = mfma(a, b, E); C2 = mfma(a, b, C2); C3 = mfma(a, b, C3); }
  } else if (mode == 3) {
#pragma unroll
    for (int i = 0; i < 64; i++) { const double d = rdlane(C[0], i & 15); C = mfma(a, d * 1e-9, C); }
  } else if (mode == 4) {
    for (int rep = 0; rep < 4; rep++) {
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const int g = j >> 2, q = j & 3;
        const double d = rdlane(C[g], q * 16 + j);
        const double r = rcp2(d);
        const double rm = (lr == q) ? -r : 0.0;
        const double u = C[g];
        const double wc = u * rm, we = E[g] * rm;
        C = mfma(u, wc, C);
        E = mfma(u, we, E);
      }
      for (int g = 0; g < 4; g++) C[g] += (lr + 4 * g == lc) ? 4.0 + lc : 0.01 * (lc + lr + 4 * g);
    }
  } else if (mode == 5) {   // pivot step without the identity copy
    for (int rep = 0; rep < 4; rep++) {
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const int g = j >> 2, q = j & 3;
        const double d = rdlane(C[g], q * 16 + j);
        const double r = rcp2(d);
        const double rm = (lr == q) ? -r : 0.0;
        const double u = C[g];
        C = mfma(u, u * rm, C);
      }
      for (int g = 0; g < 4; g++) C[g] += (lr + 4 * g == lc) ? 4.0 + lc : 0.01 * (lc + lr + 4 * g);
    }
  } else if (mode == 7 || mode == 8) {   // the look-ahead pivot loop of ldlt_mfma.hpp (8: without the identity copy)
    const int npiv = 16 + (spinners > 100);
    for (int rep = 0; rep < 4; rep++) {
      d4 Gc = E, Wc = {0, 0, 0, 0};
      double dvv = 1.0;
      double d = rdlane(C[0], 0);
      bool good = !(d == 0.0 || !(fabs(d) < INFINITY));
      double r = rcp2(d);
#pragma unroll
      for (int j = 0; j < 16; j++) {
        if (j < npiv) {
          const int g = j >> 2, q = j & 3;
          const bool in = lr == q;
          const double rm = in ? -r : 0.0;
          double u = C[g];
          asm volatile("" : "+v"(u));
          double un = u;
          if (j < 15 && ((j + 1) >> 2) != g) { un = C[(j + 1) >> 2]; asm volatile("" : "+v"(un)); }
          const double wc = u * rm;
          C = mfma(u, wc, C);
          __builtin_amdgcn_sched_barrier(0);
          double a2 = 0.0, bb = 0.0;
          if (j < 15) { a2 = rdlane(un, ((j + 1) & 3) * 16 + j + 1); bb = rdlane(u, q * 16 + j + 1); }
          const double eg = E[g];
          const double we = eg * rm;
          Gc[g] = in ? eg : Gc[g];
          Wc[g] -= wc;
          if (lane == j) dvv = r;
          if (j + 1 < npiv) d = __builtin_fma(-r * bb, bb, a2);
          __builtin_amdgcn_sched_barrier(0);
          if (mode == 7) E = mfma(u, we, E);
          __builtin_amdgcn_sched_barrier(0);
          if (j + 1 < npiv) {
            if (d == 0.0 || !(fabs(d) < INFINITY)) good = false;
            r = rcp2(d);
          }
        }
      }
      for (int g = 0; g < 4; g++) { C[g] += ((lr + 4 * g == lc) ? 4.0 + lc : 0.01 * (lc + lr + 4 * g)) + 1e-30 * (Gc[g] + Wc[g] + dvv + good); E[g] = (lr + 4 * g == lc) ? 1.0 : 0.0; }
    }
  } else if (mode == 9) {    // dependent MFMA chain + an independent chain of 8 dependent FMAs per MFMA: do they overlap?
    double v = 1.0 + lane * 1e-3;
#pragma unroll
    for (int i = 0; i < 64; i++) {
      C = mfma(a, b, C);
#pragma unroll
      for (int q = 0; q < 8; q++) v = __builtin_fma(v, 0.999, 1e-3);
    }
    C2[0] = v;
  } else if (mode == 10) {   // MFMA -> v_mov of one result register -> 8 independent FMAs -> MFMA (operand = the copy)
    double v = 1.0 + lane * 1e-3;
#pragma unroll
    for (int i = 0; i < 64; i++) {
      double u = C[0];
      asm volatile("" : "+v"(u));
#pragma unroll
      for (int q = 0; q < 8; q++) v = __builtin_fma(v, 0.999, 1e-3);
      C = mfma(u, b, C);
    }
    C2[0] = v;
  } else if (mode == 11) {   // as 10, the FMAs placed between the MFMA and the copy (in the instruction's shadow)
    double v = 1.0 + lane * 1e-3;
#pragma unroll
    for (int i = 0; i < 64; i++) {
#pragma unroll
      for (int q = 0; q < 8; q++) v = __builtin_fma(v, 0.999, 1e-3);
      asm volatile("" : "+v"(v));
      double u = C[0];
      asm volatile("" : "+v"(u));
      C = mfma(u, b, C);
      asm volatile("" ::: "memory");
    }
    C2[0] = v;
  } else if (mode == 12) {   // MFMA -> readlane of an OPERAND copy (not the result) -> dependent FMA chain; MFMA chain independent
    double v = 1.0 + lane * 1e-3, u = 1e-3 * lane;
#pragma unroll
    for (int i = 0; i < 64; i++) {
      C = mfma(u, b, C);
      const double s = rdlane(u, i);
      v = __builtin_fma(v, s, 1e-3);
    }
    C2[0] = v;
  } else if (mode >= 13 && mode <= 19) {   // mode 5 + the pieces of the real diagonal loop, one more per mode
    __shared__ double s_pub[16 * 64];
    __shared__ int s_cnt;
    const int npiv = 16 + (spinners > 100);
    bool good = true;
    d4 Wc = {0, 0, 0, 0};
    for (int rep = 0; rep < 4; rep++) {
#pragma unroll
      for (int j = 0; j < 16; j++) {
        if ((mode != 16 && mode != 17) || j < npiv) {
          const int g = j >> 2, q = j & 3;
          const double d = rdlane(C[g], q * 16 + j);
          if (mode >= 13 && mode != 18) { if (d == 0.0 || !(fabs(d) < INFINITY)) good = false; }
          const double r = rcp2(d);
          double u = C[g];
          if (mode != 19) asm volatile("" : "+v"(u));
          if (mode >= 15 && mode < 18) {
            s_pub[j * 64 + lane] = (lane == ((q + 1) & 3) * 16) ? r : u;
            asm volatile("" ::: "memory");
            __hip_atomic_store(&s_cnt, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          const double rm = (lr == q) ? -r : 0.0;
          const double wc = u * rm;
          if (mode >= 14 && mode < 18) Wc[g] -= wc;
          if (mode != 17 || j < 15) C = mfma(u, wc, C);
        }
      }
      for (int g = 0; g < 4; g++) C[g] += ((lr + 4 * g == lc) ? 4.0 + lc : 0.01 * (lc + lr + 4 * g)) + 1e-30 * (Wc[g] + good);
    }
  } else if (mode == 6) {   // dependent chain: readlane -> rcp2 only (no MFMA)
    double v = 3.0 + lane;
#pragma unroll
    for (int i = 0; i < 64; i++) { const double d = rdlane(v, i); v = v * 0.5 + rcp2(d); }
    C[0] = v;
  }
  long long t1 = clock64();
  if (lane == 0) { cyc[0] = t1 - t0; __hip_atomic_store(&s_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  double s = 0;
  for (int g = 0; g < 4; g++) s += C[g] + E[g] + C2[g] + C3[g];
  out[lane] = s;
}

int main() {
  double* out; long long* cyc;
  CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 8));
  const char* names[] = {"64 dependent MFMA f64 16x16x4", "64 x 2 independent accumulators", "64 x 4 independent accumulators",
                         "64 x (MFMA -> readlane -> MFMA)", "64 LDL^T pivots (tile + identity copy)", "64 LDL^T pivots (tile only)",
                         "64 x (readlane -> rcp2)", "64 look-ahead pivots (tile + identity copy)", "64 look-ahead pivots (tile only)", "64 x (MFMA || 8 dependent FMAs)", "64 x (copy result, 8 FMAs, MFMA)", "64 x (8 FMAs, copy result, MFMA)", "64 x (MFMA || readlane(operand) -> FMA)", "64 pivots, tile only + finite check", "  + factor row accumulate", "  + LDS publish (row + counter)", "  + runtime pivot count branch", "  + last update skipped", "tile only + operand copy (no check)", "tile only + finite check (no copy)"};
  const int per[] = {64, 128, 256, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64};
  for (int spin = 0; spin <= 0; spin++) {
    for (int mode = 13; mode < 20; mode++) {
      long long c = 0;
      for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_lat, dim3(1), dim3(512), 0, 0, mode, out, cyc, spin);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      }
      printf("spinners s_sleep(%d)%s: %-42s %7lld cycles  = %.1f per item\n", spin, spin ? "" : " (none)", names[mode], c, (double)c / per[mode]);
    }
  }
  return 0;
}
