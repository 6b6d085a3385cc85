// Dense LDL^T + solve of the reduced camera system  S x = b  on the FP64 matrix cores of gfx950
// (v_mfma_f64_16x16x4_f64): the MI355X-native stand-in for g2o's LinearSolverEigen / Eigen::SimplicialLDLT
// (G/solvers/linear_solver_eigen.h:94-124) behind BlockSolver<6,3>::solve (G/core/block_solver.hpp:447).
//
// One workgroup of 8 wavefronts.  The symmetric matrix, bordered by the right-hand side as one more row / column
// (index cb: the forward substitution then happens as part of the elimination), is cut into 16x16 tiles; the upper
// triangle's tiles are dealt cyclically to the wavefronts and stay in REGISTERS for the whole factorisation, in the
// accumulator layout of the instruction (lane l, register g <-> row (l>>4)+4g, column l&15).  Per tile row k:
//
//   diag    the owner of tile (k,k) eliminates its 16 pivots one by one; every pivot is ONE matrix instruction on the
//           tile (rank-1 update  C -= u (r u)^T: the instruction's operand broadcast replaces the LDS / readlane
//           exchange a VALU update needs) and one on a copy of the identity, which collects G = L_kk^-1.
//           Row j of a symmetric tile in accumulator layout IS column j in A-operand layout, so no data moves.
//   panel   the owners of tiles (k,j), j>k: R = G X (4 instructions), W = D^-1 R; -R and W are published in LDS in
//           operand layout, W also goes to the factor store (the unit upper-triangular factor L^T, row blocks).
//   trail   every live tile (i,j), i>k:  U_ij -= R_ki^T W_kj (4 instructions, operands straight from LDS).
//
// The wavefronts run this as a dataflow program (LDS flags, no workgroup barriers): a wavefront handles the tiles of
// row k+1 first and factors tile (k+1,k+1) before it turns to the rest of row k's trailing update, so the chain
// diag(k) -> panel (k,k+1) -> update (k+1,k+1) -> diag(k+1) never waits for bulk work.  The back-substitution
// L^T x = y (y = the border column of the factor store) runs on one wavefront with x in registers (v_readlane
// broadcast, one column per step).  A zero or non-finite pivot clears *ok (SimplicialLDLT's failure rule).
//
// Summation orders are fixed by the tile geometry alone -> results are run-to-run reproducible.
#pragma once

#include <hip/hip_runtime.h>

namespace ldltm {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kGld = 17;          // leading dimension of G in LDS (column-major, padded: reads in A-operand layout)
constexpr int kMaxT = 20;

struct Geo {
  int n, n_pad, cb, T, Tp, ntiles, N;
};

__host__ __device__ inline Geo make_geo(int n) {
  Geo g;
  g.n = n;
  g.n_pad = (n + 3) & ~3;              // pivots are taken four-aligned; padding indices get a unit diagonal
  g.cb = g.n_pad;                      // border index: the right-hand side
  g.T = (g.n_pad + 1 + 15) / 16;
  g.Tp = (g.n_pad + 15) / 16;          // tile rows that hold pivots
  g.ntiles = g.T * (g.T + 1) / 2;
  g.N = 16 * g.T;
  return g;
}

// doubles of dynamic LDS
__host__ __device__ inline size_t lds_doubles(const Geo& g, bool wlds) {
  size_t d = (size_t)2 * g.T * 512 + 2 * 16 * kGld + 32;
  if (wlds) d += (size_t)g.N * (g.N - 1) / 2 + 64;
  return d;
}
__host__ inline size_t wglob_doubles(const Geo& g) { return (size_t)g.N * (g.N - 1) / 2 + 64; }

__device__ __forceinline__ double rdlane(double v, int l) {   // l must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double rcp2(double d) {            // 1/d to ~1 ulp: hardware seed + two Newton steps
  double x = __builtin_amdgcn_rcp(d);
  x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
  x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
  return x;
}

#ifdef LDLTM_PROFILE
__device__ long long g_prof[256];
#define LDLTM_T(slot) do { if (lane == 0) g_prof[slot] = clock64(); } while (0)
#else
#define LDLTM_T(slot) do { } while (0)
#endif

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

template <int NS, int NY, bool WLDS>
__global__ __launch_bounds__(kThreads) void k_ldlt_mfma(int n, const double* __restrict__ S, const double* __restrict__ b,
                                                        double* __restrict__ x, int* __restrict__ ok_flag,
                                                        double* __restrict__ wglob) {
#pragma clang fp contract(fast)        // the solve is tolerance-checked (1e-4), not bit-compared
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ int s_diag;               // tile rows whose G / D^-1 are published
  __shared__ int s_panel[kMaxT + 4];   // per tile column j: rows k for which -R_kj / W_kj are published
  __shared__ int s_rowdone[kMaxT + 4]; // per tile row: wavefronts that finished its trailing update
  __shared__ int s_ok;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const Geo G = make_geo(n);
  const int T = G.T, n_pad = G.n_pad, cb = G.cb;
  double* const Pan = sh;                               // [2][T][2][256]
  double* const Gb = Pan + (size_t)2 * T * 512;         // [2][16 * kGld]
  double* const Dv = Gb + 2 * 16 * kGld;                // [2][16]
  double* const Wl = Dv + 32;                           // factor store in LDS (WLDS)
  auto wm_store = [&](int I, int J, double v) {         // entry (I, J), I < J, of the unit upper factor; column-packed
    const size_t o = (size_t)J * (J - 1) / 2 + I;
    if constexpr (WLDS) Wl[o] = v; else wglob[o] = v;
  };
  if (wv == 0) LDLTM_T(0);
  if (tid == 0) { s_diag = 0; s_ok = 1; }
  if (tid < kMaxT + 4) { s_panel[tid] = 0; s_rowdone[tid] = 0; }

  // ---- tiles of this wavefront: t = j(j+1)/2 + i  (i <= j),  owner t % 8, slot t / 8
  d4 acc[NS];
  int ti[NS], tj[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) {
    const int t = s * kWaves + wv;
    int i = 1 << 20, j = 1 << 20;
    if (t < G.ntiles) {
      j = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
      while ((j + 1) * (j + 2) / 2 <= t) j++;
      while (j * (j + 1) / 2 > t) j--;
      i = t - j * (j + 1) / 2;
    }
    ti[s] = i; tj[s] = j;
    d4 v = {0.0, 0.0, 0.0, 0.0};
    if (t < G.ntiles) {
      const int c = 16 * j + lc;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int r = 16 * i + lr + 4 * g;
        double e = 0.0;
        if (r < n && c < n) e = S[(size_t)r * n + c];
        else if (r == c) e = r < n_pad ? 1.0 : 0.0;
        else if (c == cb && r < n) e = b[r];
        else if (r == cb && c < n) e = b[c];
        v[g] = e;
      }
    }
    acc[s] = v;
  }
  __syncthreads();
  if (wv == 0) LDLTM_T(1);

  auto wait_gt = [&](int* w, int k) {
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) <= k) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  auto wait_free = [&](int k) {   // the LDS buffers of parity k&1 were last used by tile row k-2
    if (k >= 2) wait_gt(&s_rowdone[k - 2], kWaves - 1);
  };

  // ---- the 16 (or fewer, last row) pivots of diagonal tile k
  auto factor = [&](int k) {
    LDLTM_T(8 + 8 * k + 0);
    d4 C = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < NS; s++)
      if (ti[s] == k && tj[s] == k) C = acc[s];
    d4 E, Gc, Wc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < 4; g++) E[g] = (lr + 4 * g == lc) ? 1.0 : 0.0;
    Gc = E;
    double dvv = 1.0;
    bool good = true;
    const int npiv = min(16, n_pad - 16 * k);
#pragma unroll
    for (int j = 0; j < 16; j++) {
      if (j < npiv) {
        const int g = j >> 2, q = j & 3;
        const double d = rdlane(C[g], q * 16 + j);
        if (d == 0.0 || !(fabs(d) < INFINITY)) good = false;
        const double r = rcp2(d);
        const bool in = lr == q;
        const double rm = in ? -r : 0.0;
        const double u = C[g];
        const double wc = u * rm, we = E[g] * rm;
        Wc[g] -= wc;                           // row j of the unit upper factor (lanes of group q), others unchanged
        Gc[g] = in ? E[g] : Gc[g];             // row j of L^-1
        if (lane == j) dvv = r;
        C = mfma(u, wc, C);
        E = mfma(u, we, E);
      }
    }
    LDLTM_T(8 + 8 * k + 1);
    wait_free(k);
    const int par = k & 1;
#pragma unroll
    for (int g = 0; g < 4; g++) Gb[par * 16 * kGld + lc * kGld + lr + 4 * g] = Gc[g];
    if (lane < 16) Dv[par * 16 + lane] = dvv;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int I = 16 * k + lr + 4 * g, J = 16 * k + lc;
      if (I < J && I < n_pad && J <= cb) wm_store(I, J, Wc[g]);
    }
    if (!good) s_ok = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(&s_diag, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    LDLTM_T(8 + 8 * k + 2);
  };

  auto owns_diag = [&](int k) { return (k * (k + 1) / 2 + k) % kWaves == wv; };

  if (owns_diag(0)) factor(0);
  for (int k = 0; k < G.Tp; k++) {
    const int par = k & 1;
    // ---- panel tiles (k, j), j > k
    bool have_g = false;
    double Gf[4], dv4[4];
#pragma unroll
    for (int s = 0; s < NS; s++) {
      if (ti[s] == k && tj[s] > k) {
        if (!have_g) {
          wait_gt(&s_diag, k);
          wait_free(k);
#pragma unroll
          for (int q = 0; q < 4; q++) {
            Gf[q] = Gb[par * 16 * kGld + (4 * q + lr) * kGld + lc];
            dv4[q] = Dv[par * 16 + lr + 4 * q];
          }
          have_g = true;
        }
        if (tj[s] == k + 1) LDLTM_T(8 + 8 * k + 3);
        const d4 X = acc[s];
        d4 R0 = {0.0, 0.0, 0.0, 0.0}, R1 = {0.0, 0.0, 0.0, 0.0};
        R0 = mfma(Gf[0], X[0], R0);
        R1 = mfma(Gf[2], X[2], R1);
        R0 = mfma(Gf[1], X[1], R0);
        R1 = mfma(Gf[3], X[3], R1);
        const int j = tj[s];
        double* const pb = Pan + ((size_t)(par * T + j) * 2) * 256 + lane;
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const double r = R0[g] + R1[g];
          const double w = r * dv4[g];
          pb[g * 64] = -r;
          pb[256 + g * 64] = w;
          const int J = 16 * j + lc;
          if (J <= cb) wm_store(16 * k + lr + 4 * g, J, w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&s_panel[j], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (j == k + 1) LDLTM_T(8 + 8 * k + 4);
      }
    }
    // ---- trailing update with row k: tiles of row k+1 first, then (early) the next diagonal tile, then the rest
    auto trail = [&](bool next_row) {
#pragma unroll
      for (int s = 0; s < NS; s++) {
        const int i = ti[s], j = tj[s];
        if (i > k && i < (1 << 20) && ((i == k + 1) == next_row)) {
          wait_gt(&s_panel[i], k);
          if (j != i) wait_gt(&s_panel[j], k);
          const double* const pa = Pan + ((size_t)(par * T + i) * 2) * 256 + lane;
          const double* const pw = Pan + ((size_t)(par * T + j) * 2 + 1) * 256 + lane;
          double a[4], w[4];
#pragma unroll
          for (int q = 0; q < 4; q++) { a[q] = pa[q * 64]; w[q] = pw[q * 64]; }
          d4 c = acc[s];
#pragma unroll
          for (int q = 0; q < 4; q++) c = mfma(a[q], w[q], c);
          acc[s] = c;
          if (i == k + 1 && j == k + 1) LDLTM_T(8 + 8 * k + 5);
        }
      }
    };
    trail(true);
    if (k + 1 < G.Tp && owns_diag(k + 1)) factor(k + 1);
    trail(false);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(&s_rowdone[k], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if constexpr (!WLDS) __threadfence();
  if (wv == 0) LDLTM_T(2);
  __syncthreads();
  if (wv == 0) LDLTM_T(3);
  const int ok = s_ok;
  // ---- back-substitution  L^T x = y  on one wavefront: x in registers, one column per step
  if (wv == 0 && ok) {
    auto wm_load = [&](int I, int J) -> double {        // 0 outside the strict upper triangle
      if (I >= J) return 0.0;
      const size_t o = (size_t)J * (J - 1) / 2 + I;
      if constexpr (WLDS) return Wl[o]; else return __builtin_nontemporal_load(wglob + o);
    };
    double y[NY];
#pragma unroll
    for (int r = 0; r < NY; r++) { const int I = r * 64 + lane; y[r] = I < n_pad ? wm_load(I, cb) : 0.0; }
#pragma unroll
    for (int rg = NY - 1; rg >= 0; rg--) {
      const int lo = rg * 64, hi = min(n_pad, lo + 64);
      if (hi <= lo) continue;
      double cur[4][NY], nxt[4][NY];
      auto load_group = [&](int J0, double (*buf)[NY]) {
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
          for (int r2 = 0; r2 < NY; r2++) buf[c][r2] = r2 <= rg ? wm_load(r2 * 64 + lane, J0 + c) : 0.0;
      };
      load_group(hi - 4, cur);
      for (int J0 = hi - 4; J0 >= lo; J0 -= 4) {
        if (J0 - 4 >= lo) load_group(J0 - 4, nxt);
#pragma unroll
        for (int c = 3; c >= 0; c--) {
          const double xJ = rdlane(y[rg], J0 + c - lo);
#pragma unroll
          for (int r2 = 0; r2 < NY; r2++)
            if (r2 <= rg) y[r2] -= cur[c][r2] * xJ;
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
          for (int r2 = 0; r2 < NY; r2++) cur[c][r2] = nxt[c][r2];
      }
    }
#pragma unroll
    for (int r = 0; r < NY; r++) { const int I = r * 64 + lane; if (I < n) x[I] = y[r]; }
  }
  if (wv == 0) LDLTM_T(4);
  if (tid == 0) *ok_flag = ok;
}

// true if k_ldlt_mfma covers a system of n unknowns (n = 6 * free poses)
__host__ inline bool supports(int n) { return n >= 1 && make_geo(n).T <= 19; }

struct Launch { const void* fn; size_t lds; bool wlds; };

__host__ inline Launch pick(int n) {
  const Geo g = make_geo(n);
  Launch L;
  if (g.T <= 9) { L.fn = reinterpret_cast<const void*>(k_ldlt_mfma<6, 3, true>); L.wlds = true; }
  else if (g.T <= 13) { L.fn = reinterpret_cast<const void*>(k_ldlt_mfma<12, 4, false>); L.wlds = false; }
  else { L.fn = reinterpret_cast<const void*>(k_ldlt_mfma<24, 5, false>); L.wlds = false; }
  L.lds = lds_doubles(g, L.wlds) * sizeof(double);
  return L;
}

__host__ inline hipError_t launch(int n, const double* S, const double* b, double* x, int* ok, double* wglob, hipStream_t st) {
  const Launch L = pick(n);
  static size_t attr[3] = {0, 0, 0};
  const Geo g = make_geo(n);
  const int which = g.T <= 9 ? 0 : g.T <= 13 ? 1 : 2;
  if (L.lds > 64 * 1024 && attr[which] < L.lds) {
    hipError_t e = hipFuncSetAttribute(L.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds);
    if (e != hipSuccess) return e;
    attr[which] = L.lds;
  }
  void* args[] = {(void*)&n, (void*)&S, (void*)&b, (void*)&x, (void*)&ok, (void*)&wglob};
  return hipLaunchKernel(L.fn, dim3(1), dim3(kThreads), args, L.lds, st);
}

}  // namespace ldltm
