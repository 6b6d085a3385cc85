// Dense LDL^T + solve of the reduced camera system  S x = b  on the FP64 matrix cores of gfx950
// (v_mfma_f64_16x16x4_f64): the MI355X-native stand-in for g2o's LinearSolverEigen / Eigen::SimplicialLDLT
// (G/solvers/linear_solver_eigen.h:94-124) behind BlockSolver<6,3>::solve (G/core/block_solver.hpp:447).
//
// One workgroup of 8 wavefronts.  The symmetric matrix, bordered by the right-hand side as one more row / column
// (index cb: the forward substitution then happens as part of the elimination), is cut into 16x16 tiles; the upper
// triangle's tiles are dealt cyclically to the wavefronts and stay in REGISTERS for the whole factorisation, in the
// accumulator layout of the instruction (lane l, register g <-> row (l>>4)+4g, column l&15).  Per tile row k:
//
//   diag    the owner of tile (k,k) eliminates its 16 pivots TWO at a time; every pair is ONE matrix instruction on the
//           tile (rank-2 update: the instruction's operand broadcast replaces the LDS / readlane exchange a VALU
//           update needs) and one on a copy of the identity, which collects G = L_kk^-1.
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

#include <cstdlib>
#include <type_traits>

namespace ldltm {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
// explicit LDS pointers for volatile reads (a volatile generic pointer becomes flat_load); volatile because hipcc
// otherwise sinks a look-ahead read into the place that uses it and waits for it there
typedef __attribute__((address_space(3))) const volatile d2* lds_vd2p;
typedef __attribute__((address_space(3))) const volatile double* lds_vdp;

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
  if (wlds) d += (size_t)g.N * (g.N - 1) / 2 + 256;
  return d;
}
__host__ inline size_t wglob_doubles(const Geo& g) { return (size_t)g.N * (g.N - 1) / 2 + 512; }

// Input layout shared by the kernels below ("tile image"): the upper triangle's 16x16 tiles, tile (i, j), i <= j, at
// t = j(j+1)/2 + i, 256 doubles each, lane-major: element (row 16i + (l>>4) + 4g, column 16j + (l&15)) at
// t*256 + 4*l + g -- a lane's four accumulator values are 32 contiguous bytes (two 16-byte loads, fully coalesced:
// one CU pulls 8-byte strided loads of a row-major matrix at ~11 GB/s, 16-byte ones at ~100 GB/s).  Diagonal tiles
// hold both triangles.  Entries outside the n x n matrix are never read (the kernels synthesise padding and border).
__host__ __device__ inline size_t tile_image_doubles(int n) { const Geo g = make_geo(n); return (size_t)g.ntiles * 256; }
__host__ __device__ inline int tile_index(int i, int j) { return j * (j + 1) / 2 + i; }
// position of matrix element (r, c) in the tile image, or -1 if it lies in a lower-triangle tile
__host__ __device__ inline int tile_image_pos(int r, int c) {
  const int i = r >> 4, j = c >> 4;
  if (i > j) return -1;
  const int l = ((r & 15) & 3) * 16 + (c & 15), g = (r & 15) >> 2;
  return tile_index(i, j) * 256 + 4 * l + g;
}

// The image holds the BORDERED matrix the kernels factor, N = 16 T rows and columns:  S in rows/columns < n, ones on the
// diagonal of the padding rows n .. n_pad-1, the right-hand side in column cb = n_pad (rows < n; mirrored into row cb
// inside the last diagonal tile), zeros everywhere else.  The kernels load tiles and nothing else (synthesising padding
// and border per element cost every wavefront ~1100 vector instructions before its first pivot).  Producers write S and,
// per row, image_put_rhs(); everything that does not depend on the values is written once per geometry by k_image_pad.
__host__ __device__ inline void image_put_rhs(double* St, int n, int r, double v) {
  const Geo g = make_geo(n);
  St[tile_image_pos(r, g.cb)] = v;
  const int m = tile_image_pos(g.cb, r);
  if (m >= 0) St[m] = v;
}
// value of image element (r, c) outside S and the right-hand side, or -1.0 if (r, c) is not such an element
__host__ __device__ inline double image_pad_value(const Geo& g, int n, int r, int c) {
  if (r < n && c < n) return -1.0;
  if ((c == g.cb && r < n) || (r == g.cb && c < n)) return -1.0;
  return (r == c && r < g.n_pad) ? 1.0 : 0.0;
}
// (thread `me` of `nthreads`; also called from lba.hip's k_prep)
__device__ __forceinline__ void image_pad_range(int n, double* __restrict__ St, int me, int nthreads) {
  const Geo g = make_geo(n);
  const int j0 = n >> 4;                         // tile columns j0 .. T-1 hold everything outside S
  const int t0 = tile_index(0, j0), cnt = (g.ntiles - t0) * 256;
  for (int e = me; e < cnt; e += nthreads) {
    const int t = t0 + (e >> 8), w = e & 255, l = w >> 2, gg = w & 3;
    int j = j0;
    while (tile_index(0, j + 1) <= t) j++;
    const int i = t - tile_index(0, j);
    const double v = image_pad_value(g, n, 16 * i + (l >> 4) + 4 * gg, 16 * j + (l & 15));
    if (v >= 0.0) St[(size_t)t0 * 256 + e] = v;
  }
}
__global__ void k_image_pad(int n, double* __restrict__ St) {
  image_pad_range(n, St, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x));
}
__host__ inline hipError_t launch_image_pad(int n, double* St, hipStream_t st) {
  hipLaunchKernelGGL(k_image_pad, dim3(8), dim3(256), 0, st, n, St);
  return hipGetLastError();
}

__device__ __forceinline__ double rdlane(double v, int l) {   // l must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// lanes of the odd 16-lane groups receive the value held by the even group below them (gfx950 v_permlane16_swap:
// swaps the odd rows of its first operand with the even rows of its second; both operands are copies of v here)
__device__ __forceinline__ double row_even_to_odd(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]);
}

__device__ __forceinline__ double rcp2(double d) {            // 1/d to ~1 ulp: hardware seed + two Newton steps
  double x = __builtin_amdgcn_rcp(d);
  x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
  x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
  return x;
}

__device__ __forceinline__ double rcp1(double d) {            // hardware seed (4.5e-8) + one Newton step: 2e-15 relative
  const double x = __builtin_amdgcn_rcp(d);                   // (tools/micro/fp64_latency.hip); 0 -> NaN, NaN -> NaN
  return __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
}

#ifndef LDLTM_SLEEP
#define LDLTM_SLEEP 1
#endif
#ifdef LDLTM_PROFILE
__device__ long long g_prof[2048];
#define LDLTM_T(slot) do { if (lane == 0) g_prof[slot] = clock64(); } while (0)
#else
#define LDLTM_T(slot) do { } while (0)
#endif

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

template <int NS, int NY, bool WLDS>
__global__ __launch_bounds__(kThreads) void k_ldlt_mfma(int n, const double* __restrict__ St, double* __restrict__ x, int* __restrict__ ok_flag,
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
    const int o = J * (J - 1) / 2 + I;
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
  }
  // two 16-byte loads per lane and tile from the tile image; wave-uniform conditions only (hipcc waits for a
  // lane-predicated load before it issues the next one)
  {
    d4 sv[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
      // no branch around the loads (hipcc drains the memory counter at every join): slots past the end re-read the last tile
      const int t = min(s * kWaves + wv, G.ntiles - 1);
      sv[s] = *reinterpret_cast<const d4*>(St + (size_t)t * 256 + 4 * lane);
    }
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const bool on = tj[s] < T;
#pragma unroll
      for (int g = 0; g < 4; g++) acc[s][g] = on ? sv[s][g] : 0.0;
    }
  }
  __syncthreads();
  if (wv == 0) LDLTM_T(1);

  auto wait_gt = [&](int* w, int k) {
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) <= k) __builtin_amdgcn_s_sleep(LDLTM_SLEEP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  auto wait_free = [&](int k) {   // the LDS buffers of parity k&1 were last used by tile row k-2
    if (k >= 2) wait_gt(&s_rowdone[k - 2], kWaves - 1);
  };

  // ---- the 16 (or fewer, last row) pivots of diagonal tile k, TWO per matrix instruction (see the diagonal loop of
  // k_ldlt_cols below for the derivation): rows p0, p1 of register g sit in lane groups q0, q0+1; with r0 = 1/c00,
  // l10 = c01 r0, 1/d1 = c00 / det the rank-2 update is C -= A B with A = [-r0 row0 | -(1/d1) row1'], B = [row0; row1'],
  // row1' = row1 - l10 row0.  The same two pivots are replayed on a copy of the identity (-> G = L^-1, needed by the
  // panel owners): its second row is reduced by its first with the same l10, then one instruction with the same A.
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
    const int npiv = min(16, n_pad - 16 * k);          // a multiple of 4
    double rlast = 1.0;
#ifdef LDLTM_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
#pragma unroll
    for (int g = 0; g < 4; g++) {
      if (g == 2) LDLTM_T(8 + 8 * k + 6);
      if (4 * g < npiv) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int q0 = 2 * h, p0 = 4 * g + q0, p1 = p0 + 1;
          double u = C[g];
          asm volatile("" : "+v"(u));              // own registers: the instruction below then updates C in place
          const double c00 = rdlane(u, q0 * 16 + p0), c01 = rdlane(u, q0 * 16 + p1), c11 = rdlane(u, (q0 + 1) * 16 + p1);
          const double det = __builtin_fma(c00, c11, -(c01 * c01));
          const double r0 = rcp1(c00);
          const double rdet = rcp1(det);
          const double r1 = c00 * rdet;
          const double nl10 = -(c01 * r0);
          const double u0b = row_even_to_odd(u);
          const double u1 = __builtin_fma(nl10, u0b, u);          // row1' in the lanes of group q0 + 1
          const bool in0 = lr == q0, in1 = lr == q0 + 1;
          const double bv = in1 ? u1 : u;
          const double av = in0 ? u * -r0 : in1 ? u1 * -r1 : 0.0;
          if (p1 < 15) C = mfma(av, bv, C);        // after the 16th pivot nothing of the tile is read again
          // the identity copy: rows p0, p1 of L^-1 are final before their own pivots
          double eg = E[g];
          asm volatile("" : "+v"(eg));
          const double e0b = row_even_to_odd(eg);
          const double erow = in1 ? __builtin_fma(nl10, e0b, eg) : eg;
          Gc[g] = (in0 || in1) ? erow : Gc[g];
          if (p1 < 15) E = mfma(av, erow, E);
          Wc[g] -= av;                             // rows p0, p1 of the unit upper factor (their two lane groups)
          dvv = lane == p0 ? r0 : lane == p1 ? r1 : dvv;
          if (h == 1) rlast = r0 + r1;             // a zero or non-finite pivot turns every later reciprocal into NaN
        }
      }
    }
    const bool good = fabs(rlast) < INFINITY;
#ifdef LDLTM_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    LDLTM_T(8 + 8 * k + 1);
    wait_free(k);
    const int par = k & 1;
#pragma unroll
    for (int g = 0; g < 4; g++) Gb[par * 16 * kGld + lc * kGld + lr + 4 * g] = Gc[g];
    if (lane < 16) Dv[par * 16 + lane] = dvv;
    if (!good) s_ok = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(&s_diag, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    LDLTM_T(8 + 8 * k + 2);
#pragma unroll
    for (int g = 0; g < 4; g++) {              // the factor's rows are not needed before the back-substitution
      const int I = 16 * k + lr + 4 * g, J = 16 * k + lc;
      if (I < J && I < n_pad && J <= cb) wm_store(I, J, Wc[g]);
    }
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
        double* const pb = Pan + ((par * T + j) * 2) * 256 + lane;
        double w4[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const double rr = R0[g] + R1[g];
          w4[g] = rr * dv4[g];
          pb[g * 64] = -rr;
          pb[256 + g * 64] = w4[g];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&s_panel[j], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (j == k + 1) LDLTM_T(8 + 8 * k + 4);
        const int J = 16 * j + lc;
        if (J <= cb) {
#pragma unroll
          for (int g = 0; g < 4; g++) wm_store(16 * k + lr + 4 * g, J, w4[g]);
        }
      }
    }
    // ---- trailing update with row k.  Order: the next diagonal tile (then its pivots at once), the other tiles of
    // row k+1 (the next panel), the rest.
    auto trail = [&](int pass) {
#pragma unroll
      for (int s = 0; s < NS; s++) {
        const int i = ti[s], j = tj[s];
        const int cls = (i == k + 1) ? (j == k + 1 ? 0 : 1) : 2;
        if (i > k && i < (1 << 20) && cls == pass) {
          wait_gt(&s_panel[i], k);
          if (j != i) wait_gt(&s_panel[j], k);
          const double* const pa = Pan + ((par * T + i) * 2) * 256 + lane;
          const double* const pw = Pan + ((par * T + j) * 2 + 1) * 256 + lane;
          double a[4], w[4];
#pragma unroll
          for (int q = 0; q < 4; q++) { a[q] = pa[q * 64]; w[q] = pw[q * 64]; }
          d4 c = acc[s];
          if (pass == 0) {                       // two chains of two: this tile is on the critical path
            d4 t2 = {0.0, 0.0, 0.0, 0.0};
            c = mfma(a[0], w[0], c);
            t2 = mfma(a[2], w[2], t2);
            c = mfma(a[1], w[1], c);
            t2 = mfma(a[3], w[3], t2);
            c += t2;
          } else {
#pragma unroll
            for (int q = 0; q < 4; q++) c = mfma(a[q], w[q], c);
          }
          acc[s] = c;
          if (pass == 0) LDLTM_T(8 + 8 * k + 5);
        }
      }
    };
    const bool next_diag = k + 1 < G.Tp && owns_diag(k + 1);
    if (next_diag) { trail(0); factor(k + 1); }
    trail(1);
    if (!next_diag) trail(0);                    // a diagonal tile that holds no pivots (border only)
    trail(2);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(&s_rowdone[k], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if constexpr (!WLDS) __threadfence();
  if (wv == 0) LDLTM_T(2);
  __syncthreads();
  if (wv == 0) LDLTM_T(3);
  const int ok = s_ok;
  // ---- back-substitution  L^T x = y  on one wavefront: x in registers, one column per step (v_readlane broadcast);
  // the factor's columns are read four at a time, one group ahead, with unconditional loads (entries on or below the
  // diagonal are masked after the load)
  if (wv == 0 && ok) {
    auto wm_load = [&](int I, int J) -> double {
      const int o = J * (J - 1) / 2 + I;
      double v;
      if constexpr (WLDS) v = Wl[o]; else v = __builtin_nontemporal_load(wglob + o);
      return I < J ? v : 0.0;
    };
    double y[NY];
#pragma unroll
    for (int r = 0; r < NY; r++) { const int I = r * 64 + lane; y[r] = I < n_pad ? wm_load(I, cb) : 0.0; }
#pragma unroll
    for (int rg = NY - 1; rg >= 0; rg--) {
      const int lo = rg * 64, hi = min(n_pad, lo + 64);
      if (hi > lo) {
        double cur[4][NY], nxt[4][NY];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
          for (int r2 = 0; r2 < NY; r2++) { cur[c][r2] = r2 <= rg ? wm_load(r2 * 64 + lane, hi - 4 + c) : 0.0; nxt[c][r2] = 0.0; }
        for (int J0 = hi - 4; J0 >= lo; J0 -= 4) {
          const int Jn = max(J0 - 4, lo);        // the last group re-reads itself (harmless) instead of branching
#pragma unroll
          for (int c = 0; c < 4; c++)
#pragma unroll
            for (int r2 = 0; r2 < NY; r2++)
              if (r2 <= rg) nxt[c][r2] = wm_load(r2 * 64 + lane, Jn + c);
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
    }
#pragma unroll
    for (int r = 0; r < NY; r++) { const int I = r * 64 + lane; if (I < n) x[I] = y[r]; }
  }
  if (wv == 0) LDLTM_T(4);
  if (tid == 0) *ok_flag = ok;
}

// ---------------------------------------------------------------------------------------------------------------
// Variant for 10 .. 19 tile rows (windows of 25 .. 50 free poses, the C4 workload): FOUR wavefronts, one per SIMD, up to
// 48 tiles each.  With eight wavefronts a SIMD's 512 registers are split in two: 24 tiles fill 192 of a wavefront's 256
// and hipcc parks tiles in scratch memory; with one wavefront per SIMD the same 384 tile registers leave 128 for
// everything else, the eliminating wavefront has the FP64 pipe of its SIMD to itself (a neighbour's matrix instruction
// between two dependent FP64 instructions holds the pipe for 64 cycles: 860 instead of 400 cycles per pair of pivots),
// and the latency a second wavefront would hide is hidden by updating four tiles at a time instead.
//
// Tiles are dealt in ROW-major order (t = rowstart(i) + j - i, owner t % 4, slot t / 4): the tiles that still change after
// tile row k are a suffix of every wavefront's slots, so the bulk of a trailing update is a run over consecutive slots with
// no per-tile test, and a wavefront's tiles of one row (a panel) are consecutive slots too.  Register arrays cannot be
// indexed at run time: the slot is a compile-time constant inside the cases of a switch (slot_switch).
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// (the empty statements differ from case to case: hipcc otherwise merges the cases' moves into one access with a computed
// index, and an array that is indexed at run time lives in scratch memory)
#define LDLTM_CASE(c) case c: if constexpr (c < N) { asm volatile("" :: "n"(c)); f(std::integral_constant<int, c>{}); asm volatile("" :: "n"(c)); } break;
template <int N, class F>
__device__ __forceinline__ void slot_switch(int s, F&& f) {     // f(integral_constant<int, s>), s < N <= 48
  switch (s) {
    LDLTM_CASE(0) LDLTM_CASE(1) LDLTM_CASE(2) LDLTM_CASE(3) LDLTM_CASE(4) LDLTM_CASE(5) LDLTM_CASE(6) LDLTM_CASE(7)
    LDLTM_CASE(8) LDLTM_CASE(9) LDLTM_CASE(10) LDLTM_CASE(11) LDLTM_CASE(12) LDLTM_CASE(13) LDLTM_CASE(14) LDLTM_CASE(15)
    LDLTM_CASE(16) LDLTM_CASE(17) LDLTM_CASE(18) LDLTM_CASE(19) LDLTM_CASE(20) LDLTM_CASE(21) LDLTM_CASE(22) LDLTM_CASE(23)
    LDLTM_CASE(24) LDLTM_CASE(25) LDLTM_CASE(26) LDLTM_CASE(27) LDLTM_CASE(28) LDLTM_CASE(29) LDLTM_CASE(30) LDLTM_CASE(31)
    LDLTM_CASE(32) LDLTM_CASE(33) LDLTM_CASE(34) LDLTM_CASE(35) LDLTM_CASE(36) LDLTM_CASE(37) LDLTM_CASE(38) LDLTM_CASE(39)
    LDLTM_CASE(40) LDLTM_CASE(41) LDLTM_CASE(42) LDLTM_CASE(43) LDLTM_CASE(44) LDLTM_CASE(45) LDLTM_CASE(46) LDLTM_CASE(47)
    default: break;
  }
}
#undef LDLTM_CASE

// Tile store of k_ldlt_big: tile S < 32 lives in the accumulation registers a[8S : 8S+7], tile S >= 32 in the vector registers
// v[128 + 8(S-32) : ...], all addressed from inline assembly; matrix instructions update a tile in place, 8 moves copy it
// out.  hipcc never sees these registers as values (a tile array of its own ends up as hundreds of moves at every join
// of the row loop, or in scratch memory).  To keep its own values out of them: the kernel with 48 tiles is limited to 128
// registers of each kind (amdgpu_num_vgpr: v128.. and a128.. are then reserved), and every assembly statement names the
// accumulation registers hipcc may still use as clobbered -- accumulation and vector registers are one allocatable class on
// this part, and hipcc does park long-lived values there.
#define LDLTM_AGPR_LO \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", \
  "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", \
  "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", \
  "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", \
  "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", \
  "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", \
  "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", \
  "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", \
  "a125", "a126", "a127"
#define LDLTM_AGPR_HI \
  "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", \
  "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", \
  "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", \
  "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", \
  "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", \
  "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", \
  "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", \
  "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", \
  "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", \
  "a254", "a255"
typedef int i8v __attribute__((ext_vector_type(8)));
#define LDLTM_ASM(V, text, outs, ins)                                   \
  do {                                                                  \
    if constexpr (V) { asm volatile(text : outs : ins : LDLTM_AGPR_LO); } \
    else { asm volatile(text : outs : ins : LDLTM_AGPR_LO, LDLTM_AGPR_HI); } \
  } while (0)
#define LDLTM_COMMA ,
constexpr int tile_reg(int S) { return S < 32 ? 8 * S : 128 + 8 * (S - 32); }
template <int S, bool V>
__device__ __forceinline__ d4 tile_get_r() {
  int x0, x1, x2, x3, x4, x5, x6, x7;
  constexpr int R = tile_reg(S);
  // (18 wait states between a matrix instruction's write and a read of its result; assembly is opaque to the hazard pass)
  if constexpr (S < 32) {
    LDLTM_ASM(V, "s_nop 15\n\ts_nop 7\n\t"
                 "v_accvgpr_read_b32 %0, a[%8]\n\tv_accvgpr_read_b32 %1, a[%9]\n\tv_accvgpr_read_b32 %2, a[%10]\n\tv_accvgpr_read_b32 %3, a[%11]\n\t"
                 "v_accvgpr_read_b32 %4, a[%12]\n\tv_accvgpr_read_b32 %5, a[%13]\n\tv_accvgpr_read_b32 %6, a[%14]\n\tv_accvgpr_read_b32 %7, a[%15]",
              "=v"(x0) LDLTM_COMMA "=v"(x1) LDLTM_COMMA "=v"(x2) LDLTM_COMMA "=v"(x3) LDLTM_COMMA "=v"(x4) LDLTM_COMMA "=v"(x5) LDLTM_COMMA "=v"(x6) LDLTM_COMMA "=v"(x7),
              "n"(R) LDLTM_COMMA "n"(R + 1) LDLTM_COMMA "n"(R + 2) LDLTM_COMMA "n"(R + 3) LDLTM_COMMA "n"(R + 4) LDLTM_COMMA "n"(R + 5) LDLTM_COMMA "n"(R + 6) LDLTM_COMMA "n"(R + 7));
  } else {
    LDLTM_ASM(V, "s_nop 15\n\ts_nop 7\n\t"
                 "v_mov_b32 %0, v[%8]\n\tv_mov_b32 %1, v[%9]\n\tv_mov_b32 %2, v[%10]\n\tv_mov_b32 %3, v[%11]\n\t"
                 "v_mov_b32 %4, v[%12]\n\tv_mov_b32 %5, v[%13]\n\tv_mov_b32 %6, v[%14]\n\tv_mov_b32 %7, v[%15]",
              "=v"(x0) LDLTM_COMMA "=v"(x1) LDLTM_COMMA "=v"(x2) LDLTM_COMMA "=v"(x3) LDLTM_COMMA "=v"(x4) LDLTM_COMMA "=v"(x5) LDLTM_COMMA "=v"(x6) LDLTM_COMMA "=v"(x7),
              "n"(R) LDLTM_COMMA "n"(R + 1) LDLTM_COMMA "n"(R + 2) LDLTM_COMMA "n"(R + 3) LDLTM_COMMA "n"(R + 4) LDLTM_COMMA "n"(R + 5) LDLTM_COMMA "n"(R + 6) LDLTM_COMMA "n"(R + 7));
  }
  const i8v v = {x0, x1, x2, x3, x4, x5, x6, x7};
  return __builtin_bit_cast(d4, v);
}
template <int S, bool V>
__device__ __forceinline__ void tile_put_r(d4 c) {
  const i8v v = __builtin_bit_cast(i8v, c);
  constexpr int R = tile_reg(S);
  if constexpr (S < 32) {
    LDLTM_ASM(V, "v_accvgpr_write_b32 a[%8], %0\n\tv_accvgpr_write_b32 a[%9], %1\n\tv_accvgpr_write_b32 a[%10], %2\n\tv_accvgpr_write_b32 a[%11], %3\n\t"
                 "v_accvgpr_write_b32 a[%12], %4\n\tv_accvgpr_write_b32 a[%13], %5\n\tv_accvgpr_write_b32 a[%14], %6\n\tv_accvgpr_write_b32 a[%15], %7\n\ts_nop 3", ,
              "v"(v[0]) LDLTM_COMMA "v"(v[1]) LDLTM_COMMA "v"(v[2]) LDLTM_COMMA "v"(v[3]) LDLTM_COMMA "v"(v[4]) LDLTM_COMMA "v"(v[5]) LDLTM_COMMA "v"(v[6]) LDLTM_COMMA "v"(v[7]) LDLTM_COMMA
              "n"(R) LDLTM_COMMA "n"(R + 1) LDLTM_COMMA "n"(R + 2) LDLTM_COMMA "n"(R + 3) LDLTM_COMMA "n"(R + 4) LDLTM_COMMA "n"(R + 5) LDLTM_COMMA "n"(R + 6) LDLTM_COMMA "n"(R + 7));
  } else {
    LDLTM_ASM(V, "v_mov_b32 v[%8], %0\n\tv_mov_b32 v[%9], %1\n\tv_mov_b32 v[%10], %2\n\tv_mov_b32 v[%11], %3\n\t"
                 "v_mov_b32 v[%12], %4\n\tv_mov_b32 v[%13], %5\n\tv_mov_b32 v[%14], %6\n\tv_mov_b32 v[%15], %7\n\ts_nop 3", ,
              "v"(v[0]) LDLTM_COMMA "v"(v[1]) LDLTM_COMMA "v"(v[2]) LDLTM_COMMA "v"(v[3]) LDLTM_COMMA "v"(v[4]) LDLTM_COMMA "v"(v[5]) LDLTM_COMMA "v"(v[6]) LDLTM_COMMA "v"(v[7]) LDLTM_COMMA
              "n"(R) LDLTM_COMMA "n"(R + 1) LDLTM_COMMA "n"(R + 2) LDLTM_COMMA "n"(R + 3) LDLTM_COMMA "n"(R + 4) LDLTM_COMMA "n"(R + 5) LDLTM_COMMA "n"(R + 6) LDLTM_COMMA "n"(R + 7));
  }
}
// tile S -= A^T W in place: four dependent matrix instructions
template <int S, bool V>
__device__ __forceinline__ void tile_mfma4(const double (&a)[4], const double (&w)[4]) {
  constexpr int R = tile_reg(S);
#define LDLTM_M4(F)                                                                                                                     \
  LDLTM_ASM(V, "v_mfma_f64_16x16x4_f64 " F "[%8:%9], %0, %1, " F "[%8:%9]\n\tv_mfma_f64_16x16x4_f64 " F "[%8:%9], %2, %3, " F "[%8:%9]\n\t"   \
               "v_mfma_f64_16x16x4_f64 " F "[%8:%9], %4, %5, " F "[%8:%9]\n\tv_mfma_f64_16x16x4_f64 " F "[%8:%9], %6, %7, " F "[%8:%9]", ,       \
            "v"(a[0]) LDLTM_COMMA "v"(w[0]) LDLTM_COMMA "v"(a[1]) LDLTM_COMMA "v"(w[1]) LDLTM_COMMA "v"(a[2]) LDLTM_COMMA "v"(w[2]) LDLTM_COMMA \
            "v"(a[3]) LDLTM_COMMA "v"(w[3]) LDLTM_COMMA "n"(R) LDLTM_COMMA "n"(R + 7))
  if constexpr (S < 32) { LDLTM_M4("a"); } else { LDLTM_M4("v"); }
#undef LDLTM_M4
}
// tiles S and S+1 (S even), the two chains interleaved (an independent matrix instruction issues after 66 cycles, a dependent
// one after 82)
template <int S, bool V>
__device__ __forceinline__ void tile_mfma4x2(const double (&a0)[4], const double (&w0)[4], const double (&a1)[4], const double (&w1)[4]) {
  constexpr int R = tile_reg(S);
#define LDLTM_M8(F)                                                                                                                     \
  LDLTM_ASM(V, "v_mfma_f64_16x16x4_f64 " F "[%16:%17], %0, %1, " F "[%16:%17]\n\tv_mfma_f64_16x16x4_f64 " F "[%18:%19], %8, %9, " F "[%18:%19]\n\t"     \
               "v_mfma_f64_16x16x4_f64 " F "[%16:%17], %2, %3, " F "[%16:%17]\n\tv_mfma_f64_16x16x4_f64 " F "[%18:%19], %10, %11, " F "[%18:%19]\n\t"   \
               "v_mfma_f64_16x16x4_f64 " F "[%16:%17], %4, %5, " F "[%16:%17]\n\tv_mfma_f64_16x16x4_f64 " F "[%18:%19], %12, %13, " F "[%18:%19]\n\t"   \
               "v_mfma_f64_16x16x4_f64 " F "[%16:%17], %6, %7, " F "[%16:%17]\n\tv_mfma_f64_16x16x4_f64 " F "[%18:%19], %14, %15, " F "[%18:%19]", ,       \
            "v"(a0[0]) LDLTM_COMMA "v"(w0[0]) LDLTM_COMMA "v"(a0[1]) LDLTM_COMMA "v"(w0[1]) LDLTM_COMMA "v"(a0[2]) LDLTM_COMMA "v"(w0[2]) LDLTM_COMMA \
            "v"(a0[3]) LDLTM_COMMA "v"(w0[3]) LDLTM_COMMA "v"(a1[0]) LDLTM_COMMA "v"(w1[0]) LDLTM_COMMA "v"(a1[1]) LDLTM_COMMA "v"(w1[1]) LDLTM_COMMA \
            "v"(a1[2]) LDLTM_COMMA "v"(w1[2]) LDLTM_COMMA "v"(a1[3]) LDLTM_COMMA "v"(w1[3]) LDLTM_COMMA                                       \
            "n"(R) LDLTM_COMMA "n"(R + 7) LDLTM_COMMA "n"(R + 8) LDLTM_COMMA "n"(R + 15))
  if constexpr (S < 32) { LDLTM_M8("a"); } else { LDLTM_M8("v"); }
#undef LDLTM_M8
}

// Matrix instruction on compiler-allocated vector registers, for k_ldlt_big (the builtin lets hipcc place the result in an
// accumulation register of its choice).  Assembly is opaque to the hazard pass: two wait states cover a vector-ALU write
// of an operand just before, 18 a vector-ALU read of the result right after.
// (A wait state is one issue slot of the wavefront -- four cycles: the 18 after every instruction cost a panel tile 300 of its
// 570 cycles of matrix work and the elimination 150 of 660 per pivot pair.  WAIT = false where the next reader is another
// matrix instruction on the same registers, which the hardware interlocks, or provably far away; mfma_wait() before a
// vector-ALU read otherwise.)
template <bool V, bool WAIT = true>
__device__ __forceinline__ d4 mfma_v(double a, double b, d4 c) {
  if constexpr (WAIT) {
    LDLTM_ASM(V, "s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 2", "+v"(c), "v"(a) LDLTM_COMMA "v"(b));
  } else {
    LDLTM_ASM(V, "s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0", "+v"(c), "v"(a) LDLTM_COMMA "v"(b));
  }
  return c;
}
// 18 wait states for the results in c0 (and c1); the operands tie the statement to the instructions that produce them
template <bool V>
__device__ __forceinline__ void mfma_wait(d4& c0, d4& c1) {
  LDLTM_ASM(V, "s_nop 15\n\ts_nop 2", "+v"(c0) LDLTM_COMMA "+v"(c1), );
}

#include "ldlt_jump_tables.inc"
// run-time slot -> the tile's registers: a computed jump into a table of equally sized cases (tools/gen/gen_ldlt_jump_tables.py);
// hipcc lowers a switch over the slot to a chain of up to 48 compare-and-branch blocks, 500-900 cycles per dispatch
// (WAIT = false: without the 24 wait states in front -- for a tile whose last matrix instruction is known to be far away)
template <bool V, bool WAIT = true>
__device__ __forceinline__ d4 tile_get_jt(int sl) {
  int x0, x1, x2, x3, x4, x5, x6, x7, tmp;
  if constexpr (V && WAIT) {
    asm volatile(LDLTM_JT_GET_48 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7), "=&s"(tmp) : "s"(sl)
                 : "vcc", "scc", LDLTM_AGPR_LO);
  } else if constexpr (V) {
    asm volatile(LDLTM_JT_GETNW_48 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7), "=&s"(tmp) : "s"(sl)
                 : "vcc", "scc", LDLTM_AGPR_LO);
  } else if constexpr (WAIT) {
    asm volatile(LDLTM_JT_GET_32 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7), "=&s"(tmp) : "s"(sl)
                 : "vcc", "scc", LDLTM_AGPR_LO, LDLTM_AGPR_HI);
  } else {
    asm volatile(LDLTM_JT_GETNW_32 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7), "=&s"(tmp) : "s"(sl)
                 : "vcc", "scc", LDLTM_AGPR_LO, LDLTM_AGPR_HI);
  }
  const i8v v = {x0, x1, x2, x3, x4, x5, x6, x7};
  return __builtin_bit_cast(d4, v);
}
template <bool V>
__device__ __forceinline__ void tile_mfma4_jt(int sl, const double (&a)[4], const double (&w)[4]) {
  int tmp;
  if constexpr (V) {
    asm volatile(LDLTM_JT_MFMA4_48 : "=&s"(tmp) : "v"(a[0]), "v"(w[0]), "v"(a[1]), "v"(w[1]), "v"(a[2]), "v"(w[2]), "v"(a[3]), "v"(w[3]), "s"(sl)
                 : "vcc", "scc", LDLTM_AGPR_LO);
  } else {
    asm volatile(LDLTM_JT_MFMA4_32 : "=&s"(tmp) : "v"(a[0]), "v"(w[0]), "v"(a[1]), "v"(w[1]), "v"(a[2]), "v"(w[2]), "v"(a[3]), "v"(w[3]), "s"(sl)
                 : "vcc", "scc", LDLTM_AGPR_LO, LDLTM_AGPR_HI);
  }
}
template <bool V>
__device__ __forceinline__ void tile_mfma8_jt(int pair, const double (&a0)[4], const double (&w0)[4], const double (&a1)[4], const double (&w1)[4]) {
  int tmp;
  if constexpr (V) {
    asm volatile(LDLTM_JT_MFMA8_48 : "=&s"(tmp)
                 : "v"(a0[0]), "v"(w0[0]), "v"(a0[1]), "v"(w0[1]), "v"(a0[2]), "v"(w0[2]), "v"(a0[3]), "v"(w0[3]),
                   "v"(a1[0]), "v"(w1[0]), "v"(a1[1]), "v"(w1[1]), "v"(a1[2]), "v"(w1[2]), "v"(a1[3]), "v"(w1[3]), "s"(pair)
                 : "vcc", "scc", LDLTM_AGPR_LO);
  } else {
    asm volatile(LDLTM_JT_MFMA8_32 : "=&s"(tmp)
                 : "v"(a0[0]), "v"(w0[0]), "v"(a0[1]), "v"(w0[1]), "v"(a0[2]), "v"(w0[2]), "v"(a0[3]), "v"(w0[3]),
                   "v"(a1[0]), "v"(w1[0]), "v"(a1[1]), "v"(w1[1]), "v"(a1[2]), "v"(w1[2]), "v"(a1[3]), "v"(w1[3]), "s"(pair)
                 : "vcc", "scc", LDLTM_AGPR_LO, LDLTM_AGPR_HI);
  }
}

// The trailing update's steady state (ldlt_jump_tables.inc, BULK): `pairs` consecutive pairs of tiles starting at the EVEN pair
// `first`, everything else between the matrix instructions.  vtab: slot -> tile coordinates (one lane per slot); addr_r / addr_w:
// this lane's LDS byte addresses inside the -R / W operand images of column 0 of the row's parity.  The operand sets and the
// address registers are v60..v127 (named as clobbered: hipcc keeps its own values below).
#define LDLTM_VGPR_60_127 \
  "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
  "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", \
  "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", \
  "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
template <bool V>
__device__ __forceinline__ void tile_bulk_pairs(int first, int pairs, int vtab, unsigned addr_r, unsigned addr_w) {
  int t0, t1, t2, t3;
  if constexpr (V) {
    asm volatile(LDLTM_JT_BULK_48 : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "v"(vtab), "v"(addr_r), "v"(addr_w), "s"(first), "s"(pairs)
                 : "vcc", "scc", "memory", LDLTM_VGPR_60_127, LDLTM_AGPR_LO);
  } else {
    asm volatile(LDLTM_JT_BULK_32 : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "v"(vtab), "v"(addr_r), "v"(addr_w), "s"(first), "s"(pairs)
                 : "vcc", "scc", "memory", LDLTM_VGPR_60_127, LDLTM_AGPR_LO, LDLTM_AGPR_HI);
  }
}

constexpr int kBigWaves = 4;
constexpr int kBigThreads = 64 * kBigWaves;

template <int NS, int NY>
__device__ __forceinline__ void ldlt_big_body(int n, const double* __restrict__ St, double* __restrict__ x, int* __restrict__ ok_flag,
                                              double* __restrict__ wglob) {
#pragma clang fp contract(fast)        // the solve is tolerance-checked (1e-4), not bit-compared
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ int s_diag;               // tile rows whose G / D^-1 are published
  __shared__ int s_panel[kMaxT + 4];   // per tile column j: rows k for which -R_kj / W_kj are published
  __shared__ int s_pcount[kMaxT + 4];  // per tile row: panel tiles published
  __shared__ int s_rowdone[kMaxT + 4]; // per tile row: wavefronts that finished its trailing update
  __shared__ int s_ok;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const Geo G = make_geo(n);
  const int T = G.T, n_pad = G.n_pad, cb = G.cb;
  double* const Pan = sh;                               // [2][T][2][256]
  double* const Gb = Pan + (size_t)2 * T * 512;         // [2][16 * kGld]
  double* const Dv = Gb + 2 * 16 * kGld;                // [2][16]
  auto wm_store = [&](int I, int J, double v) { wglob[J * (J - 1) / 2 + I] = v; };   // entry (I, J), I < J, of the unit upper factor
  if (wv == 0) LDLTM_T(0);
  if (tid == 0) { s_diag = 0; s_ok = 1; }
  if (tid < kMaxT + 4) { s_panel[tid] = 0; s_rowdone[tid] = 0; s_pcount[tid] = 0; }

  auto rowstart = [&](int i) -> int { return i * T - i * (i - 1) / 2; };
  // slot of tile (i, j), i <= j, if this wavefront owns it, else -1
  auto my_slot = [&](int i, int j) -> int {
    const int t = rowstart(i) + j - i;
    return (t & (kBigWaves - 1)) == wv ? t >> 2 : -1;
  };
  // first slot of this wavefront whose tile lies in row i or below
  auto first_slot_from_row = [&](int i) -> int { return (rowstart(i) - wv + kBigWaves - 1) >> 2; };
  const int my_count = first_slot_from_row(T);          // rowstart(T) = number of tiles

  // The tiles.  A register array can only be indexed by constants, and code specialised per slot does not fit the
  // instruction cache (a wavefront then fetches every tile update from the L2: ~3000 cycles per tile measured), so the work
  // on a tile is generic code; per slot there is only a case of a switch with the tile's registers in it: four matrix
  // instructions (update in place) or eight moves (copy out for the panel / the elimination).
  constexpr bool V = NS > 32;                           // tiles in v128..v255 as well
  // (the register allocation of the kernel: hipcc counts what assembly statements name as clobbered, not what their text uses;
  // in the 48-tile kernel these registers are reserved -- which is the point -- and naming them draws a warning)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  if constexpr (V) { asm volatile("" ::: "a0", "a255", "v255"); } else { asm volatile("" ::: "a0", "a255"); }
#pragma clang diagnostic pop
  auto tile_get = [&](int sl) -> d4 { return tile_get_jt<V>(sl); };
  // U -= R^T W on the tile in slot sl, operands in a[], w[]
  auto tile_update = [&](int sl, const double (&a)[4], const double (&w)[4]) { tile_mfma4_jt<V>(sl, a, w); };
  // ... on the tiles in slots sl (even) and sl + 1, chains interleaved
  auto pair_update = [&](int sl, const double (&a0)[4], const double (&w0)[4], const double (&a1)[4], const double (&w1)[4]) {
    tile_mfma8_jt<V>(sl >> 1, a0, w0, a1, w1);
  };
  // two 16-byte loads per lane and tile from the tile image (column-major tile order there)
  auto tile_ij = [&](int sl, int& i, int& j) {       // the tile in slot sl of this wavefront
    const int t = min(sl * kBigWaves + wv, G.ntiles - 1);
    const float bq = (float)(2 * T + 1);
    i = (int)((bq - sqrtf(bq * bq - 8.0f * (float)t)) * 0.5f);
    i = max(0, min(T - 1, i));
    if (i + 1 < T && rowstart(i + 1) <= t) i++;
    if (rowstart(i) > t) i--;
    j = i + t - rowstart(i);
  };
  // lane s holds the coordinates of slot s (i | j << 8): the bulk loop gets a tile's operand addresses with one v_readlane
  // instead of stepping (i, j) through the triangle -- between two groups of matrix instructions every scalar instruction counts,
  // the wavefront issues them while the matrix pipe is idle -- and the loads below get their addresses without 75 instructions
  // of index arithmetic per tile (48 tiles: 18 k of the 21 k cycles the loading took)
  int vtab;
  { int ti, tj; tile_ij(lane, ti, tj); vtab = ti | (tj << 8); }
  auto tile_addr = [&](int sl) -> const d4* {
    const int ij = __builtin_amdgcn_readlane(vtab, sl);
    return reinterpret_cast<const d4*>(St + (size_t)tile_index(ij & 255, ij >> 8) * 256 + 4 * lane);
  };
  // rounds of LC tiles; rounds r+1 .. r+LD-1 are requested before round r is moved into the tile store (a round is one memory
  // round trip of ~1800 cycles: with one round ahead the 48-tile kernel spent 21 k cycles = 9 us loading)
  constexpr int LC = V ? 4 : 8, LD = V ? 3 : 2, NR = (NS + LC - 1) / LC;
  {
    d4 buf[LD][LC];
    static_for<0, LD - 1>([&](auto dd) {
      constexpr int d = decltype(dd)::value;
      static_for<0, LC>([&](auto uu) { constexpr int u = decltype(uu)::value; if constexpr (d * LC + u < NS) buf[d][u] = *tile_addr(d * LC + u); });
    });
    static_for<0, NR>([&](auto rr) {
      constexpr int r = decltype(rr)::value, c0 = LC * r, pb = r % LD, nb = (r + LD - 1) % LD, n0 = c0 + (LD - 1) * LC;
      if constexpr (r + LD - 1 < NR)
        static_for<0, LC>([&](auto uu) { constexpr int u = decltype(uu)::value; if constexpr (n0 + u < NS) buf[nb][u] = *tile_addr(n0 + u); });
      static_for<0, LC>([&](auto uu) { constexpr int u = decltype(uu)::value; if constexpr (c0 + u < NS) tile_put_r<c0 + u, V>(buf[pb][u]); });
    });
  }
  __syncthreads();
  if (wv == 0) LDLTM_T(1);

  // Signalling goes through LDS only, and the LDS unit executes a wavefront's accesses in program order: a flag written
  // after the data is seen after the data, a read issued after the flag read returns sees what the flag announces.  The
  // hand-overs need a compiler barrier, not a fence (a release fence would also wait for the factor's global stores).
  auto ld_flag = [&](int* w) -> int {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  };
  auto wait_gt = [&](int* w, int k) {
    while (ld_flag(w) <= k) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  };
  auto wait_free = [&](int k) {   // the LDS buffers of parity k&1 were last used by tile row k-2
    if (k >= 2) wait_gt(&s_rowdone[k - 2], kBigWaves - 1);
  };
  auto post = [&](int* w, int v) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto post_add = [&](int* w) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  // ---- the pivots of diagonal tile k, two per matrix instruction (k_ldlt_mfma::factor above; here the two rows of
  // G = L^-1 that a pair completes go to LDS at once)
  auto factor = [&](d4 C, int k) {
    LDLTM_T(8 + 8 * k + 0);
    // (no wait for the buffers of this parity: G and D^-1 of tile row k-2 were read by that row's panel tiles, and this wavefront
    // itself waited for ALL of them -- s_pcount[k-2] -- before it started row k-2's update, one loop iteration ago.  Waiting for
    // the row's UPDATE to finish everywhere, as the panel images need, put the slowest wavefront's bulk work of row k-2 on the
    // diagonal chain: 10-12 k cycles per late tile row instead of ~6.5 k.)
    const int par = k & 1;
    // (the lane-derived values of this function are recomputed from an opaque copy of the lane number: hoisted out of the row loop
    // they are live across the bulk update's assembly block, which leaves hipcc 60 vector registers, and were spilled to scratch
    // memory -- reloaded here, on the critical chain)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int lr = lane_o >> 4, lc = lane_o & 15;
    d4 E, Wc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < 4; g++) E[g] = (lr + 4 * g == lc) ? 1.0 : 0.0;
    double dvv = 1.0;
    const int npiv = min(16, n_pad - 16 * k);          // a multiple of 4
    double rlast = 1.0;
    double* const gcol = Gb + par * 16 * kGld + lc * kGld + lr;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      if (g == 2) LDLTM_T(8 + 8 * k + 6);
      if (4 * g < npiv) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int q0 = 2 * h, p0 = 4 * g + q0, p1 = p0 + 1;
          double u = C[g];
          // own registers: the instruction below then updates C in place.  (The previous pair's instruction on C was issued
          // before its instruction on E, which held the pipe for 64 cycles: with these 8 wait states more than the 18 required.)
          asm volatile("s_nop 7" : "+v"(u));
          const double c00 = rdlane(u, q0 * 16 + p0), c01 = rdlane(u, q0 * 16 + p1), c11 = rdlane(u, (q0 + 1) * 16 + p1);
          const double det = __builtin_fma(c00, c11, -(c01 * c01));
          const double r0 = rcp1(c00);
          const double rdet = rcp1(det);
          const double r1 = c00 * rdet;
          const double nl10 = -(c01 * r0);
          const double u0b = row_even_to_odd(u);
          const double u1 = __builtin_fma(nl10, u0b, u);          // row1' in the lanes of group q0 + 1
          const bool in0 = lr == q0, in1 = lr == q0 + 1;
          const double bv = in1 ? u1 : u;
          const double av = in0 ? u * -r0 : in1 ? u1 * -r1 : 0.0;
          if (p1 < 15) C = mfma_v<V, false>(av, bv, C);   // after the 16th pivot nothing of the tile is read again
          double eg = E[g];
          asm volatile("" : "+v"(eg));
          const double e0b = row_even_to_odd(eg);
          const double erow = in1 ? __builtin_fma(nl10, e0b, eg) : eg;
          if (in0 || in1) gcol[4 * g] = erow;      // rows p0, p1 of L^-1 are final before their own pivots
          if (p1 < 15) E = mfma_v<V, false>(av, erow, E);  // (read again after the next pair's chain of reciprocals: > 200 cycles)
          Wc[g] -= av;                             // rows p0, p1 of the unit upper factor (their two lane groups)
          dvv = lane == p0 ? r0 : lane == p1 ? r1 : dvv;
          if (h == 1) rlast = r0 + r1;             // a zero or non-finite pivot turns every later reciprocal into NaN
        }
      } else {
        gcol[4 * g] = E[g];                        // rows without pivots (last tile row): identity
      }
    }
    const bool good = fabs(rlast) < INFINITY;
    LDLTM_T(8 + 8 * k + 1);
    if (lane < 16) Dv[par * 16 + lane] = dvv;
    if (!good) s_ok = 0;
    post(&s_diag, k + 1);
    LDLTM_T(8 + 8 * k + 2);
#pragma unroll
    for (int g = 0; g < 4; g++) {              // the factor's rows are not needed before the back-substitution
      const int I = 16 * k + lr + 4 * g, J = 16 * k + lc;
      if (I < J && I < n_pad && J <= cb) wm_store(I, J, Wc[g]);
    }
  };

  // ---- panel tile (k, j), j > k: R = G X, W = D^-1 R; -R and W to LDS in operand layout
  auto panel_tile = [&](const d4 X, int k, int j, const double (&Gf)[4], const double (&dv4)[4]) {
    const int par = k & 1;
    if (j == k + 1) LDLTM_T(8 + 8 * k + 3);
    d4 R0 = {0.0, 0.0, 0.0, 0.0}, R1 = {0.0, 0.0, 0.0, 0.0};
    R0 = mfma_v<V, false>(Gf[0], X[0], R0);
    R1 = mfma_v<V, false>(Gf[2], X[2], R1);
    R0 = mfma_v<V, false>(Gf[1], X[1], R0);
    R1 = mfma_v<V, false>(Gf[3], X[3], R1);
    mfma_wait<V>(R0, R1);
    // operand images in LDS: a lane's four values as two 16-byte halves, [half][lane][2] -- two 128-bit accesses per operand
    // with a 16-byte lane stride (no bank conflicts) instead of four 64-bit ones
    d2* const pb = reinterpret_cast<d2*>(Pan + ((par * T + j) * 2) * 256) + lane;
    double w4[4], nr[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const double rr = R0[g] + R1[g];
      w4[g] = rr * dv4[g];
      nr[g] = -rr;
    }
    pb[0] = d2{nr[0], nr[1]}; pb[64] = d2{nr[2], nr[3]};
    pb[128] = d2{w4[0], w4[1]}; pb[192] = d2{w4[2], w4[3]};
    if (j == k + 1) post(&s_panel[j], k + 1);    // the one tile somebody waits for by itself (the next diagonal tile's update)
    if (j == k + 1) LDLTM_T(8 + 8 * k + 4);
    const int J = 16 * j + lc;
    if (J <= cb) {
#pragma unroll
      for (int g = 0; g < 4; g++) wm_store(16 * k + lr + 4 * g, J, w4[g]);
    }
  };
  // ---- U_ij -= R_ki^T W_kj (4 instructions, operands straight from LDS)
  auto update1 = [&](d4 c, int i, int j, int k, bool two_chains) -> d4 {
    const int par = k & 1;
    const d2* const pa = reinterpret_cast<const d2*>(Pan + ((par * T + i) * 2) * 256) + lane;
    const d2* const pw = reinterpret_cast<const d2*>(Pan + ((par * T + j) * 2 + 1) * 256) + lane;
    double a[4], w[4];
    { const d2 a01 = pa[0], a23 = pa[64], w01 = pw[0], w23 = pw[64];
      a[0] = a01[0]; a[1] = a01[1]; a[2] = a23[0]; a[3] = a23[1]; w[0] = w01[0]; w[1] = w01[1]; w[2] = w23[0]; w[3] = w23[1]; }
    d4 t2 = {0.0, 0.0, 0.0, 0.0};
    if (two_chains) {                        // two chains of two: this tile is on the critical path
      c = mfma_v<V, false>(a[0], w[0], c);
      t2 = mfma_v<V, false>(a[2], w[2], t2);
      c = mfma_v<V, false>(a[1], w[1], c);
      t2 = mfma_v<V, false>(a[3], w[3], t2);
      mfma_wait<V>(c, t2);
      c += t2;
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) c = mfma_v<V, false>(a[q], w[q], c);
      mfma_wait<V>(c, t2);
    }
    return c;
  };
  auto ld_ops = [&](int k, int i, int j, double (&av)[4], double (&wv4)[4]) {
    const int par = k & 1;
    // volatile LDS reads: they stay where they are written (the look-ahead of the bulk loop below)
    const lds_vd2p pa = (lds_vd2p)(Pan + ((par * T + i) * 2) * 256) + lane;
    const lds_vd2p pw = (lds_vd2p)(Pan + ((par * T + j) * 2 + 1) * 256) + lane;
    const d2 a01 = pa[0], a23 = pa[64], w01 = pw[0], w23 = pw[64];
    av[0] = a01[0]; av[1] = a01[1]; av[2] = a23[0]; av[3] = a23[1];
    wv4[0] = w01[0]; wv4[1] = w01[1]; wv4[2] = w23[0]; wv4[3] = w23[1];
  };
  auto ld_ops_slot = [&](int k, int slot, double (&av)[4], double (&wv4)[4]) {
    const int ij = __builtin_amdgcn_readlane(vtab, slot);
    ld_ops(k, ij & 255, ij >> 8, av, wv4);
  };
  // Row program.  For tile row k (k = -1: nothing to update yet):
  //   stage 0  the tile (k+1,k+1), if it is mine, gets row k's update and is eliminated at once;
  //   stage 1  my other tiles of row k+1 (the next panel) get row k's update;
  //   stage 2  my remaining live tiles, four consecutive slots per step where possible; between two steps the wavefront
  //            looks whether tile (k+1,k+1) has been published meanwhile and, if so, turns to its panel tiles of row k+1
  //            FIRST -- the chain diag -> panel -> update -> diag never waits for bulk work.
  for (int k = -1; k < G.Tp; k++) {
    LDLTM_T(512 + ((k + 1) * 4 + wv) * 8 + 0);
    if (k + 1 < T) {
      const int sd = my_slot(k + 1, k + 1);
      if (sd >= 0) {
        d4 C = tile_get(sd);
        if (k >= 0) {
          wait_gt(&s_panel[k + 1], k);
          C = update1(C, k + 1, k + 1, k, true);
          LDLTM_T(8 + 8 * k + 5);
        }
        if (k + 1 < G.Tp) factor(C, k + 1);
      }
    }
    LDLTM_T(512 + ((k + 1) * 4 + wv) * 8 + 1);
    // my live tiles from row k+1 on (consecutive slots; the eliminated diagonal tile's slot, if it is mine, is swept along: it is
    // dead).  The tiles of row k+1 -- the next panel -- come first in slot order and must have row k's update before this
    // wavefront computes its panel tiles of row k+1 (sl >= s2 below).
    int sl = 0, se = 0, s2 = 0;
    if (k >= 0) {
      wait_gt(&s_pcount[k], T - 2 - k);            // every panel tile of row k is published: no flag checks in the update
      sl = first_slot_from_row(k + 1); s2 = min(first_slot_from_row(k + 2), my_count); se = my_count;
    }
    LDLTM_T(512 + ((k + 1) * 4 + wv) * 8 + 3);
    const int seA = se;
    const unsigned lbase = (unsigned)(unsigned long)(lds_vdp)Pan + (((unsigned)(k & 1) * (unsigned)T) << 12) + 16u * (unsigned)lane;
    bool pdone = !(k + 1 < G.Tp);
    for (;;) {
      if (!pdone && sl >= s2) {
        const int kk = k + 1;
        if (ld_flag(&s_diag) > kk && (kk < 2 || ld_flag(&s_rowdone[kk - 2]) >= kBigWaves)) {
          asm volatile("" ::: "memory");
          LDLTM_T(512 + ((k + 1) * 4 + wv) * 8 + 5);
          // my panel tiles (kk, j) (consecutive slots, every fourth j)
          const int par = kk & 1;
          int ps = first_slot_from_row(kk);
          const int pe = min(first_slot_from_row(kk + 1), my_count);
          int pj = kk + (ps * kBigWaves + wv - rowstart(kk));
          if (ps < pe && pj == kk) { ps++; pj += kBigWaves; }
          if (ps < pe) {
            double Gf[4], dv4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
              Gf[q] = Gb[par * 16 * kGld + (4 * q + lr) * kGld + lc];
              dv4[q] = Dv[par * 16 + lr + 4 * q];
            }
            // (the first read of a tile waits out a matrix instruction that may just have written it; while that tile is worked
            // on the bulk update's last instructions drain, the later tiles are read without the wait states)
            const int np_mine = pe - ps;
            panel_tile(tile_get(ps), kk, pj, Gf, dv4);
            for (ps++, pj += kBigWaves; ps < pe; ps++, pj += kBigWaves) panel_tile(tile_get_jt<V, false>(ps), kk, pj, Gf, dv4);
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&s_pcount[kk], np_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          pdone = true;
          LDLTM_T(512 + ((k + 1) * 4 + wv) * 8 + 6);
        }
      }
      if (sl < seA) {
        if ((sl & 3) == 0 && sl + 2 <= seA) {
          // consecutive pairs as one block of straight-line code (tile_bulk_pairs, entered at an even pair); while the next
          // diagonal tile is still being eliminated the block is left every four pairs (~2000 cycles) to look for it
          int np = (seA - sl) >> 1;
          if (!pdone) np = min(np, 4);
          tile_bulk_pairs<V>(sl >> 1, np, vtab, lbase, lbase + 2048u);
          sl += 2 * np;
        } else if ((sl & 1) == 0 && sl + 2 <= seA) {
          double a0[4], w0[4], a1[4], w1[4];
          ld_ops_slot(k, sl, a0, w0);
          ld_ops_slot(k, sl + 1, a1, w1);
          pair_update(sl, a0, w0, a1, w1);
          sl += 2;
        } else {
          double a0[4], w0[4];
          ld_ops_slot(k, sl, a0, w0);
          tile_update(sl, a0, w0);
          sl++;
        }
      } else if (pdone) {
        break;
      } else {
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (k >= 0) post_add(&s_rowdone[k]);
    LDLTM_T(512 + ((k + 1) * 4 + wv) * 8 + 4);
  }
  __threadfence();
  if (wv == 0) LDLTM_T(2);
  __syncthreads();
  if (wv == 0) LDLTM_T(3);
  const int ok = s_ok;
  // ---- back-substitution  L^T x = y  on one wavefront: x in registers, one column per step (v_readlane broadcast);
  // the factor's columns are read four at a time, one group ahead, with unconditional loads (entries on or below the
  // diagonal are masked after the load)
  if (wv == 0 && ok) {
    // (raw loads: only the column's own row group holds entries on or below the diagonal, and they are masked where they are
    // USED -- masking them where they arrive, and copying the look-ahead registers, made every group wait for its own loads:
    // a memory round trip per four columns.  Two register sets take turns instead.)
    auto wm_raw = [&](int I, int J) -> double { return __builtin_nontemporal_load(wglob + (J * (J - 1) / 2 + I)); };
    double y[NY];
#pragma unroll
    for (int r = 0; r < NY; r++) { const int I = r * 64 + lane; y[r] = I < n_pad ? wm_raw(I, cb) : 0.0; }
    static_for<0, NY>([&](auto rr) {
      constexpr int rg = NY - 1 - decltype(rr)::value;
      const int lo = rg * 64, hi = min(n_pad, lo + 64);
      if (hi > lo) {
        double A[4][rg + 1], B[4][rg + 1];
        auto fetch = [&](double (&dst)[4][rg + 1], int J0) {
          const int Jc = max(J0, lo);            // past the row group's first column: a valid group again, never used
#pragma unroll
          for (int c = 0; c < 4; c++)
#pragma unroll
            for (int r2 = 0; r2 <= rg; r2++) dst[c][r2] = wm_raw(r2 * 64 + lane, Jc + c);
        };
        auto apply = [&](const double (&src)[4][rg + 1], int J0) {
#pragma unroll
          for (int c = 3; c >= 0; c--) {
            const double xJ = rdlane(y[rg], J0 + c - lo);
#pragma unroll
            for (int r2 = 0; r2 < rg; r2++) y[r2] -= src[c][r2] * xJ;
            y[rg] -= (lo + lane < J0 + c ? src[c][rg] : 0.0) * xJ;
          }
        };
        fetch(A, hi - 4);
        for (int J0 = hi - 4; J0 >= lo; J0 -= 8) {
          fetch(B, J0 - 4);
          apply(A, J0);
          if (J0 - 4 < lo) break;
          fetch(A, J0 - 8);
          apply(B, J0 - 4);
        }
      }
    });
#pragma unroll
    for (int r = 0; r < NY; r++) { const int I = r * 64 + lane; if (I < n) x[I] = y[r]; }
  }
  if (wv == 0) LDLTM_T(4);
  if (tid == 0) *ok_flag = ok;
}

template <int NS, int NY>
__global__ __launch_bounds__(kBigThreads) void k_ldlt_big(int n, const double* __restrict__ St, double* __restrict__ x, int* __restrict__ ok_flag,
                                                          double* __restrict__ wglob) {
  ldlt_big_body<NS, NY>(n, St, x, ok_flag, wglob);
}
// 48 tiles per wavefront: hipcc keeps to v0..v127
__global__ __launch_bounds__(kBigThreads) __attribute__((amdgpu_num_vgpr(128))) void k_ldlt_big48(int n, const double* __restrict__ St, double* __restrict__ x,
                                                                                                int* __restrict__ ok_flag, double* __restrict__ wglob) {
  ldlt_big_body<48, 5>(n, St, x, ok_flag, wglob);
}

// ---------------------------------------------------------------------------------------------------------------
// Column variant for up to 8 tile rows (windows of <= 20 free poses, the C2 workload): tile column j lives in
// wavefront j.  Measured on MI355X (tools/micro/): an FP64 matrix instruction and FP64 vector instructions of the same
// SIMD do NOT overlap (the 16x16x4 instruction holds the FP64 pipe for 64 cycles), every FP64 vector instruction costs a
// wavefront 8 cycles of issue, and a pair of pivots costs its owner ~340 cycles in isolation (readlanes, two reciprocals,
// the row reduction, one instruction) -- every further instruction on the diagonal wavefront is added serially.  Here
// the diagonal wavefront therefore does nothing but its own tile: it streams every pair (its A operand and the two
// reciprocals) through LDS and the owners of the panel tiles (k, j) REPLAY the pairs on their tile (one instruction per
// pair on another SIMD), collecting R and W row by row.  Wavefront k+1 holds both (k, k+1) and (k+1, k+1): it folds
// R^T W into its diagonal tile four rows at a time straight from registers, so the next diagonal tile is complete one
// instruction after the last replayed pair.  Only -W goes through LDS for the remaining trailing tiles (their B operand
// is the owner's own unscaled R, which stays in registers).
constexpr int kColT = 8;
constexpr int kPivRing = 4;                                   // tile rows of streamed pivots kept in LDS
constexpr int kColPanels = kColT * (kColT - 1) / 2;           // one R image per panel tile: never overwritten

// factor store of the column kernel: columns in pairs, (I, J) at  2p(p+1) + 2I + (J&1),  p = J>>1, rows I <= 2p+1;
// entries on / below the diagonal inside a pair are stored as zeros, so the back-substitution reads two columns with
// one 128-bit LDS load per lane and needs no per-entry mask
__host__ __device__ inline size_t lds_doubles_cols(const Geo& g) {
  const size_t pairs = (size_t)g.N / 2;
  return (size_t)kPivRing * 16 * 64 + (size_t)kPivRing * 16 + (size_t)kColPanels * 256 + 2 * pairs * (pairs + 1) + 512;
}


// Back-substitution of the column kernel with every pair index known at compile time (NP = n_pad / 2 pairs of columns):
// lane selects of v_readlane, LDS offsets and the bounds of the look-ahead ring become immediates, the address arithmetic and
// the loop control of the run-time form (about half of its ~35 instructions per pair) disappear.  Instantiated for the
// window size local BA runs at (20 free poses: n_pad = 120); other sizes take the loops in the kernel.
template <int NP>
__device__ __forceinline__ void backsub_pairs_unrolled(const lds_vd2p W2, double& y0, double& y1, const int lane) {
  // No masks on the factor's columns: a lane whose x is final is not protected from the entries past its pair's rows (they
  // belong to the next pair's storage: finite numbers), its x is simply kept aside the moment it becomes final (one select
  // per pair instead of two per column); nothing reads a final lane of y again.
  // Both unknowns of a pair come from ONE state of y:  x1 = y[2p+1],  x0 = y[2p] - w01 x1  with w01 = W(2p, 2p+1) read out of the
  // column itself ahead of time -- one lane read on the dependent chain per pair instead of two.
  // ONE ring of four pairs in flight over both halves (rows >= 64 live in y1 and need the second 128-bit load per pair).  The
  // scheduling barrier after every pair keeps the refill where it is written: hipcc otherwise moves two pairs of arithmetic in
  // front of it and the wait in front of a pair then covers a load issued ONE pair earlier (93 cycles per pair, an LDS round trip
  // of a lone wavefront is ~130).
  double x0f = y0, x1f = y1;
  d2 c0[4], c1[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int pi = NP - 1 - i > 0 ? NP - 1 - i : 0;
    c0[i] = W2[pi * (pi + 1)];
    c1[i] = pi >= 32 ? W2[pi * (pi + 1) + 64] : c0[i];
  }
#pragma unroll
  for (int p2 = NP - 1; p2 >= 0; p2--) {
    const int i = (NP - 1 - p2) & 3;
    if (p2 >= 32) {
      const double w01 = rdlane(c1[i][1], 2 * p2 - 64);
      const double x1 = rdlane(y1, 2 * p2 + 1 - 64), y0r = rdlane(y1, 2 * p2 - 64);
      const double x0 = __builtin_fma(-w01, x1, y0r);
      y1 = __builtin_fma(-c1[i][0], x0, __builtin_fma(-c1[i][1], x1, y1));   // entries on / below the diagonal inside the pair are stored as zeros
      y0 = __builtin_fma(-c0[i][0], x0, __builtin_fma(-c0[i][1], x1, y0));
      x1f = (lane >> 1) == p2 - 32 ? y1 : x1f;   // lanes 2 p2 - 64 and 2 p2 - 63 are final now
    } else {
      const double w01 = rdlane(c0[i][1], 2 * p2);
      const double x1 = rdlane(y0, 2 * p2 + 1), y0r = rdlane(y0, 2 * p2);
      const double x0 = __builtin_fma(-w01, x1, y0r);
      y0 = __builtin_fma(-c0[i][0], x0, __builtin_fma(-c0[i][1], x1, y0));
      x0f = (lane >> 1) == p2 ? y0 : x0f;
    }
    if (p2 - 4 >= 0) {
      c0[i] = W2[(p2 - 4) * (p2 - 3)];
      if (p2 - 4 >= 32) c1[i] = W2[(p2 - 4) * (p2 - 3) + 64];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  y0 = x0f; y1 = x1f;
}

// (a device function so that the local BA can launch it as workgroup 0 of a larger grid: lba.hip, k_ldlt_cols_update.  AGENT_X:
// the solution is stored with agent-scope atomic stores -- write-through to the level every XCD sees -- because workgroups of the
// SAME launch on other XCDs read it; a release fence at agent scope would write the whole L2 back instead)
template <bool AGENT_X>
__device__ __forceinline__ void ldlt_cols_body(int n, const double* __restrict__ St, double* __restrict__ x, int* __restrict__ ok_flag) {
#pragma clang fp contract(fast)        // the solve is tolerance-checked (1e-4), not bit-compared
  extern __shared__ __attribute__((aligned(16))) double sh[];
  __shared__ int s_piv;                // pivots published so far (16 k + pv + 1)
  __shared__ int s_rbits[kColT];       // per tile row k: bit j set once -W_kj is published
  __shared__ int s_rowdone[kColT];     // per tile row: wavefronts that finished it
  __shared__ int s_ok;
  const int tid = threadIdx.x, lane = tid & 63, hwv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const Geo G = make_geo(n);
  const int T = G.T, n_pad = G.n_pad, cb = G.cb;
  // Tile column <-> wavefront.  Wavefronts w and w + 4 of a workgroup share a SIMD (tools/micro/wave_simd_map.hip), and a
  // long column's trailing updates take the FP64 pipe away from whoever shares its SIMD: columns 0..3 go to wavefronts
  // 0..3, the rest in DESCENDING order to wavefronts 4..7, i.e. column w is paired with column T-1-w.  Every SIMD then carries
  // about the same number of matrix instructions, and from the middle of the factorisation on the diagonal wavefront's
  // partner has already finished (19.7 -> 19.2 us at 120 unknowns against the identity map; pairing 2w with 2w+1 puts the
  // wavefront that replays for the NEXT diagonal tile next to the diagonal: 22.3 us).
#ifdef LDLTM_IDMAP
  const int wv = hwv;
#else
  const int wv = hwv < 4 ? hwv : (T - 1 - (hwv - 4) >= 4 ? T - 1 - (hwv - 4) : kColT + hwv);      // >= T: no column
#endif
  double* const Piv = sh;                               // [4][8][64]   per pivot pair the A operand (-L[:, p0] | -L[:, p1] in their lane groups, zero elsewhere)
  double* const Rcp = Piv + kPivRing * 16 * 64;         // [4][16]      1/d, written pair by pair
  double* const Rb = Rcp + kPivRing * 16;               // [28][256]    -W_kj register images, tile (k, j) at k(15-k)/2 + j-k-1
  double* const Wl = Rb + kColPanels * 256;             // unit upper factor, pair-packed
  if (wv == 0) LDLTM_T(0);
  if (tid == 0) { s_piv = 0; s_ok = 1; }
  if (tid < kColT) { s_rbits[tid] = 0; s_rowdone[tid] = 0; }
  // ---- the column's tiles, two 16-byte loads per lane and tile, all in flight before the first use, no branch around
  // the loads (even a wave-uniform one makes hipcc drain the memory counter at the join: the column then arrives one
  // tile per memory round trip; slots past the column re-read its first tile and are never used).
  // Rt[j] is tile (k + j, wv) while tile row k is worked on: the trailing update of tile k+j writes its result into
  // slot j-1 (a matrix instruction's destination need not be its accumulator), so the panel tile of the current row is
  // always Rt[0] and no register is ever indexed at run time (hipcc turns that into 56 selects or into scratch memory).
  // The diagonal tile (wv, wv) has registers of its own.
  d4 Rt[kColT - 1], Dg;
  {
    const int wc_ = min(wv, T - 1);
    Dg = *reinterpret_cast<const d4*>(St + (size_t)tile_index(wc_, wc_) * 256 + 4 * lane);
#pragma unroll
    for (int i = 0; i < kColT - 1; i++) Rt[i] = *reinterpret_cast<const d4*>(St + (size_t)tile_index(i < wc_ ? i : 0, wc_) * 256 + 4 * lane);
  }
  __syncthreads();                     // the only barrier before the back-substitution (the tile loads above are in flight across
                                       // it): from here on the wavefronts run on flags
  LDLTM_T(310 + wv);
  if (wv == 0) LDLTM_T(1);
  LDLTM_T(300 + wv);

  auto poll_gt = [&](int* w, int k) {
#ifdef LDLTM_COLSLEEP
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) <= k) __builtin_amdgcn_s_sleep(LDLTM_COLSLEEP);
#else
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) <= k) { }
#endif
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  auto wait_free = [&](int k) {   // the pivot ring's slot k&3 was last used by tile row k-4 (T-1-(k-4) wavefronts replayed it)
    if (k >= kPivRing) poll_gt(&s_rowdone[k - kPivRing], T - 1 - (k - kPivRing) - 1);
  };
  auto wm_store = [&](int I, int J, double v) { const int p2 = J >> 1; Wl[2 * p2 * (p2 + 1) + 2 * I + (J & 1)] = v; };
  auto rb_of = [&](int k, int j) { return Rb + (k * (15 - k) / 2 + j - k - 1) * 256 + lane; };
  auto row_done = [&](int k) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(&s_rowdone[k], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  if (wv < T) {
    d4 nWlast = {0.0, 0.0, 0.0, 0.0};   // -W of this wavefront's last panel tile: stored after its diagonal tile is factored
    for (int k = 0; k <= wv && k < G.Tp; k++) {
      const int par = k & (kPivRing - 1);
      const int npiv = min(16, n_pad - 16 * k);          // a multiple of 4
      if (k == wv) {
        // ---- diagonal tile: pivot by pivot, each one streamed to the replaying wavefronts.  Everything that is not
        // on the pivot chain is kept out of the loop (tools/micro/pivot_variants.hip: chain 173 cycles, +13 for a
        // finite check, +26 for a branch, +16 for an accumulate, +37 for the two LDS stores -- per pivot, all serial):
        // one branch per four pivots, the factor's rows stay in their registers until the loop is over, a zero pivot
        // is detected from the last one (it poisons everything after it).
        LDLTM_T(8 + 8 * k + 0);
        d4 C = Dg;
        // Every FP64 vector instruction costs this wavefront 8 cycles whether anything depends on it or not (and the
        // matrix instruction shares their pipe), so the loop holds only what the next pivot needs: the rows of the
        // factor stay in registers until it is over, and a zero pivot is not tested for -- 1/0 becomes NaN in the
        // Newton step and NaN reaches every later pivot of the matrix, so the LAST pair's reciprocals tell.
        double wav[8];                           // the pairs' A operands = rows of the unit upper factor (negated)
#pragma unroll
        for (int i = 0; i < 8; i++) wav[i] = 0.0;
        double rlast = 1.0;
#ifndef LDLTM_ALONE
        if (k == 0) wait_free(k);       // (later rows: checked while this wavefront replayed row k-1, off the diagonal chain)
#endif
#ifdef LDLTM_COLPRIO
        __builtin_amdgcn_s_setprio(3);
#endif
#pragma unroll
        for (int g = 0; g < 4; g++) {
          if (g == 2) LDLTM_T(8 + 8 * k + 6);
          if (4 * g < npiv) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
              // TWO pivots per matrix instruction: rows p0, p1 = p0 + 1 sit in lane groups q0, q0 + 1 of register g.
              // With  [c00 c01; c01 c11]  the leading 2x2 block:  r0 = 1/c00,  l10 = c01 r0,  1/d1 = c00 / det,
              // row1' = row1 - l10 row0  (row0 copied into the lanes of group q0+1 by one v_permlane16_swap per half),
              // and the rank-2 update  C -= r0 row0^T row0 + (1/d1) row1'^T row1'  is ONE instruction with
              // A = [-r0 row0 | -(1/d1) row1'] in groups q0, q0+1 (zero elsewhere), B = [row0; row1'].
              const int q0 = 2 * h, p0 = 4 * g + q0, p1 = p0 + 1;
              double u = C[g];
              asm volatile("" : "+v"(u));          // own registers: the instruction below then updates C in place
              const double c00 = rdlane(u, q0 * 16 + p0), c01 = rdlane(u, q0 * 16 + p1), c11 = rdlane(u, (q0 + 1) * 16 + p1);
              const double det = __builtin_fma(c00, c11, -(c01 * c01));
              const double r0 = rcp1(c00);
              const double rdet = rcp1(det);
              const double r1 = c00 * rdet;
              const double nl10 = -(c01 * r0);
              const double u0b = row_even_to_odd(u);
              const double u1 = __builtin_fma(nl10, u0b, u);          // row1' in the lanes of group q0 + 1
              const bool in0 = lr == q0, in1 = lr == q0 + 1;
              const double bv = in1 ? u1 : u;
              // the A operand of this pair for everybody (the replaying wavefronts read -l10 out of it: element p1 of
              // its first column) and the two reciprocals.  A lone wavefront pays ~30 cycles per LDS instruction; LDS
              // executes a wavefront's instructions in order, so the counter (same value from every lane: no exec
              // juggling) becomes visible after the data
              const double av = in0 ? u * -r0 : in1 ? u1 * -r1 : 0.0;
              Piv[(par * 8 + 2 * g + h) * 64 + lane] = av;
              Rcp[par * 16 + p0 + (lane & 1)] = (lane & 1) ? r1 : r0;
              asm volatile("" ::: "memory");
              __hip_atomic_store(&s_piv, 16 * k + p1 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              wav[2 * g + h] = av;
              if (h == 1) rlast = r0 + r1;       // the loop ends on a group of four
              if (p1 < 15) C = mfma(av, bv, C);    // after the 16th pivot nothing of the tile is read again
            }
          }
        }
#ifdef LDLTM_COLPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        LDLTM_T(8 + 8 * k + 1);
        if (!(fabs(rlast) < INFINITY)) s_ok = 0;
        if (k > 0) {                               // the deferred stores of the panel tile (k-1, k)
          const int J = 16 * wv + lc;
          if (J <= cb) {
#pragma unroll
            for (int g = 0; g < 4; g++) wm_store(16 * (k - 1) + lr + 4 * g, J, -nWlast[g]);
          }
          row_done(k - 1);
        }
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const double w = -(wav[2 * g] + wav[2 * g + 1]);       // disjoint lane groups
          const int I = 16 * k + lr + 4 * g, J = 16 * k + lc;
          if (I <= (J | 1) && I < n_pad && J <= cb) wm_store(I, J, I < J ? w : 0.0);
        }
        LDLTM_T(8 + 8 * k + 2);
      } else {
#ifdef LDLTM_ALONE
        continue;                                  // timing experiment: nobody replays (results are garbage)
#endif
        // ---- panel tile (k, wv): replay the pivots behind the diagonal wavefront.
        // Per pair: its A operand comes from LDS, -l10 is read out of it, the second row is reduced by the first exactly as
        // on the diagonal, then ONE instruction applies both pivots (~200 cycles per pair with the update of the own
        // diagonal tile, tools/micro/replay_chain.hip; the diagonal wavefront needs ~400).  An LDS round trip costs a lone
        // wavefront 120-200 cycles and an LDS instruction ~25, so every wait reads what it waits for in the SAME round
        // trip: the counter first, then speculatively all eight operands and the reciprocals -- LDS executes this
        // wavefront's reads in order and the diagonal's writes in order, so whatever the counter value covers is the
        // published data.  Pairs it does not cover are read again, counter first, one round trip per try.
        d4 X = Rt[0];
        d4 D = Dg;
        // the pivot ring's slot of the row this wavefront will eliminate next: an LDS round trip that has no business between the
        // last replayed pair and the first pivot
        if (wv == k + 1) wait_free(k + 1);
        if (wv == k + 1) LDLTM_T(8 + 8 * k + 3);
        LDLTM_T(80 + wv * 24 + 3 * k);
        d4 Rc = {0.0, 0.0, 0.0, 0.0}, nW = {0.0, 0.0, 0.0, 0.0};
        const lds_vdp pivr = (lds_vdp)(Piv) + par * 8 * 64 + lane;
        const lds_vdp rcpr = (lds_vdp)(Rcp) + par * 16 + lr;
        auto pair_step = [&](const int g, const int h, const double a) {
          const int q0 = 2 * h, p1 = 4 * g + q0 + 1;
          double xg = X[g];
          asm volatile("" : "+v"(xg));         // own registers: X is then updated in place
          const double x0b = row_even_to_odd(xg);
          const double nl10 = rdlane(a, q0 * 16 + p1);
          const double rrow = (lr == q0 + 1) ? __builtin_fma(nl10, x0b, xg) : xg;   // rows p0, p1 of R in groups q0, q0+1
          Rc[g] = (h == 0 || lr >= 2) ? rrow : Rc[g];
          if (p1 < 15) X = mfma(a, rrow, X);   // X -= L[:, p0] R[p0, :] + L[:, p1] R[p1, :]; dead after its 16th row
        };
        double a8[8], rs[4];
        const int c0 = __hip_atomic_load(&s_piv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; i++) a8[i] = pivr[i * 64];
#pragma unroll
        for (int g = 0; g < 4; g++) rs[g] = rcpr[4 * g];
        int seen = __builtin_amdgcn_readfirstlane(c0);
#pragma unroll
        for (int g = 0; g < 4; g++) {
          if (4 * g < npiv) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const int need = 16 * k + 4 * g + 2 * h + 1;       // the counter exceeds this once the pair is published
              if (seen <= need) {
                // a wait also fetches the FOLLOWING pair's operand: the diagonal needs ~400 cycles per pair, a wait is a
                // round trip of ~200, so the next pair is often published by the time this one's wait returns and the
                // wavefront that will factor the next diagonal tile stays one pair behind instead of two
                for (;;) {
                  const int c = __hip_atomic_load(&s_piv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                  asm volatile("" ::: "memory");
                  a8[2 * g + h] = pivr[(2 * g + h) * 64];
                  if (2 * g + h + 1 < 8) a8[2 * g + h + 1] = pivr[(2 * g + h + 1) * 64];
                  rs[g] = rcpr[4 * g];
                  if (h == 1 && g + 1 < 4) rs[g + 1] = rcpr[4 * (g + 1)];
                  const int cv = __builtin_amdgcn_readfirstlane(c);
                  if (cv > need) { seen = min(cv, need + 3); break; }     // this read covers this pair and, if published, the next
                }
              }
              pair_step(g, h, a8[2 * g + h]);
            }
            // four rows complete: -W = -D^-1 R and the own diagonal tile (wv, wv) -= R^T W
            nW[g] = Rc[g] * -rs[g];
            D = mfma(Rc[g], nW[g], D);
            if (g == 0 && k > 0) {
              // the factor store of the PREVIOUS tile row (for the back-substitution) and its completion count (frees the
              // pivot ring four rows later) are nobody's next step: issued here, where a wavefront that keeps up waits anyway
              const int J = 16 * wv + lc;
              if (J <= cb) {
#pragma unroll
                for (int g2 = 0; g2 < 4; g2++) wm_store(16 * (k - 1) + lr + 4 * g2, J, -nWlast[g2]);
              }
              row_done(k - 1);
            }
          }
        }
        Dg = D;
        if (wv == k + 1) LDLTM_T(8 + 8 * k + 4);
        LDLTM_T(80 + wv * 24 + 3 * k + 1);
        // publish -W for the trailing tiles of the other columns (their B operand is their own unscaled R)
        // (measured: deferring this behind the first pivot pair for the wavefront that eliminates next -- to get ~150 cycles of
        // LDS issue off the diagonal chain -- is slower, 19.8 vs 19.2 us: the readers' late start costs more)
        double* const rb = rb_of(k, wv);
#pragma unroll
        for (int g = 0; g < 4; g++) rb[g * 64] = nW[g];
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_or(&s_rbits[k], 1 << wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (wv == k + 1) LDLTM_T(8 + 8 * k + 5);
        nWlast = nW;
        if (wv > k + 1) {
          // ---- trailing tiles (k + j, wv), 0 < j < wv - k:  -= W_k,k+j^T R_k,wv.  The row's publication bits and every
          // operand in one round trip (the same speculation as above), then the instructions, the tiles' chains
          // interleaved (a dependent matrix instruction waits 82 cycles, an independent one issues after 66).  One
          // straight-line variant per tile count: with a condition per tile hipcc carries every slot through ~100 register
          // copies per tile row.
          const int want = ((1 << wv) - 1) & ~((2 << k) - 1);       // columns k+1 .. wv-1
          auto trailing = [&](auto ntc) {
            constexpr int NT = decltype(ntc)::value;
            double ta[NT][4];
            for (;;) {
              const int f = __hip_atomic_load(&s_rbits[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              asm volatile("" ::: "memory");
#pragma unroll
              for (int j = 0; j < NT; j++) {
                const lds_vdp pa = (lds_vdp)rb_of(k, k + 1 + j);
#pragma unroll
                for (int q = 0; q < 4; q++) ta[j][q] = pa[q * 64];
              }
              if ((__builtin_amdgcn_readfirstlane(f) & want) == want) break;
            }
            d4 c[NT];
#pragma unroll
            for (int j = 0; j < NT; j++) c[j] = Rt[j + 1];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
              for (int j = 0; j < NT; j++) c[j] = mfma(ta[j][q], Rc[q], c[j]);
#pragma unroll
            for (int j = 0; j < NT; j++) Rt[j] = c[j];
          };
          switch (wv - k - 1) {
            case 1: trailing(std::integral_constant<int, 1>{}); break;
            case 2: trailing(std::integral_constant<int, 2>{}); break;
            case 3: trailing(std::integral_constant<int, 3>{}); break;
            case 4: trailing(std::integral_constant<int, 4>{}); break;
            case 5: trailing(std::integral_constant<int, 5>{}); break;
            default: trailing(std::integral_constant<int, 6>{}); break;
          }
          LDLTM_T(80 + wv * 24 + 3 * k + 2);
        }
      }
    }
    // a column without pivots of its own (the border column when n_pad is a multiple of 16) never reaches the branch
    // above that stores its last panel tile
    if (wv >= G.Tp && wv > 0) {
      const int J = 16 * wv + lc;
      if (J <= cb) {
#pragma unroll
        for (int g = 0; g < 4; g++) wm_store(16 * (wv - 1) + lr + 4 * g, J, -nWlast[g]);
      }
      row_done(wv - 1);
    }
  }
  if (wv == 0) LDLTM_T(2);
  __syncthreads();
  if (wv == 0) LDLTM_T(3);
  const int ok = s_ok;
  // ---- back-substitution  L^T x = y  on one wavefront (n_pad <= 124: two registers of x per lane), two columns per
  // 128-bit load, one pair ahead; lanes past the pair's rows are switched off (they hold finished x)
  if (wv == 0 && ok) {
    // volatile: hipcc otherwise sinks the look-ahead load into the iteration that uses it (and then waits for it there)
    const lds_vd2p W2 = (lds_vd2p)(Wl) + lane;   // entry pair (I, 2p..2p+1) at p(p+1) + I
    double y0, y1;
    {
      const int pb = (cb >> 1) * ((cb >> 1) + 1);
      const d2 a0 = W2[pb], a1 = W2[pb + 64];
      y0 = lane < n_pad ? a0[0] : 0.0; y1 = 64 + lane < n_pad ? a1[0] : 0.0;
    }
    const int np = n_pad >> 1;
    LDLTM_T(5);
    if (np == 60) backsub_pairs_unrolled<60>(W2, y0, y1, lane);          // 20 free poses
    else if (np == 58) backsub_pairs_unrolled<58>(W2, y0, y1, lane);     // 19
    else if (np == 54) backsub_pairs_unrolled<54>(W2, y0, y1, lane);     // 18
    else if (np == 48) backsub_pairs_unrolled<48>(W2, y0, y1, lane);     // 16
    else {
    // pairs 32 .. np-1: the diagonal sits in y1, y0 takes part in full.  Four pairs are in flight (an LDS round trip
    // is ~130 cycles, a pair's two dependent steps ~60): slot i of the ring is refilled right after its use
    if (np > 32) {
      d2 c0[4], c1[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { const int pi = max(np - 1 - i, 32); c0[i] = W2[pi * (pi + 1)]; c1[i] = W2[pi * (pi + 1) + 64]; }
      for (int pb = np - 1; pb >= 32; pb -= 4) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int p2 = pb - i;
          if (p2 >= 32) {
            const bool on = 64 + lane <= 2 * p2 + 1;
            const double x1 = rdlane(y1, 2 * p2 + 1 - 64);
            y1 -= (on ? c1[i][1] : 0.0) * x1;
            y0 -= c0[i][1] * x1;
            const double x0 = rdlane(y1, 2 * p2 - 64);
            y1 -= (on ? c1[i][0] : 0.0) * x0;
            y0 -= c0[i][0] * x0;
            const int pn = max(p2 - 4, 32);          // past the end: re-read a valid pair instead of branching
            c0[i] = W2[pn * (pn + 1)]; c1[i] = W2[pn * (pn + 1) + 64];
          }
        }
      }
    }
    {
      const int ps = min(np, 32) - 1;
      if (ps >= 0) {
        d2 c0[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { const int pi = max(ps - i, 0); c0[i] = W2[pi * (pi + 1)]; }
        for (int pb = ps; pb >= 0; pb -= 4) {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int p2 = pb - i;
            if (p2 >= 0) {
              const bool on = lane <= 2 * p2 + 1;
              const double x1 = rdlane(y0, 2 * p2 + 1);
              y0 -= (on ? c0[i][1] : 0.0) * x1;
              const double x0 = rdlane(y0, 2 * p2);
              y0 -= (on ? c0[i][0] : 0.0) * x0;
              const int pn = max(p2 - 4, 0);
              c0[i] = W2[pn * (pn + 1)];
            }
          }
        }
      }
    }
    }
    LDLTM_T(6);
    if constexpr (AGENT_X) {
      if (lane < n) __hip_atomic_store(&x[lane], y0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (64 + lane < n) __hip_atomic_store(&x[64 + lane], y1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane < n) x[lane] = y0;
      if (64 + lane < n) x[64 + lane] = y1;
    }
  }
  if (wv == 0) LDLTM_T(4);
  if (tid == 0) *ok_flag = ok;
}

__global__ __launch_bounds__(kThreads) void k_ldlt_cols(int n, const double* __restrict__ St, double* __restrict__ x, int* __restrict__ ok_flag) {
  ldlt_cols_body<false>(n, St, x, ok_flag);
}

// true if k_ldlt_mfma covers a system of n unknowns (n = 6 * free poses)
__host__ inline bool supports(int n) { return n >= 1 && make_geo(n).T <= 19; }

struct Launch { const void* fn; size_t lds; bool wlds; bool cols; int threads; };

// dynamic LDS beyond 64 KB has to be allowed per kernel and per DEVICE: the owner of a device context (an lba handle) keeps one
struct AttrCache { struct { const void* fn; size_t lds; } e[12] = {}; };

// Which kernel a system of n unknowns gets (measured, tools/micro/ldlt_mfma_test): up to kColT tile rows the column kernel, 9 tile rows
// the four-wavefront tile kernel with W in LDS, beyond that the 512-thread kernels (16 / 24 / 32 / 48 tiles per wavefront group).
// (Rounds 2-4 kept an eight-wavefront tile kernel and a tile kernel for small windows behind switches: slower, removed in round 5.)
__host__ inline Launch pick(int n) {
  const Geo g = make_geo(n);
  Launch L;
  if (g.T <= kColT) {
    L.fn = reinterpret_cast<const void*>(k_ldlt_cols); L.wlds = true; L.cols = true; L.threads = kThreads;
    L.lds = lds_doubles_cols(g) * sizeof(double);
    return L;
  }
  L.cols = false;
  L.threads = kThreads;
  if (g.T <= 9) { L.fn = reinterpret_cast<const void*>(k_ldlt_mfma<6, 3, true>); L.wlds = true; }
  else {
    L.threads = kBigThreads; L.wlds = false;
    if (g.T <= 10) L.fn = reinterpret_cast<const void*>(k_ldlt_big<16, 4>);
    else if (g.T <= 13) L.fn = reinterpret_cast<const void*>(k_ldlt_big<24, 4>);
    else if (g.T <= 15) L.fn = reinterpret_cast<const void*>(k_ldlt_big<32, 4>);
    else L.fn = reinterpret_cast<const void*>(k_ldlt_big48);
  }
  L.lds = lds_doubles(g, L.wlds) * sizeof(double);
  return L;
}

// St: the bordered matrix as a tile image (see image_put_rhs / k_image_pad), x: solution, wglob: wglob_doubles() of scratch
__host__ inline hipError_t launch(int n, const double* St, double* x, int* ok, double* wglob, hipStream_t st, AttrCache* cache) {
  const Launch L = pick(n);
  // dynamic LDS beyond 64 KB has to be allowed per kernel, once per device (and again if a larger system comes along)
  AttrCache local;
  auto& attr = (cache ? cache : &local)->e;
  if (L.lds > 64 * 1024) {
    int w = 0;
    while (w < 11 && attr[w].fn && attr[w].fn != L.fn) w++;
    if (attr[w].fn != L.fn || attr[w].lds < L.lds) {
      hipError_t e = hipFuncSetAttribute(L.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds);
      if (e != hipSuccess) return e;
      attr[w].fn = L.fn; attr[w].lds = L.lds;
    }
  }
  void* args[] = {(void*)&n, (void*)&St, (void*)&x, (void*)&ok, (void*)&wglob};   // the column kernel ignores wglob
  return hipLaunchKernel(L.fn, dim3(1), dim3(L.threads), args, L.lds, st);
}

// stand-alone tools (tools/micro): one device
__host__ inline hipError_t launch(int n, const double* St, double* x, int* ok, double* wglob, hipStream_t st) {
  static AttrCache cache;
  return launch(n, St, x, ok, wglob, st, &cache);
}

}  // namespace ldltm
