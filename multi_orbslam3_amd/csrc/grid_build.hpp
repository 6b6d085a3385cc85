// Shared by matcher.hip and extractor.hip: the frame parameters a kernel reads and the feature grid build
// (Frame::AssignFeaturesToGrid / PosInGrid, S/Frame.cc:360-391,699-709; CSR, cell = ix*48+iy) as a workgroup body.
// matcher.hip launches it as a kernel of its own (host-built frames) and as the last launch of the fused Frame constructors.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/orbgpu.h"
#include "wave.hpp"

namespace orbg {

constexpr int kCells = ORBG_GRID_COLS * ORBG_GRID_ROWS;
constexpr int kGridLdsItems = 4096;

struct FrameParams {
  int n;
  float min_x, max_x, min_y, max_y;
  float w_inv, h_inv;               // mfGridElementWidthInv / HeightInv (S/Frame.cc:127-144)
  float fx, fy, cx, cy, bf, b;
  int n_levels;
  float log_sf;
  float scale[ORBG_MAX_LEVELS];
};


// NT threads per workgroup (1024 as a kernel of its own; 256 as one more workgroup of the 256-thread stereo match launch): a thread
// owns kCells / NT consecutive cells of the scan and kGridLdsItems / NT strided features.  The result does not depend on NT.
template <int NT>
__device__ __forceinline__ void grid_build_body(const orbx_keypoint* __restrict__ kps, FrameParams fp,
                                                int* __restrict__ cell_of, int* __restrict__ cell_start,
                                                int* __restrict__ cell_items, const int* __restrict__ d_n) {
  constexpr int IPT = kGridLdsItems / NT, CPT = kCells / NT, NW = NT / 64;
  static_assert(kCells % NT == 0 && kGridLdsItems % NT == 0 && NT % 64 == 0 && NW <= 16, "grid_build_body: unsupported workgroup size");
  __shared__ int cnt[kCells];
  __shared__ int s_items[kGridLdsItems];   // cell_items staged in LDS (frames of up to kGridLdsItems features): fill + per-cell
                                           // sort without a global-memory round trip per step
  if (d_n) fp.n = *d_n;               // feature count produced on the device (GPU quad-tree path)
  __shared__ int wsum[16];
  __shared__ int s_total;
  const int tid = threadIdx.x;
  const bool in_lds = fp.n <= kGridLdsItems;
  for (int c = tid; c < kCells; c += NT) cnt[c] = 0;
  __syncthreads();
  int my_cell[IPT];     // cells of this thread's features (register copy; cell_of[] is still written for the API)
#pragma unroll
  for (int q = 0; q < IPT; q++) my_cell[q] = -1;
  for (int i = tid, q = 0; i < fp.n; i += NT, q++) {
    const int px = (int)roundf((kps[i].x - fp.min_x) * fp.w_inv);
    const int py = (int)roundf((kps[i].y - fp.min_y) * fp.h_inv);
    int c = -1;
    if (!(px < 0 || px >= ORBG_GRID_COLS || py < 0 || py >= ORBG_GRID_ROWS)) {
      c = px * ORBG_GRID_ROWS + py;
      atomicAdd(&cnt[c], 1);
    }
    cell_of[i] = c;
#pragma unroll
    for (int z = 0; z < IPT; z++) if (z == q) my_cell[z] = c;
  }
  __syncthreads();
  // exclusive scan of the 3072 counts: CPT per thread
  const int c0 = tid * CPT;
  int cv[CPT];
  int mine = 0;
#pragma unroll
  for (int k = 0; k < CPT; k++) { cv[k] = cnt[c0 + k]; mine += cv[k]; }
  const int lane = tid & 63, wave = tid >> 6;
  const int inc = wave_incl_scan_add(mine);
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; w++) base += wsum[w];
  const int excl = base + inc - mine;
  int st[CPT];
  {
    int run = excl;
#pragma unroll
    for (int k = 0; k < CPT; k++) { st[k] = run; cell_start[c0 + k] = run; run += cv[k]; }
    if (tid == NT - 1) { cell_start[kCells] = run; s_total = run; }
  }
  __syncthreads();
  const int n_items = s_total;
#pragma unroll
  for (int k = 0; k < CPT; k++) cnt[c0 + k] = st[k];   // running fill cursors
  __syncthreads();
  if (in_lds) {
#pragma unroll
    for (int q = 0; q < IPT; q++) {
      const int i = tid + NT * q;
      if (i < fp.n && my_cell[q] >= 0) s_items[atomicAdd(&cnt[my_cell[q]], 1)] = i;
    }
    __syncthreads();
    // restore insertion (= keypoint index) order inside every cell
#pragma unroll
    for (int q = 0; q < CPT; q++) {
      const int s0 = st[q];
      const int e = cnt[c0 + q];
      for (int i = s0 + 1; i < e; i++) {
        const int key = s_items[i];
        int j = i - 1;
        while (j >= s0 && s_items[j] > key) { s_items[j + 1] = s_items[j]; j--; }
        s_items[j + 1] = key;
      }
    }
    __syncthreads();
    for (int i = tid; i < n_items; i += NT) cell_items[i] = s_items[i];
    return;
  }
  for (int i = tid; i < fp.n; i += NT) {
    const int cc = cell_of[i];
    if (cc >= 0) cell_items[atomicAdd(&cnt[cc], 1)] = i;
  }
  __syncthreads();
  __threadfence_block();
#pragma unroll
  for (int q = 0; q < CPT; q++) {
    const int s0 = st[q];
    const int e = cnt[c0 + q];
    for (int i = s0 + 1; i < e; i++) {
      const int key = cell_items[i];
      int j = i - 1;
      while (j >= s0 && cell_items[j] > key) { cell_items[j + 1] = cell_items[j]; j--; }
      cell_items[j + 1] = key;
    }
  }
}

// Frame::UndistortKeyPoints (S/Frame.cc:721-754) for the fused monocular constructor: cv::undistortPoints(mat, mat, K, mDistCoef, Mat(), mK)
// per keypoint, in double as OpenCV 3.2's cvUndistortPoints computes it (five fixed-point iterations; the operation order is the
// oracle's, oracle/matching.cc oracle_undistort_points, and the translation unit is built with -ffp-contract=off: bit-equal results).
struct UndistortArgs {
  int on;                                    // 0: mvKeysUn = mvKeys (mDistCoef[0] == 0), nothing below is read
  double fx, fy, cx, cy, k1, k2, p1, p2, k3;
  const orbx_keypoint* src;                  // mvKeys on the device (the extractor's buffer)
  orbx_keypoint* dst;                        // mvKeysUn on the device: what the frame's grid and searches read
  orbx_keypoint* dst_host;                   // the same records in mapped pinned memory (delivered to the caller as mvKeysUn), may be NULL
};

__device__ __forceinline__ void undistort_point(const UndistortArgs& a, float u, float v, float* xo, float* yo) {
  const double ifx = 1. / a.fx, ify = 1. / a.fy;
  double x = (double)u, y = (double)v;
  x = (x - a.cx) * ifx;
  y = (y - a.cy) * ify;
  const double x0 = x, y0 = y;
#pragma unroll 1
  for (int j = 0; j < 5; j++) {
    const double r2 = x * x + y * y;
    const double icdist = 1. / (1 + ((a.k3 * r2 + a.k2) * r2 + a.k1) * r2);
    const double deltaX = 2 * a.p1 * x * y + a.p2 * (r2 + 2 * x * x);
    const double deltaY = a.p1 * (r2 + 2 * y * y) + 2 * a.p2 * x * y;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  const double xx = a.fx * x + 0. * y + a.cx;
  const double yy = 0. * x + a.fy * y + a.cy;
  const double ww = 1. / (0. * x + 0. * y + 1.);
  *xo = (float)(xx * ww);
  *yo = (float)(yy * ww);
}

}  // namespace orbg
