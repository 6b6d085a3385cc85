// liborbgpu -- ORB extractor for gfx950 (MI355X): pyramid, FAST-9/16 score + per-cell NMS/threshold
// fallback, quad-tree keypoint selection, intensity-centroid angle, 7x7 blur and 256-bit rBRIEF.
//
// Replaces ORB_SLAM3::ORBextractor (S/ORBextractor.cc) behind the C-ABI of include/orbgpu.h.
// Design (MI355X-first, not a translation of the OpenCV call sequence):
//   * left+right images of a stereo rig share every launch (blockIdx.y/z = camera);
//   * pyramid levels live in ONE bordered HBM buffer per camera (19-px REFLECT_101 border written by the
//     same kernel that resizes, so no separate copyMakeBorder pass and no separate blur image);
//   * FAST: one 256-thread workgroup per 30-px cell, the (w+6)x(h+6) sub-image staged in LDS, a full
//     max-threshold score per pixel (branch-free sliding min/max over the 16-ring), NMS inside the cell's
//     detection area, the iniTh -> minTh fallback decided per cell with one workgroup reduction, and an
//     ORDER-PRESERVING compaction (row-major inside the cell, cells in the reference's i,j order);
//   * the blurred level image is never materialised: each keypoint's 43x43 raw patch is staged in LDS by
//     one wavefront, blurred separably in LDS (u16 row pass, exact integer), and the 512 rotated pattern
//     lookups read LDS; the orientation moments use the same staged patch;
//   * small, latency-bound hand-offs (candidate lists, selected keypoints) go through mapped pinned
//     host memory written/read directly by the kernels -- no extra memcpy launches.
// The quad-tree (DistributeOctTree; serial and pointer-chasing in the reference) runs on the device too: octree_kernel, an
// LDS-resident closed form of the std::list bookkeeping per (camera, level) with a pinned deterministic tie-break
// (orbx_config.octree_oldest_first selects the other order).  An index/range based host implementation with the same
// tie-break remains for partial lapping areas and for frames whose candidate lists overflow the device buffers.
//
// Bit-exactness contract (tests/test_gpu_parity.py): pyramid bytes, FAST candidate lists, kept
// keypoints, angles (f32 bits) and descriptors are identical to the CPU oracle.
// Compile with -ffp-contract=off (strict f32 for fastAtan2 and the pattern rotation).

#include "common.hpp"
#include "wave.hpp"
#include "stereo_finalize.hpp"
#include "grid_build.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

using namespace orbg;

namespace {

// ------------------------------------------------------------------------------------------------
// geometry shared by host and kernels

struct LevelGeom {
  int w, h;        // level size (without border)
  int stride;      // bytes per row of the bordered buffer
  int off;         // byte offset of the bordered buffer inside the camera's pyramid allocation
  int xt_off;      // offset into the x resize table (entries), level >= 1
  int yt_off;      // offset into the y resize table
  int cell_begin;  // first cell record of this level
  int cell_end;
};

struct PyrGeom {
  int n_levels;
  int cam_stride;              // bytes between camera 0 and camera 1 pyramid allocations
  LevelGeom lv[ORBG_MAX_LEVELS];
  float scale[ORBG_MAX_LEVELS];
};

struct ResizeTap { short ofs, a0, a1, pad; };   // source index + the two 11-bit fixed-point weights

struct CellRec {   // one FAST cell = one workgroup   (S/ORBextractor.cc:787-853)
  short level;
  short x0, y0;    // sub-image origin in level coordinates (iniX, iniY)
  short cw, ch;    // sub-image size (maxX-iniX, maxY-iniY), <= 65
  short offx, offy;  // j*wCell, i*hCell added to the keypoint coordinates (:847-848)
  short pad;
};

struct SelKp {     // keypoint chosen by the quad-tree, in final output order
  short x, y;      // level coordinates (border offset already added, :868-869)
  short level;
  short cam;
  int out_idx;     // position in the camera's output arrays (lapping order, :1135-1144)
  float response;
};

constexpr int kTile = 66;         // max sub-image side is 65 (wCell,hCell <= 59, +6)
constexpr int kTileStride = 68;
constexpr int kQueuePerWave = (59 * 59 + 255) / 256 * 64;   // a wavefront tests at most 64 pixels per 256-pixel stride
constexpr int kCellCap = 1024;    // >= ceil(59/2)^2 = 900 survivors of a strict 3x3 NMS

__device__ __forceinline__ int reflect101(int p, int n) {
  // BORDER_REFLECT_101 for |overshoot| < n
  p = p < 0 ? -p : p;
  return p >= n ? 2 * n - 2 - p : p;
}

// ------------------------------------------------------------------------------------------------
// pyramid  (ORBextractor::ComputePyramid, S/ORBextractor.cc:1152-1177; cv::resize INTER_LINEAR 8UC1,
// SURVEY.md Appendix A-1; cv::copyMakeBorder REFLECT_101)

__global__ __launch_bounds__(256) void pyr_level0_kernel(const uint8_t* __restrict__ img0, const uint8_t* __restrict__ img1,
                                                        int img_stride, uint8_t* __restrict__ pyr, PyrGeom g) {
  const int cam = blockIdx.z;
  const LevelGeom L = g.lv[0];
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
  const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ox >= L.w + 2 * kEdge || oy >= L.h + 2 * kEdge) return;
  const uint8_t* img = cam ? img1 : img0;
  const int sx = reflect101(ox - kEdge, L.w), sy = reflect101(oy - kEdge, L.h);
  pyr[(size_t)cam * g.cam_stride + L.off + (size_t)oy * L.stride + ox] = img[(size_t)sy * img_stride + sx];
}

__global__ __launch_bounds__(256) void pyr_resize_kernel(uint8_t* __restrict__ pyr, PyrGeom g, int level,
                                                        const ResizeTap* __restrict__ xtab, const ResizeTap* __restrict__ ytab) {
  const int cam = blockIdx.z;
  const LevelGeom D = g.lv[level];
  const LevelGeom S = g.lv[level - 1];
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
  const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ox >= D.w + 2 * kEdge || oy >= D.h + 2 * kEdge) return;
  const int dx = reflect101(ox - kEdge, D.w), dy = reflect101(oy - kEdge, D.h);
  const ResizeTap tx = xtab[D.xt_off + dx], ty = ytab[D.yt_off + dy];
  uint8_t* base = pyr + (size_t)cam * g.cam_stride;
  const uint8_t* src = base + S.off + (size_t)kEdge * S.stride + kEdge;
  const int sx0 = tx.ofs, sx1 = min(sx0 + 1, S.w - 1);
  const int sy0 = min(max((int)ty.ofs, 0), S.h - 1), sy1 = min(max((int)ty.ofs + 1, 0), S.h - 1);
  const uint8_t* r0 = src + (size_t)sy0 * S.stride;
  const uint8_t* r1 = src + (size_t)sy1 * S.stride;
  const int t0 = r0[sx0] * tx.a0 + r0[sx1] * tx.a1;
  const int t1 = r1[sx0] * tx.a0 + r1[sx1] * tx.a1;
  const int v = ((((int)ty.a0 * (t0 >> 4)) >> 16) + (((int)ty.a1 * (t1 >> 4)) >> 16) + 2) >> 2;
  base[D.off + (size_t)oy * D.stride + ox] = (uint8_t)v;
}

// ------------------------------------------------------------------------------------------------
// Whole pyramid in ONE launch ("tower" per level-0 tile).  The cascade level l <- level l-1 is a pure function of four
// source pixels, so a workgroup that holds a level-0 tile plus a small halo in LDS can derive "its" pixels of every
// level without waiting for other workgroups: ownership of a level-l pixel is inherited from the level-(l-1) pixel its
// first tap reads (per axis, so owned sets are rectangles), and the extra pixels the bilinear footprints reach beyond a
// tile (1 px per level, growing by the scale factor downwards: ~17 px at level 0 for 8 levels x 1.2) are recomputed
// locally.  Same taps, same integer arithmetic as pyr_resize_kernel -> identical bytes; 8 dependent launches of a few
// microseconds each become one.  The host precomputes, per tile and axis, the owned and the needed pixel range of each
// level (TowerAxis); the reflect-101 border copies are written by the owner of the interior pixel they mirror.
constexpr int kTwSide = 64;        // largest needed extent per axis and level a workgroup can hold
constexpr int kTwThreads = 1024;   // 16 wavefronts: the level chain is latency bound, rows are spread as thin as possible
constexpr int kTwWaves = kTwThreads / 64;

struct TowerAxis { short own_lo[ORBG_MAX_LEVELS], own_hi[ORBG_MAX_LEVELS], need_lo[ORBG_MAX_LEVELS], need_hi[ORBG_MAX_LEVELS]; };
struct TowerTap { short i0, i1, a0, a1; };   // source indices relative to the previous level's LDS block + weights

__device__ __forceinline__ void tower_emit(uint8_t* __restrict__ lvl, const LevelGeom& L, int dx, int dy, uint8_t v) {
  // interior position plus the border positions that mirror (dx, dy) under REFLECT_101 (19-px border)
  const int xa = kEdge + dx, ya = kEdge + dy;
  const int xb = (dx >= 1 && dx <= kEdge) ? kEdge - dx : -1;
  const int xc = (dx >= L.w - 1 - kEdge && dx <= L.w - 2) ? kEdge + 2 * (L.w - 1) - dx : -1;
  const int yb = (dy >= 1 && dy <= kEdge) ? kEdge - dy : -1;
  const int yc = (dy >= L.h - 1 - kEdge && dy <= L.h - 2) ? kEdge + 2 * (L.h - 1) - dy : -1;
  uint8_t* r = lvl + (size_t)ya * L.stride;
  r[xa] = v;
  if (xb >= 0) r[xb] = v;
  if (xc >= 0) r[xc] = v;
  if (yb >= 0) {
    r = lvl + (size_t)yb * L.stride;
    r[xa] = v;
    if (xb >= 0) r[xb] = v;
    if (xc >= 0) r[xc] = v;
  }
  if (yc >= 0) {
    r = lvl + (size_t)yc * L.stride;
    r[xa] = v;
    if (xb >= 0) r[xb] = v;
    if (xc >= 0) r[xc] = v;
  }
}

// -DTW_PROFILE: s_memtime stamps of thread 0 of workgroup (1, 1, 0) (tools/micro/tw_prof.py prints them)
#ifdef TW_PROFILE
__device__ long long g_tw_prof[32];
#define TW_T(slot) do { if (threadIdx.x == 0 && blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 0) g_tw_prof[slot] = clock64(); } while (0)
#else
#define TW_T(slot) do { } while (0)
#endif
__global__ __launch_bounds__(kTwThreads) void pyr_tower_kernel(const uint8_t* __restrict__ img0, const uint8_t* __restrict__ img1, int img_stride,
                                                       uint8_t* __restrict__ pyr, PyrGeom g, const ResizeTap* __restrict__ xtab,
                                                       const ResizeTap* __restrict__ ytab, const TowerAxis* __restrict__ tax,
                                                       const TowerAxis* __restrict__ tay) {
  __shared__ __attribute__((aligned(16))) uint8_t buf[2][kTwSide * kTwSide];
  __shared__ TowerTap s_xt[ORBG_MAX_LEVELS][kTwSide], s_yt[ORBG_MAX_LEVELS][kTwSide];
  __shared__ TowerAxis s_ax, s_ay;
  __shared__ LevelGeom s_lv[ORBG_MAX_LEVELS];   // kernel arguments sit in cold memory: each first touch of a cache line costs
                                                // a full miss, so the per-level geometry is fetched once, up front
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TW_T(0);
  // XCD-aware workgroup -> tile map: workgroups are dealt round-robin over the 8 XCDs (linear id % 8), each with an L2 of
  // its own.  Horizontally adjacent tiles write the two halves of the cache lines that straddle their border; on different
  // XCDs every such line is written back twice as a partial line.  Logical tile L = (id % 8) * (total / 8) + id / 8 gives
  // every XCD a contiguous run of tiles (x fastest), so those lines are completed in one L2.
  int bx = blockIdx.x, by = blockIdx.y, cam = blockIdx.z;
  {
    const int total = gridDim.x * gridDim.y * gridDim.z;
    if ((total & 7) == 0) {
      const int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const int L = (id & 7) * (total >> 3) + (id >> 3);
      bx = L % gridDim.x; by = (L / gridDim.x) % gridDim.y; cam = L / (gridDim.x * gridDim.y);
    }
  }
  const int nl = g.n_levels;
  {
    constexpr int kWords = (int)sizeof(TowerAxis) / 4;
    if (tid < kWords) reinterpret_cast<int*>(&s_ax)[tid] = reinterpret_cast<const int*>(tax + bx)[tid];
    else if (tid < 2 * kWords) reinterpret_cast<int*>(&s_ay)[tid - kWords] = reinterpret_cast<const int*>(tay + by)[tid - kWords];
    constexpr int kLvWords = (int)sizeof(LevelGeom) / 4;
    const int j = tid - 2 * kWords;
    if (j >= 0 && j < nl * kLvWords) reinterpret_cast<int*>(s_lv)[j] = reinterpret_cast<const int*>(g.lv)[j];
  }
  __syncthreads();
  TW_T(1);
  // level-0 block: image -> registers (all loads in flight), then LDS + the owned part to the pyramid
  const int x0 = s_ax.need_lo[0], nw0 = s_ax.need_hi[0] - x0, y0 = s_ay.need_lo[0], nh0 = s_ay.need_hi[0] - y0;
  const uint8_t* img = cam ? img1 : img0;
  uint8_t px[kTwSide / kTwWaves];
#pragma unroll
  for (int k = 0; k < kTwSide / kTwWaves; k++) {
    const int yy = wave + kTwWaves * k;
    // clamped, unconditional: a predicated load makes hipcc branch around it and wait for each one separately
    px[k] = img[(size_t)(y0 + min(yy, nh0 - 1)) * img_stride + x0 + min(lane, nw0 - 1)];
  }
  // taps of every level for this tile's ranges, source indices made relative to the previous level's block
  for (int l0 = 1; l0 < nl; l0 += kTwWaves / 2) {
    const int l = l0 + (wave >> 1);               // waves 2j, 2j+1: x and y taps of level l0+j
    if (l < nl) {
      const bool is_y = wave & 1;
      const LevelGeom D = s_lv[l], S = s_lv[l - 1];
      const TowerAxis& A = is_y ? s_ay : s_ax;
      const int lo = A.need_lo[l], n = A.need_hi[l] - lo, plo = A.need_lo[l - 1];
      if (lane < n) {
        const ResizeTap t = is_y ? ytab[D.yt_off + lo + lane] : xtab[D.xt_off + lo + lane];
        int i0, i1;
        if (is_y) { i0 = min(max((int)t.ofs, 0), S.h - 1); i1 = min(max((int)t.ofs + 1, 0), S.h - 1); }
        else { i0 = t.ofs; i1 = min(i0 + 1, S.w - 1); }
        TowerTap o;
        o.i0 = (short)(i0 - plo); o.i1 = (short)(i1 - plo); o.a0 = t.a0; o.a1 = t.a1;
        if (is_y) s_yt[l][lane] = o; else s_xt[l][lane] = o;
      }
    }
  }
  uint8_t* cbase = pyr + (size_t)cam * g.cam_stride;
  TW_T(2);
  {
    const LevelGeom L = s_lv[0];
    const int ohx = s_ax.own_hi[0], ohy = s_ay.own_hi[0];
    const bool plain0 = s_ax.own_lo[0] > kEdge && ohx - 1 < L.w - 1 - kEdge && s_ay.own_lo[0] > kEdge && ohy - 1 < L.h - 1 - kEdge;
#pragma unroll
    for (int k = 0; k < kTwSide / kTwWaves; k++) {
      const int yy = wave + kTwWaves * k;
      if (yy < nh0 && lane < nw0) {
        buf[0][yy * kTwSide + lane] = px[k];
        if (x0 + lane < ohx && y0 + yy < ohy) {
          if (plain0) (cbase + L.off)[(size_t)(kEdge + y0 + yy) * L.stride + kEdge + x0 + lane] = px[k];
          else tower_emit(cbase + L.off, L, x0 + lane, y0 + yy, px[k]);
        }
      }
    }
  }
  __syncthreads();
  TW_T(3);
  for (int l = 1; l < nl; l++) {
    const int lox = s_ax.need_lo[l], nw = s_ax.need_hi[l] - lox, loy = s_ay.need_lo[l], nh = s_ay.need_hi[l] - loy;
    if (nw > 0 && nh > 0) {
      const LevelGeom L = s_lv[l];
      const uint8_t* src = buf[(l - 1) & 1];
      uint8_t* dst = buf[l & 1];
      const int olx = s_ax.own_lo[l], ohx = s_ax.own_hi[l], oly = s_ay.own_lo[l], ohy = s_ay.own_hi[l];
      const bool plain = olx > kEdge && ohx - 1 < L.w - 1 - kEdge && oly > kEdge && ohy - 1 < L.h - 1 - kEdge;   // workgroup-uniform
      if (lane < nw) {
        const TowerTap tx = s_xt[l][lane];
        for (int yy = wave; yy < nh; yy += kTwWaves) {
          const TowerTap ty = s_yt[l][yy];
          const uint8_t* r0 = src + ty.i0 * kTwSide;
          const uint8_t* r1 = src + ty.i1 * kTwSide;
          const int t0 = r0[tx.i0] * tx.a0 + r0[tx.i1] * tx.a1;
          const int t1 = r1[tx.i0] * tx.a0 + r1[tx.i1] * tx.a1;
          const int v = ((((int)ty.a0 * (t0 >> 4)) >> 16) + (((int)ty.a1 * (t1 >> 4)) >> 16) + 2) >> 2;
          dst[yy * kTwSide + lane] = (uint8_t)v;
          const int dx = lox + lane, dy = loy + yy;
          if (dx >= olx && dx < ohx && dy >= oly && dy < ohy) {
            // a tile whose owned range stays clear of the bands that REFLECT_101 mirrors into the border (most tiles) writes
            // the interior byte only: the eight conditional mirror stores of tower_emit are ~60 instructions per pixel
            if (plain) (cbase + L.off)[(size_t)(kEdge + dy) * L.stride + kEdge + dx] = (uint8_t)v;
            else tower_emit(cbase + L.off, L, dx, dy, (uint8_t)v);
          }
        }
      }
    }
    __syncthreads();
    TW_T(3 + l);
  }
}
#ifdef TW_PROFILE
extern "C" int orbx_debug_tw_prof(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tw_prof), sizeof(g_tw_prof)) == hipSuccess ? 0 : -1; }
#endif

// ------------------------------------------------------------------------------------------------
// FAST-9/16 per cell  (cv::FAST via S/ORBextractor.cc:808-841)

// max over the 16 arcs of 9 contiguous ring pixels of min(d) -- sliding-window minimum by doubling.
__device__ __forceinline__ int arc9_maxmin(const int (&d)[16]) {
  int m2[16], m4[16], m8[16];
#pragma unroll
  for (int i = 0; i < 16; i++) m2[i] = min(d[i], d[(i + 1) & 15]);
#pragma unroll
  for (int i = 0; i < 16; i++) m4[i] = min(m2[i], m2[(i + 2) & 15]);
#pragma unroll
  for (int i = 0; i < 16; i++) m8[i] = min(m4[i], m4[(i + 4) & 15]);
  int best = -1024;
#pragma unroll
  for (int i = 0; i < 16; i++) best = max(best, min(m8[i], d[(i + 8) & 15]));
  return best;
}

__global__ __launch_bounds__(256) void fast_cells_kernel(const uint8_t* __restrict__ pyr, PyrGeom g,
                                                        const CellRec* __restrict__ cells, int n_cells, int ini_th,
                                                        int min_th, uint32_t* __restrict__ slots, int* __restrict__ counts) {
  __shared__ __attribute__((aligned(16))) uint8_t tile_raw[kTile * kTileStride];
  __shared__ __attribute__((aligned(16))) uint8_t score[(kTile + 2) * kTileStride];   // +1 apron of zeros all around the detection area
  __shared__ unsigned wsum[4];
  __shared__ unsigned short queue[4 * kQueuePerWave];
  __shared__ int qcount[4];
  // XCD-aware block -> cell map: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD), so
  // XCD k is given a CONTIGUOUS run of cells: neighbouring cells (which share a 6-px halo and cache lines) hit the
  // same XCD's L2 instead of eight different ones.  Placement only affects speed, never results.
  const int cam = blockIdx.y;
  int cell;
  {
    const int b = blockIdx.x, per = (n_cells + 7) >> 3;
    cell = (b & 7) * per + (b >> 3);
    if (cell >= n_cells || (b >> 3) >= per) return;      // grid is rounded up to 8 * per
  }
  const CellRec c = cells[cell];
  const LevelGeom L = g.lv[c.level];
  const uint8_t* src = pyr + (size_t)cam * g.cam_stride + L.off + (size_t)(kEdge + c.y0) * L.stride + (kEdge + c.x0);
  const int cw = c.cw, ch = c.ch;
  {
    uint32_t* z = reinterpret_cast<uint32_t*>(score);
    for (int i = threadIdx.x; i < (kTile + 2) * kTileStride / 4; i += 256) z[i] = 0;
  }
  // stage the sub-image with aligned dword loads, all issued before the first LDS store (one memory round trip per
  // workgroup instead of one per row): the tile keeps the source's alignment, i.e. its column 0 sits at byte xs
  const int xs = (kEdge + c.x0) & 3;
  const uint8_t* tl = tile_raw + xs;
  {
    const uint8_t* src_al = src - xs;                       // 4-byte aligned: level offsets and strides are multiples of 64
    const int ndw = (xs + cw + 3) >> 2;                     // <= 17 = kTileStride / 4
    const int total = ch * ndw;                             // <= 65 * 17 = 1105 < 5 * 256
    uint32_t v[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const int i = threadIdx.x + k * 256;
      if (i < total) {
        const int r = i / ndw, cdw = i - r * ndw;
        v[k] = *reinterpret_cast<const uint32_t*>(src_al + (size_t)r * L.stride + 4 * cdw);
      }
    }
    uint32_t* t32 = reinterpret_cast<uint32_t*>(tile_raw);
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const int i = threadIdx.x + k * 256;
      if (i < total) {
        const int r = i / ndw, cdw = i - r * ndw;
        t32[r * (kTileStride / 4) + cdw] = v[k];
      }
    }
  }
  __syncthreads();
  const int wd = cw - 6, hd = ch - 6;       // detection area [3,cw-3) x [3,ch-3)
  const int npix = wd > 0 && hd > 0 ? wd * hd : 0;
  // (x, y) of pixel p advance incrementally: one integer division per thread instead of one per pixel
  const int wdiv = max(wd, 1);
  const int step_y = 256 / wdiv, step_x = 256 - step_y * wdiv;
  int y = (int)threadIdx.x / wdiv, x = (int)threadIdx.x - y * wdiv;
  // Pass A -- the segment test's necessary condition on the 4 compass points of the ring (any 9-arc holds one pixel of
  // every opposite pair): ~92 % of all pixels fail it and keep score 0, which is what cv::FAST's score buffer holds for
  // them too (scores below the threshold can neither become keypoints nor suppress one).  Survivors are queued in LDS.
  const int qt = min(min_th, ini_th);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned short* wq = queue + wave * kQueuePerWave;          // each wavefront appends to its own segment: the ballot is
  int wn = 0;                                                  // wave-uniform, so no atomic and no shuffle is needed
  for (int p0 = 0; p0 < npix; p0 += 256, x += step_x, y += step_y) {
    const int p = p0 + threadIdx.x;
    bool pass = false;
    if (p < npix) {
      if (x >= wd) { x -= wd; y++; }
      const uint8_t* q = &tl[(y + 3) * kTileStride + (x + 3)];
      const int v = q[0];
      const int d0 = v - q[3 * kTileStride], d8 = v - q[-3 * kTileStride], d4 = v - q[3], d12 = v - q[-3];
      pass = ((d0 > qt || d8 > qt) && (d4 > qt || d12 > qt)) || ((d0 < -qt || d8 < -qt) && (d4 < -qt || d12 < -qt));
    }
    const unsigned long long bal = __ballot(pass);
    if (pass) wq[wn + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)(x | (y << 8));
    wn += __popcll(bal);
  }
  if (lane == 0) qcount[wave] = wn;
  __syncthreads();
  // Pass B -- full cornerScore<16> of the queued pixels (order is irrelevant: results land by position)
  const int n0 = qcount[0], n1 = n0 + qcount[1], n2 = n1 + qcount[2], nq = n2 + qcount[3];
  for (int i = threadIdx.x; i < nq; i += 256) {
    const int e = i < n0 ? queue[i] : i < n1 ? queue[kQueuePerWave + i - n0] : i < n2 ? queue[2 * kQueuePerWave + i - n1]
                                                                                      : queue[3 * kQueuePerWave + i - n2];
    const int ex = e & 255, ey = e >> 8;
    const uint8_t* q = &tl[(ey + 3) * kTileStride + (ex + 3)];
    const int v = q[0];
    int d[16];
    d[0] = v - q[3 * kTileStride];          d[1] = v - q[3 * kTileStride + 1];
    d[2] = v - q[2 * kTileStride + 2];      d[3] = v - q[kTileStride + 3];
    d[4] = v - q[3];                        d[5] = v - q[-kTileStride + 3];
    d[6] = v - q[-2 * kTileStride + 2];     d[7] = v - q[-3 * kTileStride + 1];
    d[8] = v - q[-3 * kTileStride];         d[9] = v - q[-3 * kTileStride - 1];
    d[10] = v - q[-2 * kTileStride - 2];    d[11] = v - q[-kTileStride - 3];
    d[12] = v - q[-3];                      d[13] = v - q[kTileStride - 3];
    d[14] = v - q[2 * kTileStride - 2];     d[15] = v - q[3 * kTileStride - 1];
    int nd[16];
#pragma unroll
    for (int k = 0; k < 16; k++) nd[k] = -d[k];
    const int sc = max(arc9_maxmin(d), arc9_maxmin(nd)) - 1;   // cornerScore<16>: largest passing threshold
    score[(ey + 1) * kTileStride + (ex + 1)] = (uint8_t)max(sc, 0);
  }
  __syncthreads();
  // NMS + threshold decision + ordered compaction; thread t owns pixels [t*K, (t+1)*K) in row-major order
  const int K = (npix + 255) >> 8;
  const int p0 = threadIdx.x * K, p1 = min(p0 + K, npix);
  unsigned keep_ini = 0, keep_min = 0;
  const int y0 = p0 / wdiv, x0 = p0 - y0 * wdiv;
  y = y0; x = x0;
  for (int p = p0; p < p1; p++, x++) {
    if (x >= wd) { x = 0; y++; }
    const uint8_t* s = &score[(y + 1) * kTileStride + (x + 1)];
    const int v = s[0];
    if (v >= min_th) {
      const int m = max(max(max((int)s[-kTileStride - 1], (int)s[-kTileStride]), max((int)s[-kTileStride + 1], (int)s[-1])),
                        max(max((int)s[1], (int)s[kTileStride - 1]), max((int)s[kTileStride], (int)s[kTileStride + 1])));
      if (v > m) {
        keep_min |= 1u << (p - p0);
        if (v >= ini_th) keep_ini |= 1u << (p - p0);
      }
    }
  }
  unsigned total;
  const unsigned packed = (unsigned)__popc(keep_ini) | ((unsigned)__popc(keep_min) << 16);
  const unsigned excl = block_excl_scan_256(packed, &total, wsum);
  const bool use_min = (total & 0xFFFFu) == 0;             // vKeysCell.empty() after the iniTh pass (:825)
  unsigned mask = use_min ? keep_min : keep_ini;
  unsigned pos = use_min ? (excl >> 16) : (excl & 0xFFFFu);
  uint32_t* out = slots + ((size_t)cam * n_cells + cell) * kCellCap;
  while (mask) {
    const int b = __ffs(mask) - 1;
    mask &= mask - 1;
    int xx = x0 + b, yy = y0;
    while (xx >= wd) { xx -= wd; yy++; }
    const unsigned sc = score[(yy + 1) * kTileStride + (xx + 1)];
    out[pos++] = (unsigned)(xx + 3 + c.offx) | ((unsigned)(yy + 3 + c.offy) << 12) | (sc << 24);
  }
  if (threadIdx.x == 0) counts[cam * n_cells + cell] = use_min ? (total >> 16) : (total & 0xFFFFu);
}

// Compaction of the per-cell slots into one candidate list (camera-major, level-major, cell-major), written
// straight into mapped pinned host memory together with the per-(camera,level) start offsets.
__global__ __launch_bounds__(256) void gather_cells_kernel(const uint32_t* __restrict__ slots, const int* __restrict__ counts,
                                                          PyrGeom g, const CellRec* __restrict__ cells, int n_cells,
                                                          int n_cams, int* __restrict__ out_hdr, uint32_t* __restrict__ out_cand, int out_cap) {
  __shared__ int wsum[4];
  const int cell = blockIdx.x, cam = blockIdx.y;
  const int gidx = cam * n_cells + cell;
  int s = 0;
  for (int i = threadIdx.x; i < gidx; i += 256) s += counts[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  const int base = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  const int n = counts[gidx];
  const int level = cells[cell].level;
  if (threadIdx.x == 0) {
    if (cell == g.lv[level].cell_begin) out_hdr[cam * ORBG_MAX_LEVELS + level] = base;
    if (cell == n_cells - 1 && cam == n_cams - 1) out_hdr[2 * ORBG_MAX_LEVELS] = base + n;   // grand total
    if (cell == n_cells - 1) out_hdr[2 * ORBG_MAX_LEVELS + 1 + cam] = base + n;              // end of this camera
  }
  const uint32_t* in = slots + (size_t)gidx * kCellCap;
  for (int i = threadIdx.x; i < n; i += 256)
    if (base + i < out_cap) out_cand[base + i] = in[i];
}


// ------------------------------------------------------------------------------------------------
// DistributeOctTree on the GPU (S/ORBextractor.cc:479-761): one 512-thread workgroup per (camera, level).
//
// The reference's std::list bookkeeping has a closed form.  One pass splits a set P of nodes in a processing order
// p_1..p_m; children (non-empty only, order n1..n4) are pushed to the FRONT, parents erased, so
//     new list = reverse(children(p_1) ++ ... ++ children(p_m))  ++  (old list minus P, order kept).
// Phase 1 ("while not finished"): P = every node with more than one key, in list order, no early exit.
// Phase 2 (entered once size + 3*nToExpand > N): P = the nodes with more than one key in DESCENDING (key count,
// creation order), cut after the first node that brings the list to >= N nodes (the reference's `break`), which is a
// prefix sum over the children counts.  Termination: size >= N or a pass that did not change the size.
// Keys do not need to stay ordered inside a node: the per-leaf winner "first maximum response in arrival order" is
// the key with the largest response and, among equals, the smallest candidate index (arrival order is always the
// candidate order) -- one packed atomicMax.  Everything lives in LDS; results are bit-identical to the host
// implementation above (same pinned tie-break).
constexpr int kOctThreads = 256;
constexpr int kOctKeyCap = 4096;     // candidates per (camera, level)
constexpr int kOctListCap = 2048;    // nodes alive at once (<= 4*N + 8)

struct OctCfg {
  int n_levels, n_cams;
  int oldest_first;                  // quad-tree tie-break variant (orbx_config.octree_oldest_first)
  int jump;                          // 1: the first (up to three) uniform passes come from a three-level histogram (octree_kernel, "jump start")
  int n_target[ORBG_MAX_LEVELS];     // mnFeaturesPerLevel
  int reg_off[2][ORBG_MAX_LEVELS];   // start of the (camera, level) region in the selection buffer
  int reg_cap[ORBG_MAX_LEVELS];
};

struct OctSel { short x, y; float response; };   // level coordinates (border offset added)

// Exclusive prefix over the workgroup (thread order) of one value per thread, and the total.  One barrier; `wsum` must not
// be the array the previous call on this path used (callers rotate through four).
__device__ __forceinline__ int oct_block_excl(int s, int* total, int* wsum /*LDS[4]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int inc = wave_incl_scan_add(s);
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kOctThreads / 64; w++) {
    const int x = wsum[w];
    if (w < wave) base += x;
    tot += x;
  }
  *total = tot;
  return base + inc - s;
}

// Execution model (round 2; the closed form above is unchanged).  A pass used to be 9-14 workgroup barriers over LDS arrays,
// every phase a chain of dependent LDS round trips of four wavefronts: 8 k cycles per pass even for a one-node list, 73 k
// cycles (30 us) for level 0.  Now
//   * a thread keeps its keys (<= 16: candidate word, list position, quadrant) in registers; a node is ONE 16-byte LDS
//     record (bounds, key count, creation number);
//   * the nodes are dealt to the threads in contiguous chunks of the list, so the prefix sums of a pass (children before a
//     node, survivors before a node) are a thread-local walk plus one wave scan and one barrier, and the thread that scanned
//     a node creates its children without reading the scan back;
//   * "keys follow their node" of pass p and "count the children's keys" of pass p+1 are one phase; the children-count
//     words of the next list are zeroed by whoever creates a node.
// A first-phase pass is three barriers, a second-phase pass six.
constexpr int kOctKPT = kOctKeyCap / kOctThreads;     // keys per thread
// One LDS atomic per WAVEFRONT on a workgroup-wide word: hipcc's atomic optimiser turns a per-lane atomicMax / atomicMin on
// one address into a scalar loop over the active lanes (s_ff1 + v_readlane per lane: ~3 k cycles for a full wavefront);
// a DPP reduction first and a single lane's atomic is ~150.
__device__ __forceinline__ void oct_wave_atomic_add(int* w, int v) { const int s = wave_sum(v); if ((threadIdx.x & 63) == 0 && s) atomicAdd(w, s); }
__device__ __forceinline__ void oct_wave_atomic_max(int* w, int v) { const int s = wave_max(v); if ((threadIdx.x & 63) == 0) atomicMax(w, s); }
__device__ __forceinline__ void oct_wave_atomic_min(int* w, int v) { const int s = -wave_max(-v); if ((threadIdx.x & 63) == 0) atomicMin(w, s); }
// -DOCT_PROFILE: s_memtime stamps of thread 0 per workgroup (tools/micro/oct_prof.py prints them)
#ifdef OCT_PROFILE
__device__ long long g_oct_prof[32][48];
#define OCT_T(slot) do { if (threadIdx.x == 0 && (slot) < 48) g_oct_prof[blockIdx.x][slot] = clock64(); } while (0)
#else
#define OCT_T(slot) do { } while (0)
#endif

// The workgroup takes its candidates straight from fast_cells_kernel's per-cell slots -- the
// cells of a (camera, level) are a contiguous run, cell-major order IS the candidate order -- instead of from the compacted list
// gather_cells_kernel wrote (one launch less per constructor).  A thread owns a contiguous chunk of the level's cells: it requests
// their counts and, in the same trip to memory, the first eight slot entries of each (most cells hold fewer); a workgroup scan of
// the counts gives every cell its place in an LDS staging array (the not-yet-used second node list), the entries go there, and
// the keys are then dealt to the threads from LDS exactly as they were from the compacted list.
constexpr int kOctSpecCells = 4;     // cells per thread whose first entries are requested before their counts are known
__global__ __launch_bounds__(kOctThreads) void octree_kernel(const uint32_t* __restrict__ cand, const int* __restrict__ hdr,
                                                            PyrGeom g, OctCfg cfg, OctSel* __restrict__ sel_out,
                                                            int* __restrict__ lvl_count, int* __restrict__ overflow, int cand_cap,
                                                            const uint32_t* __restrict__ slots, const int* __restrict__ counts, int n_cells) {
  __shared__ uint4 node[2][kOctListCap];          // x0 | x1 << 16, y0 | y1 << 16, key count | creation number << 16, -
  __shared__ unsigned cc2[2][kOctListCap][2];     // children key counts, two u16 packed per word: [q>>1] >> 16*(q&1)
  __shared__ unsigned short child_pos[kOctListCap][4];
  __shared__ unsigned short new_pos[kOctListCap], ord[kOctListCap], pos_of_ord[kOctListCap];
  __shared__ __attribute__((aligned(16))) int scanA[kOctListCap];
  __shared__ __attribute__((aligned(16))) int scanB[kOctListCap];
  __shared__ unsigned best[kOctListCap];
  __shared__ int wsum[4][4];
  __shared__ int s_n, s_cut, s_m, s_flag[2];
  const int tid = threadIdx.x;
  const int task = blockIdx.x;
  OCT_T(0);
  const int cam = task % cfg.n_cams, level = task / cfg.n_cams;
  const LevelGeom L = g.lv[level];
  const int N = cfg.n_target[level];
  int* out_count = lvl_count + cam * ORBG_MAX_LEVELS + level;
  if (L.cell_end == L.cell_begin) { if (tid == 0) *out_count = 0; return; }
  uint32_t kw[kOctKPT];
  unsigned kn[kOctKPT];
  int nk;
  {
    uint32_t* const stage = reinterpret_cast<uint32_t*>(&node[1][0]);     // 32 KB, first written by the first pass (behind barriers)
    static_assert(sizeof(uint4) * kOctListCap >= sizeof(uint32_t) * kOctKeyCap, "staging does not fit the second node list");
    const int nc = L.cell_end - L.cell_begin;
    const int cpt = (nc + kOctThreads - 1) / kOctThreads;               // cells per thread (1-2 at 640 x 480, 4 at 1280 x 720)
    const int c0 = min(tid * cpt, nc), c1 = min(c0 + cpt, nc);
    const int* cnt_p = counts + (size_t)cam * n_cells + L.cell_begin;
    const uint32_t* slot_p = slots + ((size_t)cam * n_cells + L.cell_begin) * kCellCap;
    int cn[kOctSpecCells];
    uint4 sp[kOctSpecCells][2];
#pragma unroll
    for (int q = 0; q < kOctSpecCells; q++) {
      const int c = min(c0 + q, nc - 1);
      cn[q] = cnt_p[c];
      const uint4* row = reinterpret_cast<const uint4*>(slot_p + (size_t)c * kCellCap);
      sp[q][0] = row[0]; sp[q][1] = row[1];
    }
    int mine = 0;
#pragma unroll
    for (int q = 0; q < kOctSpecCells; q++) { if (c0 + q >= c1) cn[q] = 0; mine += cn[q]; }
    for (int c = c0 + kOctSpecCells; c < c1; c++) mine += cnt_p[c];     // (images with more than 1024 cells per level)
    int total;
    int base = oct_block_excl(mine, &total, wsum[3]);
    nk = total;
    if (nk <= 0) { if (tid == 0) *out_count = 0; return; }
    if (nk > kOctKeyCap || 4 * N + 8 > kOctListCap) { if (tid == 0) { *overflow = 1; *out_count = 0; } return; }
#pragma unroll
    for (int q = 0; q < kOctSpecCells; q++) {
      const int n_c = cn[q];
      if (n_c > 0) {
        const uint32_t e8[8] = {sp[q][0].x, sp[q][0].y, sp[q][0].z, sp[q][0].w, sp[q][1].x, sp[q][1].y, sp[q][1].z, sp[q][1].w};
#pragma unroll
        for (int j = 0; j < 8; j++) if (j < n_c) stage[base + j] = e8[j];
        if (n_c > 8) {
          const uint32_t* row = slot_p + (size_t)(c0 + q) * kCellCap;
          for (int j = 8; j < n_c; j++) stage[base + j] = row[j];
        }
        base += n_c;
      }
    }
    for (int c = c0 + kOctSpecCells; c < c1; c++) {
      const int n_c = cnt_p[c];
      const uint32_t* row = slot_p + (size_t)c * kCellCap;
      for (int j = 0; j < n_c; j++) stage[base + j] = row[j];
      base += n_c;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < kOctKPT; m++) kw[m] = stage[min(tid + kOctThreads * m, nk - 1)];
    // (the first pass writes node[1] only behind the barriers of the root set-up below)
  }
  const int minB = kEdge - 3;
  const int W = (L.w - kEdge + 3) - minB, H = (L.h - kEdge + 3) - minB;   // maxX-minX, maxY-minY
  int nIni = (int)roundf((float)W / (float)H);
  if (nIni < 1) nIni = 1;
  const float hX = (float)W / (float)nIni;
  auto nonzero4 = [](unsigned c01, unsigned c23) { return (int)((c01 & 0xFFFFu) != 0) + (int)((c01 >> 16) != 0) + (int)((c23 & 0xFFFFu) != 0) + (int)((c23 >> 16) != 0); };
  // ---- roots (:541-583)
  for (int i = tid; i < nIni; i += kOctThreads) scanA[i] = 0;
  // scratch of the jump start: h3[128] | h2[32] | h1[8] behind the root counts, the order masks in scanB
  constexpr int kJ3 = 16, kJ2 = kJ3 + 128, kJ1 = kJ2 + 32, kJEnd = kJ1 + 8;
  for (int i = kJ3 + tid; i < kJEnd; i += kOctThreads) scanA[i] = 0;
  if (tid < 32) scanB[16 + tid] = 0;
  __syncthreads();
#pragma unroll
  for (int m = 0; m < kOctKPT; m++) {
    if (tid + kOctThreads * m < nk) {
      const int x = kw[m] & 0xFFF;
      int r = (int)((float)x / hX);
      if (r >= nIni) r = nIni - 1;
      kn[m] = (unsigned)r;                    // provisional: root index
      atomicAdd(&scanA[r], 1);
    } else {
      kn[m] = 0;
    }
  }
  __syncthreads();
  // ---- jump start (round 4).  Phase-1 passes split EVERY node that holds more than one key, so the list after the first J of them
  // (J <= 3: 1 -> 4 -> 16 -> 64 nodes per root, ~5.5 k cycles each for almost no work) is a function of the key counts of the
  // 4 / 16 / 64 cells of a three-level quadrant grid per root:
  //   * a depth-d cell is a node iff it holds keys and every ancestor holds more than one; it stops splitting when it holds one;
  //   * creation order within pass j ("parents in list order, quadrants ascending", children then pushed to the front reversed):
  //     depth 1: (root, q1) ascending; depth 2: (root, q1) DESCENDING then q2 ascending; depth 3: (root, q1) ascending, q2
  //     descending, q3 ascending -- the permuted indices pi1 / pi2 / pi3 below;
  //   * list after pass J = depth-J nodes in descending pi_J, then the one-key leaves of depth J-1 in descending pi_{J-1}, ...,
  //     depth 1, then the one-key roots in root order (survivors keep their order behind the new children);
  //   * after each pass the reference's rules decide (:667-671): size >= N or unchanged -> finished; size + 3 * expandable > N ->
  //     phase 2.  They are evaluated on the counts, so J, the phase and "finished" come out exactly as the pass loop would find them.
  // One histogram (an LDS atomic per key), bit masks of "node" / "one-key leaf" per depth in creation order, positions by popcount.
  bool jumped = false, jump_finished = false;
  int jump_n = 0, jump_mode = 1;
  if (cfg.jump && nIni <= 2 && (scanA[0] > 1 || (nIni > 1 && scanA[1] > 1))) {
    int* const h3 = scanA + kJ3; int* const h2 = scanA + kJ2; int* const h1 = scanA + kJ1;
    unsigned* const EX3 = reinterpret_cast<unsigned*>(scanB) + 16; unsigned* const LF3 = EX3 + 4;          // [4] words each
    unsigned* const EX2 = EX3 + 8; unsigned* const LF2 = EX3 + 9; unsigned* const EX1 = EX3 + 10; unsigned* const LF1 = EX3 + 11;
    unsigned* const RLF = EX3 + 12;                                                                        // one-key roots (bit r)
    auto root_box = [&](int r, int* x0, int* x1) { *x0 = (int)(short)(int)(hX * (float)r); *x1 = (int)(short)(int)(hX * (float)(r + 1)); };
    auto child_of = [](int q, int* x0, int* x1, int* y0, int* y1) {
      const int mx = *x0 + ((*x1 - *x0 + 1) >> 1), my = *y0 + ((*y1 - *y0 + 1) >> 1);
      if (q & 1) *x0 = mx; else *x1 = mx;
      if (q & 2) *y0 = my; else *y1 = my;
    };
    OCT_T(20);
    unsigned code[kOctKPT];
#pragma unroll
    for (int m = 0; m < kOctKPT; m++) {
      code[m] = 0;
      if (tid + kOctThreads * m < nk) {
        const int kx = kw[m] & 0xFFF, ky = (kw[m] >> 12) & 0xFFF;
        int x0, x1, y0 = 0, y1 = H;
        root_box((int)kn[m], &x0, &x1);
        unsigned c = kn[m];
#pragma unroll
        for (int d = 0; d < 3; d++) {
          const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
          const int q = (kx < mx ? 0 : 1) + (ky < my ? 0 : 2);
          child_of(q, &x0, &x1, &y0, &y1);
          c = c * 4 + (unsigned)q;
        }
        code[m] = c;
        atomicAdd(&h3[c], 1);
      }
    }
    OCT_T(21);
    __syncthreads();
    OCT_T(22);
    const int idmax1 = nIni * 4 - 1;
    auto cnt0 = [&](int r) { return r < nIni ? scanA[r] : 0; };
    auto pi2 = [&](int id2) { return (idmax1 - (id2 >> 2)) * 4 + (id2 & 3); };
    auto pi3 = [](int id3) { return ((id3 >> 4) * 4 + (3 - ((id3 >> 2) & 3))) * 4 + (id3 & 3); };
    // the masks by wavefront ballots (no LDS atomics on shared words): thread t of wavefronts 0 / 1 stands for the depth-3 cell
    // whose creation-order index pi3 is t; wavefront 2: lanes 0..31 the depth-2 cell with pi2 = lane, lanes 32..39 the depth-1
    // cells, lanes 40..41 the roots.  A thread sums the 16 depth-3 counts of its depth-1 cell itself (four 128-bit reads in
    // flight): the counts of its parents need no pass of their own; the depth-2 / depth-1 threads leave theirs in h2 / h1.
    {
      const int lane = tid & 63, wv = tid >> 6;
      bool is_node = false, one = false;
      int id1 = -1, q2 = 0, q3 = 0, kind = -1;          // kind: 3 / 2 / 1 = depth of the thread's cell, 0 = root
      if (wv < 2) { id1 = tid >> 4; q2 = 3 - ((tid >> 2) & 3); q3 = tid & 3; kind = 3; }                   // pi3^-1(t)
      else if (wv == 2 && lane < 32) { id1 = idmax1 - (lane >> 2); q2 = lane & 3; kind = 2; }            // pi2^-1(lane)
      else if (wv == 2 && lane < 40) { id1 = lane - 32; kind = 1; }
      else if (wv == 2 && lane < 42) { kind = 0; }
      if (kind >= 1 && id1 >= 0 && (id1 >> 2) < nIni) {
        const int4* blk = reinterpret_cast<const int4*>(h3 + 16 * id1);
        const int4 b0 = blk[0], b1 = blk[1], b2 = blk[2], b3 = blk[3];
        const int s0 = b0.x + b0.y + b0.z + b0.w, s1 = b1.x + b1.y + b1.z + b1.w, s2 = b2.x + b2.y + b2.z + b2.w, s3 = b3.x + b3.y + b3.z + b3.w;
        const int c1 = (s0 + s1) + (s2 + s3);
        const int c2 = q2 == 0 ? s0 : q2 == 1 ? s1 : q2 == 2 ? s2 : s3;
        const int4 bq = q2 == 0 ? b0 : q2 == 1 ? b1 : q2 == 2 ? b2 : b3;
        const int c3 = q3 == 0 ? bq.x : q3 == 1 ? bq.y : q3 == 2 ? bq.z : bq.w;
        const bool n1 = c1 > 0 && cnt0(id1 >> 2) > 1, n2 = c2 > 0 && n1 && c1 > 1, n3 = c3 > 0 && n2 && c2 > 1;
        if (kind == 3) { is_node = n3; one = n3 && c3 == 1; }
        else if (kind == 2) { is_node = n2; one = n2 && c2 == 1; h2[id1 * 4 + q2] = c2; }
        else { is_node = n1; one = n1 && c1 == 1; h1[id1] = c1; }
      } else if (kind == 0) {
        const int r = lane - 40;
        if (r < nIni && scanA[r] == 1) one = true;
      }
      const unsigned long long bn = __ballot(is_node), bo = __ballot(one);
      if (lane == 0) {
        if (wv < 2) { EX3[2 * wv] = (unsigned)bn; EX3[2 * wv + 1] = (unsigned)(bn >> 32); LF3[2 * wv] = (unsigned)bo; LF3[2 * wv + 1] = (unsigned)(bo >> 32); }
        else if (wv == 2) {
          *EX2 = (unsigned)bn; *LF2 = (unsigned)bo;
          *EX1 = (unsigned)(bn >> 32) & 0xFFu; *LF1 = (unsigned)(bo >> 32) & 0xFFu;
          *RLF = (unsigned)(bo >> 40) & 0x3u;
        }
      }
    }
    OCT_T(23);
    __syncthreads();
    OCT_T(24);
    auto node1 = [&](int id1) { return h1[id1] > 0 && cnt0(id1 >> 2) > 1; };
    auto node2 = [&](int id2) { return h2[id2] > 0 && node1(id2 >> 2) && h1[id2 >> 2] > 1; };
    auto node3 = [&](int id3) { return h3[id3] > 0 && node2(id3 >> 2) && h2[id3 >> 2] > 1; };
    const unsigned e3[4] = {EX3[0], EX3[1], EX3[2], EX3[3]}, f3[4] = {LF3[0], LF3[1], LF3[2], LF3[3]};
    const unsigned e2 = *EX2, f2 = *LF2, e1 = *EX1, f1 = *LF1, rl = *RLF;
    const int k1 = __popc(e1), l1 = __popc(f1), k2 = __popc(e2), l2 = __popc(f2);
    const int k3 = __popc(e3[0]) + __popc(e3[1]) + __popc(e3[2]) + __popc(e3[3]);
    const int l3 = __popc(f3[0]) + __popc(f3[1]) + __popc(f3[2]) + __popc(f3[3]);
    const int l0 = __popc(rl), xr = (int)(cnt0(0) > 1) + (int)(cnt0(1) > 1), n0 = l0 + xr;
    // the pass loop's decisions on the counts
    int J = 1, nn = k1 + l0, prevn = n0, md = 1;
    bool fin = false;
    if (nn >= N || nn == prevn) fin = true;
    else if (nn + 3 * (k1 - l1) > N) md = 2;
    else if (k1 - l1 == 0) fin = true;                              // no node left to split: the next pass would leave at once
    else {
      J = 2; prevn = nn; nn = k2 + l1 + l0;
      if (nn >= N || nn == prevn) fin = true;
      else if (nn + 3 * (k2 - l2) > N) md = 2;
      else if (k2 - l2 == 0) fin = true;
      else {
        J = 3; prevn = nn; nn = k3 + l2 + l1 + l0;
        if (nn >= N || nn == prevn) fin = true;
        else if (nn + 3 * (k3 - l3) > N) md = 2;
      }
    }
    const int KJ = J == 1 ? k1 : J == 2 ? k2 : k3;
    auto below128 = [&](const unsigned* w, int p) {                 // set bits of the 128-bit mask below bit p
      int a = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) a += j < (p >> 5) ? __popc(w[j]) : j == (p >> 5) ? __popc(w[j] & ((1u << (p & 31)) - 1u)) : 0;
      return a;
    };
    auto below32 = [](unsigned w, int p) { return __popc(w & ((1u << p) - 1u)); };
    auto above32 = [](unsigned w, int p) { return p >= 31 ? 0 : __popc(w >> (p + 1)); };
    // list position of: a depth-J node, a one-key leaf of depth d < J, a one-key root
    const int off2 = KJ, off1 = KJ + (J > 2 ? l2 : 0), off0 = KJ + (J > 2 ? l2 : 0) + (J > 1 ? l1 : 0);
    auto pos_d3 = [&](int id3) { return KJ - 1 - below128(e3, pi3(id3)); };
    auto pos_d2 = [&](int id2) { const int p = pi2(id2); return J == 2 ? KJ - 1 - below32(e2, p) : off2 + above32(f2, p); };
    auto pos_d1 = [&](int id1) { return J == 1 ? KJ - 1 - below32(e1, id1) : off1 + above32(f1, id1); };
    auto pos_d0 = [&](int r) { return off0 + below32(rl, r); };
    // ---- the nodes: one thread per cell that is in the list
    {
      int depth = -1, id = 0;
      if (tid < 128) { depth = 3; id = tid; } else if (tid < 160) { depth = 2; id = tid - 128; } else if (tid < 168) { depth = 1; id = tid - 160; } else if (tid < 170) { depth = 0; id = tid - 168; }
      const int r = depth < 0 ? nIni : id >> (2 * depth);
      if (depth >= 0 && r < nIni) {
        bool in_list; int pos = 0, cnt = 0, seq = 0;
        if (depth == 3) { in_list = J == 3 && node3(id); if (in_list) { pos = pos_d3(id); cnt = h3[id]; seq = below128(e3, pi3(id)); } }
        else if (depth == 2) { in_list = J >= 2 && node2(id) && (J == 2 || h2[id] == 1); if (in_list) { pos = pos_d2(id); cnt = h2[id]; seq = below32(e2, pi2(id)); } }
        else if (depth == 1) { in_list = node1(id) && (J == 1 || h1[id] == 1); if (in_list) { pos = pos_d1(id); cnt = h1[id]; seq = below32(e1, id); } }
        else { in_list = scanA[id] == 1; if (in_list) { pos = pos_d0(id); cnt = 1; seq = (id == 1 && scanA[0] > 0) ? 1 : 0; } }
        if (in_list) {
          int x0, x1, y0 = 0, y1 = H;
          root_box(r, &x0, &x1);
          for (int d = depth - 1; d >= 0; d--) child_of((id >> (2 * d)) & 3, &x0, &x1, &y0, &y1);
          node[0][pos] = make_uint4((unsigned)(unsigned short)(short)x0 | ((unsigned)(unsigned short)(short)x1 << 16),
                                    (unsigned)(unsigned short)(short)y0 | ((unsigned)(unsigned short)(short)y1 << 16), (unsigned)cnt | ((unsigned)seq << 16), 0u);
          cc2[0][pos][0] = 0; cc2[0][pos][1] = 0;
        }
      }
    }
    OCT_T(25);
    // ---- the keys: the deepest node on the key's path (it stops at the first cell that holds one key, or at depth J)
#pragma unroll
    for (int m = 0; m < kOctKPT; m++) {
      unsigned pos = 0;
      if (tid + kOctThreads * m < nk) {
        const int id3 = (int)code[m], id2 = id3 >> 2, id1 = id3 >> 4, r = id3 >> 6;
        if (scanA[r] == 1) pos = (unsigned)pos_d0(r);
        else if (J == 1 || h1[id1] == 1) pos = (unsigned)pos_d1(id1);
        else if (J == 2 || h2[id2] == 1) pos = (unsigned)pos_d2(id2);
        else pos = (unsigned)pos_d3(id3);
      }
      kn[m] = pos;
    }
    jumped = true; jump_finished = fin; jump_n = nn; jump_mode = md;
    OCT_T(26);
    __syncthreads();
    OCT_T(27);
  }
  // one thread per root: list position = number of non-empty roots before it (empty roots are dropped, :579-580)
  if (!jumped)
  for (int r = tid; r < nIni; r += kOctThreads) {
    const int c = scanA[r];
    int rank = 0;
    for (int q = 0; q < r; q++) rank += scanA[q] > 0;
    scanB[r] = c > 0 ? rank : -1;
    if (c > 0) {
      const unsigned x0 = (unsigned)(unsigned short)(short)(int)(hX * (float)r), x1 = (unsigned)(unsigned short)(short)(int)(hX * (float)(r + 1));
      node[0][rank] = make_uint4(x0 | (x1 << 16), 0u | ((unsigned)(unsigned short)(short)H << 16), (unsigned)c | ((unsigned)rank << 16), 0u);
      cc2[0][rank][0] = 0; cc2[0][rank][1] = 0;
    }
    if (r == nIni - 1) s_n = rank + (c > 0);
  }
  __syncthreads();
  if (!jumped) {
#pragma unroll
    for (int m = 0; m < kOctKPT; m++) kn[m] = tid + kOctThreads * m < nk ? (unsigned)scanB[kn[m]] : 0u;
  }
  int cur = 0, n = jumped ? jump_n : s_n, mode = jumped ? jump_mode : 1, par = 0, ws = 0;
  // loop invariant: node[cur][0..n) is the list, cc2[cur][0..n) is zero, kn[m] & 0xFFFF is the position of key m's node
  OCT_T(1);
  for (int pass = 0; pass < 64 && !jump_finished; pass++) {
    const int prev = n;
    const int nxt = cur ^ 1;
    OCT_T(4 + 2 * pass);
#ifdef OCT_PROFILE
    if (tid == 0 && 5 + 2 * pass < 30) g_oct_prof[blockIdx.x][5 + 2 * pass] = n * 4 + mode;
#endif
    // ---- keys: children key counts of every node with more than one key (:601-640 / :686-727 split the node's keys).
    // The records of four keys are requested before the first is used: the phases of this kernel are chains of dependent LDS
    // round trips (200-300 cycles each with four wavefronts doing the same), so independent reads are issued together.
#pragma unroll
    for (int m0 = 0; m0 < kOctKPT; m0 += 4) {
      if (kOctThreads * m0 < nk) {             // uniform
        uint4 rec4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) rec4[u] = node[cur][kn[m0 + u] & 0xFFFFu];      // inactive keys sit on position 0
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int m = m0 + u;
          if (tid + kOctThreads * m < nk) {
            const unsigned p = kn[m] & 0xFFFFu;
            const uint4 rec = rec4[u];
            if ((rec.z & 0xFFFFu) > 1u) {
              const int x0 = rec.x & 0xFFFF, x1 = rec.x >> 16, y0 = rec.y & 0xFFFF, y1 = rec.y >> 16;
              const int mx = x0 + ((x1 - x0 + 1) >> 1);             // UL.x + ceil((UR.x-UL.x)/2)
              const int my = y0 + ((y1 - y0 + 1) >> 1);
              const int kx = kw[m] & 0xFFF, ky = (kw[m] >> 12) & 0xFFF;
              const int q = (kx < mx ? 0 : 1) + (ky < my ? 0 : 2);
              atomicAdd(&cc2[cur][p][q >> 1], q & 1 ? 0x10000u : 1u);          // packed u16 pair add (counts < 65536)
              kn[m] = p | ((unsigned)q << 16) | (1u << 18);
            } else {
              kn[m] = p;
            }
          }
        }
      }
    }
    if (mode == 2) {
      // ranking keys of the expandable nodes: count << 16 | creation number (0 = not expandable), padded to a multiple of 4
      const int n4 = (n + 3) & ~3;
      for (int i = tid; i < n4; i += kOctThreads) {
        unsigned key = 0;
        if (i < n) {
          const unsigned cs = node[cur][i].z;
          if ((cs & 0xFFFFu) > 1u) key = ((cs & 0xFFFFu) << 16) | (cfg.oldest_first ? 0xFFFFu - (cs >> 16) : (cs >> 16));
        }
        scanB[i] = (int)key;
      }
    }
    if (tid == 0) { s_flag[par] = 0; s_cut = 0x7fffffff; s_m = 0; }
    if (pass == 2) OCT_T(30);
    if (pass == 4) OCT_T(36);
    __syncthreads();
    if (pass == 2) OCT_T(31);
    if (pass == 4) OCT_T(37);
    // contiguous chunk of the list for this thread
    const int per = (n + kOctThreads - 1) / kOctThreads;
    const int b = min(tid * per, n), e = min(b + per, n);
    int n_new, nexp;
    if (mode == 1) {
      // ---- phase 1: every node with more than one key splits, in list order.  One packed scan (expandable | children << 12)
      int sv = 0;
      for (int i = b; i < e; i++)
        if ((node[cur][i].z & 0xFFFFu) > 1u) sv += 1 | (nonzero4(cc2[cur][i][0], cc2[cur][i][1]) << 12);
      int tot;
      const int ex = oct_block_excl(sv, &tot, wsum[ws]); ws = (ws + 1) & 3;
      if (pass == 2) OCT_T(32);
      const int m1 = tot & 0xFFF, C1 = tot >> 12;
      if (m1 == 0) break;
      n_new = C1 + (n - m1);
      if (n_new > kOctListCap) { if (tid == 0) { *overflow = 1; *out_count = 0; } return; }
      int pe = ex & 0xFFF, ci = ex >> 12, my_exp = 0;         // expandable nodes / children before node i
      for (int i = b; i < e; i++) {
        const uint4 rec = node[cur][i];
        if ((rec.z & 0xFFFFu) > 1u) {
          const int x0 = rec.x & 0xFFFF, x1 = rec.x >> 16, y0 = rec.y & 0xFFFF, y1 = rec.y >> 16;
          const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
          const unsigned c01 = cc2[cur][i][0], c23 = cc2[cur][i][1];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const unsigned c = ((q & 2 ? c23 : c01) >> ((q & 1) * 16)) & 0xFFFFu;
            if (c == 0) continue;
            const int np = C1 - 1 - ci;                      // children end up reversed at the front
            node[nxt][np] = make_uint4((unsigned)((q & 1) ? mx : x0) | ((unsigned)((q & 1) ? x1 : mx) << 16),
                                       (unsigned)((q & 2) ? my : y0) | ((unsigned)((q & 2) ? y1 : my) << 16), c | ((unsigned)ci << 16), 0u);
            cc2[nxt][np][0] = 0; cc2[nxt][np][1] = 0;
            child_pos[i][q] = (unsigned short)np;
            my_exp += c > 1;
            ci++;
          }
          pe++;
        } else {
          const int np = C1 + (i - pe);
          node[nxt][np] = rec;
          cc2[nxt][np][0] = 0; cc2[nxt][np][1] = 0;
          new_pos[i] = (unsigned short)np;
        }
      }
      oct_wave_atomic_add(&s_flag[par], my_exp);
      if (pass == 2) OCT_T(33);
      __syncthreads();
      if (pass == 2) OCT_T(34);
      nexp = s_flag[par];
      // keys follow their node
#pragma unroll
      for (int m0 = 0; m0 < kOctKPT; m0 += 4) {
        if (kOctThreads * m0 < nk) {
          unsigned short cp[4], npv[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { const unsigned p = kn[m0 + u] & 0xFFFFu; cp[u] = child_pos[p][(kn[m0 + u] >> 16) & 3u]; npv[u] = new_pos[p]; }
#pragma unroll
          for (int u = 0; u < 4; u++) kn[m0 + u] = tid + kOctThreads * (m0 + u) < nk ? ((kn[m0 + u] >> 18) & 1u ? cp[u] : npv[u]) : 0u;
        }
      }
    } else {
      // ---- phase 2: processing order = descending (key count, creation number): rank by counting.  Every wavefront takes the
      // whole key list into registers (lane l: keys l, l+64, ...) and broadcasts it lane by lane (v_readlane): n compares per
      // node without a single further LDS access (n/4 128-bit LDS reads per node before: 7 k cycles at n = 174)
      if (n <= kOctThreads) {
        // one node per thread (the usual case: n <= N + 3, N <= 250 for up to ~1150 features per image)
        const int lane = tid & 63;
        const int nreg = (n + 63) >> 6;
        const unsigned mykey = tid < n ? (unsigned)scanB[tid] : 0u;
        int myrank = 0;
        for (int rg = 0; rg < nreg; rg++) {
          const int idx = rg * 64 + lane;
          const unsigned kreg = idx < n ? (unsigned)scanB[idx] : 0u;
#pragma unroll
          for (int l = 0; l < 64; l++) myrank += (unsigned)__builtin_amdgcn_readlane((int)kreg, l) > mykey;
        }
        if (mykey) { ord[tid] = (unsigned short)myrank; pos_of_ord[myrank] = (unsigned short)tid; }
        oct_wave_atomic_max(&s_m, mykey ? myrank + 1 : 0);
      } else {
        const int n4 = (n + 3) & ~3;
        int mloc = 0;
        for (int i = tid; i < n; i += kOctThreads) {
          const unsigned key = (unsigned)scanB[i];
          if (key) {
            int r = 0;
            const uint4* k4p = reinterpret_cast<const uint4*>(scanB);
            const int nq = n4 / 4;
            int j = 0;
            for (; j + 8 <= nq; j += 8) {                     // eight reads in flight (an LDS round trip each otherwise)
              uint4 kk[8];
#pragma unroll
              for (int u = 0; u < 8; u++) kk[u] = k4p[j + u];
#pragma unroll
              for (int u = 0; u < 8; u++) r += (kk[u].x > key) + (kk[u].y > key) + (kk[u].z > key) + (kk[u].w > key);
            }
            for (; j < nq; j++) {
              const uint4 kk = k4p[j];
              r += (kk.x > key) + (kk.y > key) + (kk.z > key) + (kk.w > key);
            }
            ord[i] = (unsigned short)r; pos_of_ord[r] = (unsigned short)i;
            mloc = max(mloc, r + 1);
          }
        }
        oct_wave_atomic_max(&s_m, mloc);
      }
      if (pass == 4) OCT_T(38);
      __syncthreads();
      if (pass == 4) OCT_T(39);
      const int m = s_m;
      if (m == 0) break;                                      // every node holds one key: size unchanged -> finished
      // children created before each processed node (processing order, chunks), and the cut of the reference's `break`
      const int pero = (m + kOctThreads - 1) / kOctThreads;
      const int bo = min(tid * pero, m), eo = min(bo + pero, m);
      int so = 0;
      unsigned k4pack = 0;                                    // the chunk's children counts, 3 bits each (chunks are <= 8 long)
      for (int o = bo; o < eo; o++) {
        const int p = pos_of_ord[o];
        const int k4 = nonzero4(cc2[cur][p][0], cc2[cur][p][1]);
        so += k4; k4pack |= (unsigned)k4 << (3 * (o - bo));
      }
      int totK;
      int before = oct_block_excl(so, &totK, wsum[ws]); ws = (ws + 1) & 3;
      (void)totK;
      int cutloc = 0x7fffffff;
      for (int o = bo; o < eo; o++) {
        const int k4 = (k4pack >> (3 * (o - bo))) & 7;
        scanA[o] = before;
        if (n + (before + k4) - (o + 1) >= N) cutloc = min(cutloc, o);
        before += k4;
      }
      oct_wave_atomic_min(&s_cut, cutloc);
      if (pass == 4) OCT_T(40);
      __syncthreads();
      const int cut = min(s_cut, m - 1);
      int C;
      {
        const int p = pos_of_ord[cut];
        C = scanA[cut] + nonzero4(cc2[cur][p][0], cc2[cur][p][1]);      // children created by nodes 0..cut
      }
      // survivors keep their order behind the children
      int ss = 0;
      for (int i = b; i < e; i++) ss += ((node[cur][i].z & 0xFFFFu) > 1u && ord[i] <= cut) ? 0 : 1;
      int nsurv;
      int sb = oct_block_excl(ss, &nsurv, wsum[ws]); ws = (ws + 1) & 3;
      if (pass == 4) OCT_T(41);
      n_new = C + nsurv;
      if (n_new > kOctListCap) { if (tid == 0) { *overflow = 1; *out_count = 0; } return; }
      int my_exp = 0;
      for (int i = b; i < e; i++) {
        const uint4 rec = node[cur][i];
        if ((rec.z & 0xFFFFu) > 1u && ord[i] <= cut) {
          const int x0 = rec.x & 0xFFFF, x1 = rec.x >> 16, y0 = rec.y & 0xFFFF, y1 = rec.y >> 16;
          const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
          const unsigned c01 = cc2[cur][i][0], c23 = cc2[cur][i][1];
          int ci = scanA[ord[i]];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const unsigned c = ((q & 2 ? c23 : c01) >> ((q & 1) * 16)) & 0xFFFFu;
            if (c == 0) continue;
            const int np = C - 1 - ci;
            node[nxt][np] = make_uint4((unsigned)((q & 1) ? mx : x0) | ((unsigned)((q & 1) ? x1 : mx) << 16),
                                       (unsigned)((q & 2) ? my : y0) | ((unsigned)((q & 2) ? y1 : my) << 16), c | ((unsigned)ci << 16), 0u);
            cc2[nxt][np][0] = 0; cc2[nxt][np][1] = 0;
            child_pos[i][q] = (unsigned short)np;
            my_exp += c > 1;
            ci++;
          }
        } else {
          const int np = C + sb;
          sb++;
          node[nxt][np] = rec;
          cc2[nxt][np][0] = 0; cc2[nxt][np][1] = 0;
          new_pos[i] = (unsigned short)np;
        }
      }
      oct_wave_atomic_add(&s_flag[par], my_exp);
      if (pass == 4) OCT_T(42);
      __syncthreads();
      nexp = s_flag[par];
#pragma unroll
      for (int m0 = 0; m0 < kOctKPT; m0 += 4) {
        if (kOctThreads * m0 < nk) {
          unsigned short cp[4], npv[4], od[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { const unsigned p = kn[m0 + u] & 0xFFFFu; cp[u] = child_pos[p][(kn[m0 + u] >> 16) & 3u]; npv[u] = new_pos[p]; od[u] = ord[p]; }
#pragma unroll
          for (int u = 0; u < 4; u++) kn[m0 + u] = tid + kOctThreads * (m0 + u) < nk ? (((kn[m0 + u] >> 18) & 1u) && (int)od[u] <= cut ? cp[u] : npv[u]) : 0u;
        }
      }
    }
    if (pass == 2) OCT_T(35);
    if (pass == 4) OCT_T(43);
    cur = nxt; n = n_new; par ^= 1;
    if (n >= N || n == prev) break;                           // :667 / :732
    if (mode == 1 && n + 3 * nexp > N) mode = 2;              // :671
  }
  OCT_T(2);
#ifdef OCT_PROFILE
  if (tid == 0) g_oct_prof[blockIdx.x][46] = nk * 65536 + n;
#endif
  // ---- retain the best key of every node, in list order (:742-758): largest response, among equals the first candidate
  if (n > cfg.reg_cap[level]) { if (tid == 0) { *overflow = 1; *out_count = 0; } return; }
  __syncthreads();                                            // a thread that left the loop early may still be read above
  for (int i = tid; i < n; i += kOctThreads) best[i] = 0;
  __syncthreads();
  unsigned kv[kOctKPT];
#pragma unroll
  for (int m = 0; m < kOctKPT; m++) {
    kv[m] = 0;
    if (tid + kOctThreads * m < nk) {
      kv[m] = ((kw[m] >> 24) << 16) | (unsigned)(0xFFFF - (tid + kOctThreads * m));
      atomicMax(&best[kn[m] & 0xFFFFu], kv[m]);
    }
  }
  __syncthreads();
  OctSel* out = sel_out + cfg.reg_off[cam][level];
#pragma unroll
  for (int m = 0; m < kOctKPT; m++) {
    if (tid + kOctThreads * m < nk && best[kn[m] & 0xFFFFu] == kv[m]) {
      OctSel o;
      o.x = (short)((int)(kw[m] & 0xFFF) + minB); o.y = (short)((int)((kw[m] >> 12) & 0xFFF) + minB); o.response = (float)(kw[m] >> 24);
      out[kn[m] & 0xFFFFu] = o;
    }
  }
  if (tid == 0) *out_count = n;
  OCT_T(3);
}
#ifdef OCT_PROFILE
extern "C" int orbx_debug_oct_prof(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_oct_prof), sizeof(g_oct_prof)) == hipSuccess ? 0 : -1; }
#endif

// ------------------------------------------------------------------------------------------------
// orientation + blur + rBRIEF, one wavefront per keypoint
// (IC_Angle S/ORBextractor.cc:75-102, GaussianBlur :1114-1115 / Appendix A-4, computeOrbDescriptor :106-145)

__device__ const signed char d_pattern[1024] = {
#include "orb_pattern_data.inc"
};

// cv::fastAtan2, Appendix A-5 (strict f32, no contraction)
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = __fdiv_rn(ay, ax + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = __fdiv_rn(ax, ay + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

struct UMax { int v[16]; int gauss[4]; };   // circle table of IC_Angle + the blur taps (orbx_config.gauss_taps)

constexpr int kPR = 21;              // patch radius: 18 (max rotated pattern reach) + 3 (blur)
constexpr int kPW = 2 * kPR + 1;     // 43
constexpr int kPS = 48;              // raw row stride: 12 dwords hold 43 bytes at any source alignment
constexpr int kBR = 18;
constexpr int kBW = 2 * kBR + 1;     // 37
constexpr int kHS = 38;              // row-pass stride (u16)
constexpr int kBS = 40;              // blurred stride
constexpr int kKpPerBlock = 4;

// One wavefront: orientation + blur + descriptor of the keypoint (cam, level, x, y) -> slot `out`.
// TAPS selects the blur kernel at compile time: 0 = {18,34,49,55} (OpenCV < 4.5, the reference's), 1 = {18,34,48,56}
// (OpenCV >= 4.5), 2 = the values of orbx_config.gauss_taps at run time.  With run-time taps the two blur passes lose the
// constant-multiplier forms and the kernel takes 27 instead of 20 us (bench.py census, pipelined step), so the two known
// variants are instantiated.
template <int TAPS>
__device__ __forceinline__ void orient_desc_wave(const uint8_t* __restrict__ pyr, const PyrGeom& g, int cam, int level, int kx_, int ky_,
                                                 float response, const UMax& um, size_t out, uint8_t* raw_al, unsigned short* hrow,
                                                 uint8_t* blur, orbx_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
                                                 orbx_keypoint* __restrict__ kps_host, uint8_t* __restrict__ desc_host) {
  const int lane = threadIdx.x & 63;
  const LevelGeom L = g.lv[level];
  const uint8_t* src = pyr + (size_t)cam * g.cam_stride + L.off + (size_t)(kEdge + ky_ - kPR) * L.stride + (kEdge + kx_ - kPR);
  // stage the 43x43 patch with aligned dword loads, all in flight before the first LDS store; the LDS copy keeps the
  // source alignment (column 0 at byte xs of each 48-byte row)
  const int xs = (kEdge + kx_ - kPR) & 3;                   // level offsets and strides are multiples of 64
  {
    const uint8_t* src_al = src - xs;
    constexpr int kDw = kPS / 4, kTot = kPW * kDw, kPer = (kTot + 63) / 64;
    uint32_t v[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int i = lane + 64 * k;
      if (i < kTot) v[k] = *reinterpret_cast<const uint32_t*>(src_al + (size_t)(i / kDw) * L.stride + 4 * (i % kDw));
    }
    uint32_t* r32 = reinterpret_cast<uint32_t*>(raw_al);
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int i = lane + 64 * k;
      if (i < kTot) r32[i] = v[k];
    }
  }
  const uint8_t* raw = raw_al + xs;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  // --- intensity centroid (exact integer sums -> order independent)
  int m01 = 0, m10 = 0;
  {
    const uint8_t* c = raw + kPR * kPS + kPR;
    for (int i = lane; i < 31 * 31; i += 64) {
      const int v = i / 31 - 15, u = i - (v + 15) * 31 - 15;
      const int av = v < 0 ? -v : v;
      if ((u < 0 ? -u : u) <= um.v[av]) {
        const int val = c[v * kPS + u];
        m10 += u * val;
        m01 += v * val;
      }
    }
    m01 = wave_sum(m01); m10 = wave_sum(m10);
  }
  const float angle = fast_atan2_deg((float)m01, (float)m10);
  // --- separable 7x7 Gaussian, Q8 taps (default {18,34,49,55,49,34,18}; orbx_config.gauss_taps): the row pass fits u16 (257*255 = 65535)
  const int t0 = TAPS == 2 ? um.gauss[0] : 18, t1 = TAPS == 2 ? um.gauss[1] : 34;
  const int t2 = TAPS == 2 ? um.gauss[2] : TAPS == 1 ? 48 : 49, t3 = TAPS == 2 ? um.gauss[3] : TAPS == 1 ? 56 : 55;
  for (int i = lane; i < kPW * kBW; i += 64) {
    const int y = i / kBW, x = i - y * kBW;
    const uint8_t* r = raw + y * kPS + x;       // x is already offset by -3 relative to the blurred column
    const int acc = t0 * (r[0] + r[6]) + t1 * (r[1] + r[5]) + t2 * (r[2] + r[4]) + t3 * r[3];
    hrow[y * kHS + x] = (unsigned short)acc;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  for (int i = lane; i < kBW * kBW; i += 64) {
    const int y = i / kBW, x = i - y * kBW;
    const unsigned short* r = hrow + y * kHS + x;
    const int acc = t0 * ((int)r[0] + r[6 * kHS]) + t1 * ((int)r[kHS] + r[5 * kHS]) + t2 * ((int)r[2 * kHS] + r[4 * kHS]) + t3 * (int)r[3 * kHS];
    const int v = (acc + 32768) >> 16;
    blur[y * kBS + x] = (uint8_t)min(v, 255);
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0);
  // --- steered BRIEF: lane l evaluates tests 4l..4l+3 (16 pattern bytes = one 128-bit load)
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  const float ang = angle * factorPI;
  const float a = (float)cos((double)ang), b = (float)sin((double)ang);
  const int4 praw = *reinterpret_cast<const int4*>(d_pattern + 16 * lane);
  const signed char* pp = reinterpret_cast<const signed char*>(&praw);
  const uint8_t* c = blur + kBR * kBS + kBR;
  unsigned nib = 0;
#pragma unroll
  for (int t = 0; t < 4; t++) {
    int val[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const float px = (float)pp[4 * t + 2 * e], py = (float)pp[4 * t + 2 * e + 1];
      const int r = (int)rintf(px * b + py * a);      // cvRound: half-to-even of the f32 value
      const int cc = (int)rintf(px * a - py * b);
      val[e] = c[r * kBS + cc];
    }
    nib |= (unsigned)(val[0] < val[1]) << t;
  }
  const unsigned hi = __shfl_down(nib, 1, 64);
  if ((lane & 1) == 0) {
    const uint8_t byte = (uint8_t)(nib | (hi << 4));
    desc[out * 32 + (lane >> 1)] = byte;
    if (desc_host) desc_host[out * 32 + (lane >> 1)] = byte;
  }
  if (lane == 0) {
    const float s = g.scale[level];
    orbx_keypoint o;
    o.x = level ? (float)kx_ * s : (float)kx_;        // keypoint->pt *= scale  (:1131-1133)
    o.y = level ? (float)ky_ * s : (float)ky_;
    o.size = (float)(int)((float)kPatch * s);         // scaledPatchSize (:862,871)
    o.angle = angle;
    o.response = response;
    o.octave = level;
    kps[out] = o;
    if (kps_host) kps_host[out] = o;
  }
}

// front-end 1: keypoints chosen by the host quad-trees (SelKp records in final order)
template <int TAPS>
__global__ __launch_bounds__(256) void orient_desc_kernel(const uint8_t* __restrict__ pyr, PyrGeom g,
                                                         const SelKp* __restrict__ sel, int n_sel, UMax um, int cam1_base,
                                                         orbx_keypoint* __restrict__ kps, uint8_t* __restrict__ desc) {
  __shared__ __attribute__((aligned(16))) uint8_t raw_s[kKpPerBlock][kPW * kPS];
  __shared__ unsigned short hrow_s[kKpPerBlock][kPW * kHS];
  __shared__ uint8_t blur_s[kKpPerBlock][kBW * kBS];
  const int wv = threadIdx.x >> 6;
  const int k = blockIdx.x * kKpPerBlock + wv;
  if (k >= n_sel) return;            // whole wavefront exits together; no block-wide barrier below
  const SelKp kp = sel[k];
  const size_t out = (size_t)(kp.cam ? cam1_base : 0) + kp.out_idx;
  orient_desc_wave<TAPS>(pyr, g, kp.cam, kp.level, kp.x, kp.y, kp.response, um, out, raw_s[wv], hrow_s[wv], blur_s[wv], kps, desc, nullptr, nullptr);
}

// front-end 2: keypoints chosen by octree_kernel.  Wavefront w of the (over-sized) grid finds its (camera, level,
// position) from the per-level counts; slot = keypoints before it in (camera, level) order -- or the mirror of it when
// the whole image lies in the lapping area (mono Frame ctor, S/Frame.cc:289).  Also publishes the keypoint totals.
template <int TAPS>
__global__ __launch_bounds__(256) void orient_desc_gpu_kernel(const uint8_t* __restrict__ pyr, PyrGeom g, OctCfg cfg,
                                                             const OctSel* __restrict__ sel, const int* __restrict__ lvl_count,
                                                             UMax um, int reverse0, int reverse1, orbx_keypoint* __restrict__ kps,
                                                             uint8_t* __restrict__ desc, orbx_keypoint* __restrict__ kps_host,
                                                             uint8_t* __restrict__ desc_host, int* __restrict__ d_nkp,
                                                             int* __restrict__ h_nkp, const int* __restrict__ d_overflow) {
  __shared__ __attribute__((aligned(16))) uint8_t raw_s[kKpPerBlock][kPW * kPS];
  __shared__ unsigned short hrow_s[kKpPerBlock][kPW * kHS];
  __shared__ uint8_t blur_s[kKpPerBlock][kBW * kBS];
  const int wv = threadIdx.x >> 6;
  const int w = blockIdx.x * kKpPerBlock + wv;
  int n0 = 0, n1 = 0;
  for (int l = 0; l < cfg.n_levels; l++) { n0 += lvl_count[l]; if (cfg.n_cams > 1) n1 += lvl_count[ORBG_MAX_LEVELS + l]; }
  const int ovf = *d_overflow;
  if (w == 0 && (threadIdx.x & 63) == 0) {
    d_nkp[0] = ovf ? 0 : n0; d_nkp[1] = ovf ? 0 : n1;     // zero counts keep the chained stereo / grid kernels idle on overflow
    h_nkp[0] = n0; h_nkp[1] = n1; h_nkp[2] = ovf;
  }
  if (ovf || w >= n0 + n1) return;                        // overflow: the host redoes the frame, nothing here is kept
  const int cam = w >= n0 ? 1 : 0;
  int s = cam ? w - n0 : w, level = 0;
  for (; level < cfg.n_levels; level++) {
    const int c = lvl_count[cam * ORBG_MAX_LEVELS + level];
    if (s < c) break;
    s -= c;
  }
  const OctSel kp = sel[cfg.reg_off[cam][level] + s];
  const int seq = cam ? w - n0 : w;                       // index in (level, list) order within the camera
  const int ncam = cam ? n1 : n0;
  const int slot = (cam ? reverse1 : reverse0) ? ncam - 1 - seq : seq;
  const size_t out = (size_t)(cam ? n0 : 0) + slot;
  orient_desc_wave<TAPS>(pyr, g, cam, level, kp.x, kp.y, kp.response, um, out, raw_s[wv], hrow_s[wv], blur_s[wv], kps, desc, kps_host, desc_host);
}

// ------------------------------------------------------------------------------------------------
// stereo matching on the device-resident features (Frame::ComputeStereoMatches, S/Frame.cc:785-963)

__device__ __forceinline__ int hamming256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1) {
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// One wavefront per left keypoint: lanes sweep all right keypoints (row band + octave + disparity gates,
// Hamming, wavefront min on the (dist, iR) key = "first minimum wins"), then the 11x11 SAD over 11 offsets.
// the wavefront's left keypoint iL (one wavefront per keypoint; no LDS, no barriers)
__device__ __forceinline__ void stereo_match_wave(const int iL, const uint8_t* __restrict__ pyr, const PyrGeom& g,
                                                  const orbx_keypoint* __restrict__ kl, const uint8_t* __restrict__ dl, int nl,
                                                  const orbx_keypoint* kr, const uint8_t* dr, int nr,
                                                  float bf, float b, float* __restrict__ uright, float* __restrict__ depth,
                                                  int* __restrict__ best_sad, const int* __restrict__ d_nkp) {
  const int lane = threadIdx.x & 63;
  if (d_nkp) {        // counts produced on the device (GPU quad-tree path): right camera starts behind the left one
    nl = d_nkp[0]; nr = d_nkp[1];
    kr = kl + nl; dr = dl + (size_t)nl * 32;
  }
  if (iL >= nl) return;
  const orbx_keypoint kpL = kl[iL];
  float out_u = -1.0f, out_d = -1.0f;
  int out_sad = -1;
  const int levelL = kpL.octave;
  const float vL = kpL.y, uL = kpL.x;
  const float minZ = b, minD = 0, maxD = bf / minZ;
  const float minU = uL - maxD, maxU = uL - minD;
  const int row = (int)vL;
  const int nRows = g.lv[0].h;
  unsigned bestKey = 0xFFFFFFFFu;
  if (!(maxU < 0) && row >= 0 && row < nRows) {
    const uint4 a0 = *reinterpret_cast<const uint4*>(dl + (size_t)iL * 32);
    const uint4 a1 = *reinterpret_cast<const uint4*>(dl + (size_t)iL * 32 + 16);
    // the right keypoint of the NEXT round is requested before this round's gates and descriptor (each a dependent trip to the L2)
    orbx_keypoint kpN = kr[min(lane, nr - 1)];
    for (int iR = lane; iR < nr; iR += 64) {
      const orbx_keypoint kpR = kpN;
      kpN = kr[min(iR + 64, nr - 1)];
      const float r = 2.0f * g.scale[kpR.octave];
      const int maxr = (int)ceilf(kpR.y + r), minr = (int)floorf(kpR.y - r);   // vRowIndices band (:806-811)
      if (row < minr || row > maxr) continue;
      if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
      if (!(kpR.x >= minU && kpR.x <= maxU)) continue;
      const uint4 b0 = *reinterpret_cast<const uint4*>(dr + (size_t)iR * 32);
      const uint4 b1 = *reinterpret_cast<const uint4*>(dr + (size_t)iR * 32 + 16);
      const unsigned key = ((unsigned)hamming256(a0, a1, b0, b1) << 16) | (unsigned)iR;
      bestKey = min(bestKey, key);
    }
  }
  bestKey = wave_min(bestKey);
  const int bestDist = bestKey == 0xFFFFFFFFu ? 100 : (int)(bestKey >> 16);
  // bestDist starts at TH_HIGH = 100 with a strict '<' (:841,862); accepted below (100+50)/2 = 75 (:790,871)
  if (bestDist < 75) {
    const int bestIdxR = (int)(bestKey & 0xFFFF);
    const float uR0 = kr[bestIdxR].x;
    const float scaleFactor = __fdiv_rn(1.0f, g.scale[levelL]);       // mvInvScaleFactors
    const float scaleduL = roundf(kpL.x * scaleFactor);
    const float scaledvL = roundf(kpL.y * scaleFactor);
    const float scaleduR0 = roundf(uR0 * scaleFactor);
    const LevelGeom L = g.lv[levelL];
    const float iniu = scaleduR0 + 5 - 5, endu = scaleduR0 + 5 + 5 + 1;
    if (!(iniu < 0 || endu >= (float)L.w)) {
      const uint8_t* IL = pyr + L.off + (size_t)kEdge * L.stride + kEdge;
      const uint8_t* IR = IL + g.cam_stride;
      const int cy = (int)scaledvL, cxL = (int)scaleduL, cxR0 = (int)scaleduR0;
      const int centreL = IL[(size_t)cy * L.stride + cxL];
      int sad[11];
#pragma unroll
      for (int o = 0; o < 11; o++) sad[o] = 0;
      for (int i = lane; i < 121; i += 64) {
        const int dy = i / 11 - 5, dx = i - (dy + 5) * 11 - 5;
        const int a = (int)IL[(size_t)(cy + dy) * L.stride + cxL + dx] - centreL;
        const uint8_t* rrow = IR + (size_t)(cy + dy) * L.stride;
#pragma unroll
        for (int o = 0; o < 11; o++) {
          const int cxR = cxR0 + o - 5;
          const int centreR = IR[(size_t)cy * L.stride + cxR];
          const int c = (int)rrow[cxR + dx] - centreR;
          sad[o] += abs(a - c);
        }
      }
#pragma unroll
      for (int o = 0; o < 11; o++) sad[o] = wave_sum(sad[o]);
      int bestS = 0x7FFFFFFF, bestinc = 0;
#pragma unroll
      for (int o = 0; o < 11; o++)
        if (sad[o] < bestS) { bestS = sad[o]; bestinc = o - 5; }
      if (!(bestinc == -5 || bestinc == 5)) {
        float d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
        for (int o = 1; o < 10; o++)
          if (o - 5 == bestinc) { d1 = (float)sad[o - 1]; d2 = (float)sad[o]; d3 = (float)sad[o + 1]; }
        const float deltaR = __fdiv_rn(d1 - d3, 2.0f * (d1 + d3 - 2.0f * d2));
        if (!(deltaR < -1 || deltaR > 1)) {
          float bestuR = g.scale[levelL] * ((float)scaleduR0 + (float)bestinc + deltaR);
          float disparity = uL - bestuR;
          if (disparity >= minD && disparity < maxD) {
            if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01); }
            out_d = __fdiv_rn(bf, disparity);
            out_u = bestuR;
            out_sad = bestS;
          }
        }
      }
    }
  }
  if (lane == 0) { uright[iL] = out_u; depth[iL] = out_d; best_sad[iL] = out_sad; }
}

__global__ __launch_bounds__(256) void stereo_match_kernel(const uint8_t* __restrict__ pyr, PyrGeom g,
                                                          const orbx_keypoint* __restrict__ kl, const uint8_t* __restrict__ dl, int nl,
                                                          const orbx_keypoint* kr, const uint8_t* dr, int nr,
                                                          float bf, float b, float* __restrict__ uright, float* __restrict__ depth,
                                                          int* __restrict__ best_sad, const int* __restrict__ d_nkp) {
  stereo_match_wave(blockIdx.x * 4 + (threadIdx.x >> 6), pyr, g, kl, dl, nl, kr, dr, nr, bf, b, uright, depth, best_sad, d_nkp);
}

// (stereo_finalize_body: stereo_finalize.hpp)
__global__ __launch_bounds__(256) void stereo_finalize_kernel(float* __restrict__ uright, float* __restrict__ depth,
                                                             const int* __restrict__ best_sad, int nl, const int* __restrict__ d_nkp,
                                                             float* __restrict__ host_out) {
  stereo_finalize_body(uright, depth, best_sad, nl, d_nkp, host_out);
}

// ------------------------------------------------------------------------------------------------
// host: quad-tree keypoint selection (DistributeOctTree, S/ORBextractor.cc:479-761)
//
// Nodes are axis-aligned boxes over a contiguous range of a key array (children = stable 4-way partition of
// the parent's range), linked in a list with the reference's push_front order.  Tie-break of the
// "largest node first" phase: (size, creation order), newest first -- the reference's order there depends
// on heap addresses, so it is pinned, identically to the oracle.

struct Cand { int x, y, score; };

class QuadTree {
 public:
  void run(const Cand* c, int n, int minX, int maxX, int minY, int maxY, int N, std::vector<int>& out) {
    out.clear();
    if (n == 0) return;
    c_ = c;
    nodes_.clear();
    keys_.resize(n);
    tmp_.resize(n);
    head_ = tail_ = -1;
    size_ = 0;
    int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    if (nIni < 1) nIni = 1;
    const float hX = static_cast<float>(maxX - minX) / nIni;
    // bucket candidates by root (stable)
    std::vector<int> cnt(nIni + 1, 0);
    root_of_.resize(n);
    for (int i = 0; i < n; i++) {
      int r = (int)((float)c[i].x / hX);
      if (r >= nIni) r = nIni - 1;
      root_of_[i] = r;
      cnt[r + 1]++;
    }
    for (int r = 0; r < nIni; r++) cnt[r + 1] += cnt[r];
    {
      std::vector<int> fill(cnt.begin(), cnt.end() - 1);
      for (int i = 0; i < n; i++) keys_[fill[root_of_[i]]++] = i;
    }
    for (int r = 0; r < nIni; r++) {
      Node nd;
      nd.x0 = (int)(hX * static_cast<float>(r)); nd.x1 = (int)(hX * static_cast<float>(r + 1));
      nd.y0 = 0; nd.y1 = maxY - minY;
      nd.b = cnt[r]; nd.e = cnt[r + 1];
      nd.no_more = (nd.e - nd.b) == 1;
      const int id = new_node(nd);
      push_back(id);
    }
    for (int it = head_; it >= 0;) {
      const int nx = nodes_[it].next;
      if (nodes_[it].e == nodes_[it].b) erase(it);
      it = nx;
    }
    bool finish = false;
    while (!finish) {
      const int prevSize = size_;
      int nToExpand = 0;
      work_.clear();
      for (int it = head_; it >= 0;) {
        if (nodes_[it].no_more) { it = nodes_[it].next; continue; }
        const int nx = nodes_[it].next;
        split(it, &nToExpand);
        erase(it);
        it = nx;
      }
      if (size_ >= N || size_ == prevSize) {
        finish = true;
      } else if (size_ + nToExpand * 3 > N) {
        while (!finish) {
          const int prev = size_;
          prev_work_.swap(work_);
          work_.clear();
          // (size, node id) ascending, walked from the back; id == creation order (oldest_first_: ids descending within a size)
          if (oldest_first_) std::sort(prev_work_.begin(), prev_work_.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first != b.first ? a.first < b.first : a.second > b.second; });
          else std::sort(prev_work_.begin(), prev_work_.end());
          for (int j = (int)prev_work_.size() - 1; j >= 0; j--) {
            const int id = prev_work_[j].second;
            split(id, nullptr);
            erase(id);
            if (size_ >= N) break;
          }
          if (size_ >= N || size_ == prev) finish = true;
        }
      }
    }
    out.reserve(size_);
    for (int it = head_; it >= 0; it = nodes_[it].next) {
      const Node& nd = nodes_[it];
      int best = keys_[nd.b];
      for (int k = nd.b + 1; k < nd.e; k++)
        if (c_[keys_[k]].score > c_[best].score) best = keys_[k];
      out.push_back(best);
    }
  }

 private:
  struct Node { int x0, y0, x1, y1, b, e, prev, next; bool no_more; };
  const Cand* c_ = nullptr;
  std::vector<Node> nodes_;
  std::vector<int> keys_, tmp_, root_of_;
  std::vector<std::pair<int, int>> work_, prev_work_;
 public:
  bool oldest_first_ = false;        // orbx_config.octree_oldest_first
 private:
  int head_ = -1, tail_ = -1, size_ = 0;

  int new_node(const Node& n) { nodes_.push_back(n); return (int)nodes_.size() - 1; }
  void push_front(int id) {
    nodes_[id].prev = -1; nodes_[id].next = head_;
    if (head_ >= 0) nodes_[head_].prev = id; else tail_ = id;
    head_ = id; size_++;
  }
  void push_back(int id) {
    nodes_[id].next = -1; nodes_[id].prev = tail_;
    if (tail_ >= 0) nodes_[tail_].next = id; else head_ = id;
    tail_ = id; size_++;
  }
  void erase(int id) {
    const int p = nodes_[id].prev, n = nodes_[id].next;
    if (p >= 0) nodes_[p].next = n; else head_ = n;
    if (n >= 0) nodes_[n].prev = p; else tail_ = p;
    size_--;
  }
  // ExtractorNode::DivideNode (:479-535) + the push_front of non-empty children (:619-658)
  void split(int id, int* nToExpand) {
    const Node p = nodes_[id];
    const int halfX = (int)std::ceil(static_cast<float>(p.x1 - p.x0) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(p.y1 - p.y0) / 2);
    const int mx = p.x0 + halfX, my = p.y0 + halfY;
    int cnt[4] = {0, 0, 0, 0};
    for (int k = p.b; k < p.e; k++) {
      const Cand& c = c_[keys_[k]];
      const int q = ((float)c.x < (float)mx ? 0 : 1) + ((float)c.y < (float)my ? 0 : 2);
      tmp_[k] = q;
      cnt[q]++;
    }
    int start[4] = {p.b, p.b + cnt[0], p.b + cnt[0] + cnt[1], p.b + cnt[0] + cnt[1] + cnt[2]};
    {
      int fill[4] = {start[0], start[1], start[2], start[3]};
      scratch_.resize(p.e - p.b);
      for (int k = p.b; k < p.e; k++) scratch_[fill[tmp_[k]]++ - p.b] = keys_[k];
      std::copy(scratch_.begin(), scratch_.begin() + (p.e - p.b), keys_.begin() + p.b);
    }
    const int bx[4][4] = {{p.x0, p.y0, mx, my}, {mx, p.y0, p.x1, my}, {p.x0, my, mx, p.y1}, {mx, my, p.x1, p.y1}};
    for (int q = 0; q < 4; q++) {
      if (!cnt[q]) continue;
      Node ch;
      ch.x0 = bx[q][0]; ch.y0 = bx[q][1]; ch.x1 = bx[q][2]; ch.y1 = bx[q][3];
      ch.b = start[q]; ch.e = start[q] + cnt[q];
      ch.no_more = cnt[q] == 1;
      const int cid = new_node(ch);
      push_front(cid);
      if (cnt[q] > 1) {
        if (nToExpand) (*nToExpand)++;
        work_.push_back({cnt[q], cid});
      }
    }
  }
  std::vector<int> scratch_;
};

// Small persistent worker pool for the per-(camera, level) quad-trees (16 independent serial problems per stereo
// frame).  The calling thread takes part, so pool size 0 degenerates to a plain loop.  prepare() wakes the workers
// ahead of time: they spin (bounded) until run() publishes the tasks, which hides the condition-variable wake-up
// latency behind the GPU phase that precedes the quad-trees.
class WorkerPool {
 public:
  explicit WorkerPool(int n) {
    for (int i = 0; i < n; i++) threads_.emplace_back([this, i] { worker(i); });
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  int size() const { return (int)threads_.size(); }
  void prepare() {
    if (threads_.empty()) return;
    { std::lock_guard<std::mutex> lk(m_); arm_id_.fetch_add(1, std::memory_order_release); }
    cv_.notify_all();
  }
  void run(int ntasks, const std::function<void(int, int)>& fn) {
    if (ntasks <= 0) return;
    {
      std::unique_lock<std::mutex> lk(m_);
      while (active_.load(std::memory_order_acquire) != 0) { lk.unlock(); std::this_thread::yield(); lk.lock(); }
      fn_ = &fn; ntasks_ = ntasks; next_.store(0); pending_.store(ntasks);
      run_id_.fetch_add(1, std::memory_order_release);
    }
    if (!threads_.empty()) cv_.notify_all();
    work((int)threads_.size());
    while (pending_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
  }

 private:
  void work(int wid) {
    for (;;) {
      const int t = next_.fetch_add(1);
      if (t >= ntasks_) break;
      (*fn_)(t, wid);
      pending_.fetch_sub(1, std::memory_order_release);
    }
  }
  void worker(int wid) {
    unsigned long long done = 0, seen_arm = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || run_id_.load() != done || arm_id_.load() != seen_arm; });
        if (stop_) return;
        seen_arm = arm_id_.load();
      }
      const auto t0 = std::chrono::steady_clock::now();
      while (run_id_.load(std::memory_order_acquire) == done) {       // armed: spin until the tasks are published
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
      }
      if (run_id_.load(std::memory_order_acquire) == done) continue;  // nothing came: sleep again (run() also notifies)
      {
        std::lock_guard<std::mutex> lk(m_);
        done = run_id_.load();
        active_.fetch_add(1);
      }
      work(wid);
      active_.fetch_sub(1, std::memory_order_release);
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_;
  const std::function<void(int, int)>* fn_ = nullptr;
  int ntasks_ = 0;
  std::atomic<int> next_{0}, pending_{0}, active_{0};
  std::atomic<unsigned long long> run_id_{0}, arm_id_{0};
  bool stop_ = false;
};

}  // namespace

// ------------------------------------------------------------------------------------------------
// handle

static inline double host_now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return 1e6 * (double)t.tv_sec + 1e-3 * (double)t.tv_nsec; }

struct orbx_handle {
  bool ext_stream = false;                    // `stream` was handed in through orbx_set_stream (never destroyed here)
  orbx_config cfg;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[12] = {};
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> feats_per_level;
  UMax umax;
  // geometry for the current image size
  int cur_w = 0, cur_h = 0;
  PyrGeom geom;
  std::vector<CellRec> cells;
  DevBuf<uint8_t> d_pyr, d_img;
  DevBuf<ResizeTap> d_xtab, d_ytab;
  DevBuf<TowerAxis> d_tower_x, d_tower_y;    // per level-0 tile ranges of pyr_tower_kernel
  int tower_T = 0, tower_ntx = 0, tower_nty = 0;   // tile side (0 = halo too large: one launch per level instead)
  DevBuf<CellRec> d_cells;
  DevBuf<uint32_t> d_slots;
  DevBuf<int> d_counts;
  PinnedBuf<int> hdr;            // [2*MAX_LEVELS] level starts, [2*MAX_LEVELS] total, +1,+2 camera ends
  PinnedBuf<uint32_t> cand;      // packed candidates
  PinnedBuf<SelKp> sel;
  DevBuf<orbx_keypoint> d_kps;   // [cam0 | cam1]
  DevBuf<uint8_t> d_desc;
  PinnedBuf<orbx_keypoint> h_kps;
  PinnedBuf<uint8_t> h_desc;
  DevBuf<float> d_uright, d_depth;
  DevBuf<orbx_keypoint> d_kps_un;   // mvKeysUn of the monocular constructor for a distorted camera (the frame views these)
  PinnedBuf<orbx_keypoint> h_kps_un;
  DevBuf<int> d_sad;
  PinnedBuf<float> h_stereo;
  int n_kp[2] = {0, 0};
  int cand_cap = 0;
  // per-level candidate views of the last extraction (for orbx_get_candidates)
  std::vector<Cand> last_cands[2][ORBG_MAX_LEVELS];
  std::vector<SelKp> level_sel[2][ORBG_MAX_LEVELS];
  std::vector<QuadTree> qts;               // one per pool thread + the caller
  std::unique_ptr<WorkerPool> pool;
  // GPU quad-tree path
  DevBuf<uint32_t> d_cand;
  DevBuf<int> d_hdr, d_lvlcount, d_nkp, d_overflow;
  DevBuf<OctSel> d_selreg;
  PinnedBuf<int> h_nkp;          // [0] n left, [1] n right, [2] overflow flag
  StreamSignal sig;
  OctCfg octcfg;
  int sel_bound = 0;             // upper bound of selected keypoints (sum of region capacities)
  bool gpu_octree = true;
  bool last_was_gpu = false;
  struct ExtractPending* pending = nullptr;    // orbx_frame_stereo_dev_submit .. _wait
  float timings[8] = {0};
  int profile = 1;   // 0: no events, 1: only the FAST kernel is bracketed (bench roofline), 2: every stage
  int profile_interval = 1;         // level-1 brackets on every k-th extraction only (an event pair costs ~5 us of stream time)
  unsigned long long extract_calls = 0;
  double fast_ms_sum = 0; long fast_ms_n = 0;   // accumulated bracket times of the bracketed kernel (fast_cells_kernel by default)
  int taps_variant = 0;             // 0 / 1: one of the two compiled-in blur kernels, 2: run-time taps (orient_desc_wave)
  int prof_kernel = 0;              // which kernel of the chain the level-1 event pair brackets: ORBX_PROF_*
  // host images (orbx_frame_stereo_submit / orbx_frame_stereo / orbx_extract*): one pinned staging slot per handle -- a handle
  // has one submission in flight, and the slot is free again when that submission has been waited for
  PinnedBuf<uint8_t> h_img;
  PinnedBuf<unsigned> up_ready;                // [0] "second image packed" word of img_upload_pair_kernel, [8] its error word
  unsigned up_seq = 0;
  // orbx_set_frame_outputs: host arrays the two-halves constructor delivers the left image's features into at _wait
  orbx_keypoint* out_kps = nullptr; uint8_t* out_desc = nullptr; float* out_uright = nullptr; float* out_depth = nullptr; int out_cap = 0;
  orbx_keypoint* out_kps_un = nullptr;         // orbx_set_frame_outputs_un: mvKeysUn of the monocular constructor
  std::atomic<int> ingest_state{0};            // 0 idle, 1 handed to the ingest thread, 2 submitted by it (ingest_rc valid)
  int ingest_rc = 0;
  // host-side timeline of the last submissions (orbx_get_ctor_timeline): per submission, microseconds
  //   [0] queue   hand-over to the ingest thread -> it starts (0 for synchronous submissions)
  //   [1] pack    rows copied into the pinned staging slot (both images)
  //   [2] enqueue the launches of the copy kernels + constructor chain (host time, pack excluded)
  //   [3] wait    time the collecting thread was blocked in the wait
  //   [4] latency hand-over -> constructor complete as seen by the collecting thread
  static constexpr int kTlCap = 512, kTlFields = 5;
  float tl[kTlCap][kTlFields] = {};
  unsigned long long tl_n = 0;                 // submissions recorded so far (ring index = tl_n % kTlCap)
  double tl_t_handover = 0, tl_pack_acc = 0, tl_queue = 0, tl_enqueue = 0;   // scratch of the submission in flight
  void tl_begin() { tl_t_handover = host_now_us(); tl_pack_acc = 0; tl_queue = 0; tl_enqueue = 0; }
  void tl_commit(double wait_us) {
    float* e = tl[tl_n % kTlCap];
    e[0] = (float)tl_queue; e[1] = (float)tl_pack_acc; e[2] = (float)tl_enqueue; e[3] = (float)wait_us; e[4] = (float)(host_now_us() - tl_t_handover);
    tl_n++;
  }
};
static void delete_pending(struct ExtractPending* p);   // (defined behind the type)

// Owned / needed pixel ranges of every level for the level-0 tiles of one axis (see pyr_tower_kernel).
// s0[l][d], s1[l][d]: the two source indices (level l-1) the taps of pixel d of level l read.  Returns the largest
// needed extent.
static int build_tower_axis(int T, int nl, const int* size, const std::vector<std::vector<int>>& s0,
                            const std::vector<std::vector<int>>& s1, std::vector<TowerAxis>& out) {
  const int nt = (size[0] + T - 1) / T;
  out.assign(nt, TowerAxis{});
  std::vector<int> owner_prev(size[0]), owner;
  for (int d = 0; d < size[0]; d++) owner_prev[d] = d / T;
  for (int t = 0; t < nt; t++) { out[t].own_lo[0] = (short)(t * T); out[t].own_hi[0] = (short)std::min((t + 1) * T, size[0]); }
  for (int l = 1; l < nl; l++) {
    owner.assign(size[l], 0);
    for (int t = 0; t < nt; t++) { out[t].own_lo[l] = 0; out[t].own_hi[l] = 0; }
    std::vector<char> seen(nt, 0);
    for (int d = 0; d < size[l]; d++) {
      const int t = owner_prev[s0[l][d]];
      owner[d] = t;
      if (!seen[t]) { seen[t] = 1; out[t].own_lo[l] = (short)d; }
      out[t].own_hi[l] = (short)(d + 1);
    }
    owner_prev.swap(owner);
  }
  int max_need = 0;
  for (int t = 0; t < nt; t++) {
    TowerAxis& A = out[t];
    A.need_lo[nl - 1] = A.own_lo[nl - 1]; A.need_hi[nl - 1] = A.own_hi[nl - 1];
    for (int l = nl - 1; l >= 1; l--) {
      int lo = A.own_lo[l - 1], hi = A.own_hi[l - 1];
      if (A.need_hi[l] > A.need_lo[l]) {
        const int a = s0[l][A.need_lo[l]], b = s1[l][A.need_hi[l] - 1] + 1;
        if (hi > lo) { lo = std::min(lo, a); hi = std::max(hi, b); } else { lo = a; hi = b; }
      }
      A.need_lo[l - 1] = (short)lo; A.need_hi[l - 1] = (short)hi;
    }
    for (int l = 0; l < nl; l++) max_need = std::max(max_need, A.need_hi[l] - A.need_lo[l]);
  }
  return max_need;
}

static int setup_geometry(orbx_handle* h, int w, int hgt) {
  if (w == h->cur_w && hgt == h->cur_h) return ORBG_OK;
  // the geometry is rebuilt in place: until the rebuild has succeeded the handle has NO valid geometry (a refused size -- a
  // pyramid level inside the border -- must not leave the tables of two image sizes mixed behind the old cur_w / cur_h)
  h->cur_w = h->cur_h = 0;
  const int nl = h->cfg.n_levels;
  PyrGeom& g = h->geom;
  g.n_levels = nl;
  int off = 0, xt = 0, yt = 0;
  h->cells.clear();
  std::vector<ResizeTap> xtab, ytab;
  for (int l = 0; l < nl; l++) {
    LevelGeom& L = g.lv[l];
    const float s = h->inv_scale[l];
    L.w = (int)std::nearbyint((double)((float)w * s));       // cvRound((float)cols*scale) :1157
    L.h = (int)std::nearbyint((double)((float)hgt * s));
    if (L.w <= kEdge || L.h <= kEdge) return ORBG_BAD_ARG;   // REFLECT_101 of a 19-px border needs more than 19 px
    L.stride = (L.w + 2 * kEdge + 63) & ~63;
    L.off = off;
    off += L.stride * (L.h + 2 * kEdge);
    off = (off + 255) & ~255;
    L.xt_off = xt; L.yt_off = yt;
    g.scale[l] = h->scale[l];
    if (l > 0) {
      const LevelGeom& S = g.lv[l - 1];
      // cv::resize coefficient tables (Appendix A-1)
      const double scale_x = 1. / ((double)L.w / S.w), scale_y = 1. / ((double)L.h / S.h);
      for (int dx = 0; dx < L.w; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)std::floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= S.w - 1) { fx = 0; sx = S.w - 1; }
        ResizeTap t;
        t.ofs = (short)sx;
        t.a0 = (short)std::nearbyintf((1.f - fx) * 2048);
        t.a1 = (short)std::nearbyintf(fx * 2048);
        t.pad = 0;
        xtab.push_back(t);
      }
      for (int dy = 0; dy < L.h; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)std::floor(fy);
        fy -= sy;
        ResizeTap t;
        t.ofs = (short)sy;
        t.a0 = (short)std::nearbyintf((1.f - fy) * 2048);
        t.a1 = (short)std::nearbyintf(fy * 2048);
        t.pad = 0;
        ytab.push_back(t);
      }
      xt += L.w; yt += L.h;
    }
    // FAST cells (:771-804)
    L.cell_begin = (int)h->cells.size();
    const int minB = kEdge - 3, maxBX = L.w - kEdge + 3, maxBY = L.h - kEdge + 3;
    const float fw = (float)(maxBX - minB), fh = (float)(maxBY - minB);
    const int nCols = (int)(fw / 30.f), nRows = (int)(fh / 30.f);
    if (nCols >= 1 && nRows >= 1) {
      const int wCell = (int)std::ceil(fw / nCols), hCell = (int)std::ceil(fh / nRows);
      for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minB + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBY - 3) continue;
        if (maxY > maxBY) maxY = (float)maxBY;
        for (int j = 0; j < nCols; j++) {
          const float iniX = (float)(minB + j * wCell);
          float maxX = iniX + wCell + 6;
          if (iniX >= maxBX - 6) continue;
          if (maxX > maxBX) maxX = (float)maxBX;
          CellRec c;
          c.level = (short)l;
          c.x0 = (short)(int)iniX; c.y0 = (short)(int)iniY;
          c.cw = (short)((int)maxX - (int)iniX); c.ch = (short)((int)maxY - (int)iniY);
          c.offx = (short)(j * wCell); c.offy = (short)(i * hCell);
          c.pad = 0;
          if (c.cw > kTile - 1 || c.ch > kTile - 1) return ORBG_INTERNAL;
          h->cells.push_back(c);
        }
      }
    }
    L.cell_end = (int)h->cells.size();
  }
  g.cam_stride = off;
  const int nc = h->cfg.n_cams;
  int rc;
  if ((rc = h->d_pyr.reserve((size_t)off * nc))) return rc;
  if ((rc = h->d_img.reserve((size_t)w * hgt * nc))) return rc;
  if ((rc = h->d_xtab.reserve(xtab.size() + 1))) return rc;
  if ((rc = h->d_ytab.reserve(ytab.size() + 1))) return rc;
  if ((rc = h->d_cells.reserve(h->cells.size() + 1))) return rc;
  if ((rc = h->d_slots.reserve(h->cells.size() * (size_t)nc * kCellCap))) return rc;
  if ((rc = h->d_counts.reserve(h->cells.size() * (size_t)nc))) return rc;
  h->cand_cap = (int)std::min<size_t>(h->cells.size() * (size_t)nc * 64 + 4096, (size_t)1 << 22);
  if ((rc = h->cand.reserve(h->cand_cap))) return rc;
  if ((rc = h->hdr.reserve(2 * ORBG_MAX_LEVELS + 4))) return rc;
  {
    OctCfg& oc = h->octcfg;
    oc.n_levels = nl; oc.n_cams = nc;
    oc.oldest_first = h->cfg.octree_oldest_first != 0;
    oc.jump = getenv("ORBG_OCT_NO_JUMP") ? 0 : 1;      // (read when the geometry is set up: per handle)
    int roff = 0;
    bool fits = true;
    for (int l = 0; l < nl; l++) {
      oc.n_target[l] = h->feats_per_level[l];
      oc.reg_cap[l] = std::min(4 * h->feats_per_level[l] + 8, kOctListCap);
      if (4 * h->feats_per_level[l] + 8 > kOctListCap) fits = false;
    }
    for (int c = 0; c < 2; c++)
      for (int l = 0; l < nl; l++) { oc.reg_off[c][l] = roff; if (c < nc) roff += oc.reg_cap[l]; }
    h->sel_bound = roff;
    if (!fits) h->gpu_octree = false;          // per-level quota too large for the LDS-resident quad-tree
    if ((rc = h->d_cand.reserve(h->cand_cap)) || (rc = h->d_hdr.reserve(2 * ORBG_MAX_LEVELS + 4)) ||
        (rc = h->d_lvlcount.reserve(2 * ORBG_MAX_LEVELS)) || (rc = h->d_nkp.reserve(4)) || (rc = h->d_overflow.reserve(4)) ||
        (rc = h->d_selreg.reserve(std::max(roff, 1))) || (rc = h->h_nkp.reserve(4)))
      return rc;
    // (on the handle's own stream, in front of everything that uses them: the null stream's hipMemset may return before the fill has
    // run, and the library's non-blocking streams do not join the null stream)
    ORBG_HIP(hipMemsetAsync(h->d_lvlcount.p, 0, 2 * ORBG_MAX_LEVELS * sizeof(int), h->stream));
    ORBG_HIP(hipMemsetAsync(h->d_overflow.p, 0, 4 * sizeof(int), h->stream));   // [0] overflow flag, [2] ticket of the constructor's last launch
    // output buffers must hold the worst case of the device-side selection
    const size_t need = (size_t)roff + 64;
    if ((rc = h->d_kps.reserve(need)) || (rc = h->d_desc.reserve(need * 32)) || (rc = h->h_kps.reserve(need)) ||
        (rc = h->h_desc.reserve(need * 32)) || (rc = h->d_uright.reserve(need)) || (rc = h->d_depth.reserve(need)) ||
        (rc = h->d_sad.reserve(need)) || (rc = h->h_stereo.reserve(2 * need)))
      return rc;
  }
  // one-launch pyramid: largest level-0 tile whose halo still fits the LDS blocks of pyr_tower_kernel
  h->tower_T = 0;
  if (!getenv("ORBG_NO_TOWER")) {
    std::vector<std::vector<int>> sx0(nl), sx1(nl), sy0(nl), sy1(nl);
    int ws[ORBG_MAX_LEVELS], hs[ORBG_MAX_LEVELS];
    for (int l = 0; l < nl; l++) { ws[l] = g.lv[l].w; hs[l] = g.lv[l].h; }
    for (int l = 1; l < nl; l++) {
      const LevelGeom& D = g.lv[l]; const LevelGeom& S = g.lv[l - 1];
      sx0[l].resize(D.w); sx1[l].resize(D.w); sy0[l].resize(D.h); sy1[l].resize(D.h);
      for (int d = 0; d < D.w; d++) { const int o = xtab[D.xt_off + d].ofs; sx0[l][d] = o; sx1[l][d] = std::min(o + 1, S.w - 1); }
      for (int d = 0; d < D.h; d++) {
        const int o = ytab[D.yt_off + d].ofs;
        sy0[l][d] = std::min(std::max(o, 0), S.h - 1); sy1[l][d] = std::min(std::max(o + 1, 0), S.h - 1);
      }
    }
    for (int T = 56; T >= 16; T -= 4) {
      std::vector<TowerAxis> ax, ay;
      if (build_tower_axis(T, nl, ws, sx0, sx1, ax) > kTwSide || build_tower_axis(T, nl, hs, sy0, sy1, ay) > kTwSide) continue;
      if ((rc = h->d_tower_x.reserve(ax.size())) || (rc = h->d_tower_y.reserve(ay.size()))) return rc;
      ORBG_HIP(hipMemcpy(h->d_tower_x.p, ax.data(), ax.size() * sizeof(TowerAxis), hipMemcpyHostToDevice));
      ORBG_HIP(hipMemcpy(h->d_tower_y.p, ay.data(), ay.size() * sizeof(TowerAxis), hipMemcpyHostToDevice));
      h->tower_T = T; h->tower_ntx = (int)ax.size(); h->tower_nty = (int)ay.size();
      break;
    }
  }
  if (!xtab.empty()) ORBG_HIP(hipMemcpy(h->d_xtab.p, xtab.data(), xtab.size() * sizeof(ResizeTap), hipMemcpyHostToDevice));
  if (!ytab.empty()) ORBG_HIP(hipMemcpy(h->d_ytab.p, ytab.data(), ytab.size() * sizeof(ResizeTap), hipMemcpyHostToDevice));
  if (!h->cells.empty()) ORBG_HIP(hipMemcpy(h->d_cells.p, h->cells.data(), h->cells.size() * sizeof(CellRec), hipMemcpyHostToDevice));
  h->cur_w = w; h->cur_h = hgt;
  return ORBG_OK;
}

extern "C" int orbx_create(const orbx_config* cfg, orbx_handle** out) {
  if (!cfg || !out) return ORBG_BAD_ARG;
  if (cfg->n_levels < 1 || cfg->n_levels > ORBG_MAX_LEVELS || cfg->n_features < 1 || cfg->n_cams < 1 || cfg->n_cams > 2 ||
      !(cfg->scale_factor > 1.0f) || cfg->max_width > 4000 || cfg->max_height > 4000 || cfg->n_features > 3500)
    return ORBG_BAD_ARG;
  int rc = select_device(cfg->device);
  if (rc) return rc;
  orbx_handle* h = new orbx_handle();
  h->cfg = *cfg;
  {
    // blur taps: all zero selects the default kernel; the row pass of the descriptor kernel is a u16 (sum * 255 <= 65535)
    int* t = h->cfg.gauss_taps;
    if (t[0] == 0 && t[1] == 0 && t[2] == 0 && t[3] == 0) { t[0] = 18; t[1] = 34; t[2] = 49; t[3] = 55; }
    if (t[0] < 0 || t[1] < 0 || t[2] < 0 || t[3] < 1 || 2 * (t[0] + t[1] + t[2]) + t[3] > 257) { delete h; return ORBG_BAD_ARG; }
  }
  h->device = cfg->device;
  const int nl = cfg->n_levels;
  // ORBextractor::ORBextractor (S/ORBextractor.cc:413-468)
  h->scale.resize(nl); h->inv_scale.resize(nl); h->sigma2.resize(nl); h->inv_sigma2.resize(nl);
  h->scale[0] = 1.0f; h->sigma2[0] = 1.0f;
  for (int i = 1; i < nl; i++) { h->scale[i] = h->scale[i - 1] * cfg->scale_factor; h->sigma2[i] = h->scale[i] * h->scale[i]; }
  for (int i = 0; i < nl; i++) { h->inv_scale[i] = 1.0f / h->scale[i]; h->inv_sigma2[i] = 1.0f / h->sigma2[i]; }
  h->feats_per_level.resize(nl);
  const float factor = 1.0f / cfg->scale_factor;
  float desired = cfg->n_features * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
  int sum = 0;
  for (int l = 0; l < nl - 1; l++) {
    h->feats_per_level[l] = (int)std::nearbyint((double)desired);
    sum += h->feats_per_level[l];
    desired *= factor;
  }
  h->feats_per_level[nl - 1] = std::max(cfg->n_features - sum, 0);
  {
    int um[kHalfPatch + 2] = {0};
    int v, v0;
    const int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    const int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (v = 0; v <= vmax; ++v) um[v] = (int)std::nearbyint(std::sqrt(hp2 - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
      while (um[v0] == um[v0 + 1]) ++v0;
      um[v] = v0;
      ++v0;
    }
    for (int i = 0; i < 16; i++) h->umax.v[i] = um[i];
    for (int i = 0; i < 4; i++) h->umax.gauss[i] = h->cfg.gauss_taps[i];
    const int* gt = h->cfg.gauss_taps;
    h->taps_variant = (gt[0] == 18 && gt[1] == 34 && gt[2] == 49 && gt[3] == 55) ? 0 : (gt[0] == 18 && gt[1] == 34 && gt[2] == 48 && gt[3] == 56) ? 1 : 2;
  }
  {
    const int nthreads = 5;                        // host quad-tree workers (the fallback path; they sleep unless it is taken)
    if (const char* env = getenv("ORBG_HOST_OCTREE")) h->gpu_octree = atoi(env) == 0;
    h->pool.reset(new WorkerPool(nthreads));
    h->qts.resize(nthreads + 1);
    for (auto& q : h->qts) q.oldest_first_ = cfg->octree_oldest_first != 0;
  }
  // (ORBG_CTOR_GRAPH=1 captures the constructor chain on the handle's stream: that stream must not be shared with other handles)
  if (orbg::create_stream(&h->stream, "ex") != hipSuccess) { delete h; return ORBG_HIP_ERROR; }
  for (auto& e : h->ev)
    if (hipEventCreate(&e) != hipSuccess) { delete h; return ORBG_HIP_ERROR; }
  const int cap = 2 * (cfg->n_features + 4 * nl + 64);
  if ((rc = h->sel.reserve(cap)) || (rc = h->d_kps.reserve(cap)) || (rc = h->d_desc.reserve((size_t)cap * 32)) ||
      (rc = h->h_kps.reserve(cap)) || (rc = h->h_desc.reserve((size_t)cap * 32)) || (rc = h->d_uright.reserve(cap)) ||
      (rc = h->d_depth.reserve(cap)) || (rc = h->d_sad.reserve(cap)) || (rc = h->h_stereo.reserve(2 * (size_t)cap))) {
    delete h;
    return rc;
  }
  if (cfg->max_width > 0 && cfg->max_height > 0) {
    rc = setup_geometry(h, cfg->max_width, cfg->max_height);
    if (rc) { delete h; return rc; }
  }
  *out = h;
  return ORBG_OK;
}



extern "C" int orbx_destroy(orbx_handle* h) {
  if (!h) return ORBG_BAD_ARG;
  (void)hipSetDevice(h->device);
  while (h->ingest_state.load(std::memory_order_acquire) == 1) std::this_thread::yield();   // an asynchronous submission is being enqueued
  (void)hipStreamSynchronize(h->stream);
  h->d_pyr.release(); h->d_img.release(); h->h_img.release(); h->up_ready.release(); h->d_xtab.release(); h->d_ytab.release(); h->d_tower_x.release(); h->d_tower_y.release(); h->d_cells.release();
  h->d_slots.release(); h->d_counts.release(); h->hdr.release(); h->cand.release(); h->sel.release();
  h->d_kps.release(); h->d_desc.release(); h->h_kps.release(); h->h_desc.release();
  h->d_uright.release(); h->d_depth.release(); h->d_sad.release(); h->h_stereo.release();
  h->d_cand.release(); h->d_hdr.release(); h->d_lvlcount.release(); h->d_nkp.release(); h->d_overflow.release();
  h->d_selreg.release(); h->h_nkp.release(); h->sig.release();
  for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
  if (!h->ext_stream) orbg::release_stream(h->stream);
  delete_pending(h->pending);
  delete h;
  return ORBG_OK;
}

extern "C" int orbx_get_tables(const orbx_handle* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                               int32_t* fpl) {
  if (!h) return ORBG_BAD_ARG;
  for (int i = 0; i < h->cfg.n_levels; i++) {
    if (scale) scale[i] = h->scale[i];
    if (inv_scale) inv_scale[i] = h->inv_scale[i];
    if (sigma2) sigma2[i] = h->sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = h->inv_sigma2[i];
    if (fpl) fpl[i] = h->feats_per_level[i];
  }
  return ORBG_OK;
}

// Work chained behind the descriptor kernel on the same stream, before the single final synchronisation:
// the rest of the stereo Frame constructor (S/Frame.cc:117 ComputeStereoMatches, :160 AssignFeaturesToGrid).
struct PostOps {
  bool stereo = false;
  float bf = 0, b = 0;
  float* uright = nullptr;
  float* depth = nullptr;
  orbm_frame* frame = nullptr;
  const orbm_frame_view* view = nullptr;
  // monocular constructor (S/Frame.cc:260-358): no stereo data; undist: Frame::UndistortKeyPoints on the device (:721-754)
  bool mono = false, undist = false;
  orbx_distortion dist{};
  orbx_keypoint* kps_un_out = nullptr;       // host copy of mvKeysUn (cap[0] entries), may be NULL
};
// Everything the second half of a GPU-path extraction needs (it runs either right away or in orbx_frame_stereo_dev_wait)
struct ExtractPending {
  bool active = false;           // submitted, not yet waited for
  bool finished = false;         // the submission ran synchronously (host quad-trees): results are already in n_res
  unsigned cams_mask = 0;
  const uint8_t* d_img0 = nullptr; const uint8_t* d_img1 = nullptr;
  int w = 0, hgt = 0, stride = 0, ncams = 0, prof = 0;
  int lap[2][2] = {{0, 0}, {0, 0}};
  orbx_keypoint* kps_out[2] = {nullptr, nullptr};
  uint8_t* desc_out[2] = {nullptr, nullptr};
  int cap[2] = {0, 0};
  int* n_out[2] = {nullptr, nullptr};
  int* n_mono_out[2] = {nullptr, nullptr};
  bool has_post = false;
  PostOps post;
  orbm_frame_view view_copy;     // the caller's view may be gone by the time of the wait
  int reverse[2] = {0, 0};
  bool stereo_out = false;
  int n_res[2] = {0, 0};
};
static void delete_pending(ExtractPending* p) { delete p; }
int orbm_internal_attach(orbm_frame* f, orbx_handle* h, const orbm_frame_view* v, int n, hipStream_t stream, const int* d_n,
                         volatile unsigned* done_flag, unsigned done_seq, const StereoFinalizeArgs* fin, const orbg::UndistortArgs* un = nullptr,
                         bool mono = false);
// arguments of the undistortion in front of the grid build (the handle's mvKeysUn buffers are sized here)
static int make_undistort_args(orbx_handle* h, const PostOps* post, orbg::UndistortArgs* ua) {
  memset(ua, 0, sizeof(*ua));
  if (!post || !post->undist) return ORBG_OK;
  const size_t cap = h->d_kps.cap;
  int rc;
  if ((rc = h->d_kps_un.reserve(cap)) || (rc = h->h_kps_un.reserve(cap))) return rc;
  ua->on = 1;
  ua->fx = post->view->fx; ua->fy = post->view->fy; ua->cx = post->view->cx; ua->cy = post->view->cy;
  ua->k1 = post->dist.k1; ua->k2 = post->dist.k2; ua->p1 = post->dist.p1; ua->p2 = post->dist.p2; ua->k3 = post->dist.k3;
  ua->src = h->d_kps.p; ua->dst = h->d_kps_un.p; ua->dst_host = h->h_kps_un.d;
  return ORBG_OK;
}
void orbm_internal_set_n(orbm_frame* f, int n);
int orbx_internal_kp_capacity(orbx_handle* h);
static int launch_stereo(orbx_handle* h, float bf, float b, hipStream_t st, bool device_counts, float* host_mirror,
                         StereoFinalizeArgs* defer_finalize = nullptr);

// Core: cams_mask selects which cameras of the rig are processed; d_img are device pointers.
static int extract_core(orbx_handle* h, unsigned cams_mask, const uint8_t* d_img0, const uint8_t* d_img1, int w, int hgt,
                        int stride, const int lap[2][2], orbx_keypoint* kps_out[2], uint8_t* desc_out[2], const int cap[2],
                        int* n_out[2], int* n_mono_out[2], const PostOps* post = nullptr, bool force_host = false,
                        bool submit_only = false);
static int extract_finish_gpu(orbx_handle* h, ExtractPending& c);

static int extract_core(orbx_handle* h, unsigned cams_mask, const uint8_t* d_img0, const uint8_t* d_img1, int w, int hgt,
                        int stride, const int lap[2][2], orbx_keypoint* kps_out[2], uint8_t* desc_out[2], const int cap[2],
                        int* n_out[2], int* n_mono_out[2], const PostOps* post, bool force_host, bool submit_only) {
  // a submitted Frame constructor owns the handle (stream, pyramid, feature buffers) until it has been waited for
  if (h->pending && h->pending->active) return ORBG_BAD_ARG;
  int rc = setup_geometry(h, w, hgt);
  if (rc) return rc;
  const PyrGeom& g = h->geom;
  const int nl = g.n_levels;
  const int ncams = (cams_mask & 2) ? 2 : 1;       // cameras [0, ncams) are launched; mask 2 alone is not supported
  if (cams_mask == 2) return ORBG_BAD_ARG;
  const int n_cells = (int)h->cells.size();
  hipStream_t st = h->stream;
  int prof = h->profile;
  if (prof == 1 && h->profile_interval > 1 && (h->extract_calls % (unsigned)h->profile_interval) != 0) prof = 0;
  h->extract_calls++;
  // The quad-trees run on the GPU unless the lapping area splits an image (only the fisheye-stereo path does that):
  // reverse = whole image inside [lap0, lap1] (mono Frame ctor), plain = nothing inside.
  bool use_gpu = h->gpu_octree && n_cells > 0 && !force_host;
  int reverse[2] = {0, 0};
  for (int c = 0; c < ncams && use_gpu; c++) {
    const float xmin = (float)kEdge, xmax = (float)w;          // keypoint x range in level-0 pixels: [19, w)
    if ((float)lap[c][1] < xmin || (float)lap[c][0] > xmax) reverse[c] = 0;
    else if ((float)lap[c][0] <= xmin && (float)lap[c][1] >= xmax) reverse[c] = 1;
    else use_gpu = false;
  }
  h->last_was_gpu = use_gpu;
  if (!use_gpu) h->pool->prepare();              // wake the host quad-tree workers only when they will be used
  const bool want_desc = desc_out[0] || desc_out[1];
  const bool do_stereo = use_gpu && post && post->stereo && ncams == 2;
  const bool stereo_out = do_stereo && (post->uright || post->depth);
  OctCfg oc = h->octcfg;
  oc.n_cams = ncams;                          // cameras processed by THIS call (a rig handle may extract one image)
  const uint8_t* const img1 = d_img1 ? d_img1 : d_img0;
  // every launch of the chain up to (not including) the grid build; prof_ > 0 adds the event brackets
  auto launch_chain = [&](int prof_) -> int {
    if (prof_ >= 2) ORBG_HIP(hipEventRecord(h->ev[0], st));
    const int pk = h->prof_kernel;
    auto br0 = [&](int which) -> int { if (prof_ >= 1 && pk == which) ORBG_HIP(hipEventRecord(h->ev[1], st)); return ORBG_OK; };
    auto br1 = [&](int which) -> int { if (prof_ >= 1 && pk == which) ORBG_HIP(hipEventRecord(h->ev[7], st)); return ORBG_OK; };
    if (br0(ORBX_PROF_PYRAMID)) return ORBG_HIP_ERROR;
    if (h->tower_T > 0) {
      hipLaunchKernelGGL(pyr_tower_kernel, dim3(h->tower_ntx, h->tower_nty, ncams), dim3(kTwThreads), 0, st, d_img0, img1,
                         stride, h->d_pyr.p, g, h->d_xtab.p, h->d_ytab.p, h->d_tower_x.p, h->d_tower_y.p);
    } else {
      const LevelGeom& L0 = g.lv[0];
      dim3 grid((L0.w + 2 * kEdge + 63) / 64, (L0.h + 2 * kEdge + 3) / 4, ncams);
      hipLaunchKernelGGL(pyr_level0_kernel, grid, dim3(256), 0, st, d_img0, img1, stride, h->d_pyr.p, g);
      for (int l = 1; l < nl; l++) {
        const LevelGeom& L = g.lv[l];
        dim3 gr((L.w + 2 * kEdge + 63) / 64, (L.h + 2 * kEdge + 3) / 4, ncams);
        hipLaunchKernelGGL(pyr_resize_kernel, gr, dim3(256), 0, st, h->d_pyr.p, g, l, h->d_xtab.p, h->d_ytab.p);
      }
    }
    if (br1(ORBX_PROF_PYRAMID)) return ORBG_HIP_ERROR;
    if (prof_ >= 2 && pk != ORBX_PROF_FAST) ORBG_HIP(hipEventRecord(h->ev[8], st));      // end of the pyramid for the stage timings
    if (n_cells > 0) {
      if (br0(ORBX_PROF_FAST)) return ORBG_HIP_ERROR;
      hipLaunchKernelGGL(fast_cells_kernel, dim3(8 * ((n_cells + 7) / 8), ncams), dim3(256), 0, st, h->d_pyr.p, g, h->d_cells.p, n_cells,
                         std::max(h->cfg.ini_th_fast, 1), std::max(h->cfg.min_th_fast, 1), h->d_slots.p, h->d_counts.p);
      if (br1(ORBX_PROF_FAST)) return ORBG_HIP_ERROR;
      // the GPU quad-trees take the candidates from the per-cell slots themselves (octree_kernel): the compacted list is only
      // built for the host quad-trees (and, on demand, for orbx_get_candidates)
      if (!use_gpu)
        hipLaunchKernelGGL(gather_cells_kernel, dim3(n_cells, ncams), dim3(256), 0, st, h->d_slots.p, h->d_counts.p, g,
                           h->d_cells.p, n_cells, ncams, use_gpu ? h->d_hdr.p : h->hdr.d, use_gpu ? h->d_cand.p : h->cand.d, h->cand_cap);
    }
    if (use_gpu) {
      // ---- everything stays on the device: quad-trees -> descriptors -> (stereo, grid) -> ONE synchronisation
      if (br0(ORBX_PROF_OCTREE)) return ORBG_HIP_ERROR;
      hipLaunchKernelGGL(octree_kernel, dim3(ncams * nl), dim3(kOctThreads), 0, st, h->d_cand.p, h->d_hdr.p, g, oc,
                         h->d_selreg.p, h->d_lvlcount.p, h->d_overflow.p, h->cand_cap, h->d_slots.p, h->d_counts.p, n_cells);
      if (br1(ORBX_PROF_OCTREE)) return ORBG_HIP_ERROR;
      if (br0(ORBX_PROF_ORIENT_DESC)) return ORBG_HIP_ERROR;
      auto launch_od = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3((h->sel_bound + kKpPerBlock - 1) / kKpPerBlock), dim3(256), 0, st, h->d_pyr.p, g,
                           oc, h->d_selreg.p, h->d_lvlcount.p, h->umax, reverse[0], reverse[1], h->d_kps.p, h->d_desc.p, h->h_kps.d,
                           want_desc ? h->h_desc.d : (uint8_t*)nullptr, h->d_nkp.p, h->h_nkp.d, h->d_overflow.p);
      };
      if (h->taps_variant == 0) launch_od(orient_desc_gpu_kernel<0>);
      else if (h->taps_variant == 1) launch_od(orient_desc_gpu_kernel<1>);
      else launch_od(orient_desc_gpu_kernel<2>);
      if (br1(ORBX_PROF_ORIENT_DESC)) return ORBG_HIP_ERROR;
      if (do_stereo) {
        // with a device frame to build, the median rejection runs as the second workgroup of the grid build (one launch less)
        StereoFinalizeArgs unused;
        const int rcs = launch_stereo(h, post->bf, post->b, st, true, stereo_out ? h->h_stereo.d : nullptr,
                                      post->frame ? &unused : nullptr);
        if (rcs) return rcs;
      }
    }
    return ORBG_OK;
  };
  if ((rc = launch_chain(prof))) return rc;
  if (use_gpu) {
    bool posted = false;
    if (post) {
      if (post->frame) {
        // the grid build is the last kernel of the chain: it posts the completion word itself (no signal kernel)
        unsigned seq; volatile unsigned* flag;
        if ((rc = h->sig.arm(&seq, &flag))) return rc;
        StereoFinalizeArgs fin{};
        const bool with_fin = do_stereo && h->sel_bound > 0;
        if (with_fin)
          fin = StereoFinalizeArgs{h->d_uright.p, h->d_depth.p, h->d_sad.p, h->sel_bound, h->d_nkp.p, stereo_out ? h->h_stereo.d : nullptr,
                                   reinterpret_cast<unsigned*>(h->d_overflow.p + 2)};
        orbg::UndistortArgs ua;
        if ((rc = make_undistort_args(h, post, &ua))) return rc;
        if ((rc = orbm_internal_attach(post->frame, h, post->view, -1, st, h->d_nkp.p, flag, seq, with_fin ? &fin : nullptr, &ua, post->mono))) return rc;
        posted = true;
      }
    }
    ORBG_HIP(hipGetLastError());
    if (!posted && (rc = h->sig.post(st))) return rc;          // completion word in pinned memory
    ExtractPending local;
    if (submit_only && !h->pending) h->pending = new ExtractPending();
    ExtractPending& c = submit_only ? *h->pending : local;
    c.active = submit_only; c.finished = false;
    c.cams_mask = cams_mask; c.d_img0 = d_img0; c.d_img1 = d_img1; c.w = w; c.hgt = hgt; c.stride = stride; c.ncams = ncams; c.prof = prof;
    for (int a = 0; a < 2; a++) {
      c.lap[a][0] = lap[a][0]; c.lap[a][1] = lap[a][1];
      c.kps_out[a] = kps_out[a]; c.desc_out[a] = desc_out[a]; c.cap[a] = cap[a]; c.n_out[a] = n_out[a]; c.n_mono_out[a] = n_mono_out[a];
      c.reverse[a] = reverse[a];
    }
    c.has_post = post != nullptr;
    if (post) {
      c.post = *post;
      if (post->view) { c.view_copy = *post->view; c.post.view = &c.view_copy; }
    }
    c.stereo_out = stereo_out;
    if (submit_only) return ORBG_OK;
    return extract_finish_gpu(h, c);
  }
  if (prof >= 2) ORBG_HIP(hipEventRecord(h->ev[2], st));
  ORBG_HIP(hipStreamSynchronize(st));
  const auto t_host0 = std::chrono::steady_clock::now();
  // ---- host: quad-tree per (camera, level), lapping order (:1104-1146)
  int total = n_cells > 0 ? h->hdr.h[2 * ORBG_MAX_LEVELS] : 0;
  if (total > h->cand_cap) {
    // candidate list did not fit the mapped buffer: grow it and ask the caller to retry is not acceptable for a
    // drop-in, so re-run the gather with a larger buffer.
    h->cand_cap = total + total / 2;
    if ((rc = h->cand.reserve(h->cand_cap))) return rc;
    hipLaunchKernelGGL(gather_cells_kernel, dim3(n_cells, ncams), dim3(256), 0, st, h->d_slots.p, h->d_counts.p, g,
                       h->d_cells.p, n_cells, ncams, h->hdr.d, h->cand.d, h->cand_cap);
    ORBG_HIP(hipStreamSynchronize(st));
  }
  int n_sel_total = 0;
  // one task per (camera, level), biggest levels first; results land in level_sel[cam][level]
  const int n_tasks = ncams * nl;
  auto octree_task = [&](int task, int wid) {
    const int cam = task % ncams, l = task / ncams;
    std::vector<Cand>& cv = h->last_cands[cam][l];
    std::vector<SelKp>& out = h->level_sel[cam][l];
    cv.clear();
    out.clear();
    const LevelGeom& L = g.lv[l];
    if (L.cell_end == L.cell_begin) return;
    const int b = h->hdr.h[cam * ORBG_MAX_LEVELS + l];
    int e;
    int nlv = l + 1;
    while (nlv < nl && g.lv[nlv].cell_end == g.lv[nlv].cell_begin) nlv++;
    if (nlv < nl) e = h->hdr.h[cam * ORBG_MAX_LEVELS + nlv];
    else e = h->hdr.h[2 * ORBG_MAX_LEVELS + 1 + cam];
    cv.resize(e - b);
    for (int i = b; i < e; i++) {
      const uint32_t p = h->cand.h[i];
      cv[i - b] = Cand{(int)(p & 0xFFF), (int)((p >> 12) & 0xFFF), (int)(p >> 24)};
    }
    const int minB = kEdge - 3;
    std::vector<int> keep;
    h->qts[wid].run(cv.data(), (int)cv.size(), minB, L.w - kEdge + 3, minB, L.h - kEdge + 3, h->feats_per_level[l], keep);
    out.reserve(keep.size());
    for (int k : keep) {
      SelKp s;
      s.x = (short)(cv[k].x + minB); s.y = (short)(cv[k].y + minB);
      s.level = (short)l; s.cam = (short)cam; s.out_idx = 0; s.response = (float)cv[k].score;
      out.push_back(s);
    }
  };
  h->pool->run(n_tasks, octree_task);
  for (int cam = 0; cam < ncams; cam++) {
    std::vector<SelKp> level_kps;   // in level order, octree list order
    for (int l = 0; l < nl; l++) level_kps.insert(level_kps.end(), h->level_sel[cam][l].begin(), h->level_sel[cam][l].end());
    const int nk = (int)level_kps.size();
    h->n_kp[cam] = nk;
    if (n_out[cam]) *n_out[cam] = nk;
    if ((kps_out[cam] || desc_out[cam]) && nk > cap[cam]) return ORBG_CAP_EXCEEDED;
    if ((size_t)(n_sel_total + nk) > h->sel.cap) return ORBG_CAP_EXCEEDED;
    int monoIndex = 0, stereoIndex = nk - 1;
    for (SelKp& s : level_kps) {
      float px = (float)s.x;
      if (s.level != 0) px *= h->scale[s.level];
      if (px >= (float)lap[cam][0] && px <= (float)lap[cam][1]) s.out_idx = stereoIndex--;
      else s.out_idx = monoIndex++;
      h->sel.h[n_sel_total++] = s;
    }
    if (n_mono_out[cam]) *n_mono_out[cam] = monoIndex;
  }
  if (ncams == 1) h->n_kp[1] = 0;
  const auto t_host1 = std::chrono::steady_clock::now();
  // ---- GPU phase 2: orientation + descriptors, written in final order
  if (prof >= 2) ORBG_HIP(hipEventRecord(h->ev[3], st));
  if (n_sel_total > 0) {
    auto launch_odh = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3((n_sel_total + kKpPerBlock - 1) / kKpPerBlock), dim3(256), 0, st,
                         h->d_pyr.p, g, h->sel.d, n_sel_total, h->umax, h->n_kp[0], h->d_kps.p, h->d_desc.p);
    };
    if (h->taps_variant == 0) launch_odh(orient_desc_kernel<0>);
    else if (h->taps_variant == 1) launch_odh(orient_desc_kernel<1>);
    else launch_odh(orient_desc_kernel<2>);
  }
  if (prof >= 2) ORBG_HIP(hipEventRecord(h->ev[4], st));
  bool stereo_out_host = false;
  if (post) {
    if (post->stereo && ncams == 2) {
      if ((rc = launch_stereo(h, post->bf, post->b, st, false, nullptr))) return rc;
      if (h->n_kp[0] > 0 && (post->uright || post->depth)) {
        ORBG_HIP(hipMemcpyAsync(h->h_stereo.h, h->d_uright.p, (size_t)h->n_kp[0] * 4, hipMemcpyDeviceToHost, st));
        ORBG_HIP(hipMemcpyAsync(h->h_stereo.h + h->n_kp[0], h->d_depth.p, (size_t)h->n_kp[0] * 4, hipMemcpyDeviceToHost, st));
        stereo_out_host = true;
      }
    }
    if (post->frame) {
      orbg::UndistortArgs ua;
      if ((rc = make_undistort_args(h, post, &ua))) return rc;
      if ((rc = orbm_internal_attach(post->frame, h, post->view, h->n_kp[0], st, nullptr, nullptr, 0u, nullptr, &ua, post->mono))) return rc;
    }
  }
  const bool want_out = kps_out[0] || desc_out[0] || kps_out[1] || desc_out[1];
  if (n_sel_total > 0) {
    // the keypoints are always mirrored into pinned host memory: the matchers' serial commit needs octave / angle
    ORBG_HIP(hipMemcpyAsync(h->h_kps.h, h->d_kps.p, (size_t)n_sel_total * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, st));
    if (desc_out[0] || desc_out[1])
      ORBG_HIP(hipMemcpyAsync(h->h_desc.h, h->d_desc.p, (size_t)n_sel_total * 32, hipMemcpyDeviceToHost, st));
  }
  ORBG_HIP(hipStreamSynchronize(st));
  if (want_out) {
    int base = 0;
    for (int cam = 0; cam < ncams; cam++) {
      if (kps_out[cam]) memcpy(kps_out[cam], h->h_kps.h + base, (size_t)h->n_kp[cam] * sizeof(orbx_keypoint));
      if (desc_out[cam]) memcpy(desc_out[cam], h->h_desc.h + (size_t)base * 32, (size_t)h->n_kp[cam] * 32);
      base += h->n_kp[cam];
    }
  }
  if (stereo_out_host) {
    if (post->uright) memcpy(post->uright, h->h_stereo.h, (size_t)h->n_kp[0] * 4);
    if (post->depth) memcpy(post->depth, h->h_stereo.h + h->n_kp[0], (size_t)h->n_kp[0] * 4);
  }
  if (post && post->kps_un_out) {
    if (h->n_kp[0] > cap[0]) return ORBG_CAP_EXCEEDED;
    memcpy(post->kps_un_out, (post->undist && post->frame) ? h->h_kps_un.h : h->h_kps.h, (size_t)h->n_kp[0] * sizeof(orbx_keypoint));
  }
  float ms;
  h->timings[2] = std::chrono::duration<float, std::milli>(t_host1 - t_host0).count();  // host quad-tree
  if (prof >= 2) {
    hipEvent_t pyr_end = h->prof_kernel == ORBX_PROF_FAST ? h->ev[1] : h->ev[8];
    if (hipEventElapsedTime(&ms, h->ev[0], pyr_end) == hipSuccess) h->timings[0] = ms;   // pyramid
    if (hipEventElapsedTime(&ms, pyr_end, h->ev[2]) == hipSuccess) h->timings[1] = ms;    // FAST + gather
    if (hipEventElapsedTime(&ms, h->ev[3], h->ev[4]) == hipSuccess) h->timings[3] = ms;   // orientation + descriptors
  }
  if (prof >= 1 && n_cells > 0 && hipEventElapsedTime(&ms, h->ev[1], h->ev[7]) == hipSuccess) { h->timings[5] = ms; h->fast_ms_sum += ms; h->fast_ms_n++; }
  return ORBG_OK;
}

// Host images reach the device through the handle's pinned staging slot: the rows are packed into it by the calling (or the
// ingest) thread and ONE copy kernel on the extractor's stream moves the slot into HBM -- every PCIe read of the transfer is in
// flight at once, the copy is ordered before the pyramid kernel by the stream, and it overlaps whatever other streams run
// (the runtime's pageable-memory copy stages through its own buffers synchronously: ~35 us of the calling thread for two
// 640 x 480 images, and its blit kernel takes ~26 us for what this kernel moves in a few).
__global__ __launch_bounds__(256) void img_upload_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}

// Round 4: ONE copy kernel per stereo constructor, launched when the FIRST image is in the staging slot.  Its first workgroups copy
// that image at once; the others belong to the second image: their first thread polls one word in pinned memory, which the host
// writes (the submission's sequence number) when the second image is packed, and then they copy -- four 16-byte blocks per thread,
// every read in flight before the first store.  The word is written once, behind the pack, and is the only host memory the device
// reads while the host is still packing (a first form that polled a word per 38 KB chunk from a kernel launched before the pack was
// measured at 64 us per pair: the device's reads of lines the host was writing slowed the pack itself to 21-47 us).  A poll that does
// not see its word within ~2 s (the packing thread died) raises the handle's error word instead of hanging the device.
// (Also measured, round 4: the right image packed by the SUBMITTING thread while the ingest thread packs the left one -- the ingest
// thread's pack halves, 15 -> 8 us, the constructor's latency does not move, 147 vs 146 us: next to the searches and the local BA the
// chain is not waiting for the second image.)
constexpr int kUpThreads = 256, kUpPerThread = 4;
__global__ __launch_bounds__(kUpThreads) void img_upload_pair_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int first16, int total16,
                                                                     int wg_first, int wg_stride, const unsigned* ready, unsigned seq, unsigned* err) {
  __shared__ int s_ok;
  int b0, b1, wg;
  if ((int)blockIdx.x < wg_first) { b0 = 0; b1 = first16; wg = blockIdx.x; }
  else {
    b0 = first16; b1 = total16; wg = blockIdx.x - wg_first;
    if (threadIdx.x == 0) {
      int ok = 0;
      for (int spin = 0; spin < (1 << 20); spin++) {          // ~2 s: a packing thread that lost its core for a few time slices is no error
        if (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == seq) { ok = 1; break; }
        __builtin_amdgcn_s_sleep(4);
      }
      if (!ok) __hip_atomic_store(err, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) return;
    // The image is read after the word: the loads below are ISSUED after the spin loop has seen it (LDS hand-over + barrier), and the
    // staging slot is coherent host memory (hipHostMallocMapped: uncached in the GPU's L2), so no cache line can hold an older
    // copy.  A system-scope acquire fence here (round 4 until the end) invalidated the XCD's L2 once per wavefront of the right
    // image's workgroups -- ~80 invalidations per frame under the local BA's kernels.
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  // a workgroup moves 16 KB per trip and then its next chunk (wg_stride workgroups per image): the reads in flight over PCIe are
  // wg_stride x 16 KB per image instead of the whole image (see g_upload_wgs)
  for (int c = wg;; c += wg_stride) {
    const int i0 = b0 + (c * kUpPerThread) * kUpThreads + (int)threadIdx.x;
    if (i0 - (int)threadIdx.x >= b1) break;
    uint4 v[kUpPerThread];
#pragma unroll
    for (int q = 0; q < kUpPerThread; q++) { const int i = i0 + q * kUpThreads; if (i < b1) v[q] = src[i]; }
#pragma unroll
    for (int q = 0; q < kUpPerThread; q++) { const int i = i0 + q * kUpThreads; if (i < b1) dst[i] = v[q]; }
  }
}

// a submitted Frame constructor owns the handle (stream, staging slot, pyramid, feature buffers) until it has been waited for
static inline bool handle_busy(const orbx_handle* h) {
  return h->ingest_state.load(std::memory_order_acquire) != 0 || (h->pending && (h->pending->active || h->pending->finished));
}

// images[c] for c < n_img -> h->d_img (packed rows, image c at c * w * hgt)
static int stage_images(orbx_handle* h, const uint8_t* const* images, int n_img, int w, int hgt, int stride) {
  const size_t per = (size_t)w * hgt, total = per * n_img;
  int rc;
  if ((rc = h->h_img.reserve(total + 16)) || (rc = h->d_img.reserve(total + 16))) return rc;
  if (n_img == 2) {
    if ((rc = h->up_ready.reserve(16))) return rc;
    if (!h->up_seq) memset(h->up_ready.h, 0, 16 * sizeof(unsigned));
    const unsigned seq = ++h->up_seq;
    // blocks of 16 bytes: the first image's blocks end at the block that holds its last byte; that block (shared with the second
    // image when the image size is no multiple of 16) belongs to the SECOND range, which is copied once both images are packed
    const int first16 = (int)(per / 16), total16 = (int)((total + 15) / 16);
    const int per_wg = kUpThreads * kUpPerThread;
    int wg_first = (first16 + per_wg - 1) / per_wg, wg_second = (total16 - first16 + per_wg - 1) / per_wg;
    int wg_stride = std::max(wg_first, wg_second);
    double tp0 = host_now_us();
    auto pack = [&](int c) {
      uint8_t* dst = h->h_img.h + c * per;
      if (stride == w) memcpy(dst, images[c], per);
      else for (int y = 0; y < hgt; y++) memcpy(dst + (size_t)y * w, images[c] + (size_t)y * stride, w);
    };
    pack(0);
    h->tl_pack_acc += host_now_us() - tp0;
    hipLaunchKernelGGL(img_upload_pair_kernel, dim3(wg_first + wg_second), dim3(kUpThreads), 0, h->stream, reinterpret_cast<const uint4*>(h->h_img.d),
                       reinterpret_cast<uint4*>(h->d_img.p), first16, total16, wg_first, wg_stride, h->up_ready.d, seq, h->up_ready.d + 8);
    ORBG_HIP(hipGetLastError());
    tp0 = host_now_us();
    pack(1);
    __atomic_store_n(&h->up_ready.h[0], seq, __ATOMIC_RELEASE);
    h->tl_pack_acc += host_now_us() - tp0;
    return ORBG_OK;
  }
  // one copy kernel per image, launched as soon as that image is packed: the device copies the left image while the host packs the
  // right one.  A launch moves the 16-byte blocks that cover its image: the few bytes it shares with a neighbour's blocks are either
  // already staged (the image before) or rewritten by the next launch (the image after), which runs behind it on the stream.
  for (int c = 0; c < n_img; c++) {
    uint8_t* dst = h->h_img.h + c * per;
    const double tp0 = host_now_us();
    // (non-temporal stores were measured, round 4: under memory pressure from the same L3 slice 106 vs 112 us per pair, quiet 16.9 vs 16.9)
    if (stride == w) memcpy(dst, images[c], per);
    else for (int y = 0; y < hgt; y++) memcpy(dst + (size_t)y * w, images[c] + (size_t)y * stride, w);
    h->tl_pack_acc += host_now_us() - tp0;
    const size_t b0 = (c * per) / 16, b1 = ((c + 1) * per + 15) / 16;
    const int n16 = (int)(b1 - b0);
    hipLaunchKernelGGL(img_upload_kernel, dim3((n16 + 255) / 256), dim3(256), 0, h->stream, reinterpret_cast<const uint4*>(h->h_img.d) + b0,
                       reinterpret_cast<uint4*>(h->d_img.p) + b0, n16);
  }
  return ORBG_OK;
}

extern "C" int orbx_extract(orbx_handle* h, int cam, const uint8_t* img, int width, int height, int stride, int lap0,
                            int lap1, orbx_keypoint* kps, uint8_t* desc, int cap, int* n, int* n_mono) {
  if (!h || !n) return ORBG_BAD_ARG;
  if (!img || width <= 0 || height <= 0) return ORBG_EMPTY;     // S/ORBextractor.cc:1072-1073
  if (cam != 0 || stride < width) return ORBG_BAD_ARG;          // one camera per call goes through slot 0
  if (handle_busy(h)) return ORBG_BAD_ARG;
  int rc = select_device(h->device);
  if (rc) return rc;
  if ((rc = setup_geometry(h, width, height))) return rc;
  if ((rc = stage_images(h, &img, 1, width, height, stride))) return rc;
  const int lap[2][2] = {{lap0, lap1}, {0, 0}};
  orbx_keypoint* ko[2] = {kps, nullptr};
  uint8_t* dout[2] = {desc, nullptr};
  const int caps[2] = {cap, 0};
  int* no[2] = {n, nullptr};
  int* nm[2] = {n_mono, nullptr};
  return extract_core(h, 1, h->d_img.p, nullptr, width, height, width, lap, ko, dout, caps, no, nm);
}

static int extract_stereo_impl(orbx_handle* h, const uint8_t* d0, const uint8_t* d1, int width, int height, int stride,
                               orbx_keypoint* kl, uint8_t* dl, int cl, int* nl, orbx_keypoint* kr, uint8_t* dr, int cr, int* nr) {
  const int lap[2][2] = {{0, 0}, {0, 0}};      // vLapping = {0,0} for the rectified stereo Frame ctor (S/Frame.cc:92-95)
  orbx_keypoint* ko[2] = {kl, kr};
  uint8_t* dout[2] = {dl, dr};
  const int caps[2] = {cl, cr};
  int* no[2] = {nl, nr};
  int* nm[2] = {nullptr, nullptr};
  return extract_core(h, 3, d0, d1, width, height, stride, lap, ko, dout, caps, no, nm);
}

extern "C" int orbx_extract_stereo(orbx_handle* h, const uint8_t* img_left, const uint8_t* img_right, int width, int height,
                                   int stride, orbx_keypoint* kps_left, uint8_t* desc_left, int cap_left, int* n_left,
                                   orbx_keypoint* kps_right, uint8_t* desc_right, int cap_right, int* n_right) {
  if (!h || h->cfg.n_cams != 2) return ORBG_BAD_ARG;
  if (!img_left || !img_right || width <= 0 || height <= 0) return ORBG_EMPTY;
  if (stride < width || handle_busy(h)) return ORBG_BAD_ARG;
  int rc = select_device(h->device);
  if (rc) return rc;
  if ((rc = setup_geometry(h, width, height))) return rc;
  const uint8_t* const both[2] = {img_left, img_right};
  if ((rc = stage_images(h, both, 2, width, height, stride))) return rc;
  return extract_stereo_impl(h, h->d_img.p, h->d_img.p + (size_t)width * height, width, height, width, kps_left, desc_left,
                             cap_left, n_left, kps_right, desc_right, cap_right, n_right);
}

extern "C" int orbx_extract_stereo_dev(orbx_handle* h, const uint8_t* d_img_left, const uint8_t* d_img_right, int width,
                                       int height, int stride, orbx_keypoint* kps_left, uint8_t* desc_left, int cap_left,
                                       int* n_left, orbx_keypoint* kps_right, uint8_t* desc_right, int cap_right, int* n_right) {
  if (!h || h->cfg.n_cams != 2) return ORBG_BAD_ARG;
  if (!d_img_left || !d_img_right || width <= 0 || height <= 0) return ORBG_EMPTY;
  if (stride < width) return ORBG_BAD_ARG;
  int rc = select_device(h->device);
  if (rc) return rc;
  return extract_stereo_impl(h, d_img_left, d_img_right, width, height, stride, kps_left, desc_left, cap_left, n_left,
                             kps_right, desc_right, cap_right, n_right);
}

// The fused constructors feed the extracted keypoints straight into the grid: mvKeysUn = mvKeys, i.e. mDistCoef[0] == 0
// (S/Frame.cc:723-727).  The reference's bounds are the image rectangle exactly then (S/Frame.cc:775-783); anything else is a
// distorted camera, which this entry point does not undistort: refused.
static inline bool view_is_undistorted(const orbm_frame_view* v, int width, int height) {
  return v->min_x == 0.0f && v->min_y == 0.0f && v->max_x == (float)width && v->max_y == (float)height;
}

static int frame_stereo_impl(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const uint8_t* img_left,
                             const uint8_t* img_right, bool on_device, int width, int height, int stride, float bf, float b,
                             orbx_keypoint* kps_left, uint8_t* desc_left, float* uright, float* depth, int cap_left,
                             int* n_left, int* n_right) {
  if (!h || h->cfg.n_cams != 2 || !n_left) return ORBG_BAD_ARG;
  if (frame && !view) return ORBG_BAD_ARG;
  if (frame && width > 0 && height > 0 && !view_is_undistorted(view, width, height)) return ORBG_BAD_ARG;
  if (!img_left || !img_right || width <= 0 || height <= 0) return ORBG_EMPTY;
  if (stride < width || handle_busy(h)) return ORBG_BAD_ARG;
  int rc = select_device(h->device);
  if (rc) return rc;
  const uint8_t* d_img_left = img_left;
  const uint8_t* d_img_right = img_right;
  if (!on_device) {
    if ((rc = setup_geometry(h, width, height))) return rc;
    const uint8_t* const both[2] = {img_left, img_right};
    if ((rc = stage_images(h, both, 2, width, height, stride))) return rc;
    d_img_left = h->d_img.p;
    d_img_right = h->d_img.p + (size_t)width * height;
    stride = width;
  }
  PostOps post;
  post.stereo = true; post.bf = bf; post.b = b; post.uright = uright; post.depth = depth; post.frame = frame; post.view = view;
  const int lap[2][2] = {{0, 0}, {0, 0}};
  orbx_keypoint* ko[2] = {kps_left, nullptr};
  uint8_t* dout[2] = {desc_left, nullptr};
  const int caps[2] = {cap_left, 0};
  int* no[2] = {n_left, n_right};
  int* nm[2] = {nullptr, nullptr};
  return extract_core(h, 3, d_img_left, d_img_right, width, height, stride, lap, ko, dout, caps, no, nm, &post);
}

extern "C" int orbx_frame_stereo_dev(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const uint8_t* d_img_left,
                                     const uint8_t* d_img_right, int width, int height, int stride, float bf, float b,
                                     orbx_keypoint* kps_left, uint8_t* desc_left, float* uright, float* depth, int cap_left,
                                     int* n_left, int* n_right) {
  return frame_stereo_impl(h, frame, view, d_img_left, d_img_right, true, width, height, stride, bf, b, kps_left, desc_left, uright,
                           depth, cap_left, n_left, n_right);
}

// Frame constructor split in two so that the caller can overlap it with work on other streams (tracking of the previous
// frame): submit enqueues the whole chain and returns, wait completes it.
// mono: the monocular constructor (img_right unused; dist = mDistCoef or NULL)
static int frame_submit_core(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const uint8_t* img_left,
                             const uint8_t* img_right, bool on_device, int width, int height, int stride, float bf, float b, bool mono,
                             const orbx_distortion* dist);
static int frame_submit_impl(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const uint8_t* img_left,
                             const uint8_t* img_right, bool on_device, int width, int height, int stride, float bf, float b, bool mono = false,
                             const orbx_distortion* dist = nullptr) {
  const double t0 = host_now_us();
  h->tl_queue = t0 - h->tl_t_handover;
  const int rc = frame_submit_core(h, frame, view, img_left, img_right, on_device, width, height, stride, bf, b, mono, dist);
  h->tl_enqueue = host_now_us() - t0 - h->tl_pack_acc;
  return rc;
}
// Frame::Frame(mono): ExtractORB(0, imGray, 0, 1000) (S/Frame.cc:289) -- camera 0 only, lapping area {0, 1000}
static const int kMonoLap[2][2] = {{0, 1000}, {0, 0}};
static inline bool dist_active(const orbx_distortion* d) { return d && d->k1 != 0.0f; }      // S/Frame.cc:723
static void fill_mono_post(PostOps* post, orbm_frame* frame, const orbm_frame_view* view, const orbx_distortion* dist, orbx_keypoint* kps_un_out) {
  post->stereo = false; post->mono = true; post->frame = frame; post->view = view; post->kps_un_out = kps_un_out;
  post->undist = dist_active(dist) && frame != nullptr;
  if (post->undist) post->dist = *dist;
}
static int frame_submit_core(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const uint8_t* img_left,
                             const uint8_t* img_right, bool on_device, int width, int height, int stride, float bf, float b, bool mono,
                             const orbx_distortion* dist) {
  int rc = select_device(h->device);
  if (rc) return rc;
  if (!h->pending) h->pending = new ExtractPending();
  ExtractPending& P = *h->pending;
  if (P.active || P.finished) return ORBG_BAD_ARG;      // the previous submission has not been collected
  const uint8_t* d_img_left = img_left;
  const uint8_t* d_img_right = img_right;
  if (!on_device) {
    if ((rc = setup_geometry(h, width, height))) return rc;
    const uint8_t* const both[2] = {img_left, img_right};
    if ((rc = stage_images(h, both, mono ? 1 : 2, width, height, stride))) return rc;
    d_img_left = h->d_img.p;
    d_img_right = mono ? nullptr : h->d_img.p + (size_t)width * height;
    stride = width;
  }
  PostOps post;
  if (mono) fill_mono_post(&post, frame, view, dist, h->out_kps_un);
  else {
    post.stereo = true; post.bf = bf; post.b = b; post.frame = frame; post.view = view;
    post.uright = h->out_uright; post.depth = h->out_depth;             // (orbx_set_frame_outputs: delivered by _wait)
  }
  const int lap_stereo[2][2] = {{0, 0}, {0, 0}};
  orbx_keypoint* ko[2] = {h->out_kps, nullptr};
  uint8_t* dout[2] = {h->out_desc, nullptr};
  const int caps[2] = {h->out_cap, 0};
  int* no[2] = {&P.n_res[0], &P.n_res[1]};
  int* nm[2] = {nullptr, nullptr};
  P.n_res[1] = 0;
  rc = extract_core(h, mono ? 1u : 3u, d_img_left, d_img_right, width, height, stride, mono ? kMonoLap : lap_stereo, ko, dout, caps, no, nm, &post,
                    false, true);
  if (rc) { P.active = false; return rc; }
  if (!P.active) P.finished = true;                      // host quad-tree path: it ran to completion inside the call
  return ORBG_OK;
}

// Host arrays for the features of the two-halves constructor (include/orbgpu.h): taken by every later _submit of this handle.
extern "C" int orbx_set_frame_outputs(orbx_handle* h, orbx_keypoint* kps_left, uint8_t* desc_left, float* uright, float* depth, int cap_left) {
  if (!h || cap_left < 0 || ((kps_left || desc_left || uright || depth) && cap_left == 0)) return ORBG_BAD_ARG;
  if (handle_busy(h)) return ORBG_BAD_ARG;
  h->out_kps = kps_left; h->out_desc = desc_left; h->out_uright = uright; h->out_depth = depth; h->out_cap = cap_left;
  if (cap_left == 0) h->out_kps_un = nullptr;               // (the mvKeysUn array shares the capacity: off together)
  return ORBG_OK;
}

extern "C" int orbx_frame_stereo_dev_submit(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const uint8_t* d_img_left,
                                            const uint8_t* d_img_right, int width, int height, int stride, float bf, float b) {
  if (!h || h->cfg.n_cams != 2) return ORBG_BAD_ARG;
  if (frame && !view) return ORBG_BAD_ARG;
  if (!d_img_left || !d_img_right || width <= 0 || height <= 0) return ORBG_EMPTY;
  if (stride < width) return ORBG_BAD_ARG;
  if (frame && !view_is_undistorted(view, width, height)) return ORBG_BAD_ARG;
  if (h->ingest_state.load(std::memory_order_acquire) != 0) return ORBG_BAD_ARG;
  h->tl_begin();
  return frame_submit_impl(h, frame, view, d_img_left, d_img_right, true, width, height, stride, bf, b);
}

// ---- the ingest thread: ONE per process, shared by every extractor handle.  A camera driver / image-grabber thread of a
// deployment plays this role itself (it calls orbx_frame_stereo_submit with flags 0 when a stereo pair arrives); for callers
// that hand images over on their tracking thread, ORBX_SUBMIT_ASYNC moves the staging copy (~0.6 MB at 640 x 480) and the
// launches of the constructor chain off that thread.  The thread is created by the first asynchronous submission and
// inherits that caller's CPU affinity.
namespace {
struct IngestJob {
  orbx_handle* h; orbm_frame* frame; orbm_frame_view view; bool has_view; const uint8_t* L; const uint8_t* R; int w, hgt, stride; float bf, b;
  bool mono = false, has_dist = false; orbx_distortion dist{};
};
struct IngestWorker {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<IngestJob> q;
  std::atomic<int> queued{0};
  bool quit = false;
  std::thread th;
  static bool spin_ok() { return orbg::poll_allowed(); }
  void run() {
    orbg::set_thread_role(orbg::kRoleIngest);
    for (;;) {
      // the next pair usually arrives within a frame time: spin for a while, then sleep
      bool have = false;
      if (spin_ok()) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; spins++) {
          if (queued.load(std::memory_order_acquire) > 0) { have = true; break; }
#if defined(__x86_64__)
          __builtin_ia32_pause();
#endif
          if ((spins & 0xFF) == 0xFF && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) break;
        }
      }
      IngestJob job;
      {
        std::unique_lock<std::mutex> lk(mu);
        if (!have) cv.wait(lk, [this]() { return quit || !q.empty(); });
        if (quit && q.empty()) return;
        if (q.empty()) continue;
        job = q.front();
        q.erase(q.begin());
        queued.fetch_sub(1, std::memory_order_acq_rel);
      }
      job.h->ingest_rc = frame_submit_impl(job.h, job.frame, job.has_view ? &job.view : nullptr, job.L, job.R, false, job.w, job.hgt,
                                           job.stride, job.bf, job.b, job.mono, job.has_dist ? &job.dist : nullptr);
      job.h->ingest_state.store(2, std::memory_order_release);
    }
  }
  void push(const IngestJob& j) {
    {
      std::unique_lock<std::mutex> lk(mu);
      if (!th.joinable()) th = std::thread([this]() { run(); });
      q.push_back(j);
      queued.fetch_add(1, std::memory_order_acq_rel);
    }
    cv.notify_one();
  }
  ~IngestWorker() {
    { std::unique_lock<std::mutex> lk(mu); quit = true; }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};
IngestWorker& ingest_worker() { static IngestWorker w; return w; }
}  // namespace

extern "C" int orbx_frame_stereo_submit(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const uint8_t* img_left,
                                        const uint8_t* img_right, int width, int height, int stride, float bf, float b, int flags) {
  if (!h || h->cfg.n_cams != 2) return ORBG_BAD_ARG;
  if (frame && !view) return ORBG_BAD_ARG;
  if (!img_left || !img_right || width <= 0 || height <= 0) return ORBG_EMPTY;
  if (stride < width || width > h->cfg.max_width || height > h->cfg.max_height) return ORBG_BAD_ARG;
  if (frame && !view_is_undistorted(view, width, height)) return ORBG_BAD_ARG;
  if (h->ingest_state.load(std::memory_order_acquire) != 0) return ORBG_BAD_ARG;       // one submission per handle at a time
  if (h->pending && (h->pending->active || h->pending->finished)) return ORBG_BAD_ARG;
  h->tl_begin();
  if (!(flags & ORBX_SUBMIT_ASYNC)) return frame_submit_impl(h, frame, view, img_left, img_right, false, width, height, stride, bf, b);
  int rc = select_device(h->device);                    // a missing device is reported by the call, not by the wait
  if (rc) return rc;
  IngestJob j;
  j.h = h; j.frame = frame; j.has_view = view != nullptr; j.L = img_left; j.R = img_right; j.w = width; j.hgt = height; j.stride = stride;
  j.bf = bf; j.b = b;
  if (view) j.view = *view;
  h->ingest_state.store(1, std::memory_order_release);
  ingest_worker().push(j);
  return ORBG_OK;
}

extern "C" int orbx_frame_stereo_dev_wait(orbx_handle* h, int* n_left, int* n_right) {
  if (!h) return ORBG_BAD_ARG;
  int rc;
  const double t_wait0 = host_now_us();
  if (h->ingest_state.load(std::memory_order_acquire) != 0) {
    // handed to the ingest thread: wait until it has enqueued the chain (it is usually done long before)
    for (unsigned spins = 0; h->ingest_state.load(std::memory_order_acquire) != 2; spins++) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
      if (!IngestWorker::spin_ok() || (spins & 0xFFFF) == 0xFFFF) std::this_thread::yield();
    }
    rc = h->ingest_rc;
    h->ingest_state.store(0, std::memory_order_release);
    if (rc) return rc;
  }
  if (!h->pending || !(h->pending->active || h->pending->finished)) return ORBG_BAD_ARG;
  if ((rc = select_device(h->device))) return rc;
  ExtractPending& P = *h->pending;
  if (P.active) {
    P.active = false;
    rc = extract_finish_gpu(h, P);
  }
  P.finished = false;
  if (rc) return rc;
  h->tl_commit(host_now_us() - t_wait0);
  if (n_left) *n_left = P.n_res[0];
  if (n_right) *n_right = P.n_res[1];
  return ORBG_OK;
}

// Host-side timeline of the handle's last submissions, oldest first: out[i * 5 + f], f = queue / pack / enqueue / wait / latency in
// microseconds (see orbx_handle::tl); *n = entries written (<= cap, <= 512).  reset != 0 forgets them afterwards.
extern "C" int orbx_get_ctor_timeline(orbx_handle* h, float* out, int cap, int* n, int reset) {
  if (!h || !n || cap < 0 || (cap > 0 && !out)) return ORBG_BAD_ARG;
  const unsigned long long have = std::min<unsigned long long>(h->tl_n, orbx_handle::kTlCap);
  const int take = (int)std::min<unsigned long long>(have, (unsigned long long)cap);
  for (int i = 0; i < take; i++) {
    const float* e = h->tl[(h->tl_n - take + i) % orbx_handle::kTlCap];
    for (int f = 0; f < orbx_handle::kTlFields; f++) out[i * orbx_handle::kTlFields + f] = e[f];
  }
  *n = take;
  if (reset) h->tl_n = 0;
  return ORBG_OK;
}
extern "C" int orbx_set_stream(orbx_handle* h, void* hip_stream) {
  if (!h) return ORBG_BAD_ARG;
  if (h->ingest_state.load(std::memory_order_acquire) != 0 || (h->pending && (h->pending->active || h->pending->finished))) return ORBG_BAD_ARG;
  int rc = select_device(h->device);
  if (rc) return rc;
  return orbg::swap_stream(&h->stream, &h->ext_stream, hip_stream, "ex");
}

extern "C" int orbx_frame_stereo_wait(orbx_handle* h, int* n_left, int* n_right) { return orbx_frame_stereo_dev_wait(h, n_left, n_right); }

extern "C" int orbx_frame_stereo(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const uint8_t* img_left,
                                 const uint8_t* img_right, int width, int height, int stride, float bf, float b,
                                 orbx_keypoint* kps_left, uint8_t* desc_left, float* uright, float* depth, int cap_left,
                                 int* n_left, int* n_right) {
  if (h) h->tl_begin();
  const int rc = frame_stereo_impl(h, frame, view, img_left, img_right, false, width, height, stride, bf, b, kps_left, desc_left, uright,
                                   depth, cap_left, n_left, n_right);
  if (h && rc == ORBG_OK) h->tl_commit(0.0);            // synchronous: queue / enqueue / wait are not separated, latency = the call
  return rc;
}

// ---- the monocular Frame constructor (include/orbgpu.h): camera 0, lapping area {0, 1000}, undistortion + grid as the chain's tail
static int mono_args_ok(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const orbx_distortion* dist, const uint8_t* img,
                        int width, int height, int stride) {
  if (!h) return ORBG_BAD_ARG;
  if (frame && !view) return ORBG_BAD_ARG;
  if (dist_active(dist) && !frame) return ORBG_BAD_ARG;          // mvKeysUn of a distorted camera exists as the frame's features
  if (!img || width <= 0 || height <= 0) return ORBG_EMPTY;      // (the reference returns before it touches anything: S/Frame.cc:297-298)
  if (stride < width || width > h->cfg.max_width || height > h->cfg.max_height) return ORBG_BAD_ARG;
  // without distortion the reference's bounds are the image rectangle (S/Frame.cc:776-782); with it they are the undistorted corners
  if (frame && !dist_active(dist) && !view_is_undistorted(view, width, height)) return ORBG_BAD_ARG;
  if (frame && dist_active(dist) && !(view->fx > 0.0f && view->fy > 0.0f && view->max_x > view->min_x && view->max_y > view->min_y)) return ORBG_BAD_ARG;
  return ORBG_OK;
}

static int frame_mono_impl(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const orbx_distortion* dist, const uint8_t* img,
                           bool on_device, int width, int height, int stride, orbx_keypoint* kps, orbx_keypoint* kps_un, uint8_t* desc, int cap,
                           int* n) {
  if (!n) return ORBG_BAD_ARG;
  int rc = mono_args_ok(h, frame, view, dist, img, width, height, stride);
  if (rc) return rc;
  if (handle_busy(h)) return ORBG_BAD_ARG;
  if ((rc = select_device(h->device))) return rc;
  const uint8_t* d_img = img;
  if (!on_device) {
    if ((rc = setup_geometry(h, width, height))) return rc;
    if ((rc = stage_images(h, &img, 1, width, height, stride))) return rc;
    d_img = h->d_img.p;
    stride = width;
  }
  PostOps post;
  fill_mono_post(&post, frame, view, dist, kps_un);
  orbx_keypoint* ko[2] = {kps, nullptr};
  uint8_t* dout[2] = {desc, nullptr};
  const int caps[2] = {cap, 0};
  int* no[2] = {n, nullptr};
  int* nm[2] = {nullptr, nullptr};
  return extract_core(h, 1, d_img, nullptr, width, height, stride, kMonoLap, ko, dout, caps, no, nm, &post);
}

extern "C" int orbx_frame_mono(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const orbx_distortion* dist, const uint8_t* img,
                               int width, int height, int stride, orbx_keypoint* kps, orbx_keypoint* kps_un, uint8_t* desc, int cap, int* n) {
  if (h) h->tl_begin();
  const int rc = frame_mono_impl(h, frame, view, dist, img, false, width, height, stride, kps, kps_un, desc, cap, n);
  if (h && rc == ORBG_OK) h->tl_commit(0.0);
  return rc;
}
extern "C" int orbx_frame_mono_dev(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const orbx_distortion* dist, const uint8_t* d_img,
                                   int width, int height, int stride, orbx_keypoint* kps, orbx_keypoint* kps_un, uint8_t* desc, int cap, int* n) {
  return frame_mono_impl(h, frame, view, dist, d_img, true, width, height, stride, kps, kps_un, desc, cap, n);
}

extern "C" int orbx_frame_mono_dev_submit(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const orbx_distortion* dist,
                                          const uint8_t* d_img, int width, int height, int stride) {
  int rc = mono_args_ok(h, frame, view, dist, d_img, width, height, stride);
  if (rc) return rc;
  if (h->ingest_state.load(std::memory_order_acquire) != 0) return ORBG_BAD_ARG;
  h->tl_begin();
  return frame_submit_impl(h, frame, view, d_img, nullptr, true, width, height, stride, 0.f, 0.f, true, dist);
}

extern "C" int orbx_frame_mono_submit(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const orbx_distortion* dist, const uint8_t* img,
                                      int width, int height, int stride, int flags) {
  int rc = mono_args_ok(h, frame, view, dist, img, width, height, stride);
  if (rc) return rc;
  if (h->ingest_state.load(std::memory_order_acquire) != 0) return ORBG_BAD_ARG;       // one submission per handle at a time
  if (h->pending && (h->pending->active || h->pending->finished)) return ORBG_BAD_ARG;
  h->tl_begin();
  if (!(flags & ORBX_SUBMIT_ASYNC)) return frame_submit_impl(h, frame, view, img, nullptr, false, width, height, stride, 0.f, 0.f, true, dist);
  if ((rc = select_device(h->device))) return rc;
  IngestJob j;
  j.h = h; j.frame = frame; j.has_view = view != nullptr; j.L = img; j.R = nullptr; j.w = width; j.hgt = height; j.stride = stride; j.bf = 0; j.b = 0;
  if (view) j.view = *view;
  j.mono = true; j.has_dist = dist != nullptr;
  if (dist) j.dist = *dist;
  h->ingest_state.store(1, std::memory_order_release);
  ingest_worker().push(j);
  return ORBG_OK;
}

extern "C" int orbx_frame_mono_wait(orbx_handle* h, int* n) { return orbx_frame_stereo_dev_wait(h, n, nullptr); }

extern "C" int orbx_set_frame_outputs_un(orbx_handle* h, orbx_keypoint* kps_un) {
  if (!h || (kps_un && h->out_cap == 0)) return ORBG_BAD_ARG;
  if (handle_busy(h)) return ORBG_BAD_ARG;
  h->out_kps_un = kps_un;
  return ORBG_OK;
}

// cv::undistortPoints on n host points, on the device (Frame::ComputeImageBounds runs it on the four corners, S/Frame.cc:756-783)
__global__ __launch_bounds__(256) void undistort_points_kernel(orbg::UndistortArgs ua, const float2* __restrict__ in, float2* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float2 p = in[i], q;
  orbg::undistort_point(ua, p.x, p.y, &q.x, &q.y);
  out[i] = q;
}
extern "C" int orbx_undistort_points(int device, const float* xy_in, int n, float fx, float fy, float cx, float cy, const orbx_distortion* dist,
                                     float* xy_out) {
  if (n < 0 || (n > 0 && (!xy_in || !xy_out))) return ORBG_BAD_ARG;
  int rc = select_device(device);
  if (rc) return rc;
  if (n == 0) return ORBG_OK;
  if (!dist_active(dist)) { if (xy_out != xy_in) memmove(xy_out, xy_in, (size_t)n * 8); return ORBG_OK; }     // S/Frame.cc:723-727
  orbg::MiscStream ms;
  if ((rc = ms.open())) return rc;
  DevBuf<float2> d_in, d_out;
  if ((rc = d_in.reserve(n)) || (rc = d_out.reserve(n))) { d_in.release(); d_out.release(); return rc; }
  orbg::UndistortArgs ua;
  memset(&ua, 0, sizeof(ua));
  ua.on = 1; ua.fx = fx; ua.fy = fy; ua.cx = cx; ua.cy = cy; ua.k1 = dist->k1; ua.k2 = dist->k2; ua.p1 = dist->p1; ua.p2 = dist->p2; ua.k3 = dist->k3;
  hipError_t e = hipMemcpyAsync(d_in.p, xy_in, (size_t)n * 8, hipMemcpyHostToDevice, ms.s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(undistort_points_kernel, dim3((n + 255) / 256), dim3(256), 0, ms.s, ua, d_in.p, d_out.p, n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(xy_out, d_out.p, (size_t)n * 8, hipMemcpyDeviceToHost, ms.s);
  if (e == hipSuccess) e = hipStreamSynchronize(ms.s);
  d_in.release(); d_out.release();
  return e == hipSuccess ? ORBG_OK : ORBG_HIP_ERROR;
}

extern "C" int orbx_get_level(orbx_handle* h, int cam, int level, uint8_t* host_out, int* width, int* height) {
  if (!h || cam < 0 || cam >= h->cfg.n_cams || level < 0 || level >= h->cfg.n_levels || h->cur_w == 0) return ORBG_BAD_ARG;
  const LevelGeom& L = h->geom.lv[level];
  if (width) *width = L.w;
  if (height) *height = L.h;
  if (host_out) {
    int rc = select_device(h->device);
    if (rc) return rc;
    const uint8_t* src = h->d_pyr.p + (size_t)cam * h->geom.cam_stride + L.off + (size_t)kEdge * L.stride + kEdge;
    ORBG_HIP(hipMemcpy2D(host_out, L.w, src, L.stride, L.w, L.h, hipMemcpyDeviceToHost));
  }
  return ORBG_OK;
}

extern "C" int orbx_get_level_bordered(orbx_handle* h, int cam, int level, uint8_t* host_out, int* width, int* height) {
  if (!h || cam < 0 || cam >= h->cfg.n_cams || level < 0 || level >= h->cfg.n_levels || h->cur_w == 0) return ORBG_BAD_ARG;
  const LevelGeom& L = h->geom.lv[level];
  if (width) *width = L.w;
  if (height) *height = L.h;
  if (host_out) {
    int rc = select_device(h->device);
    if (rc) return rc;
    const uint8_t* src = h->d_pyr.p + (size_t)cam * h->geom.cam_stride + L.off;
    ORBG_HIP(hipMemcpy2D(host_out, L.w + 2 * kEdge, src, L.stride, L.w + 2 * kEdge, L.h + 2 * kEdge, hipMemcpyDeviceToHost));
  }
  return ORBG_OK;
}

extern "C" int orbx_get_candidates(orbx_handle* h, int cam, int level, int32_t* xys, int cap, int* n) {
  if (!h || cam < 0 || cam >= h->cfg.n_cams || level < 0 || level >= h->cfg.n_levels || !n) return ORBG_BAD_ARG;
  if (h->last_was_gpu) {
    // device-resident candidate list of the last extraction: fetch and decode on demand
    int rc = select_device(h->device);
    if (rc) return rc;
    if (!h->cells.empty()) {
      // the constructor chain no longer builds the compacted list (the quad-trees read the per-cell slots): build it now from the
      // slots of the last extraction, which stay valid until the next one
      const int ncams = h->cfg.n_cams, n_cells = (int)h->cells.size();
      hipLaunchKernelGGL(gather_cells_kernel, dim3(n_cells, ncams), dim3(256), 0, h->stream, h->d_slots.p, h->d_counts.p, h->geom,
                         h->d_cells.p, n_cells, ncams, h->d_hdr.p, h->d_cand.p, h->cand_cap);
      ORBG_HIP(hipGetLastError());
      ORBG_HIP(hipStreamSynchronize(h->stream));
    }
    std::vector<int> hdr(2 * ORBG_MAX_LEVELS + 4);
    ORBG_HIP(hipMemcpy(hdr.data(), h->d_hdr.p, hdr.size() * sizeof(int), hipMemcpyDeviceToHost));
    const PyrGeom& g = h->geom;
    const int nl = g.n_levels;
    std::vector<Cand>& cvw = h->last_cands[cam][level];
    cvw.clear();
    if (g.lv[level].cell_end != g.lv[level].cell_begin) {
      const int b = hdr[cam * ORBG_MAX_LEVELS + level];
      int nlv = level + 1;
      while (nlv < nl && g.lv[nlv].cell_end == g.lv[nlv].cell_begin) nlv++;
      const int e = nlv < nl ? hdr[cam * ORBG_MAX_LEVELS + nlv] : hdr[2 * ORBG_MAX_LEVELS + 1 + cam];
      std::vector<uint32_t> raw(std::max(e - b, 0));
      if (e > b) ORBG_HIP(hipMemcpy(raw.data(), h->d_cand.p + b, (size_t)(e - b) * 4, hipMemcpyDeviceToHost));
      for (uint32_t pk : raw) cvw.push_back(Cand{(int)(pk & 0xFFF), (int)((pk >> 12) & 0xFFF), (int)(pk >> 24)});
    }
  }
  const std::vector<Cand>& cv = h->last_cands[cam][level];
  *n = (int)cv.size();
  for (int i = 0; i < *n && i < cap; i++) { xys[3 * i] = cv[i].x; xys[3 * i + 1] = cv[i].y; xys[3 * i + 2] = cv[i].score; }
  return *n > cap ? ORBG_CAP_EXCEEDED : ORBG_OK;
}

// Second half of the GPU-path extraction: wait for the completion word, fall back to the host quad-trees if a level
// overflowed the device lists, hand the counts (and the optional host copies) over.
static int extract_finish_gpu(orbx_handle* h, ExtractPending& c) {
  int rc;
  hipStream_t st = h->stream;
  if ((rc = h->sig.wait(st))) return rc;            // the overflow flag came with the keypoint counts
  // (img_upload_pair_kernel gave up waiting for the host's pack: the images never reached the device)
  if (h->up_seq && h->up_ready.h && h->up_ready.h[8] == h->up_seq) return ORBG_INTERNAL;
  const PostOps* post = c.has_post ? &c.post : nullptr;
  if (h->h_nkp.h[2]) {
    // a level had more candidates / nodes than the LDS-resident quad-tree holds: redo this frame with the host trees
    ORBG_HIP(hipMemsetAsync(h->d_overflow.p, 0, sizeof(int), st));
    return extract_core(h, c.cams_mask, c.d_img0, c.d_img1, c.w, c.hgt, c.stride, c.lap, c.kps_out, c.desc_out, c.cap, c.n_out, c.n_mono_out,
                        post, true, false);
  }
  const int ncams = c.ncams;
  h->n_kp[0] = h->h_nkp.h[0];
  h->n_kp[1] = ncams == 2 ? h->h_nkp.h[1] : 0;
  int base = 0;
  for (int cam = 0; cam < ncams; cam++) {
    const int nk = h->n_kp[cam];
    if (c.n_out[cam]) *c.n_out[cam] = nk;
    if (c.n_mono_out[cam]) *c.n_mono_out[cam] = c.reverse[cam] ? 0 : nk;
    if ((c.kps_out[cam] || c.desc_out[cam]) && nk > c.cap[cam]) return ORBG_CAP_EXCEEDED;
    if (c.kps_out[cam]) memcpy(c.kps_out[cam], h->h_kps.h + base, (size_t)nk * sizeof(orbx_keypoint));
    if (c.desc_out[cam]) memcpy(c.desc_out[cam], h->h_desc.h + (size_t)base * 32, (size_t)nk * 32);
    base += nk;
  }
  if (post && post->frame) orbm_internal_set_n(post->frame, h->n_kp[0]);
  if (post && post->kps_un_out) {
    if (h->n_kp[0] > c.cap[0]) return ORBG_CAP_EXCEEDED;
    memcpy(post->kps_un_out, (post->undist && post->frame) ? h->h_kps_un.h : h->h_kps.h, (size_t)h->n_kp[0] * sizeof(orbx_keypoint));
  }
  if (c.stereo_out && h->n_kp[0] > c.cap[0]) return ORBG_CAP_EXCEEDED;      // uright / depth hold cap_left entries like the other outputs
  if (c.stereo_out) {
    if (post->uright) memcpy(post->uright, h->h_stereo.h, (size_t)h->n_kp[0] * 4);
    if (post->depth) memcpy(post->depth, h->h_stereo.h + h->n_kp[0], (size_t)h->n_kp[0] * 4);
  }
  float ms;
  h->timings[2] = 0;
  if (c.prof >= 1 && hipEventElapsedTime(&ms, h->ev[1], h->ev[7]) == hipSuccess) { h->timings[5] = ms; h->fast_ms_sum += ms; h->fast_ms_n++; }
  return ORBG_OK;
}

// defer_finalize: do not launch the median rejection; hand its arguments back (it then runs next to the grid build)
static int launch_stereo(orbx_handle* h, float bf, float b, hipStream_t st, bool device_counts, float* host_mirror,
                         StereoFinalizeArgs* defer_finalize) {
  const int nl = device_counts ? h->sel_bound : h->n_kp[0], nr = h->n_kp[1];
  // the stereo key packs the right keypoint's index into 16 bits (dist << 16 | iR)
  if (nl >= ORBG_MAX_FRAME_FEATURES || nr >= ORBG_MAX_FRAME_FEATURES || (device_counts && orbx_internal_kp_capacity(h) >= 2 * ORBG_MAX_FRAME_FEATURES))
    return ORBG_CAP_EXCEEDED;
  if (h->profile >= 2) ORBG_HIP(hipEventRecord(h->ev[5], st));
  if (nl > 0) {
    const int* dn = device_counts ? h->d_nkp.p : nullptr;
    hipLaunchKernelGGL(stereo_match_kernel, dim3((nl + 3) / 4), dim3(256), 0, st, h->d_pyr.p, h->geom, h->d_kps.p, h->d_desc.p,
                       nl, h->d_kps.p + nl, h->d_desc.p + (size_t)nl * 32, nr, bf, b, h->d_uright.p, h->d_depth.p, h->d_sad.p, dn);
    if (defer_finalize) *defer_finalize = StereoFinalizeArgs{h->d_uright.p, h->d_depth.p, h->d_sad.p, nl, dn, host_mirror, nullptr};
    else hipLaunchKernelGGL(stereo_finalize_kernel, dim3(1), dim3(256), 0, st, h->d_uright.p, h->d_depth.p, h->d_sad.p, nl, dn, host_mirror);
  }
  if (h->profile >= 2) ORBG_HIP(hipEventRecord(h->ev[6], st));
  return ORBG_OK;
}

extern "C" int orbx_stereo_match(orbx_handle* h, float bf, float b, float* uright, float* depth) {
  if (!h || h->cfg.n_cams != 2 || h->cur_w == 0) return ORBG_BAD_ARG;
  int rc = select_device(h->device);
  if (rc) return rc;
  const int nl = h->n_kp[0];
  hipStream_t st = h->stream;
  if ((rc = launch_stereo(h, bf, b, st, false, nullptr))) return rc;
  if (nl > 0 && (uright || depth)) {
    ORBG_HIP(hipMemcpyAsync(h->h_stereo.h, h->d_uright.p, (size_t)nl * 4, hipMemcpyDeviceToHost, st));
    ORBG_HIP(hipMemcpyAsync(h->h_stereo.h + nl, h->d_depth.p, (size_t)nl * 4, hipMemcpyDeviceToHost, st));
  }
  ORBG_HIP(hipStreamSynchronize(st));
  if (nl > 0 && uright) memcpy(uright, h->h_stereo.h, (size_t)nl * 4);
  if (nl > 0 && depth) memcpy(depth, h->h_stereo.h + nl, (size_t)nl * 4);
  float ms;
  if (h->profile >= 2 && hipEventElapsedTime(&ms, h->ev[5], h->ev[6]) == hipSuccess) h->timings[4] = ms;
  return ORBG_OK;
}

// level-1 brackets on every `interval`-th extraction; reset = 1 clears the accumulated fast_cells_kernel statistics
extern "C" int orbx_set_profile_kernel(orbx_handle* h, int which) {
  if (!h || which < ORBX_PROF_FAST || which > ORBX_PROF_PYRAMID) return ORBG_BAD_ARG;
  h->prof_kernel = which;
  h->fast_ms_sum = 0; h->fast_ms_n = 0;
  return ORBG_OK;
}

extern "C" int orbx_set_profile_interval(orbx_handle* h, int interval, int reset) {
  if (!h || interval < 1) return ORBG_BAD_ARG;
  h->profile_interval = interval;
  if (reset) { h->fast_ms_sum = 0; h->fast_ms_n = 0; h->extract_calls = 0; }
  return ORBG_OK;
}

extern "C" int orbx_get_fast_kernel_stats(orbx_handle* h, double* sum_ms, int64_t* n) {
  if (!h || !sum_ms || !n) return ORBG_BAD_ARG;
  *sum_ms = h->fast_ms_sum; *n = (int64_t)h->fast_ms_n;
  return ORBG_OK;
}

extern "C" int orbx_set_profiling(orbx_handle* h, int level) {
  if (!h || level < 0 || level > 2) return ORBG_BAD_ARG;
  h->profile = level;
  return ORBG_OK;
}

// Cost of an empty hipEventRecord pair on the handle's stream (what the profiling brackets add to a measured kernel time)
extern "C" int orbx_event_overhead(orbx_handle* h, int reps, float* ms) {
  if (!h || !ms || reps < 1) return ORBG_BAD_ARG;
  int rc = select_device(h->device);
  if (rc) return rc;
  double acc = 0;
  for (int i = 0; i < reps; i++) {
    ORBG_HIP(hipEventRecord(h->ev[1], h->stream));
    ORBG_HIP(hipEventRecord(h->ev[7], h->stream));
    ORBG_HIP(hipStreamSynchronize(h->stream));
    float e = 0;
    ORBG_HIP(hipEventElapsedTime(&e, h->ev[1], h->ev[7]));
    acc += e;
  }
  *ms = (float)(acc / reps);
  return ORBG_OK;
}

extern "C" int orbx_get_timings(orbx_handle* h, float* ms) {
  if (!h || !ms) return ORBG_BAD_ARG;
  for (int i = 0; i < 8; i++) ms[i] = h->timings[i];
  return ORBG_OK;
}

// accessors for matcher.hip (device-resident hand-over, same shared object)
extern "C++" {
// capacity (keypoints, both cameras) of the device feature buffers: the bound of any device-side keypoint count
int orbx_internal_kp_capacity(orbx_handle* h) { return h ? (int)std::min<size_t>(h->d_kps.cap, 0x7fffffff) : 0; }

int orbx_internal_left_features(orbx_handle* h, const orbx_keypoint** d_kps, const uint8_t** d_desc, const float** d_uright,
                                const float** d_depth, const orbx_keypoint** h_kps, int* n, hipStream_t* stream) {
  if (!h) return ORBG_BAD_ARG;
  *d_kps = h->d_kps.p; *d_desc = h->d_desc.p; *d_uright = h->d_uright.p; *d_depth = h->d_depth.p; *n = h->n_kp[0];
  *h_kps = h->h_kps.h;
  *stream = h->stream;
  return ORBG_OK;
}
}
