// liborbgpu -- Hamming matchers for gfx950 (MI355X): feature grid, isInFrustum, SearchByProjection (map and
// frame variants) and SearchByBoW.  Replaces the ORBmatcher searches of S/ORBmatcher.cc and the Frame helpers
// of S/Frame.cc behind the C-ABI of include/orbgpu.h.
//
// Design: frame features (keypoints, descriptors, uRight, 64x48 CSR grid) and the local map stay resident in HBM;
// one 64-lane wavefront serves one query (map point / last-frame point / keyframe feature):
//   * the query's window of grid cells (or its BoW node bucket) is enumerated in the reference's order, every
//     candidate gets its position `pos` in that order;
//   * the 256-bit descriptors are compared with v_popc on 8 dwords; each lane keeps its two smallest
//     (dist<<20 | pos) keys and a butterfly merge over the wavefront yields the best / second best --
//     identical to the reference's sequential strict-'<' scan (tests pin the equivalence, ties included);
//   * the reference's matchers are greedy and order dependent (a feature claimed by an earlier query is skipped
//     by later ones, S/ORBmatcher.cc:89-91,324-325,2045-2047).  Kernels evaluate all queries against the state at
//     entry and also emit each query's (feature, distance) candidate list into mapped pinned memory; the host
//     commits results in the reference's serial order and re-scans a query's list only if its best / second
//     best feature has been claimed meanwhile.  Results are identical to the serial loop.
// Float conventions mirror the oracle's documented OpenCV small-matrix rules (see oracle/matching.cc header).
// Compile with -ffp-contract=off.

#include "common.hpp"
#include "wave.hpp"
#include "stereo_finalize.hpp"

#include <algorithm>
#include <cmath>

using namespace orbg;

int orbx_internal_left_features(orbx_handle* h, const orbx_keypoint** d_kps, const uint8_t** d_desc, const float** d_uright,
                                const float** d_depth, const orbx_keypoint** h_kps, int* n, hipStream_t* stream);

#include "grid_build.hpp"

namespace {

constexpr int TH_HIGH = 100;      // S/ORBmatcher.cc:36
constexpr int TH_LOW = 50;        // :37
constexpr int HISTO_LENGTH = 30;  // :38
struct PoseF {   // Tcw split as the reference does (S/Frame.cc:439-445)
  float R[9], t[3], Ow[3];
};

constexpr unsigned short kQCountMask = 0x7FFF;   // QResult.count: entries of the list segment (capped) ...
constexpr unsigned short kQVisible = 0x8000;     // ... and, from search_local_kernel, "isInFrustum() returned true for this point"
struct QResult {   // per query, written by the kernel into mapped pinned memory (24 B)
  unsigned base;            // its segment of the candidate list (device memory; fetched only if the top-4 cannot decide)
  unsigned short count;     // entries in the segment, filtered ones included (& kQCountMask); bit 15: kQVisible
  unsigned short n_top;     // valid entries below (4 = there may be more candidates than listed here)
  unsigned short idx[4];    // the four best candidates in the order a sequential strict-'<' scan ranks them
  unsigned short dist[4];
};

__device__ __forceinline__ int popc256(const uint4 a0, const uint4 a1, const uint4 b0, const uint4 b1) {
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__device__ __forceinline__ void pose_map(const PoseF& P, const float* X, float* out) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float t0 = P.R[3 * i] * X[0] + P.R[3 * i + 1] * X[1] + P.R[3 * i + 2] * X[2];
    out[i] = t0 + P.t[i];
  }
}

__device__ __forceinline__ float norm3d(const float* v) {
  const double s = (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2];
  return (float)sqrt(s);
}

// ------------------------------------------------------------------------------------------------
// grid  (Frame::AssignFeaturesToGrid / PosInGrid, S/Frame.cc:360-391,699-709); CSR, cell = ix*48+iy

// done_flag != nullptr: the Frame constructor's completion word.  The last kernel of the chain posts it itself: everything the
// earlier kernels wrote for the host is complete at their end, and nothing this kernel writes is read by the host.
__global__ __launch_bounds__(1024) void grid_build_kernel(const orbx_keypoint* __restrict__ kps, FrameParams fp,
                                                         int* __restrict__ cell_of, int* __restrict__ cell_start,
                                                         int* __restrict__ cell_items, const int* __restrict__ d_n,
                                                         volatile unsigned* done_flag, unsigned done_seq) {
  grid_build_body<1024>(kps, fp, cell_of, cell_start, cell_items, d_n);
  if (done_flag) {
    // The host may act on the word BEFORE this kernel has ended (it launches the frame's searches on another stream): every
    // wavefront's grid stores must have reached the L2 (vmcnt) and the L2 must have been written back (one agent-scope release by the
    // posting thread: a search on another XCD reads the grid through ITS L2) before the word is written.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      *done_flag = done_seq;
    }
  }
}

// Last launch of the MONOCULAR Frame constructor for a distorted camera (orbx_frame_mono*): Frame::UndistortKeyPoints (S/Frame.cc:721-754)
// and AssignFeaturesToGrid (:344) in one workgroup -- a thread undistorts exactly the keypoints whose cells it then computes.  The
// undistorted records also go to mapped pinned memory (mvKeysUn for the caller), so every wavefront releases to system scope before
// the completion word is posted.
__global__ __launch_bounds__(1024) void undistort_grid_kernel(orbg::UndistortArgs ua, FrameParams fp, int* __restrict__ cell_of,
                                                             int* __restrict__ cell_start, int* __restrict__ cell_items,
                                                             const int* __restrict__ d_n, volatile unsigned* done_flag, unsigned done_seq) {
  const int n = d_n ? *d_n : fp.n;
  for (int i = threadIdx.x; i < n; i += 1024) {
    orbx_keypoint k = ua.src[i];
    float xu, yu;
    orbg::undistort_point(ua, k.x, k.y, &xu, &yu);
    k.x = xu; k.y = yu;
    ua.dst[i] = k;
    if (ua.dst_host) ua.dst_host[i] = k;
  }
  // (grid_build_body's thread t reads the records t, t + 1024, ... -- the ones it has just written)
  grid_build_body<1024>(ua.dst, fp, cell_of, cell_start, cell_items, d_n);
  if (done_flag) {
    if (ua.dst_host) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");           // mvKeysUn for the host: system scope, every wavefront
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (see grid_build_kernel)
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      *done_flag = done_seq;
    }
  }
}

// Last launch of the stereo Frame constructor: workgroup 0 builds the grid, workgroup 1 (its first four wavefronts; the
// others leave at once) runs the median rejection of ComputeStereoMatches -- the two do not depend on each other, and as two
// launches they cost the tracking thread a submission and the stream a kernel boundary more.  The rejection mirrors its
// result into pinned host memory, so its wavefronts release to system scope before they take the ticket; the workgroup
// that takes it second posts the constructor's completion word.
__global__ __launch_bounds__(1024) void grid_build_finalize_kernel(const orbx_keypoint* __restrict__ kps, FrameParams fp,
                                                                  int* __restrict__ cell_of, int* __restrict__ cell_start,
                                                                  int* __restrict__ cell_items, const int* __restrict__ d_n,
                                                                  volatile unsigned* done_flag, unsigned done_seq, StereoFinalizeArgs fin) {
  if (blockIdx.x == 0) {
    grid_build_body<1024>(kps, fp, cell_of, cell_start, cell_items, d_n);
    // every wavefront's grid stores have reached the L2 before thread 0's agent-scope release (the ticket below) writes it back: the
    // host may launch the frame's searches -- on another stream, possibly another XCD -- as soon as it sees the completion word
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    if (threadIdx.x >= 256) return;              // (whole wavefronts: a barrier only counts wavefronts that are still alive)
    stereo_finalize_body(fin.uright, fin.depth, fin.best_sad, fin.nl, fin.d_nkp, fin.host_out);
    // system-scope RELEASE (write-back + wait), not __threadfence_system(): the acquire half of a full fence invalidates the XCD's L2
    // under every kernel running next to this one (the local BA's: 0.583 -> 0.575 ms per solve with the invalidations gone)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned before = __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (before == 1u) {
      __hip_atomic_store(fin.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (done_flag) *done_flag = done_seq;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// isInFrustum (S/Frame.cc:466-543, Nleft == -1) + MapPoint::PredictScale (S/MapPoint.cc:646-661)

struct TrackFields { int in_view; float px, py, pxr, depth, view_cos; int level; };

__device__ __forceinline__ TrackFields frustum_check(const FrameParams& fp, const PoseF& P, const float* X, const float* Pn,
                                                     float min_raw, float max_raw, float limit) {
  TrackFields f;
  f.in_view = 0; f.px = -1.f; f.py = -1.f; f.pxr = 0.f; f.depth = 0.f; f.view_cos = 0.f; f.level = 0;
  float Pc[3];
  pose_map(P, X, Pc);
  const float Pc_dist = norm3d(Pc);
  const float PcZ = Pc[2];
  const float invz = 1.0f / PcZ;
  if (PcZ < 0.0f) return f;
  const float u = fp.fx * Pc[0] / Pc[2] + fp.cx;
  const float v = fp.fy * Pc[1] / Pc[2] + fp.cy;
  if (u < fp.min_x || u > fp.max_x) return f;
  if (v < fp.min_y || v > fp.max_y) return f;
  f.px = u; f.py = v;
  const float maxDistance = 1.2f * max_raw, minDistance = 0.8f * min_raw;
  const float PO[3] = {X[0] - P.Ow[0], X[1] - P.Ow[1], X[2] - P.Ow[2]};
  const float dist = norm3d(PO);
  if (dist < minDistance || dist > maxDistance) return f;
  const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
  const float viewCos = (float)(dot / (double)dist);
  if (viewCos < limit) return f;
  const float ratio = max_raw / dist;
  // logf(ratio): correctly rounded via the f64 log (matches glibc logf wherever that is correctly rounded)
  const float lg = (float)log((double)ratio);
  int nScale = (int)ceilf(lg / fp.log_sf);
  if (nScale < 0) nScale = 0;
  else if (nScale >= fp.n_levels) nScale = fp.n_levels - 1;
  f.in_view = 1;
  f.pxr = u - fp.bf * invz;
  f.depth = Pc_dist;
  f.level = nScale;
  f.view_cos = viewCos;
  return f;
}

struct WorldPtsDev {
  int m;
  const float* pos; const float* normal; const float* min_dist; const float* max_dist;
  const uint8_t* desc; const uint8_t* bad; const uint8_t* skip;
};

struct TrackDev {   // SoA track fields on the device
  uint8_t* in_view; float* px; float* py; float* pxr; float* depth; int* level; float* view_cos;
};

__global__ __launch_bounds__(256) void frustum_kernel(FrameParams fp, PoseF P, WorldPtsDev w, float limit, TrackDev t) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= w.m) return;
  const float X[3] = {w.pos[3 * i], w.pos[3 * i + 1], w.pos[3 * i + 2]};
  const float N[3] = {w.normal[3 * i], w.normal[3 * i + 1], w.normal[3 * i + 2]};
  const TrackFields f = frustum_check(fp, P, X, N, w.min_dist[i], w.max_dist[i], limit);
  t.in_view[i] = (uint8_t)f.in_view; t.px[i] = f.px; t.py[i] = f.py; t.pxr[i] = f.pxr; t.depth[i] = f.depth;
  t.level[i] = f.level; t.view_cos[i] = f.view_cos;
}

// ---- two-camera rig (Frame::Nleft != -1): Frame::isInFrustumChecks for either camera (S/Frame.cc:1154-1231)
struct RigCamF { int model; float fx, fy, cx, cy, k[4]; };
struct RigSideF { float R[9], t[3], twc[3]; RigCamF cam; };   // mR, mt, twc of :1158-1170 and the camera the side projects through
static RigCamF rig_cam_of(const orbg_camera& c) {
  RigCamF r;
  r.model = c.model; r.fx = c.fx; r.fy = c.fy; r.cx = c.cx; r.cy = c.cy;
  for (int i = 0; i < 4; i++) r.k[i] = c.k[i];
  return r;
}

// GeometricCamera::project(cv::Mat) -> project(cv::Point3f): Pinhole.cpp:41-47, KannalaBrandt8.cpp:28-44 (float32 throughout; the
// float32 atan2 / cos / sin are taken as the rounded float64 ones)
__device__ __forceinline__ void rig_project(const RigCamF& c, const float* p, float* uv) {
  if (c.model == ORBG_CAM_KANNALA_BRANDT8) {
    const float x2_plus_y2 = p[0] * p[0] + p[1] * p[1];
    const float theta = (float)atan2((double)sqrtf(x2_plus_y2), (double)p[2]);
    const float psi = (float)atan2((double)p[1], (double)p[0]);
    const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
    const float r = theta + c.k[0] * theta3 + c.k[1] * theta5 + c.k[2] * theta7 + c.k[3] * theta9;
    uv[0] = c.fx * r * (float)cos((double)psi) + c.cx;
    uv[1] = c.fy * r * (float)sin((double)psi) + c.cy;
  } else {
    uv[0] = c.fx * p[0] / p[2] + c.cx;
    uv[1] = c.fy * p[1] / p[2] + c.cy;
  }
}

__device__ __forceinline__ TrackFields rig_frustum_check(const FrameParams& fp, const RigSideF& S, const float* X, const float* Pn,
                                                         float min_raw, float max_raw, float limit) {
  TrackFields f;
  f.in_view = 0; f.px = 0.f; f.py = 0.f; f.pxr = 0.f; f.depth = 0.f; f.view_cos = 0.f; f.level = -1;
  float Pc[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float t0 = S.R[3 * i] * X[0] + S.R[3 * i + 1] * X[1] + S.R[3 * i + 2] * X[2];
    Pc[i] = t0 + S.t[i];
  }
  const float Pc_dist = norm3d(Pc);
  if (Pc[2] < 0.0f) return f;
  float uv[2];
  rig_project(S.cam, Pc, uv);
  if (uv[0] < fp.min_x || uv[0] > fp.max_x) return f;
  if (uv[1] < fp.min_y || uv[1] > fp.max_y) return f;
  const float maxDistance = 1.2f * max_raw, minDistance = 0.8f * min_raw;
  const float PO[3] = {X[0] - S.twc[0], X[1] - S.twc[1], X[2] - S.twc[2]};
  const float dist = norm3d(PO);
  if (dist < minDistance || dist > maxDistance) return f;
  const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
  const float viewCos = (float)(dot / (double)dist);
  if (viewCos < limit) return f;
  const float ratio = max_raw / dist;
  const float lg = (float)log((double)ratio);
  int nScale = (int)ceilf(lg / fp.log_sf);
  if (nScale < 0) nScale = 0;
  else if (nScale >= fp.n_levels) nScale = fp.n_levels - 1;
  f.in_view = 1; f.px = uv[0]; f.py = uv[1]; f.depth = Pc_dist; f.level = nScale; f.view_cos = viewCos;
  return f;
}

__global__ __launch_bounds__(256) void frustum_rig_kernel(FrameParams fp, RigSideF L, RigSideF R, int has_right, WorldPtsDev w, float limit, TrackDev tl, TrackDev tr) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= w.m) return;
  const float X[3] = {w.pos[3 * i], w.pos[3 * i + 1], w.pos[3 * i + 2]};
  const float N[3] = {w.normal[3 * i], w.normal[3 * i + 1], w.normal[3 * i + 2]};
  const float mn = w.min_dist[i], mx = w.max_dist[i];
  const TrackFields a = rig_frustum_check(fp, L, X, N, mn, mx, limit);
  tl.in_view[i] = (uint8_t)a.in_view; tl.px[i] = a.px; tl.py[i] = a.py; tl.depth[i] = a.depth; tl.level[i] = a.level; tl.view_cos[i] = a.view_cos;
  if (!has_right) return;                                   // a single camera behind a model (Nleft == -1, mpCamera a fisheye)
  const TrackFields b = rig_frustum_check(fp, R, X, N, mn, mx, limit);
  tr.in_view[i] = (uint8_t)b.in_view; tr.px[i] = b.px; tr.py[i] = b.py; tr.depth[i] = b.depth; tr.level[i] = b.level; tr.view_cos[i] = b.view_cos;
}

// ------------------------------------------------------------------------------------------------
// window search: one wavefront per query

struct Top4 { unsigned k[4]; int i[4]; };   // keys = dist<<20 | pos, ascending; 0xFFFFFFFF = empty

__device__ __forceinline__ void top4_init(Top4& t) {
#pragma unroll
  for (int j = 0; j < 4; j++) { t.k[j] = 0xFFFFFFFFu; t.i[j] = -1; }
}

__device__ __forceinline__ void top4_insert(Top4& t, unsigned key, int idx) {
  if (key >= t.k[3]) return;
#pragma unroll
  for (int j = 3; j >= 0; j--) {
    const bool shift = j > 0 && key < t.k[j - 1];
    if (shift) { t.k[j] = t.k[j - 1]; t.i[j] = t.i[j - 1]; }
    else { t.k[j] = key; t.i[j] = idx; break; }
  }
}

// The four smallest keys over the wavefront (each lane holds its own four smallest): four rounds of "wave minimum of the
// lane heads, owner pops".  Keys are unique (pos is), so the owner is unique.  All 64 lanes must be active.
__device__ __forceinline__ void top4_wave_emit(Top4& t, QResult& res) {
  res.n_top = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    res.idx[r] = 0xFFFF; res.dist[r] = 256;
    const unsigned m = wave_min(t.k[0]);
    if (m != 0xFFFFFFFFu) {
      const bool own = t.k[0] == m;
      const int owner = __ffsll((unsigned long long)__ballot(own)) - 1;
      const int idx = __builtin_amdgcn_readlane(t.i[0], owner);
      res.idx[r] = (unsigned short)idx; res.dist[r] = (unsigned short)(m >> 20);
      res.n_top = (unsigned short)(r + 1);
      if (own) {
        t.k[0] = t.k[1]; t.i[0] = t.i[1]; t.k[1] = t.k[2]; t.i[1] = t.i[2]; t.k[2] = t.k[3]; t.i[2] = t.i[3];
        t.k[3] = 0xFFFFFFFFu; t.i[3] = -1;
      }
    }
  }
}

constexpr int kOccBits = 4096;
struct FrameDev {
  const orbx_keypoint* kps; const uint8_t* desc; const float* uright;   // uright may be NULL
  const int* cell_start; const int* cell_items;
  // "feature already holds a map point" at entry: for frames of up to kOccBits features the flags travel as a bitmask
  // inside the kernel arguments (no upload, no copy command); larger frames read the two arrays from device memory
  const int* assigned_mp; const int* assigned_obs;
  int use_mask;
  uint32_t occ[kOccBits / 32];
};

__device__ __forceinline__ bool feature_occupied(const FrameDev& F, int idx) {
  if (F.use_mask) return (F.occ[idx >> 5] >> (idx & 31)) & 1u;
  return F.assigned_mp[idx] >= 0 && (!F.assigned_obs || F.assigned_obs[idx] > 0);
}

struct Query { int valid; float x, y, r; int min_level, max_level; float ur_ref; };

// GetFeaturesInArea (S/Frame.cc:628-697) + the candidate loop of the projection searches.
// Emits the candidate list [base, base+count) (entry = idx | dist<<16, or 0xFFFFFFFF when filtered).
constexpr int kSlot = 16;
constexpr int kListStage = 64;

struct QDesc { uint4 a0, a1; };
// a query's descriptor: requested at the top of a kernel, together with its other fields (for the frame search they sit in
// pinned host memory: every dependent access there is a ~2 us PCIe round trip)
__device__ __forceinline__ QDesc load_qdesc(const uint8_t* p) {
  QDesc d;
  d.a0 = *reinterpret_cast<const uint4*>(p); d.a1 = *reinterpret_cast<const uint4*>(p + 16);
  return d;
}

__device__ __forceinline__ void window_search(const FrameParams& fp, const FrameDev& F, const Query& q, const QDesc& qd,
                                              int qid, int n_queries, int* __restrict__ list_counter,
                                              uint32_t* __restrict__ list, int list_cap, QResult* __restrict__ out,
                                              uint32_t* __restrict__ s_list /*LDS, kListStage entries of this wavefront*/,
                                              unsigned short flags = 0) {
  const int lane = threadIdx.x & 63;
  QResult res;
  res.base = 0; res.count = flags; res.n_top = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) { res.idx[j] = 0xFFFF; res.dist[j] = 256; }
  bool empty = !q.valid;
  int nMinCellX = 0, nMaxCellX = -1, nMinCellY = 0, nMaxCellY = -1;
  if (!empty) {
    nMinCellX = max(0, (int)floorf((q.x - fp.min_x - q.r) * fp.w_inv));
    nMaxCellX = min(ORBG_GRID_COLS - 1, (int)ceilf((q.x - fp.min_x + q.r) * fp.w_inv));
    nMinCellY = max(0, (int)floorf((q.y - fp.min_y - q.r) * fp.h_inv));
    nMaxCellY = min(ORBG_GRID_ROWS - 1, (int)ceilf((q.y - fp.min_y + q.r) * fp.h_inv));
    if (nMinCellX >= ORBG_GRID_COLS || nMaxCellX < 0 || nMinCellY >= ORBG_GRID_ROWS || nMaxCellY < 0) empty = true;
  }
  if (empty) {
    if (lane == 0) *out = res;
    return;
  }
  const int ncy = nMaxCellY - nMinCellY + 1, ncx = nMaxCellX - nMinCellX + 1;
  const int ncell = ncx * ncy;
  // pass 1: total number of items in the window (upper bound of the list length)
  int total = 0;
  for (int c = lane; c < ncell; c += 64) {
    const int ix = nMinCellX + c / ncy, iy = nMinCellY + c % ncy;
    const int cell = ix * ORBG_GRID_ROWS + iy;
    total += F.cell_start[cell + 1] - F.cell_start[cell];
  }
  total = wave_sum(total);
  if (total == 0) {
    if (lane == 0) *out = res;
    return;
  }
  // short lists (the common case) live in a fixed slot of kSlot entries per query; only longer ones take a segment of
  // the shared overflow region behind the slots -- one device-scope atomic word saturates at ~88 ops/us on MI355X.
  int base = qid * kSlot;
  if (total > kSlot) {
    if (lane == 0) base = n_queries * kSlot + atomicAdd(list_counter, total);
    base = __shfl(base, 0, 64);
  }
  const uint4 a0 = qd.a0, a1 = qd.a1;
  const bool bCheckLevels = (q.min_level > 0) || (q.max_level >= 0);
  Top4 t;
  top4_init(t);
  int run = 0;   // candidates before the current chunk of 64 cells
  for (int c0 = 0; c0 < ncell; c0 += 64) {
    const int c = c0 + lane;
    int s = 0, n = 0;
    if (c < ncell) {
      const int ix = nMinCellX + c / ncy, iy = nMinCellY + c % ncy;
      const int cell = ix * ORBG_GRID_ROWS + iy;
      s = F.cell_start[cell];
      n = F.cell_start[cell + 1] - s;
    }
    const int inc = wave_incl_scan_add(n);
    const int chunk_total = __builtin_amdgcn_readlane(inc, 63);
    const int my_pos0 = run + inc - n;
    const int nmax = wave_max(n);
    // The candidate loop is a chain of dependent loads (item -> keypoint -> right coordinate -> descriptor, ~0.3 us each from
    // the L2): the next item's index is fetched one step ahead, and a candidate's keypoint, right coordinate and descriptor
    // are requested together (the descriptor of a candidate the gates reject is then simply not used)
    int idx_next = n > 0 ? F.cell_items[s] : 0;
    for (int j = 0; j < nmax; j++) {
      const int idx = idx_next;
      if (j + 1 < n) idx_next = F.cell_items[s + j + 1];
      if (j < n) {
        const int pos = my_pos0 + j;
        const orbx_keypoint kp = F.kps[idx];
        const float ur_c = F.uright ? F.uright[idx] : 0.f;
        const uint4 b0 = *reinterpret_cast<const uint4*>(F.desc + (size_t)idx * 32);
        const uint4 b1 = *reinterpret_cast<const uint4*>(F.desc + (size_t)idx * 32 + 16);
        bool ok = true;
        if (bCheckLevels) {
          if (kp.octave < q.min_level) ok = false;
          if (q.max_level >= 0 && kp.octave > q.max_level) ok = false;
        }
        const float distx = kp.x - q.x, disty = kp.y - q.y;
        if (!(fabsf(distx) < q.r && fabsf(disty) < q.r)) ok = false;
        if (ok && feature_occupied(F, idx)) ok = false;
        if (ok && F.uright) {
          const float ur = ur_c;
          if (ur > 0) {
            const float er = fabsf(q.ur_ref - ur);
            if (er > q.r) ok = false;
          }
        }
        unsigned entry = 0xFFFFFFFFu;
        if (ok) {
          const int d = popc256(a0, a1, b0, b1);
          top4_insert(t, ((unsigned)d << 20) | (unsigned)pos, idx);
          entry = (unsigned)idx | ((unsigned)d << 16);
        }
        // the head of the list is staged in LDS and leaves as ONE contiguous burst (the list lives in host memory)
        if (pos < kListStage) s_list[pos] = entry;
        else if (base + pos < list_cap) list[base + pos] = entry;
      }
    }
    run += chunk_total;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < min(total, kListStage) && base + lane < list_cap) list[base + lane] = s_list[lane];
  top4_wave_emit(t, res);
  if (lane == 0) {
    res.base = (unsigned)base; res.count = (unsigned short)(min(total, (int)kQCountMask) | flags);
    *out = res;
  }
}

// MODE 0: SearchByProjection(Frame, vector<MapPoint*>) with track fields given (S/ORBmatcher.cc:44-143)
struct MpsDev {
  int m;
  const uint8_t* in_view; const uint8_t* bad; const float* px; const float* py; const float* pxr; const float* depth;
  const int* level; const float* view_cos; const uint8_t* desc;
};

__global__ __launch_bounds__(256) void search_mps_kernel(FrameParams fp, FrameDev F, MpsDev mp, float th, int far_points,
                                                        float th_far, int* list_counter, int* counter_next, uint32_t* list, int list_cap,
                                                        QResult* results) {
  __shared__ uint32_t s_stage[4][kListStage];
  if (blockIdx.x == 0 && threadIdx.x == 0) *counter_next = 0;   // the overflow counter the NEXT search on this frame will use
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= mp.m) return;
  Query q;
  q.valid = mp.in_view[i] && !(far_points && mp.depth[i] > th_far) && !mp.bad[i];
  q.x = q.y = q.r = 0; q.min_level = q.max_level = 0; q.ur_ref = 0;
  if (q.valid) {
    const int lvl = mp.level[i];
    float r = (mp.view_cos[i] > 0.998) ? 2.5f : 4.0f;      // RadiusByViewingCos, :216-222 (double literal compare)
    if (th != 1.0) r *= th;
    q.x = mp.px[i]; q.y = mp.py[i];
    q.r = r * fp.scale[lvl];
    q.min_level = lvl - 1; q.max_level = lvl;
    q.ur_ref = mp.pxr[i];
  }
  window_search(fp, F, q, load_qdesc(mp.desc + (size_t)i * 32), i, mp.m, list_counter, list, list_cap, results + i, s_stage[threadIdx.x >> 6]);
}


// Tracking::SearchLocalPoints in one launch: isInFrustum(pMP, 0.5) (S/Frame.cc:466-543) for the wavefront's own point,
// then the query of search_mps_kernel -- the track fields never leave registers.
__global__ __launch_bounds__(256) void search_local_kernel(FrameParams fp, FrameDev F, WorldPtsDev w, const uint8_t* __restrict__ skip_call,
                                                          PoseF P, float th, int far_points, float th_far, int* list_counter,
                                                          int* counter_next, uint32_t* list, int list_cap, QResult* results) {
  __shared__ uint32_t s_stage[4][kListStage];
  if (blockIdx.x == 0 && threadIdx.x == 0) *counter_next = 0;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= w.m) return;
  Query q;
  q.valid = 0; q.x = q.y = q.r = 0; q.min_level = q.max_level = 0; q.ur_ref = 0;
  // the point's fields and its descriptor are requested together (flags -> position -> distances -> descriptor were dependent)
  const uint8_t w_bad = w.bad[i], w_skip = w.skip[i], c_skip = skip_call ? skip_call[i] : (uint8_t)0;
  const float X[3] = {w.pos[3 * i], w.pos[3 * i + 1], w.pos[3 * i + 2]};
  const float N[3] = {w.normal[3 * i], w.normal[3 * i + 1], w.normal[3 * i + 2]};
  const float w_min = w.min_dist[i], w_max = w.max_dist[i];
  const QDesc qd = load_qdesc(w.desc + (size_t)i * 32);
  unsigned short vis = 0;
  if (!(w_bad || w_skip || c_skip)) {
    const TrackFields t = frustum_check(fp, P, X, N, w_min, w_max, 0.5f);
    if (t.in_view) vis = kQVisible;                        // what Tracking::SearchLocalPoints counts (IncreaseVisible, nToMatch)
    if (t.in_view && !(far_points && t.depth > th_far)) {
      float r = (t.view_cos > 0.998) ? 2.5f : 4.0f;        // RadiusByViewingCos, S/ORBmatcher.cc:216-222
      if (th != 1.0) r *= th;
      q.valid = 1;
      q.x = t.px; q.y = t.py;
      q.r = r * fp.scale[t.level];
      q.min_level = t.level - 1; q.max_level = t.level;
      q.ur_ref = t.pxr;
    }
  }
  window_search(fp, F, q, qd, i, w.m, list_counter, list, list_cap, results + i, s_stage[threadIdx.x >> 6], vis);
}

// ---- large local maps (a merged server map: tens of thousands of points, a few per cent of them in view).  One wavefront per
// map point -- what search_local_kernel does -- spends its time on points that fail isInFrustum (26 ns per point at 76 k points).
// Here the frustum test runs one THREAD per point (cull_count_kernel), the points that become queries are compacted in point
// order (cull_fill_kernel: slot -> point index, on the device and mirrored for the host), and the window search runs one
// wavefront per SLOT (search_culled_kernel).  The results are the ones search_local_kernel gives for the same points: the same
// frustum_check, the same window_search, the commit walks the slots in ascending point order.
__device__ __forceinline__ bool local_query_of(const FrameParams& fp, const WorldPtsDev& w, const uint8_t* __restrict__ skip_call,
                                               const PoseF& P, float th, int far_points, float th_far, int i, Query* q, bool* in_view) {
  q->valid = 0; q->x = q->y = q->r = 0; q->min_level = q->max_level = 0; q->ur_ref = 0;
  *in_view = false;
  const uint8_t w_bad = w.bad[i], w_skip = w.skip[i], c_skip = skip_call ? skip_call[i] : (uint8_t)0;
  const float X[3] = {w.pos[3 * i], w.pos[3 * i + 1], w.pos[3 * i + 2]};
  const float N[3] = {w.normal[3 * i], w.normal[3 * i + 1], w.normal[3 * i + 2]};
  const float w_min = w.min_dist[i], w_max = w.max_dist[i];
  if (w_bad || w_skip || c_skip) return false;
  const TrackFields t = frustum_check(fp, P, X, N, w_min, w_max, 0.5f);
  *in_view = t.in_view != 0;
  if (!(t.in_view && !(far_points && t.depth > th_far))) return false;
  float r = (t.view_cos > 0.998) ? 2.5f : 4.0f;            // RadiusByViewingCos, S/ORBmatcher.cc:216-222
  if (th != 1.0) r *= th;
  q->valid = 1;
  q->x = t.px; q->y = t.py;
  q->r = r * fp.scale[t.level];
  q->min_level = t.level - 1; q->max_level = t.level;
  q->ur_ref = t.pxr;
  return true;
}

// pass 1: per point "becomes a query" (flag byte, device) and "isInFrustum" (byte, host mirror, optional); per block the count
__global__ __launch_bounds__(256) void cull_count_kernel(FrameParams fp, WorldPtsDev w, const uint8_t* __restrict__ skip_call, PoseF P, float th,
                                                        int far_points, float th_far, uint8_t* __restrict__ qflag,
                                                        uint8_t* __restrict__ vis_host, int* __restrict__ blk_cnt) {
  __shared__ int wcnt[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool valid = false, vis = false;
  if (i < w.m) {
    Query q;
    valid = local_query_of(fp, w, skip_call, P, th, far_points, th_far, i, &q, &vis);
    qflag[i] = valid ? 1 : 0;
    if (vis_host) vis_host[i] = vis ? 1 : 0;
  }
  const int c = __popcll(__ballot(valid));
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

// pass 2: order-preserving compaction.  Block b's first slot = the counts of the blocks before it; inside the block by ballot rank.
__global__ __launch_bounds__(256) void cull_fill_kernel(int m, const uint8_t* __restrict__ qflag, const int* __restrict__ blk_cnt,
                                                       int* __restrict__ slot_pt, int* __restrict__ slot_pt_host, int* __restrict__ total_dev,
                                                       int* __restrict__ total_host) {
  __shared__ int red[4];
  __shared__ int wcnt[4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int sacc = 0;
  for (int j = threadIdx.x; j < b; j += 256) sacc += blk_cnt[j];
  sacc = wave_sum(sacc);
  const int i = b * 256 + threadIdx.x;
  const bool valid = i < m && qflag[i] != 0;
  const unsigned long long bal = __ballot(valid);
  if (lane == 0) { red[wave] = sacc; wcnt[wave] = __popcll(bal); }
  __syncthreads();
  const int base = red[0] + red[1] + red[2] + red[3];
  int woff = 0;
  for (int k = 0; k < wave; k++) woff += wcnt[k];
  if (valid) {
    const int s = base + woff + __popcll(bal & ((1ull << lane) - 1ull));
    slot_pt[s] = i;
    slot_pt_host[s] = i;
  }
  if (b == (int)gridDim.x - 1 && threadIdx.x == 0) {
    const int tot = base + wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    *total_dev = tot;
    *total_host = tot;
  }
}

// pass 3: one wavefront per compacted query (grid-stride over the slots); query id = slot
__global__ __launch_bounds__(256) void search_culled_kernel(FrameParams fp, FrameDev F, WorldPtsDev w, const uint8_t* __restrict__ skip_call, PoseF P,
                                                           float th, int far_points, float th_far, const int* __restrict__ slot_pt,
                                                           const int* __restrict__ total_p, int slot_cap, int* list_counter, int* counter_next,
                                                           uint32_t* list, int list_cap, QResult* results) {
  __shared__ uint32_t s_stage[4][kListStage];
  if (blockIdx.x == 0 && threadIdx.x == 0) *counter_next = 0;
  const int total = min(*total_p, slot_cap);
  const int nwaves = (int)gridDim.x * 4;
  for (int s = blockIdx.x * 4 + (threadIdx.x >> 6); s < total; s += nwaves) {
    const int i = slot_pt[s];
    Query q;
    bool vis;
    local_query_of(fp, w, skip_call, P, th, far_points, th_far, i, &q, &vis);
    const QDesc qd = load_qdesc(w.desc + (size_t)i * 32);
    window_search(fp, F, q, qd, s, slot_cap, list_counter, list, list_cap, results + s, s_stage[threadIdx.x >> 6], vis ? kQVisible : (unsigned short)0);
  }
}

// SearchByProjection(KeyFrame*, Scw, ...) candidate tests (S/ORBmatcher.cc:495-548 / :612-667) fused with the window search
// kModel: pKF->mpCamera->project (:515) is a camera model's (a fisheye keyframe) instead of the pinhole of the keyframe's view
template <bool kModel>
__global__ __launch_bounds__(256) void search_sim3_kernel(FrameParams fp, FrameDev F, WorldPtsDev w, const uint8_t* __restrict__ found,
                                                         PoseF P, int camera_project, int th, int* list_counter, int* counter_next, uint32_t* list,
                                                         int list_cap, QResult* results, RigCamF cam) {
  __shared__ uint32_t s_stage[4][kListStage];
  if (blockIdx.x == 0 && threadIdx.x == 0) *counter_next = 0;   // the overflow counter the NEXT search on this frame will use
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= w.m) return;
  Query q;
  q.valid = 0; q.x = q.y = q.r = 0; q.min_level = q.max_level = 0; q.ur_ref = 0;
  const QDesc qd = load_qdesc(w.desc + (size_t)i * 32);
  if (!(w.bad[i] || w.skip[i] || (found && found[i]))) {
    const float X[3] = {w.pos[3 * i], w.pos[3 * i + 1], w.pos[3 * i + 2]};
    float Pc[3];
    pose_map(P, X, Pc);
    if (!(Pc[2] < 0.0f)) {
      float u, v;
      if constexpr (kModel) {
        float uv[2]; rig_project(cam, Pc, uv); u = uv[0]; v = uv[1];
      } else if (camera_project) {
        u = fp.fx * Pc[0] / Pc[2] + fp.cx;
        v = fp.fy * Pc[1] / Pc[2] + fp.cy;
      } else {
        const float invz = 1.0f / Pc[2];
        const float x = Pc[0] * invz, y = Pc[1] * invz;
        u = fp.fx * x + fp.cx;
        v = fp.fy * y + fp.cy;
      }
      if (u >= fp.min_x && u < fp.max_x && v >= fp.min_y && v < fp.max_y) {           // KeyFrame::IsInImage
        const float max_raw = w.max_dist[i];
        const float maxDistance = 1.2f * max_raw, minDistance = 0.8f * w.min_dist[i];
        const float PO[3] = {X[0] - P.Ow[0], X[1] - P.Ow[1], X[2] - P.Ow[2]};
        const float dist = norm3d(PO);
        if (!(dist < minDistance || dist > maxDistance)) {
          const double dot = (double)PO[0] * w.normal[3 * i] + (double)PO[1] * w.normal[3 * i + 1] + (double)PO[2] * w.normal[3 * i + 2];
          if (!(dot < 0.5 * (double)dist)) {
            const float ratio = max_raw / dist;                                        // PredictScale(dist, pKF)
            const float lg = (float)log((double)ratio);
            int lvl = (int)ceilf(lg / fp.log_sf);
            if (lvl < 0) lvl = 0;
            else if (lvl >= fp.n_levels) lvl = fp.n_levels - 1;
            q.valid = 1;
            q.x = u; q.y = v;
            q.r = (float)th * fp.scale[lvl];
            q.min_level = lvl - 1; q.max_level = lvl;
          }
        }
      }
    }
  }
  window_search(fp, F, q, qd, i, w.m, list_counter, list, list_cap, results + i, s_stage[threadIdx.x >> 6]);
}

// SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist) -- the relocalisation overload,
// S/ORBmatcher.cc:2188-2310: one wavefront per feature of the keyframe.  No depth-sign test in front of Pinhole::project (:2214-2217),
// bounds inclusive on both sides (:2219-2222), distance range with the 0.8 / 1.2 invariance factors (:2228-2233), PredictScale on
// the FRAME's scale tables (:2235), window th * scale, levels nPredictedLevel - 1 .. + 1 (:2238-2240); a feature that holds ANY map
// point is not a candidate (:2246-2247: the occupancy the host stages has no "observations > 0" condition here).
// kModel: CurrentFrame.mpCamera->project is a camera model's (a monocular fisheye frame) instead of the pinhole of the frame view
template <bool kModel>
__global__ __launch_bounds__(256) void search_reloc_kernel(FrameParams fp, FrameDev F, WorldPtsDev w, const uint8_t* __restrict__ found,
                                                          PoseF P, float th, int* list_counter, int* counter_next, uint32_t* list,
                                                          int list_cap, QResult* results, RigCamF cam) {
  __shared__ uint32_t s_stage[4][kListStage];
  if (blockIdx.x == 0 && threadIdx.x == 0) *counter_next = 0;   // the overflow counter the NEXT search on this frame will use
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= w.m) return;
  Query q;
  q.valid = 0; q.x = q.y = q.r = 0; q.min_level = q.max_level = 0; q.ur_ref = 0;
  const uint8_t w_bad = w.bad[i], w_skip = w.skip[i], w_found = found ? found[i] : (uint8_t)0;
  const float X[3] = {w.pos[3 * i], w.pos[3 * i + 1], w.pos[3 * i + 2]};
  const float max_raw = w.max_dist[i], min_raw = w.min_dist[i];
  const QDesc qd = load_qdesc(w.desc + (size_t)i * 32);
  if (!(w_bad || w_skip || w_found)) {
    float Pc[3];
    pose_map(P, X, Pc);
    float u, v;
    if constexpr (kModel) { float uv[2]; rig_project(cam, Pc, uv); u = uv[0]; v = uv[1]; }
    else { u = fp.fx * Pc[0] / Pc[2] + fp.cx; v = fp.fy * Pc[1] / Pc[2] + fp.cy; }
    if (!(u < fp.min_x || u > fp.max_x) && !(v < fp.min_y || v > fp.max_y)) {
      const float maxDistance = 1.2f * max_raw, minDistance = 0.8f * min_raw;
      const float PO[3] = {X[0] - P.Ow[0], X[1] - P.Ow[1], X[2] - P.Ow[2]};
      const float dist = norm3d(PO);
      if (!(dist < minDistance || dist > maxDistance)) {
        const float ratio = max_raw / dist;                                        // PredictScale(dist3D, &CurrentFrame)
        const float lg = (float)log((double)ratio);
        int lvl = (int)ceilf(lg / fp.log_sf);
        if (lvl < 0) lvl = 0;
        else if (lvl >= fp.n_levels) lvl = fp.n_levels - 1;
        q.valid = 1;
        q.x = u; q.y = v;
        q.r = th * fp.scale[lvl];
        q.min_level = lvl - 1; q.max_level = lvl + 1;
      }
    }
  }
  window_search(fp, F, q, qd, i, w.m, list_counter, list, list_cap, results + i, s_stage[threadIdx.x >> 6]);
}

// MODE 2: SearchByProjection(CurrentFrame, LastFrame) (S/ORBmatcher.cc:1993-2066)
struct LastDev {
  int n;
  const uint8_t* mp_valid; const uint8_t* outlier; const float* world_pos; const uint8_t* desc; const int* octave;
};

__global__ __launch_bounds__(256) void search_frame_kernel(FrameParams fp, FrameDev F, LastDev L, PoseF Pc, float th,
                                                          int forward, int backward, int* list_counter, int* counter_next, uint32_t* list,
                                                          int list_cap, QResult* results) {
  __shared__ uint32_t s_stage[4][kListStage];
  if (blockIdx.x == 0 && threadIdx.x == 0) *counter_next = 0;   // the overflow counter the NEXT search on this frame will use
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= L.n) return;
  Query q;
  q.valid = 0; q.x = q.y = q.r = 0; q.min_level = q.max_level = 0; q.ur_ref = 0;
  // every field of the query in ONE trip to host memory (valid -> position -> octave -> descriptor were four)
  const uint8_t l_valid = L.mp_valid[i], l_outlier = L.outlier[i];
  const float X[3] = {L.world_pos[3 * i], L.world_pos[3 * i + 1], L.world_pos[3 * i + 2]};
  const int l_oct = L.octave[i];
  const QDesc qd = load_qdesc(L.desc + (size_t)i * 32);
  if (l_valid && !l_outlier) {
    float x3Dc[3];
    pose_map(Pc, X, x3Dc);
    const float invzc = (float)(1.0 / (double)x3Dc[2]);
    if (!(invzc < 0)) {
      const float u = fp.fx * x3Dc[0] / x3Dc[2] + fp.cx;
      const float v = fp.fy * x3Dc[1] / x3Dc[2] + fp.cy;
      if (!(u < fp.min_x || u > fp.max_x) && !(v < fp.min_y || v > fp.max_y)) {
        const int oct = l_oct;
        q.valid = 1;
        q.x = u; q.y = v;
        q.r = th * fp.scale[oct];
        if (forward) { q.min_level = oct; q.max_level = -1; }
        else if (backward) { q.min_level = 0; q.max_level = oct; }
        else { q.min_level = oct - 1; q.max_level = oct + 1; }
        q.ur_ref = u - fp.bf * invzc;
      }
    }
  }
  window_search(fp, F, q, qd, i, L.n, list_counter, list, list_cap, results + i, s_stage[threadIdx.x >> 6]);
}

// SearchByProjection(CurrentFrame, LastFrame) on a two-camera frame (S/ORBmatcher.cc:1996-2160): `right` = 0: the left camera's
// query (:2001-2031); 1: the query of :2093-2110 -- the point taken into the right camera's frame by mTrl and projected through
// mpCamera (as the text has it), searched in the right camera's grid without an image-bounds test.  Either is made only for a point the
// left camera's tests let through.
struct TrlF { float m[12]; };
__global__ __launch_bounds__(256) void search_frame_rig_kernel(FrameParams fp, FrameDev F, LastDev L, PoseF Pc, RigCamF cam, TrlF Trl, int right,
                                                              float th, int forward, int backward, int* list_counter, int* counter_next,
                                                              uint32_t* list, int list_cap, QResult* results) {
  __shared__ uint32_t s_stage[4][kListStage];
  if (blockIdx.x == 0 && threadIdx.x == 0) *counter_next = 0;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= L.n) return;
  Query q;
  q.valid = 0; q.x = q.y = q.r = 0; q.min_level = q.max_level = 0; q.ur_ref = 0;
  const uint8_t l_valid = L.mp_valid[i], l_outlier = L.outlier[i];
  const float X[3] = {L.world_pos[3 * i], L.world_pos[3 * i + 1], L.world_pos[3 * i + 2]};
  const int l_oct = L.octave[i];
  const QDesc qd = load_qdesc(L.desc + (size_t)i * 32);
  if (l_valid && !l_outlier) {
    float x3Dc[3];
    pose_map(Pc, X, x3Dc);
    const float invzc = (float)(1.0 / (double)x3Dc[2]);
    if (!(invzc < 0)) {
      float uv[2];
      rig_project(cam, x3Dc, uv);
      if (!(uv[0] < fp.min_x || uv[0] > fp.max_x) && !(uv[1] < fp.min_y || uv[1] > fp.max_y)) {
        if (right) {
          float x3Dr[3];
#pragma unroll
          for (int a = 0; a < 3; a++) {
            const float t0 = Trl.m[4 * a] * x3Dc[0] + Trl.m[4 * a + 1] * x3Dc[1] + Trl.m[4 * a + 2] * x3Dc[2];
            x3Dr[a] = t0 + Trl.m[4 * a + 3];
          }
          rig_project(cam, x3Dr, uv);
        }
        // (a projection that is not a number selects no cell in the reference: (int)floor(NaN) is INT_MIN there)
        if (isfinite(uv[0]) && isfinite(uv[1])) {
          q.valid = 1;
          q.x = uv[0]; q.y = uv[1];
          q.r = th * fp.scale[l_oct];
          if (forward) { q.min_level = l_oct; q.max_level = -1; }
          else if (backward) { q.min_level = 0; q.max_level = l_oct; }
          else { q.min_level = l_oct - 1; q.max_level = l_oct + 1; }
        }
      }
    }
  }
  window_search(fp, F, q, qd, i, L.n, list_counter, list, list_cap, results + i, s_stage[threadIdx.x >> 6]);
}

// SearchByBoW inner loops (S/ORBmatcher.cc:297-371): one wavefront per keyframe feature of a shared node.
struct BowJob { int kf_idx; int f_begin, f_end; };   // frame-side bucket [f_begin,f_end) in fvF.feat_idx

__global__ __launch_bounds__(256) void search_bow_kernel(const uint8_t* __restrict__ fdesc, const uint32_t* __restrict__ f_feat_idx,
                                                        const uint8_t* __restrict__ tvalid /*target eligibility or NULL*/,
                                                        const uint8_t* __restrict__ kf_desc, const BowJob* __restrict__ jobs,
                                                        int n_jobs, int* list_counter, int* counter_next, uint32_t* list, int list_cap,
                                                        QResult* results, int n_left = -1) {
  // n_left >= 0 (a two-camera Frame, S/ORBmatcher.cc:342-370): the first half of the jobs ranks the left camera's features of a
  // bucket (index < n_left), the second half -- the same queries again -- the right camera's
  if (blockIdx.x == 0 && threadIdx.x == 0) *counter_next = 0;   // the overflow counter the NEXT search on this frame will use
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= n_jobs) return;
  const BowJob job = jobs[j];
  const int side = n_left >= 0 ? (j >= (n_jobs >> 1) ? 1 : 0) : -1;
  const int total = job.f_end - job.f_begin;
  QResult res;
  int base = j * kSlot;
  if (total > kSlot) {
    if (lane == 0) base = n_jobs * kSlot + atomicAdd(list_counter, total);
    base = __shfl(base, 0, 64);
  }
  const uint4 a0 = *reinterpret_cast<const uint4*>(kf_desc + (size_t)job.kf_idx * 32);
  const uint4 a1 = *reinterpret_cast<const uint4*>(kf_desc + (size_t)job.kf_idx * 32 + 16);
  Top4 t;
  top4_init(t);
  for (int p = lane; p < total; p += 64) {
    const int idx = (int)f_feat_idx[job.f_begin + p];
    unsigned entry = 0xFFFFFFFFu;
    if ((!tvalid || tvalid[idx]) && (side < 0 || (idx >= n_left) == (side == 1))) {
      const uint4 b0 = *reinterpret_cast<const uint4*>(fdesc + (size_t)idx * 32);
      const uint4 b1 = *reinterpret_cast<const uint4*>(fdesc + (size_t)idx * 32 + 16);
      const int d = popc256(a0, a1, b0, b1);
      top4_insert(t, ((unsigned)d << 20) | (unsigned)p, idx);
      entry = (unsigned)idx | ((unsigned)d << 16);
    }
    if (base + p < list_cap) list[base + p] = entry;
  }
  top4_wave_emit(t, res);
  if (lane == 0) {
    res.base = (unsigned)base; res.count = (unsigned short)min(total, (int)kQCountMask);
    results[j] = res;
  }
}

// ------------------------------------------------------------------------------------------------
// raw Hamming kernels (DescriptorDistance, S/ORBmatcher.cc:2358-2374)

__global__ __launch_bounds__(256) void hamming_matrix_kernel(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t,
                                                            int nt, int* __restrict__ dist) {
  __shared__ uint4 qs[16][2];
  const int q0 = blockIdx.y * 16;
  if (threadIdx.x < 32) {
    const int qi = q0 + (threadIdx.x >> 1);
    if (qi < nq) qs[threadIdx.x >> 1][threadIdx.x & 1] = *reinterpret_cast<const uint4*>(q + (size_t)qi * 32 + 16 * (threadIdx.x & 1));
  }
  __syncthreads();
  const int ti = blockIdx.x * 256 + threadIdx.x;
  if (ti >= nt) return;
  const uint4 b0 = *reinterpret_cast<const uint4*>(t + (size_t)ti * 32);
  const uint4 b1 = *reinterpret_cast<const uint4*>(t + (size_t)ti * 32 + 16);
  for (int k = 0; k < 16 && q0 + k < nq; k++) dist[(size_t)(q0 + k) * nt + ti] = popc256(qs[k][0], qs[k][1], b0, b1);
}

__global__ __launch_bounds__(256) void hamming_best2_kernel(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t,
                                                           int nt, int* __restrict__ out4) {
  const int lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= nq) return;
  const uint4 a0 = *reinterpret_cast<const uint4*>(q + (size_t)qi * 32);
  const uint4 a1 = *reinterpret_cast<const uint4*>(q + (size_t)qi * 32 + 16);
  Top4 tt;
  top4_init(tt);
  for (int j = lane; j < nt; j += 64) {
    const uint4 b0 = *reinterpret_cast<const uint4*>(t + (size_t)j * 32);
    const uint4 b1 = *reinterpret_cast<const uint4*>(t + (size_t)j * 32 + 16);
    top4_insert(tt, ((unsigned)popc256(a0, a1, b0, b1) << 20) | (unsigned)j, j);
  }
  int od[2] = {256, 256}, oi[2] = {-1, -1};
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const unsigned m = wave_min(tt.k[0]);
    if (m != 0xFFFFFFFFu) {
      od[r] = (int)(m >> 20); oi[r] = (int)(m & 0xFFFFFu);
      if (tt.k[0] == m) { tt.k[0] = tt.k[1]; tt.k[1] = tt.k[2]; tt.k[2] = tt.k[3]; tt.k[3] = 0xFFFFFFFFu; }
    }
  }
  if (lane == 0) {
    out4[4 * qi] = od[0]; out4[4 * qi + 1] = oi[0];
    out4[4 * qi + 2] = od[1]; out4[4 * qi + 3] = oi[1];
  }
}

// ------------------------------------------------------------------------------------------------
// KeyFrame wire block: what orb_slam3_ros/KF carries per feature -- CvKeyPoint (R/msg/CvKeyPoint.msg:1-9,
// serialised packed: f32 x, f32 y, u8 size, f32 angle, u8 response, i8 octave = 15 bytes) and Descriptor (u8[32]) -- as ONE
// contiguous byte block [N x 15 | N x 32].  Conversions as Converter::toCvKeyPointMsg / fromCvKeyPointMsg (S/Converter.cc:217-245):
// size and response travel as (u_int8_t) casts and come back as floats.
constexpr int kWireKp = 15, kWireDesc = 32;

__device__ __forceinline__ void wire_put_f32(uint8_t* p, float v) {
  const unsigned u = __float_as_uint(v);
  p[0] = (uint8_t)u; p[1] = (uint8_t)(u >> 8); p[2] = (uint8_t)(u >> 16); p[3] = (uint8_t)(u >> 24);
}
__device__ __forceinline__ float wire_get_f32(const uint8_t* p) {
  return __uint_as_float((unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24));
}

__global__ __launch_bounds__(256) void wire_pack_kernel(const orbx_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc, int n,
                                                       uint8_t* __restrict__ wire) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const orbx_keypoint k = kps[i];
  uint8_t* p = wire + (size_t)i * kWireKp;
  wire_put_f32(p, k.x); wire_put_f32(p + 4, k.y);
  p[8] = (uint8_t)k.size;
  wire_put_f32(p + 9, k.angle);
  p[13] = (uint8_t)k.response;
  p[14] = (uint8_t)(int8_t)k.octave;
  const uint4* s = reinterpret_cast<const uint4*>(desc + (size_t)i * 32);
  uint8_t* d = wire + (size_t)n * kWireKp + (size_t)i * kWireDesc;     // byte-aligned destination (15 n is odd for odd n)
  const uint4 a = s[0], b = s[1];
  const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int q = 0; q < 8; q++) { d[4 * q] = (uint8_t)w[q]; d[4 * q + 1] = (uint8_t)(w[q] >> 8); d[4 * q + 2] = (uint8_t)(w[q] >> 16); d[4 * q + 3] = (uint8_t)(w[q] >> 24); }
}

__global__ __launch_bounds__(256) void wire_unpack_kernel(const uint8_t* __restrict__ wire, int n, orbx_keypoint* __restrict__ kps,
                                                         uint8_t* __restrict__ desc, orbx_keypoint* __restrict__ kps_host) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint8_t* p = wire + (size_t)i * kWireKp;
  orbx_keypoint k;
  k.x = wire_get_f32(p); k.y = wire_get_f32(p + 4);
  k.size = (float)p[8];
  k.angle = wire_get_f32(p + 9);
  k.response = (float)p[13];
  k.octave = (int)(int8_t)p[14];
  kps[i] = k;
  if (kps_host) kps_host[i] = k;
  const uint8_t* d = wire + (size_t)n * kWireKp + (size_t)i * kWireDesc;
  for (int q = 0; q < 32; q++) desc[(size_t)i * 32 + q] = d[q];
}

// ------------------------------------------------------------------------------------------------
// host helpers

void make_pose(const float* T, PoseF* P) {
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) P->R[3 * i + j] = T[4 * i + j];
    P->t[i] = T[4 * i + 3];
  }
  for (int i = 0; i < 3; i++) {      // mOw = -mRcw.t()*mtcw, general gemm path: double accumulation
    double s = 0;
    for (int k = 0; k < 3; k++) s += (double)P->R[3 * k + i] * (double)P->t[k];
    P->Ow[i] = (float)(-s);
  }
}

// Rotation histogram without per-call allocations: bin counts + one reusable (bin, index) list in push order.
struct RotHist {
  int cnt[HISTO_LENGTH];
  std::vector<uint32_t>& e;
  explicit RotHist(std::vector<uint32_t>& store) : e(store) { for (int& c : cnt) c = 0; e.clear(); }
  void add(int bin, int idx) { cnt[bin]++; e.push_back(((uint32_t)bin << 24) | (uint32_t)idx); }
  // ORBmatcher::ComputeThreeMaxima, S/ORBmatcher.cc:2312-2353, on the bin sizes; calls drop(idx) for every entry outside
  template <typename DropFn>
  void reject_outside_three_maxima(DropFn drop) const {
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < HISTO_LENGTH; i++) {
      const int s = cnt[i];
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    for (const uint32_t v : e) {
      const int bin = (int)(v >> 24);
      if (bin != ind1 && bin != ind2 && bin != ind3) drop((int)(v & 0xFFFFFFu));
    }
  }
};

inline int rot_bin(float a1, float a2) {   // factor = 1/HISTO_LENGTH
  const float factor = 1.0f / HISTO_LENGTH;
  float rot = a1 - a2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)std::round(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// frame object

struct orbm_frame {
  // `stream` is the stream the frame's work is enqueued on: its own one, or -- while it views an extractor's features -- that
  // extractor's (the features were produced there, so no cross-stream ordering is needed, and the agent keeps fewer
  // streams busy: the runtime multiplexes streams onto a handful of hardware queues, and a search kernel that shares a queue
  // with the local BA's chain waits behind 48 us solves)
  hipStream_t own_stream = nullptr;
  bool ext_stream = false;               // own_stream was handed in through orbm_frame_set_stream (never destroyed here)
  std::vector<uint8_t> claimed_buf;      // reusable host scratch of the serial commits
  std::vector<uint32_t> rot_entries;
  // octave / angle of the frame's keypoints in ordinary (cached) host memory: the serial commits index them at random,
  // and the pinned mirror the GPU has just written costs a DRAM round trip per touched line
  std::vector<float> hk_angle;
  std::vector<int8_t> hk_oct;
  const orbx_keypoint* hk_cached_from = nullptr;
  int hk_cached_n = -1;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[2] = {};
  int cap = 0;
  FrameParams fp;
  bool has_uright = false;
  DevBuf<orbx_keypoint> d_kps;
  DevBuf<uint8_t> d_desc;
  DevBuf<float> d_uright, d_depth;
  DevBuf<int> d_cell_of, d_cell_start, d_cell_items;
  // features the kernels read: own buffers after orbm_frame_upload, the extractor's after orbm_frame_from_extractor
  const orbx_keypoint* kps_p = nullptr;
  const uint8_t* desc_p = nullptr;
  const float* uright_p = nullptr;
  const float* depth_p = nullptr;
  std::vector<orbx_keypoint> h_kps_own;
  PinnedBuf<orbx_keypoint> h_kps_pin;    // host mirror written by the unpack kernel itself (orbk_frame_from_wire)
  // large-map path of SearchLocalPoints (cull_* kernels)
  DevBuf<uint8_t> d_qflag;
  DevBuf<int> d_blk_cnt, d_slot_pt, d_total;
  PinnedBuf<int> h_slot_pt;              // [0] = number of queries, [4 ..] slot -> point index
  PinnedBuf<uint8_t> h_vis;
  const orbx_keypoint* hk = nullptr;     // host mirror (octave / angle for the serial commit)
  // per-call inputs are packed into ONE pinned staging block and moved with ONE H2D copy
  PinnedBuf<uint8_t> stage;
  DevBuf<uint8_t> d_stage;
  size_t stage_off = 0;
  const int* d_assigned_mp = nullptr;
  const int* d_assigned_obs = nullptr;
  bool occ_mask = false;                 // occupancy travels in the kernel arguments (frames of <= kOccBits features)
  uint32_t occ[kOccBits / 32];
  size_t copy_lo = 0, copy_hi = 0;       // part of the staging block that needs a device copy
  unsigned search_seq = 0;               // selects the overflow counter; the kernel clears the other one
  StreamSignal sig;                      // completion word in pinned memory (host spins instead of hipStreamSynchronize)
  // query-side scratch
  DevBuf<int> d_counter;
  PinnedBuf<uint32_t> list;              // candidate lists: written by the kernels (one coalesced burst per query) into
                                         // mapped pinned memory, read by the host only when a query's top-4 cannot decide
  PinnedBuf<QResult> results;
  float last_ms = 0;
};

static int frame_set_params(orbm_frame* f, const orbm_frame_view* v, int n) {
  if (v->n_levels < 1 || v->n_levels > ORBG_MAX_LEVELS) return ORBG_BAD_ARG;
  // feature indices travel in 16 bits (candidate lists idx | dist << 16, QResult.idx with 0xFFFF = none)
  if (n >= ORBG_MAX_FRAME_FEATURES) return ORBG_CAP_EXCEEDED;
  FrameParams& p = f->fp;
  p.n = n;
  p.min_x = v->min_x; p.max_x = v->max_x; p.min_y = v->min_y; p.max_y = v->max_y;
  p.w_inv = static_cast<float>(ORBG_GRID_COLS) / static_cast<float>(v->max_x - v->min_x);
  p.h_inv = static_cast<float>(ORBG_GRID_ROWS) / static_cast<float>(v->max_y - v->min_y);
  p.fx = v->fx; p.fy = v->fy; p.cx = v->cx; p.cy = v->cy; p.bf = v->bf; p.b = v->b;
  p.n_levels = v->n_levels;
  p.log_sf = std::log(v->scale_factor);
  p.scale[0] = 1.0f;
  for (int i = 1; i < v->n_levels; i++) p.scale[i] = p.scale[i - 1] * v->scale_factor;
  for (int i = v->n_levels; i < ORBG_MAX_LEVELS; i++) p.scale[i] = 0;
  return ORBG_OK;
}

static int frame_reserve(orbm_frame* f, int n) {
  int rc;
  const size_t c = (size_t)std::max(n, 1);
  if ((rc = f->d_kps.reserve(c)) || (rc = f->d_desc.reserve(c * 32)) || (rc = f->d_uright.reserve(c)) ||
      (rc = f->d_depth.reserve(c)) || (rc = f->d_cell_of.reserve(c)) || (rc = f->d_cell_start.reserve(kCells + 1)) ||
      (rc = f->d_cell_items.reserve(c)))
    return rc;
  if (!f->d_counter.p) {                        // both overflow counters start at zero; from then on the kernels keep them so
    if ((rc = f->d_counter.reserve(4))) return rc;
    // (once per frame object, and complete before anything can use it: the null stream's hipMemset may return before the fill has run,
    // and the library's non-blocking streams do not join the null stream)
    ORBG_HIP(hipMemsetAsync(f->d_counter.p, 0, f->d_counter.cap * sizeof(int), f->stream));
    ORBG_HIP(hipStreamSynchronize(f->stream));
  }
  return ORBG_OK;
}

static int frame_build_grid(orbm_frame* f) {
  hipLaunchKernelGGL(grid_build_kernel, dim3(1), dim3(1024), 0, f->stream, f->kps_p, f->fp, f->d_cell_of.p,
                     f->d_cell_start.p, f->d_cell_items.p, (const int*)nullptr, (volatile unsigned*)nullptr, 0u);
  ORBG_HIP(hipGetLastError());
  return ORBG_OK;
}

extern "C" int orbm_frame_create(int device, int cap_features, orbm_frame** out) {
  if (!out || cap_features < 0) return ORBG_BAD_ARG;
  int rc = select_device(device);
  if (rc) return rc;
  orbm_frame* f = new orbm_frame();
  f->device = device;
  f->cap = cap_features;
  memset(&f->fp, 0, sizeof(f->fp));
  if (orbg::create_stream(&f->own_stream, "fr") != hipSuccess) { delete f; return ORBG_HIP_ERROR; }
  f->stream = f->own_stream;
  for (auto& e : f->ev) if (hipEventCreate(&e) != hipSuccess) { delete f; return ORBG_HIP_ERROR; }
  if ((rc = frame_reserve(f, cap_features))) { delete f; return rc; }
  *out = f;
  return ORBG_OK;
}

extern "C" int orbm_frame_destroy(orbm_frame* f) {
  if (!f) return ORBG_BAD_ARG;
  (void)hipSetDevice(f->device);
  (void)hipStreamSynchronize(f->stream);
  if (f->own_stream != f->stream) (void)hipStreamSynchronize(f->own_stream);
  f->d_kps.release(); f->d_desc.release(); f->d_uright.release(); f->d_depth.release(); f->d_cell_of.release();
  f->d_cell_start.release(); f->d_cell_items.release(); f->stage.release(); f->d_stage.release();
  f->d_counter.release(); f->list.release(); f->results.release(); f->sig.release(); f->h_kps_pin.release(); f->d_qflag.release(); f->d_blk_cnt.release(); f->d_slot_pt.release(); f->d_total.release(); f->h_slot_pt.release(); f->h_vis.release();
  for (auto& e : f->ev) if (e) (void)hipEventDestroy(e);
  if (!f->ext_stream) orbg::release_stream(f->own_stream);
  delete f;
  return ORBG_OK;
}

extern "C" int orbm_frame_set_stream(orbm_frame* f, void* hip_stream) {
  if (!f) return ORBG_BAD_ARG;
  int rc = select_device(f->device);
  if (rc) return rc;
  const bool on_own = f->stream == f->own_stream;         // (a frame whose constructor is still in flight stays on that extractor's stream)
  if (!on_own) ORBG_HIP(hipStreamSynchronize(f->stream));
  if ((rc = orbg::swap_stream(&f->own_stream, &f->ext_stream, hip_stream, "fr"))) return rc;
  if (on_own) f->stream = f->own_stream;
  return ORBG_OK;
}

extern "C" int orbm_frame_upload(orbm_frame* f, const orbm_frame_view* v) {
  if (!f || !v || v->n < 0 || (v->n > 0 && (!v->kps || !v->desc))) return ORBG_BAD_ARG;
  int rc = select_device(f->device);
  if (rc) return rc;
  if ((rc = frame_set_params(f, v, v->n))) return rc;
  if ((rc = frame_reserve(f, v->n))) return rc;
  const int n = v->n;
  f->h_kps_own.assign(v->kps, v->kps + n);
  f->hk = f->h_kps_own.data(); f->hk_cached_n = -1;
  f->stream = f->own_stream;       // features of its own: back on the frame's own stream
  f->kps_p = f->d_kps.p; f->desc_p = f->d_desc.p; f->uright_p = f->d_uright.p; f->depth_p = f->d_depth.p;
  f->has_uright = v->uright != nullptr;
  if (n > 0) {
    ORBG_HIP(hipMemcpyAsync(f->d_kps.p, v->kps, (size_t)n * sizeof(orbx_keypoint), hipMemcpyHostToDevice, f->stream));
    ORBG_HIP(hipMemcpyAsync(f->d_desc.p, v->desc, (size_t)n * 32, hipMemcpyHostToDevice, f->stream));
    if (v->uright) ORBG_HIP(hipMemcpyAsync(f->d_uright.p, v->uright, (size_t)n * 4, hipMemcpyHostToDevice, f->stream));
    if (v->depth) ORBG_HIP(hipMemcpyAsync(f->d_depth.p, v->depth, (size_t)n * 4, hipMemcpyHostToDevice, f->stream));
  }
  if ((rc = frame_build_grid(f))) return rc;
  ORBG_HIP(hipStreamSynchronize(f->stream));
  return ORBG_OK;
}

extern "C" int orbm_frame_from_extractor(orbm_frame* f, orbx_handle* h, const orbm_frame_view* v) {
  if (!f || !h || !v) return ORBG_BAD_ARG;
  int rc = select_device(f->device);
  if (rc) return rc;
  const orbx_keypoint* dk; const uint8_t* dd; const float* du; const float* dz; const orbx_keypoint* hk; int n; hipStream_t xs;
  if ((rc = orbx_internal_left_features(h, &dk, &dd, &du, &dz, &hk, &n, &xs))) return rc;
  if (v->n >= 0 && v->n != n) return ORBG_BAD_ARG;
  if ((rc = frame_set_params(f, v, n))) return rc;
  if ((rc = frame_reserve(f, n))) return rc;
  // Zero-copy hand-over: the frame aliases the extractor's device-resident left features (valid until the next
  // extraction on that handle) and its pinned host mirror of the keypoints.  Every orbx_* entry point synchronises
  // its stream before returning, so the data is complete here.
  f->has_uright = true;
  f->kps_p = dk; f->desc_p = dd; f->uright_p = du; f->depth_p = dz; f->hk = hk; f->hk_cached_n = -1;
  f->stream = xs;                  // work on this frame is enqueued on the extractor's stream from now on
  return frame_build_grid(f);      // asynchronous on that stream; the searches run on the same stream
}

// Used by orbx_frame_stereo_dev (extractor.hip): alias the extractor's left features and launch the grid build on the
// EXTRACTOR's stream, behind the descriptor / stereo kernels; the caller synchronises that stream once.
int orbx_internal_kp_capacity(orbx_handle* h);   // extractor.hip
int orbm_internal_attach(orbm_frame* f, orbx_handle* h, const orbm_frame_view* v, int n, hipStream_t stream, const int* d_n,
                         volatile unsigned* done_flag, unsigned done_seq, const StereoFinalizeArgs* fin, const orbg::UndistortArgs* un,
                         bool mono) {
  if (!f || !h || !v) return ORBG_BAD_ARG;
  const orbx_keypoint* dk; const uint8_t* dd; const float* du; const float* dz; const orbx_keypoint* hk; int n0; hipStream_t xs;
  int rc = orbx_internal_left_features(h, &dk, &dd, &du, &dz, &hk, &n0, &xs);
  if (rc) return rc;
  // n < 0: the count only exists on the device yet (d_n): the grid buffers are sized for the most keypoints the
  // extractor's own buffers can hold (grid_build_kernel writes cell_of[i] / cell_items[i] for every i < *d_n)
  if ((rc = frame_set_params(f, v, n < 0 ? 0 : n))) return rc;
  const int xcap = orbx_internal_kp_capacity(h);
  if (n < 0 && xcap >= ORBG_MAX_FRAME_FEATURES) return ORBG_CAP_EXCEEDED;
  if ((rc = frame_reserve(f, n < 0 ? std::max(std::max(f->cap, 4096), xcap) : n))) return rc;
  f->has_uright = !mono;           // monocular frame: mvuRight = -1 for every feature (S/Frame.cc:303)
  f->kps_p = (un && un->on) ? un->dst : dk;      // the grid and the searches read mvKeysUn
  f->desc_p = dd; f->uright_p = du; f->depth_p = dz; f->hk = hk; f->hk_cached_n = -1;      // (octave / angle of mvKeys == those of mvKeysUn)
  f->stream = stream;              // the extractor's stream: searches on this frame follow its constructor in order
  if (un && un->on)
    hipLaunchKernelGGL(undistort_grid_kernel, dim3(1), dim3(1024), 0, stream, *un, f->fp, f->d_cell_of.p, f->d_cell_start.p, f->d_cell_items.p,
                       d_n, done_flag, done_seq);
  else if (fin)
    hipLaunchKernelGGL(grid_build_finalize_kernel, dim3(2), dim3(1024), 0, stream, f->kps_p, f->fp, f->d_cell_of.p, f->d_cell_start.p,
                       f->d_cell_items.p, d_n, done_flag, done_seq, *fin);
  else
    hipLaunchKernelGGL(grid_build_kernel, dim3(1), dim3(1024), 0, stream, f->kps_p, f->fp, f->d_cell_of.p, f->d_cell_start.p,
                       f->d_cell_items.p, d_n, done_flag, done_seq);
  ORBG_HIP(hipGetLastError());
  return ORBG_OK;
}

// Called when the host has COLLECTED the constructor (completion word seen: every result of the chain is released to system scope).
// From here on the frame's searches need no ordering against the extractor's stream, and they must not sit behind the constructors
// of LATER frames that a pipelined caller has already enqueued there (two frames ahead: SearchLocalPoints waited 129 us behind
// Frame(t+2)): the frame goes back to its own stream.
void orbm_internal_set_n(orbm_frame* f, int n) {
  f->fp.n = n; f->hk_cached_n = -1;
  f->stream = f->own_stream;
}

// device-resident descriptors of the frame's features (for the vocabulary transform in bow.hip)
int orbm_internal_features(orbm_frame* f, const uint8_t** d_desc, int* n, hipStream_t* stream) {
  if (!f || !f->desc_p) return ORBG_BAD_ARG;
  *d_desc = f->desc_p; *n = f->fp.n; *stream = f->stream;
  return ORBG_OK;
}

extern "C" int orbm_frame_get_grid(orbm_frame* f, int32_t* cell_start, int32_t* cell_items) {
  if (!f || !cell_start) return ORBG_BAD_ARG;
  int rc = select_device(f->device);
  if (rc) return rc;
  ORBG_HIP(hipStreamSynchronize(f->stream));              // (the grid build may still be in flight on the frame's stream; the copies below use the null stream)
  ORBG_HIP(hipMemcpy(cell_start, f->d_cell_start.p, (kCells + 1) * sizeof(int), hipMemcpyDeviceToHost));
  const int total = cell_start[kCells];
  if (cell_items && total > 0) ORBG_HIP(hipMemcpy(cell_items, f->d_cell_items.p, (size_t)total * sizeof(int), hipMemcpyDeviceToHost));
  return ORBG_OK;
}

extern "C" int orbk_wire_bytes(int n) { return n < 0 ? ORBG_BAD_ARG : n * (kWireKp + kWireDesc); }

// Frame / KeyFrame features -> wire block.  `wire` is device memory (for the RCCL exchange) when wire_on_device != 0,
// otherwise host memory.
extern "C" int orbk_pack_frame(orbm_frame* f, uint8_t* wire, int wire_on_device) {
  if (!f || !f->kps_p || (f->fp.n > 0 && !wire)) return ORBG_BAD_ARG;
  int rc = select_device(f->device);
  if (rc) return rc;
  const int n = f->fp.n;
  if (n == 0) return ORBG_OK;
  const size_t bytes = (size_t)n * (kWireKp + kWireDesc);
  uint8_t* d_wire = wire;
  if (!wire_on_device) {
    if ((rc = f->d_stage.reserve(bytes))) return rc;
    d_wire = f->d_stage.p;
  }
  hipLaunchKernelGGL(wire_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, f->stream, f->kps_p, f->desc_p, n, d_wire);
  ORBG_HIP(hipGetLastError());
  if (!wire_on_device) ORBG_HIP(hipMemcpyAsync(wire, d_wire, bytes, hipMemcpyDeviceToHost, f->stream));
  ORBG_HIP(hipStreamSynchronize(f->stream));
  return ORBG_OK;
}

// Wire block -> device-resident KeyFrame (features + grid), ready for the KeyFrame matchers.  view supplies what the KF message
// carries next to the features (bounds, calibration, scale pyramid; view->n / kps / desc are ignored).
extern "C" int orbk_frame_from_wire(orbm_frame* f, const orbm_frame_view* v, const uint8_t* wire, int n, int wire_on_device) {
  if (!f || !v || n < 0 || (n > 0 && !wire)) return ORBG_BAD_ARG;
  int rc = select_device(f->device);
  if (rc) return rc;
  if ((rc = frame_set_params(f, v, n))) return rc;
  if ((rc = frame_reserve(f, n))) return rc;
  f->kps_p = f->d_kps.p; f->desc_p = f->d_desc.p; f->uright_p = f->d_uright.p; f->depth_p = f->d_depth.p;
  f->has_uright = false;
  // the host mirror of the keypoints (octave / angle for the serial commits) is written by the unpack kernel straight into
  // pinned memory: no copy command, and -- for a block that is already on the device -- no wait here at all: whatever uses the
  // frame next is enqueued on the same stream, and the commits read the mirror only after their own search has completed
  if ((rc = f->h_kps_pin.reserve((size_t)std::max(n, 1)))) return rc;
  f->hk = f->h_kps_pin.h; f->hk_cached_n = -1; f->hk_cached_from = nullptr;
  f->stream = f->own_stream;       // features of its own: back on the frame's own stream
  if (n > 0) {
    const size_t bytes = (size_t)n * (kWireKp + kWireDesc);
    const uint8_t* d_wire = wire;
    if (!wire_on_device) {
      if ((rc = f->d_stage.reserve(bytes))) return rc;
      ORBG_HIP(hipMemcpyAsync(f->d_stage.p, wire, bytes, hipMemcpyHostToDevice, f->stream));
      d_wire = f->d_stage.p;
    }
    hipLaunchKernelGGL(wire_unpack_kernel, dim3((n + 255) / 256), dim3(256), 0, f->stream, d_wire, n, f->d_kps.p, f->d_desc.p,
                       f->h_kps_pin.d);
    ORBG_HIP(hipGetLastError());
  }
  if ((rc = frame_build_grid(f))) return rc;
  if (!wire_on_device) ORBG_HIP(hipStreamSynchronize(f->stream));     // the caller's host block may go away
  return ORBG_OK;
}

// host copies of a device-resident frame's features (e.g. after orbk_frame_from_wire)
extern "C" int orbm_frame_download(orbm_frame* f, orbx_keypoint* kps, uint8_t* desc) {
  if (!f || !f->kps_p) return ORBG_BAD_ARG;
  int rc = select_device(f->device);
  if (rc) return rc;
  const int n = f->fp.n;
  if (n > 0 && kps) ORBG_HIP(hipMemcpyAsync(kps, f->kps_p, (size_t)n * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, f->stream));
  if (n > 0 && desc) ORBG_HIP(hipMemcpyAsync(desc, f->desc_p, (size_t)n * 32, hipMemcpyDeviceToHost, f->stream));
  ORBG_HIP(hipStreamSynchronize(f->stream));
  return ORBG_OK;
}

// ------------------------------------------------------------------------------------------------
// raw Hamming entry points

static int hamming_common(int device, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* out, bool matrix) {
  if (nq < 0 || nt < 0 || !out || (nq > 0 && !q) || (nt > 0 && !t)) return ORBG_BAD_ARG;
  int rc = select_device(device);
  if (rc) return rc;
  if (nq == 0) return ORBG_OK;
  uint8_t *dq = nullptr, *dt = nullptr;
  int* dout = nullptr;
  const size_t out_n = matrix ? (size_t)nq * nt : (size_t)nq * 4;
  ORBG_HIP(hipMalloc((void**)&dq, (size_t)nq * 32));
  ORBG_HIP(hipMalloc((void**)&dt, (size_t)std::max(nt, 1) * 32));
  ORBG_HIP(hipMalloc((void**)&dout, std::max<size_t>(out_n, 1) * sizeof(int)));
  orbg::MiscStream ms;                                 // the library's M stream (never the legacy null stream)
  if ((rc = ms.open())) return rc;
  ORBG_HIP(hipMemcpyAsync(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice, ms.s));
  if (nt > 0) ORBG_HIP(hipMemcpyAsync(dt, t, (size_t)nt * 32, hipMemcpyHostToDevice, ms.s));
  if (matrix) {
    if (nt > 0) hipLaunchKernelGGL(hamming_matrix_kernel, dim3((nt + 255) / 256, (nq + 15) / 16), dim3(256), 0, ms.s, dq, nq, dt, nt, dout);
  } else {
    hipLaunchKernelGGL(hamming_best2_kernel, dim3((nq + 3) / 4), dim3(256), 0, ms.s, dq, nq, dt, nt, dout);
  }
  ORBG_HIP(hipGetLastError());
  if (out_n > 0) ORBG_HIP(hipMemcpyAsync(out, dout, out_n * sizeof(int), hipMemcpyDeviceToHost, ms.s));
  ORBG_HIP(hipStreamSynchronize(ms.s));
  (void)hipFree(dq); (void)hipFree(dt); (void)hipFree(dout);
  return ORBG_OK;
}

extern "C" int orbm_hamming_matrix(int device, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* dist) {
  return hamming_common(device, q, nq, t, nt, dist, true);
}
// ------------------------------------------------------------------------------------------------
// Frame::ComputeStereoFishEyeMatches (S/Frame.cc:1093-1150): brute-force 2-nearest-neighbour Hamming match of the two cameras'
// lapping-area features, Lowe's ratio, KannalaBrandt8::TriangulateMatches per surviving pair.  One wavefront per left feature: the
// lanes stride over the right descriptors, the two smallest (distance, index) keys of the wavefront are its knnMatch row; lane 0
// then runs the triangulation (a 4 x 4 null vector: Jacobi rotations on A^T A in float64 where the reference calls cv::SVD).

struct FisheyeDev {
  int nq, nt, mono_left, mono_right;
  const orbx_keypoint* kl; const orbx_keypoint* kr;        // the lapping-area keypoints only
  const uint8_t* dl; const uint8_t* dr;
  const float* sigma2;
  RigCamF cam1, cam2;
  float Tlr[12];
  int* l2r; float* depth; float* p3d;                      // nq entries each (p3d: 3 nq): mapped pinned memory, read by the host after the signal
};

// KannalaBrandt8::unproject, S/CameraModels/KannalaBrandt8.cpp:103-133
__device__ __forceinline__ void kb8_unproject(const RigCamF& c, float u, float v, float* ray) {
  const float pwx = (u - c.cx) / c.fx, pwy = (v - c.cy) / c.fy;
  float scale = 1.f;
  float theta_d = sqrtf(pwx * pwx + pwy * pwy);
  theta_d = fminf(fmaxf((float)(-3.1415926535897932384626433832795 / 2.f), theta_d), (float)(3.1415926535897932384626433832795 / 2.f));
  if (theta_d > 1e-8) {
    float theta = theta_d;
    for (int j = 0; j < 10; j++) {
      const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
      const float k0 = c.k[0] * theta2, k1 = c.k[1] * theta4, k2 = c.k[2] * theta6, k3 = c.k[3] * theta8;
      const float theta_fix = (theta * (1 + k0 + k1 + k2 + k3) - theta_d) / (1 + 3 * k0 + 5 * k1 + 7 * k2 + 9 * k3);
      theta = theta - theta_fix;
      if (fabsf(theta_fix) < 1e-6f) break;
    }
    scale = (float)tan((double)theta) / theta_d;
  }
  ray[0] = pwx * scale; ray[1] = pwy * scale; ray[2] = 1.f;
}

__device__ __forceinline__ void null_vector4(double (&S)[4][4], double (&v)[4]) {   // eigenvector of the smallest eigenvalue of a symmetric 4 x 4: cyclic Jacobi
  double V[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0, diag = 0;
#pragma unroll
    for (int p = 0; p < 4; p++) {
      diag += S[p][p] * S[p][p];
#pragma unroll
      for (int q = p + 1; q < 4; q++) off += S[p][q] * S[p][q];
    }
    if (off <= 1e-28 * diag) break;                         // eigenvectors to ~1e-14: far below the float32 the result is rounded to
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
      for (int q = p + 1; q < 4; q++) {
        if (S[p][q] == 0.0) continue;
        const double tau = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        const double cs = 1.0 / sqrt(1.0 + t * t), sn = t * cs;
#pragma unroll
        for (int k = 0; k < 4; k++) { const double a = S[k][p], b = S[k][q]; S[k][p] = cs * a - sn * b; S[k][q] = sn * a + cs * b; }
#pragma unroll
        for (int k = 0; k < 4; k++) { const double a = S[p][k], b = S[q][k]; S[p][k] = cs * a - sn * b; S[q][k] = sn * a + cs * b; }
#pragma unroll
        for (int k = 0; k < 4; k++) { const double a = V[k][p], b = V[k][q]; V[k][p] = cs * a - sn * b; V[k][q] = sn * a + cs * b; }
      }
  }
  int m = 0;
  double smallest = S[0][0];                                // (compile-time indices only: a dynamically indexed array lives in scratch memory)
#pragma unroll
  for (int i = 1; i < 4; i++) if (S[i][i] < smallest) { smallest = S[i][i]; m = i; }
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = m == 0 ? V[k][0] : m == 1 ? V[k][1] : m == 2 ? V[k][2] : V[k][3];
}

// KannalaBrandt8::TriangulateMatches, :335-403 (Triangulate :405-420)
__device__ __forceinline__ float kb8_triangulate_matches(const FisheyeDev& D, const orbx_keypoint& kp1, const orbx_keypoint& kp2, float sigmaLevel, float unc, float (&p3D)[3]) {
  float r1[3], r2[3], r21[3];
  kb8_unproject(D.cam1, kp1.x, kp1.y, r1);
  kb8_unproject(D.cam2, kp2.x, kp2.y, r2);
  const float* T = D.Tlr;
#pragma unroll
  for (int i = 0; i < 3; i++) r21[i] = T[4 * i] * r2[0] + T[4 * i + 1] * r2[1] + T[4 * i + 2] * r2[2];
  auto dot = [](const float* a, const float* b) { return (double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2]; };
  const float cosParallaxRays = (float)(dot(r1, r21) / (sqrt(dot(r1, r1)) * sqrt(dot(r21, r21))));
  if (cosParallaxRays > 0.9998) return -1;
  float R21[9], t21[3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) R21[3 * i + j] = T[4 * j + i];
#pragma unroll
  for (int i = 0; i < 3; i++) { const float t0 = R21[3 * i] * T[3] + R21[3 * i + 1] * T[7] + R21[3 * i + 2] * T[11]; t21[i] = -t0; }
  float Tcw2[12];
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) Tcw2[4 * i + j] = R21[3 * i + j];
    Tcw2[4 * i + 3] = t21[i];
  }
  float A[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const float e2 = j == 2 ? 1.f : 0.f, e0 = j == 0 ? 1.f : 0.f, e1 = j == 1 ? 1.f : 0.f;      // rows of Tcw1 = [I | 0]
    A[0][j] = r1[0] * e2 - e0;
    A[1][j] = r1[1] * e2 - e1;
    A[2][j] = r2[0] * Tcw2[8 + j] - Tcw2[j];
    A[3][j] = r2[1] * Tcw2[8 + j] - Tcw2[4 + j];
  }
  double S[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) { double acc = 0; for (int k = 0; k < 4; k++) acc += (double)A[k][i] * (double)A[k][j]; S[i][j] = acc; }
  double v[4];
  null_vector4(S, v);
  float x3D[3];
#pragma unroll
  for (int i = 0; i < 3; i++) x3D[i] = (float)(v[i] / v[3]);
  const float z1 = x3D[2];
  if (!(z1 > 0)) return -1;
  const float z2 = (float)(dot(R21 + 6, x3D) + (double)t21[2]);
  if (!(z2 > 0)) return -1;
  float uv1[2];
  rig_project(D.cam1, x3D, uv1);
  const float errX1 = uv1[0] - kp1.x, errY1 = uv1[1] - kp1.y;
  if ((double)(errX1 * errX1 + errY1 * errY1) > 5.991 * (double)sigmaLevel) return -1;
  float x3D2[3];
#pragma unroll
  for (int i = 0; i < 3; i++) { const float t0 = R21[3 * i] * x3D[0] + R21[3 * i + 1] * x3D[1] + R21[3 * i + 2] * x3D[2]; x3D2[i] = t0 + t21[i]; }
  float uv2[2];
  rig_project(D.cam2, x3D2, uv2);
  const float errX2 = uv2[0] - kp2.x, errY2 = uv2[1] - kp2.y;
  if ((double)(errX2 * errX2 + errY2 * errY2) > 5.991 * (double)unc) return -1;
  p3D[0] = x3D[0]; p3D[1] = x3D[1]; p3D[2] = x3D[2];
  return z1;
}

__global__ __launch_bounds__(256) void fisheye_stereo_kernel(FisheyeDev D) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= D.nq) return;
  const uint4 a0 = *reinterpret_cast<const uint4*>(D.dl + (size_t)q * 32), a1 = *reinterpret_cast<const uint4*>(D.dl + (size_t)q * 32 + 16);
  unsigned k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;                  // (distance << 16 | index): equal distances rank by index
  for (int t = lane; t < D.nt; t += 64) {
    const uint4 b0 = *reinterpret_cast<const uint4*>(D.dr + (size_t)t * 32), b1 = *reinterpret_cast<const uint4*>(D.dr + (size_t)t * 32 + 16);
    const unsigned key = ((unsigned)popc256(a0, a1, b0, b1) << 16) | (unsigned)t;
    if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {                     // the two smallest keys of the wavefront
    const unsigned o1 = __shfl_xor(k1, off, 64), o2 = __shfl_xor(k2, off, 64);
    const unsigned lo = min(k1, o1), hi = max(k1, o1);
    k2 = min(hi, min(k2, o2));
    k1 = lo;
  }
  if (lane != 0) return;
  D.l2r[q] = -1; D.depth[q] = -1.0f;
  if (D.nt < 2) return;
  const int d1 = (int)(k1 >> 16), i1 = (int)(k1 & 0xFFFF), d2 = (int)(k2 >> 16);
  if (!((double)(float)d1 < (double)(float)d2 * 0.7)) return;  // Lowe's ratio, :1137
  const orbx_keypoint kl = D.kl[q], kr = D.kr[i1];
  float p3D[3];
  const float z = kb8_triangulate_matches(D, kl, kr, D.sigma2[kl.octave], D.sigma2[kr.octave], p3D);
  if (z > 0.0001f) {
    D.l2r[q] = i1 + D.mono_right;
    D.depth[q] = z;
    D.p3d[3 * q] = p3D[0]; D.p3d[3 * q + 1] = p3D[1]; D.p3d[3 * q + 2] = p3D[2];
  }
}

extern "C" int orbx_fisheye_stereo_matches(int device, const orbx_fisheye_stereo_view* v, int32_t* left_to_right, int32_t* right_to_left, float* depth,
                                           float* points3d, int* n_matches) {
  if (!v || !left_to_right || !right_to_left || !depth || !points3d || v->n_left < 0 || v->n_right < 0 || v->mono_left < 0 || v->mono_left > v->n_left ||
      v->mono_right < 0 || v->mono_right > v->n_right || v->n_levels < 1 || !v->level_sigma2)
    return ORBG_BAD_ARG;
  if (v->left.model != ORBG_CAM_KANNALA_BRANDT8 || v->right.model != ORBG_CAM_KANNALA_BRANDT8) return ORBG_BAD_ARG;
  const int nq = v->n_left - v->mono_left, nt = v->n_right - v->mono_right;
  if (nt > 65535) return ORBG_CAP_EXCEEDED;
  if ((nq > 0 && (!v->kps_left || !v->desc_left)) || (nt > 0 && (!v->kps_right || !v->desc_right))) return ORBG_BAD_ARG;
  for (int i = 0; i < nq; i++) { const int o = v->kps_left[v->mono_left + i].octave; if (o < 0 || o >= v->n_levels) return ORBG_BAD_ARG; }
  for (int i = 0; i < nt; i++) { const int o = v->kps_right[v->mono_right + i].octave; if (o < 0 || o >= v->n_levels) return ORBG_BAD_ARG; }
  int rc = select_device(device);
  if (rc) return rc;
  for (int i = 0; i < v->n_left; i++) { left_to_right[i] = -1; depth[i] = -1.0f; }
  for (int i = 0; i < v->n_right; i++) right_to_left[i] = -1;
  if (n_matches) *n_matches = 0;
  if (nq == 0 || nt < 2) return ORBG_OK;
  // one pinned block in (one copy to the device), results straight into mapped pinned memory (buffers of the calling thread, kept from
  // frame to frame); mvRightToLeftMatch and nMatches follow from mvLeftToRightMatch on the host (:1144-1145: the last left feature stays)
  struct Scratch { int device = -1; PinnedBuf<uint8_t> in, out; DevBuf<uint8_t> din; StreamSignal sig; hipStream_t st = nullptr; };
  static thread_local Scratch S;
  if (S.device != device) {
    S.in.release(); S.out.release(); S.din = DevBuf<uint8_t>(); S.sig.release(); S.sig = StreamSignal(); S.device = device;
    if (S.st) { release_stream(S.st); S.st = nullptr; }
    ORBG_HIP(create_stream(&S.st, "misc"));             // the library's M stream (kept: a stream object per call costs a synchronisation)
  }
  auto up = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t o_kl = 0, o_kr = up(o_kl + (size_t)nq * sizeof(orbx_keypoint)), o_dl = up(o_kr + (size_t)nt * sizeof(orbx_keypoint)), o_dr = up(o_dl + (size_t)nq * 32),
               o_sg = up(o_dr + (size_t)nt * 32), in_bytes = up(o_sg + (size_t)v->n_levels * 4);
  const size_t p_l2r = 0, p_dep = up(p_l2r + (size_t)nq * 4), p_p3d = up(p_dep + (size_t)nq * 4), out_bytes = up(p_p3d + (size_t)nq * 12);
  if ((rc = S.in.reserve(in_bytes)) || (rc = S.out.reserve(out_bytes)) || (rc = S.din.reserve(in_bytes))) return rc;
  memcpy(S.in.h + o_kl, v->kps_left + v->mono_left, (size_t)nq * sizeof(orbx_keypoint));
  memcpy(S.in.h + o_kr, v->kps_right + v->mono_right, (size_t)nt * sizeof(orbx_keypoint));
  memcpy(S.in.h + o_dl, v->desc_left + (size_t)v->mono_left * 32, (size_t)nq * 32);
  memcpy(S.in.h + o_dr, v->desc_right + (size_t)v->mono_right * 32, (size_t)nt * 32);
  memcpy(S.in.h + o_sg, v->level_sigma2, (size_t)v->n_levels * 4);
  ORBG_HIP(hipMemcpyAsync(S.din.p, S.in.h, in_bytes, hipMemcpyHostToDevice, S.st));
  FisheyeDev D;
  D.nq = nq; D.nt = nt; D.mono_left = v->mono_left; D.mono_right = v->mono_right;
  D.kl = reinterpret_cast<const orbx_keypoint*>(S.din.p + o_kl); D.kr = reinterpret_cast<const orbx_keypoint*>(S.din.p + o_kr);
  D.dl = S.din.p + o_dl; D.dr = S.din.p + o_dr; D.sigma2 = reinterpret_cast<const float*>(S.din.p + o_sg);
  D.cam1 = rig_cam_of(v->left); D.cam2 = rig_cam_of(v->right);
  memcpy(D.Tlr, v->Tlr, sizeof(D.Tlr));
  D.l2r = reinterpret_cast<int*>(S.out.d + p_l2r); D.depth = reinterpret_cast<float*>(S.out.d + p_dep); D.p3d = reinterpret_cast<float*>(S.out.d + p_p3d);
  hipLaunchKernelGGL(fisheye_stereo_kernel, dim3((nq + 3) / 4), dim3(256), 0, S.st, D);
  ORBG_HIP(hipGetLastError());
  if ((rc = S.sig.sync(S.st))) return rc;
  const int* h_l2r = reinterpret_cast<const int*>(S.out.h + p_l2r); const float* h_dep = reinterpret_cast<const float*>(S.out.h + p_dep);
  const float* h_p3d = reinterpret_cast<const float*>(S.out.h + p_p3d);
  int nm = 0;
  for (int q = 0; q < nq; q++) {
    left_to_right[v->mono_left + q] = h_l2r[q]; depth[v->mono_left + q] = h_dep[q];
    if (h_l2r[q] < 0) continue;
    memcpy(points3d + 3 * (size_t)(v->mono_left + q), h_p3d + 3 * (size_t)q, 12);
    right_to_left[h_l2r[q]] = v->mono_left + q;             // ascending q: the last one stays
    nm++;
  }
  if (n_matches) *n_matches = nm;
  return ORBG_OK;
}

extern "C" int orbm_hamming_best2(int device, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* out4) {
  return hamming_common(device, q, nq, t, nt, out4, false);
}

// ------------------------------------------------------------------------------------------------
// isInFrustum entry point + map object

struct orbm_map {
  int device = 0;
  int m = 0;
  // point fields: ONE device block filled by ONE copy from a pinned staging block (offsets below, all 16-byte aligned)
  DevBuf<uint8_t> arena;
  PinnedBuf<uint8_t> stage;
  hipStream_t stream = nullptr;
  bool ext_stream = false;
  size_t o_pos = 0, o_normal = 0, o_min = 0, o_max = 0, o_desc = 0, o_bad = 0, o_skip = 0, arena_bytes = 0;
  std::vector<int> n_obs;
  std::vector<uint8_t> h_bad;
  // device track fields
  DevBuf<uint8_t> t_in_view;
  DevBuf<float> t_px, t_py, t_pxr, t_depth, t_vc;
  DevBuf<int> t_level;
  // the upload is asynchronous: the copy is followed by an event the consumers' streams wait for (no host wait per upload;
  // UpdateLocalMap runs once per keyframe, the next search on another stream follows ~100 us later)
  hipEvent_t up_ev = nullptr;
  bool up_pending = false;
};

// Orders the kernels enqueued on `st` after the map's last upload.  Host cost: one hipStreamWaitEvent while an upload is pending.
static int map_sync_to(orbm_map* m, hipStream_t st) {
  if (!m->up_pending) return ORBG_OK;
  if (hipEventQuery(m->up_ev) == hipSuccess) { m->up_pending = false; return ORBG_OK; }
  (void)hipGetLastError();                       // hipErrorNotReady is not an error here
  ORBG_HIP(hipStreamWaitEvent(st, m->up_ev, 0));
  return ORBG_OK;
}

static int map_reserve(orbm_map* m, int n) {
  const size_t c = (size_t)std::max(n, 1);
  int rc;
  const size_t c16 = (c + 15) & ~(size_t)15;
  m->o_pos = 0; m->o_normal = m->o_pos + 12 * c16; m->o_min = m->o_normal + 12 * c16; m->o_max = m->o_min + 4 * c16;
  m->o_desc = m->o_max + 4 * c16; m->o_bad = m->o_desc + 32 * c16; m->o_skip = m->o_bad + c16; m->arena_bytes = m->o_skip + c16;
  if ((rc = m->arena.reserve(m->arena_bytes)) || (rc = m->stage.reserve(m->arena_bytes)) || (rc = m->t_in_view.reserve(c)) ||
      (rc = m->t_px.reserve(c)) || (rc = m->t_py.reserve(c)) || (rc = m->t_pxr.reserve(c)) || (rc = m->t_depth.reserve(c)) ||
      (rc = m->t_vc.reserve(c)) || (rc = m->t_level.reserve(c)))
    return rc;
  return ORBG_OK;
}

extern "C" int orbm_map_create(int device, int cap_points, orbm_map** out) {
  if (!out || cap_points < 0) return ORBG_BAD_ARG;
  int rc = select_device(device);
  if (rc) return rc;
  orbm_map* m = new orbm_map();
  m->device = device;
  if (orbg::create_stream(&m->stream, "map") != hipSuccess) { delete m; return ORBG_HIP_ERROR; }
  if (hipEventCreateWithFlags(&m->up_ev, hipEventDisableTiming) != hipSuccess) { orbg::release_stream(m->stream); delete m; return ORBG_HIP_ERROR; }
  if ((rc = map_reserve(m, cap_points))) { orbm_map_destroy(m); return rc; }
  *out = m;
  return ORBG_OK;
}

extern "C" int orbm_map_set_stream(orbm_map* m, void* hip_stream) {
  if (!m) return ORBG_BAD_ARG;
  int rc = select_device(m->device);
  if (rc) return rc;
  return orbg::swap_stream(&m->stream, &m->ext_stream, hip_stream, "map");
}

extern "C" int orbm_map_destroy(orbm_map* m) {
  if (!m) return ORBG_BAD_ARG;
  (void)hipSetDevice(m->device);
  (void)hipDeviceSynchronize();
  m->arena.release(); m->stage.release();
  if (m->up_ev) (void)hipEventDestroy(m->up_ev);
  if (!m->ext_stream) orbg::release_stream(m->stream);
  m->t_in_view.release(); m->t_px.release(); m->t_py.release(); m->t_pxr.release(); m->t_depth.release();
  m->t_vc.release(); m->t_level.release();
  delete m;
  return ORBG_OK;
}

// pinned staging block -> device arena, 16 bytes per thread, every PCIe read in flight at once (map uploads, last-frame views): the
// runtime's hipMemcpyAsync takes ~26 us to move 150 KB (its blit kernel), this a few
__global__ __launch_bounds__(256) void lastview_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}

extern "C" int orbm_map_upload(orbm_map* m, const orbm_worldpoints_view* p) {
  if (!m || !p || p->m < 0) return ORBG_BAD_ARG;
  if (p->m > 0 && (!p->pos || !p->normal || !p->min_dist || !p->max_dist || !p->desc || !p->n_obs || !p->bad)) return ORBG_BAD_ARG;
  int rc = select_device(m->device);
  if (rc) return rc;
  // the staging block (and, if the buffers grow, the arena) must not be touched while the previous copy is in flight
  if (m->up_pending) { ORBG_HIP(hipEventSynchronize(m->up_ev)); m->up_pending = false; }
  if ((rc = map_reserve(m, p->m))) return rc;
  const size_t n = (size_t)p->m;
  m->m = p->m;
  m->n_obs.assign(p->n_obs, p->n_obs + n);
  m->h_bad.assign(p->bad, p->bad + n);
  if (n > 0) {
    uint8_t* S = m->stage.h;
    memcpy(S + m->o_pos, p->pos, n * 12); memcpy(S + m->o_normal, p->normal, n * 12);
    memcpy(S + m->o_min, p->min_dist, n * 4); memcpy(S + m->o_max, p->max_dist, n * 4);
    memcpy(S + m->o_desc, p->desc, n * 32); memcpy(S + m->o_bad, p->bad, n);
    if (p->skip) memcpy(S + m->o_skip, p->skip, n); else memset(S + m->o_skip, 0, n);
    const int n16 = (int)((m->arena_bytes + 15) / 16);
    hipLaunchKernelGGL(lastview_copy_kernel, dim3((n16 + 255) / 256), dim3(256), 0, m->stream, reinterpret_cast<const uint4*>(m->stage.d),
                       reinterpret_cast<uint4*>(m->arena.p), n16);
    ORBG_HIP(hipGetLastError());
    ORBG_HIP(hipEventRecord(m->up_ev, m->stream));
    m->up_pending = true;
  }
  return ORBG_OK;
}

// Per-frame refresh of the one field of an uploaded map that lives on the HOST side of the search (MapPoint::Observations(), read by
// the serial commit: S/ORBmatcher.cc:89-91): no device traffic.  For callers that keep a local map resident across frames and pass
// the per-frame exclusions (already matched / became bad) through the `skip` argument of orbm_search_local_points*.
extern "C" int orbm_map_set_observations(orbm_map* m, const int32_t* n_obs) {
  if (!m || (m->m > 0 && !n_obs)) return ORBG_BAD_ARG;
  m->n_obs.assign(n_obs, n_obs + m->m);
  return ORBG_OK;
}

static WorldPtsDev map_dev(const orbm_map* m) {
  WorldPtsDev w;
  const uint8_t* A = m->arena.p;
  w.m = m->m; w.pos = reinterpret_cast<const float*>(A + m->o_pos); w.normal = reinterpret_cast<const float*>(A + m->o_normal);
  w.min_dist = reinterpret_cast<const float*>(A + m->o_min); w.max_dist = reinterpret_cast<const float*>(A + m->o_max);
  w.desc = A + m->o_desc; w.bad = A + m->o_bad; w.skip = A + m->o_skip;
  return w;
}
static TrackDev map_track(const orbm_map* m) {
  TrackDev t;
  t.in_view = m->t_in_view.p; t.px = m->t_px.p; t.py = m->t_py.p; t.pxr = m->t_pxr.p; t.depth = m->t_depth.p;
  t.level = m->t_level.p; t.view_cos = m->t_vc.p;
  return t;
}

// Stand-alone isInFrustum for a list of points (the loop of Tracking::SearchLocalPoints): inputs and outputs travel through the
// frame's pinned staging block, which the kernel reads and writes in place (every field is touched once): no allocation, no
// copy command, one launch and one completion word per call.  Defined behind the staging helpers.
static int frustum_zero_copy(orbm_frame* f, const float* Tcw, const orbm_worldpoints_view* pts, float limit, uint8_t* track_in_view,
                             float* proj_x, float* proj_y, float* proj_xr, float* track_depth, int32_t* scale_level, float* view_cos);
extern "C" int orbm_is_in_frustum(orbm_frame* f, const float* Tcw, const orbm_worldpoints_view* pts, float limit,
                                  uint8_t* track_in_view, float* proj_x, float* proj_y, float* proj_xr, float* track_depth,
                                  int32_t* scale_level, float* view_cos) {
  if (!f || !Tcw || !pts || pts->m < 0) return ORBG_BAD_ARG;
  if (pts->m > 0 && (!pts->pos || !pts->normal || !pts->min_dist || !pts->max_dist || !track_in_view || !proj_x || !proj_y || !proj_xr ||
                     !track_depth || !scale_level || !view_cos))
    return ORBG_BAD_ARG;
  int rc = select_device(f->device);
  if (rc) return rc;
  if (pts->m == 0) return ORBG_OK;
  return frustum_zero_copy(f, Tcw, pts, limit, track_in_view, proj_x, proj_y, proj_xr, track_depth, scale_level, view_cos);
}

// ------------------------------------------------------------------------------------------------
// search drivers

static FrameDev frame_dev(orbm_frame* f) {
  FrameDev F;
  F.kps = f->kps_p; F.desc = f->desc_p; F.uright = f->has_uright ? f->uright_p : nullptr;
  F.cell_start = f->d_cell_start.p; F.cell_items = f->d_cell_items.p;
  F.assigned_mp = f->d_assigned_mp; F.assigned_obs = f->d_assigned_obs;
  F.use_mask = f->occ_mask;
  if (f->occ_mask) memcpy(F.occ, f->occ, sizeof(F.occ));
  return F;
}

// ---- staging: pack every per-call host array into one pinned block, one H2D copy
// (re)builds the cached octave / angle arrays; `fresh` forces it (a new frame was attached to the same buffers)
static void cache_keypoint_fields(orbm_frame* f) {
  const int n = f->fp.n;
  if (f->hk_cached_from == f->hk && f->hk_cached_n == n) return;
  f->hk_angle.resize((size_t)std::max(n, 1));
  f->hk_oct.resize((size_t)std::max(n, 1));
  for (int i = 0; i < n; i++) { f->hk_angle[i] = f->hk[i].angle; f->hk_oct[i] = (int8_t)f->hk[i].octave; }
  f->hk_cached_from = f->hk; f->hk_cached_n = n;
}

static int stage_begin(orbm_frame* f, size_t total_bytes) {
  int rc;
  const size_t need = total_bytes + 64 * 16;
  if ((rc = f->stage.reserve(need)) || (rc = f->d_stage.reserve(need))) return rc;
  f->stage_off = 0;
  f->copy_lo = f->copy_hi = 0;
  return ORBG_OK;
}
// Packs a per-call input into the pinned staging block.  Inputs every query reads ONCE (descriptors, projections, flags)
// are read by the kernel straight from that block (it is mapped into the device address space): no copy command at all.
// Inputs that are re-read by many queries ask for a device copy (dev_copy): one H2D over their contiguous range.
template <typename T>
static const T* stage_add(orbm_frame* f, const T* src, size_t count, bool dev_copy = false) {
  f->stage_off = (f->stage_off + 15) & ~(size_t)15;
  if (count) memcpy(f->stage.h + f->stage_off, src, count * sizeof(T));
  const T* dev = reinterpret_cast<const T*>((dev_copy ? f->d_stage.p : f->stage.d) + f->stage_off);
  if (dev_copy && count) {
    if (f->copy_hi == f->copy_lo) f->copy_lo = f->stage_off;
    f->copy_hi = f->stage_off + count * sizeof(T);
  }
  f->stage_off += count * sizeof(T);
  return dev;
}
static int stage_commit(orbm_frame* f) {
  if (f->copy_hi > f->copy_lo)
    ORBG_HIP(hipMemcpyAsync(f->d_stage.p + f->copy_lo, f->stage.h + f->copy_lo, f->copy_hi - f->copy_lo, hipMemcpyHostToDevice, f->stream));
  return ORBG_OK;
}
static int frustum_zero_copy(orbm_frame* f, const float* Tcw, const orbm_worldpoints_view* pts, float limit, uint8_t* track_in_view,
                             float* proj_x, float* proj_y, float* proj_xr, float* track_depth, int32_t* scale_level, float* view_cos) {
  const size_t n = (size_t)pts->m;
  int rc;
  if ((rc = stage_begin(f, n * (32 + 25) + 16 * 16))) return rc;
  WorldPtsDev w;
  w.m = pts->m;
  w.pos = stage_add(f, pts->pos, 3 * n); w.normal = stage_add(f, pts->normal, 3 * n);
  w.min_dist = stage_add(f, pts->min_dist, n); w.max_dist = stage_add(f, pts->max_dist, n);
  w.desc = nullptr; w.bad = nullptr; w.skip = nullptr;
  // outputs: regions of the same block
  auto out_region = [&](size_t bytes) { f->stage_off = (f->stage_off + 15) & ~(size_t)15; const size_t o = f->stage_off; f->stage_off += bytes; return o; };
  const size_t o_iv = out_region(n), o_px = out_region(4 * n), o_py = out_region(4 * n), o_pxr = out_region(4 * n), o_dep = out_region(4 * n),
               o_lvl = out_region(4 * n), o_vc = out_region(4 * n);
  uint8_t* D = f->stage.d;
  TrackDev t;
  t.in_view = D + o_iv; t.px = reinterpret_cast<float*>(D + o_px); t.py = reinterpret_cast<float*>(D + o_py);
  t.pxr = reinterpret_cast<float*>(D + o_pxr); t.depth = reinterpret_cast<float*>(D + o_dep); t.level = reinterpret_cast<int*>(D + o_lvl);
  t.view_cos = reinterpret_cast<float*>(D + o_vc);
  PoseF P;
  make_pose(Tcw, &P);
  hipLaunchKernelGGL(frustum_kernel, dim3((pts->m + 255) / 256), dim3(256), 0, f->stream, f->fp, P, w, limit, t);
  ORBG_HIP(hipGetLastError());
  if ((rc = f->sig.sync(f->stream))) return rc;
  const uint8_t* Hh = f->stage.h;
  memcpy(track_in_view, Hh + o_iv, n); memcpy(proj_x, Hh + o_px, 4 * n); memcpy(proj_y, Hh + o_py, 4 * n); memcpy(proj_xr, Hh + o_pxr, 4 * n);
  memcpy(track_depth, Hh + o_dep, 4 * n); memcpy(scale_level, Hh + o_lvl, 4 * n); memcpy(view_cos, Hh + o_vc, 4 * n);
  return ORBG_OK;
}

// F.mvpMapPoints occupancy at entry (S/ORBmatcher.cc:89-91 / :556): bitmask in the kernel arguments, or device arrays
static void stage_occupancy(orbm_frame* f, const int32_t* amp, const int32_t* aob, int n) {
  f->occ_mask = n <= kOccBits;
  if (f->occ_mask) {
    memset(f->occ, 0, sizeof(f->occ));
    for (int i = 0; i < n; i++)
      if (amp[i] >= 0 && (!aob || aob[i] > 0)) f->occ[i >> 5] |= 1u << (i & 31);
    f->d_assigned_mp = nullptr; f->d_assigned_obs = nullptr;
  } else {
    f->d_assigned_mp = stage_add(f, amp, n, true);
    f->d_assigned_obs = aob ? stage_add(f, aob, n, true) : nullptr;
  }
}

// Launches `launch(list_cap)` until the candidate list fits; leaves results + list in pinned memory.
// n_scan_host: the number of queries that actually ran is written by the launch into pinned memory (large-map path); NULL: n_queries
template <typename LaunchFn>
static int run_search(orbm_frame* f, int n_queries, LaunchFn launch, const volatile int* n_scan_host = nullptr) {
  int rc;
  if ((rc = f->results.reserve((size_t)std::max(n_queries, 1)))) return rc;
  const size_t slots = (size_t)n_queries * kSlot;
  if (f->list.cap < slots + (1 << 16) && (rc = f->list.reserve(slots + (1 << 18)))) return rc;
  for (int attempt = 0; attempt < 3; attempt++) {
    // two overflow counters take turns: every search kernel clears the one its successor will use
    int* cur = f->d_counter.p + (f->search_seq & 1);
    int* nxt = f->d_counter.p + ((f->search_seq + 1) & 1);
    f->search_seq++;
    launch((int)std::min<size_t>(f->list.cap, (size_t)1 << 30), cur, nxt);
    ORBG_HIP(hipGetLastError());
    // (the completion word cannot be posted by the search kernel itself: its results go to pinned memory from wavefronts on
    // eight XCDs, and only a system-scope release per wavefront -- a whole-L2 write-back each, 2.5x on the kernel -- orders
    // them before the word; a drained memory counter does not: measured, the host saw the word before the results)
    if ((rc = f->sig.sync(f->stream))) return rc;
    // the end of the furthest list segment tells whether the overflow region was large enough
    size_t total = 0;
    const QResult* R = f->results.h;
    const int n_scan = n_scan_host ? std::min((int)*n_scan_host, n_queries) : n_queries;
    for (int i = 0; i < n_scan; i++) total = std::max(total, (size_t)R[i].base + (size_t)(R[i].count & kQCountMask));
    if (total <= f->list.cap) return ORBG_OK;
    if ((rc = f->list.reserve(total + total / 4))) return rc;
  }
  return ORBG_CAP_EXCEEDED;
}

// Two searches -- the two cameras of a rig frame, each on its own frame object -- in flight together: both launched, then both awaited
// (one host round trip instead of two; the frames' streams may be the same one, then the kernels run back to back).  A list that
// did not fit is rare (run_search's slack): that side alone is repeated through run_search.
template <typename LaunchA, typename LaunchB>
static int run_search_pair(orbm_frame* a, orbm_frame* b, int n_queries, LaunchA launch_a, LaunchB launch_b) {
  int rc;
  orbm_frame* fs[2] = {a, b};
  for (orbm_frame* f : fs) {
    if ((rc = f->results.reserve((size_t)std::max(n_queries, 1)))) return rc;
    const size_t slots = (size_t)n_queries * kSlot;
    if (f->list.cap < slots + (1 << 16) && (rc = f->list.reserve(slots + (1 << 18)))) return rc;
  }
  for (int k = 0; k < 2; k++) {
    orbm_frame* f = fs[k];
    int* cur = f->d_counter.p + (f->search_seq & 1);
    int* nxt = f->d_counter.p + ((f->search_seq + 1) & 1);
    f->search_seq++;
    const int cap = (int)std::min<size_t>(f->list.cap, (size_t)1 << 30);
    if (k == 0) launch_a(cap, cur, nxt); else launch_b(cap, cur, nxt);
    ORBG_HIP(hipGetLastError());
    if ((rc = f->sig.post(f->stream))) return rc;
  }
  for (orbm_frame* f : fs)
    if ((rc = f->sig.wait(f->stream))) return rc;
  for (int k = 0; k < 2; k++) {
    orbm_frame* f = fs[k];
    size_t total = 0;
    const QResult* R = f->results.h;
    for (int i = 0; i < n_queries; i++) total = std::max(total, (size_t)R[i].base + (size_t)(R[i].count & kQCountMask));
    if (total <= f->list.cap) continue;
    if ((rc = f->list.reserve(total + total / 4))) return rc;
    if ((rc = k == 0 ? run_search(f, n_queries, launch_a) : run_search(f, n_queries, launch_b))) return rc;
  }
  return ORBG_OK;
}

// The two (or one) best candidates of query r that have not been claimed since the kernel ran, in the order the
// reference's sequential scan would find them.  The kernel's top-4 decides whenever it can (it holds every candidate,
// or enough unclaimed ones); otherwise the query's full list is re-scanned.
struct Pick { int idx1 = -1, dist1 = 256, idx2 = -1, dist2 = 256; };
template <typename ClaimedFn>
static int pick_unclaimed(orbm_frame* f, const QResult& r, int want, ClaimedFn claimed, Pick* out) {
  Pick p;
  int found = 0;
  for (int k = 0; k < r.n_top && found < want; k++) {
    const int idx = r.idx[k];
    if (claimed(idx)) continue;
    if (found == 0) { p.idx1 = idx; p.dist1 = r.dist[k]; } else { p.idx2 = idx; p.dist2 = r.dist[k]; }
    found++;
  }
  if (found < want && r.n_top == 4) {
    p = Pick();
    const uint32_t* list = f->list.h + r.base;
    for (int k = 0; k < (r.count & kQCountMask); k++) {
      const uint32_t e = list[k];
      if (e == 0xFFFFFFFFu) continue;
      const int idx = (int)(e & 0xFFFF), dist = (int)(e >> 16);
      if (claimed(idx)) continue;
      if (dist < p.dist1) { p.dist2 = p.dist1; p.idx2 = p.idx1; p.dist1 = dist; p.idx1 = idx; }
      else if (dist < p.dist2) { p.dist2 = dist; p.idx2 = idx; }
    }
  }
  *out = p;
  return ORBG_OK;
}

// Serial commit of SearchByProjection(Frame, MapPoints): S/ORBmatcher.cc:85-141 replayed on the GPU results.
// slot_pt: the queries are a compacted subset of the points (large-map path), R[s] belongs to point slot_pt[s]; NULL: R[i] is point i
static int commit_mps(orbm_frame* f, int m, const int32_t* n_obs, float nnratio, int32_t* amp, int32_t* aob, int* nmatches_out,
                      const int* slot_pt = nullptr) {
  const int n = f->fp.n;
  cache_keypoint_fields(f);
  f->claimed_buf.assign((size_t)std::max(n, 1), 0);  // features newly assigned in this call to an MP with Observations()>0
  uint8_t* claimed = f->claimed_buf.data();
  int nmatches = 0;
  const QResult* R = f->results.h;
  auto is_claimed = [&](int idx) { return claimed[idx] != 0; };
  for (int s = 0; s < m; s++) {
    __builtin_prefetch(&R[s + 16]);                     // results sit in pinned memory the GPU has just written
    const QResult& r = R[s];
    if (r.n_top == 0) continue;
    const int i = slot_pt ? slot_pt[s] : s;
    Pick pk;
    int rc = pick_unclaimed(f, r, 2, is_claimed, &pk);   // features claimed since the kernel ran are skipped (:89-91)
    if (rc) return rc;
    if (pk.idx1 < 0) continue;
    const int bestDist = pk.dist1, bestIdx = pk.idx1, bestDist2 = pk.dist2, idx2 = pk.idx2;
    const int bestLevel = f->hk_oct[bestIdx];
    const int bestLevel2 = idx2 >= 0 ? f->hk_oct[idx2] : -1;
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
      if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
        amp[bestIdx] = i;
        aob[bestIdx] = n_obs[i];
        if (n_obs[i] > 0) claimed[bestIdx] = 1;
        nmatches++;
      }
    }
  }
  if (nmatches_out) *nmatches_out = nmatches;
  return ORBG_OK;
}

extern "C" int orbm_search_by_projection_mps(orbm_frame* f, const orbm_mappoints_view* mps, float th, int far_points,
                                             float th_far_points, float nnratio, int32_t* assigned_mp, int32_t* assigned_obs,
                                             int* nmatches) {
  if (!f || !mps || !assigned_mp || !assigned_obs || mps->m < 0) return ORBG_BAD_ARG;
  int rc = select_device(f->device);
  if (rc) return rc;
  const int m = mps->m;
  if (nmatches) *nmatches = 0;
  if (m == 0) return ORBG_OK;
  const int n = f->fp.n;
  if ((rc = stage_begin(f, (size_t)n * 8 + (size_t)m * (2 + 32 + 6 * 4)))) return rc;
  hipStream_t st = f->stream;
  stage_occupancy(f, assigned_mp, assigned_obs, n);
  MpsDev mp;
  mp.m = m;
  mp.in_view = stage_add(f, mps->track_in_view, m); mp.bad = stage_add(f, mps->bad, m);
  mp.desc = stage_add(f, mps->desc, (size_t)m * 32);
  mp.px = stage_add(f, mps->proj_x, m); mp.py = stage_add(f, mps->proj_y, m); mp.pxr = stage_add(f, mps->proj_xr, m);
  mp.depth = stage_add(f, mps->track_depth, m); mp.view_cos = stage_add(f, mps->view_cos, m);
  mp.level = stage_add(f, mps->scale_level, m);
  if ((rc = stage_commit(f))) return rc;
  rc = run_search(f, m, [&](int list_cap, int* cnt, int* cnt_next) {
    hipLaunchKernelGGL(search_mps_kernel, dim3((m + 3) / 4), dim3(256), 0, st, f->fp, frame_dev(f), mp, th, far_points,
                       th_far_points, cnt, cnt_next, f->list.d, list_cap, f->results.d);
  });
  if (rc) return rc;
  return commit_mps(f, m, mps->n_obs, nnratio, assigned_mp, assigned_obs, nmatches);
}

extern "C" int orbm_search_local_points(orbm_frame* f, orbm_map* mp, const float* Tcw, const uint8_t* skip, float th,
                                        int far_points, float th_far_points, float nnratio, int32_t* assigned_mp,
                                        int32_t* assigned_obs, int* nmatches) {
  return orbm_search_local_points_vis(f, mp, Tcw, skip, th, far_points, th_far_points, nnratio, assigned_mp, assigned_obs, nmatches, nullptr);
}

extern "C" int orbm_search_local_points_vis(orbm_frame* f, orbm_map* mp, const float* Tcw, const uint8_t* skip, float th,
                                            int far_points, float th_far_points, float nnratio, int32_t* assigned_mp,
                                            int32_t* assigned_obs, int* nmatches, uint8_t* in_frustum) {
  if (!f || !mp || !Tcw || !assigned_mp || !assigned_obs || f->device != mp->device) return ORBG_BAD_ARG;
  int rc = select_device(f->device);
  if (rc) return rc;
  const int m = mp->m;
  if (nmatches) *nmatches = 0;
  if (m == 0) return ORBG_OK;
  const int n = f->fp.n;
  if ((rc = stage_begin(f, (size_t)n * 8 + (size_t)m))) return rc;
  hipStream_t st = f->stream;
  stage_occupancy(f, assigned_mp, assigned_obs, n);
  PoseF P;
  make_pose(Tcw, &P);
  const uint8_t* d_skip_call = skip ? stage_add(f, skip, m) : nullptr;
  if ((rc = stage_commit(f))) return rc;
  if ((rc = map_sync_to(mp, st))) return rc;       // the map's last upload may still be in flight on its own stream
  const WorldPtsDev w = map_dev(mp);
  // large maps: frustum test one thread per point, compaction, window search only for the points that become queries
  static const int cull_min = []() { const char* e = getenv("ORBG_CULL_MIN"); return e ? atoi(e) : 8192; }();
  if (m >= cull_min) {
    const int nb = (m + 255) / 256;
    if ((rc = f->d_qflag.reserve((size_t)m)) || (rc = f->d_blk_cnt.reserve((size_t)nb)) || (rc = f->d_slot_pt.reserve((size_t)m)) ||
        (rc = f->d_total.reserve(4)) || (rc = f->h_slot_pt.reserve((size_t)m + 4)) || (in_frustum && (rc = f->h_vis.reserve((size_t)m))))
      return rc;
    const int grid_search = std::min((m + 3) / 4, 2048);
    rc = run_search(f, m, [&](int list_cap, int* cnt, int* cnt_next) {
      hipLaunchKernelGGL(cull_count_kernel, dim3(nb), dim3(256), 0, st, f->fp, w, d_skip_call, P, th, far_points, th_far_points, f->d_qflag.p,
                         in_frustum ? f->h_vis.d : (uint8_t*)nullptr, f->d_blk_cnt.p);
      hipLaunchKernelGGL(cull_fill_kernel, dim3(nb), dim3(256), 0, st, m, f->d_qflag.p, f->d_blk_cnt.p, f->d_slot_pt.p, f->h_slot_pt.d + 4,
                         f->d_total.p, f->h_slot_pt.d);
      hipLaunchKernelGGL(search_culled_kernel, dim3(grid_search), dim3(256), 0, st, f->fp, frame_dev(f), w, d_skip_call, P, th, far_points,
                         th_far_points, f->d_slot_pt.p, f->d_total.p, m, cnt, cnt_next, f->list.d, list_cap, f->results.d);
    }, f->h_slot_pt.h);
    if (rc) return rc;
    const int total = std::min(f->h_slot_pt.h[0], m);
    if (in_frustum) memcpy(in_frustum, f->h_vis.h, (size_t)m);
    return commit_mps(f, total, mp->n_obs.data(), nnratio, assigned_mp, assigned_obs, nmatches, f->h_slot_pt.h + 4);
  }
  rc = run_search(f, m, [&](int list_cap, int* cnt, int* cnt_next) {
    hipLaunchKernelGGL(search_local_kernel, dim3((m + 3) / 4), dim3(256), 0, st, f->fp, frame_dev(f), w, d_skip_call, P, th, far_points,
                       th_far_points, cnt, cnt_next, f->list.d, list_cap, f->results.d);
  });
  if (rc) return rc;
  if (in_frustum) {
    const QResult* R = f->results.h;
    for (int i = 0; i < m; i++) in_frustum[i] = (R[i].count & kQVisible) ? 1 : 0;
  }
  return commit_mps(f, m, mp->n_obs.data(), nnratio, assigned_mp, assigned_obs, nmatches);
}

// ---- two-camera rig frames (Frame::Nleft != -1)

// S/Frame.cc:1158-1170: the cv::Mat products of the right camera (float32 small-matrix rules, as make_pose)
static RigSideF rig_side_of(const PoseF& P, const orbg_camera_rig* rig, const float* Tlr, bool right) {
  RigSideF s;
  if (!right) {
    memcpy(s.R, P.R, sizeof(s.R)); memcpy(s.t, P.t, sizeof(s.t)); memcpy(s.twc, P.Ow, sizeof(s.twc));
    s.cam = rig_cam_of(rig->left);
    return s;
  }
  const float* Trl = rig->Trl;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) s.R[3 * i + j] = Trl[4 * i] * P.R[j] + Trl[4 * i + 1] * P.R[3 + j] + Trl[4 * i + 2] * P.R[6 + j];
    const float t0 = Trl[4 * i] * P.t[0] + Trl[4 * i + 1] * P.t[1] + Trl[4 * i + 2] * P.t[2];
    s.t[i] = t0 + Trl[4 * i + 3];
    const float w0 = P.R[i] * Tlr[3] + P.R[3 + i] * Tlr[7] + P.R[6 + i] * Tlr[11];
    s.twc[i] = w0 + P.Ow[i];
  }
  s.cam = rig_cam_of(rig->right);
  return s;
}

extern "C" int orbm_is_in_frustum_rig(orbm_frame* f, const float* Tcw, const orbg_camera_rig* rig, const float* Tlr, const orbm_worldpoints_view* pts,
                                      float limit, uint8_t* in_view, float* proj_x, float* proj_y, float* track_depth, int32_t* scale_level,
                                      float* view_cos, uint8_t* in_view_r, float* proj_x_r, float* proj_y_r, float* track_depth_r,
                                      int32_t* scale_level_r, float* view_cos_r) {
  if (!f || !Tcw || !rig || !pts || pts->m < 0 || (rig->has_right && !Tlr)) return ORBG_BAD_ARG;
  const bool two = rig->has_right != 0;                     // one camera behind a model (Nleft == -1, S/Frame.cc:466-543 with mpCamera->project): left outputs only
  if ((rig->left.model != ORBG_CAM_PINHOLE && rig->left.model != ORBG_CAM_KANNALA_BRANDT8) ||
      (two && rig->right.model != ORBG_CAM_PINHOLE && rig->right.model != ORBG_CAM_KANNALA_BRANDT8))
    return ORBG_BAD_ARG;
  if (pts->m > 0 && (!pts->pos || !pts->normal || !pts->min_dist || !pts->max_dist || !in_view || !proj_x || !proj_y || !track_depth ||
                     !scale_level || !view_cos || (two && (!in_view_r || !proj_x_r || !proj_y_r || !track_depth_r || !scale_level_r || !view_cos_r))))
    return ORBG_BAD_ARG;
  int rc = select_device(f->device);
  if (rc) return rc;
  if (pts->m == 0) return ORBG_OK;
  const size_t n = (size_t)pts->m;
  if ((rc = stage_begin(f, n * (32 + 2 * 21) + 16 * 24))) return rc;
  WorldPtsDev w;
  w.m = pts->m;
  w.pos = stage_add(f, pts->pos, 3 * n); w.normal = stage_add(f, pts->normal, 3 * n);
  w.min_dist = stage_add(f, pts->min_dist, n); w.max_dist = stage_add(f, pts->max_dist, n);
  w.desc = nullptr; w.bad = nullptr; w.skip = nullptr;
  auto out_region = [&](size_t bytes) { f->stage_off = (f->stage_off + 15) & ~(size_t)15; const size_t o = f->stage_off; f->stage_off += bytes; return o; };
  size_t off[2][6];
  TrackDev t[2];
  uint8_t* D = f->stage.d;
  for (int sd = 0; sd < 2; sd++) {
    off[sd][0] = out_region(n);
    for (int k = 1; k < 6; k++) off[sd][k] = out_region(4 * n);
    t[sd].in_view = D + off[sd][0]; t[sd].px = reinterpret_cast<float*>(D + off[sd][1]); t[sd].py = reinterpret_cast<float*>(D + off[sd][2]);
    t[sd].depth = reinterpret_cast<float*>(D + off[sd][3]); t[sd].level = reinterpret_cast<int*>(D + off[sd][4]);
    t[sd].view_cos = reinterpret_cast<float*>(D + off[sd][5]); t[sd].pxr = nullptr;
  }
  PoseF P;
  make_pose(Tcw, &P);
  hipLaunchKernelGGL(frustum_rig_kernel, dim3((pts->m + 255) / 256), dim3(256), 0, f->stream, f->fp, rig_side_of(P, rig, Tlr, false),
                     rig_side_of(P, rig, Tlr, two), two ? 1 : 0, w, limit, t[0], t[1]);
  ORBG_HIP(hipGetLastError());
  if ((rc = f->sig.sync(f->stream))) return rc;
  const uint8_t* Hh = f->stage.h;
  void* outs[2][6] = {{in_view, proj_x, proj_y, track_depth, scale_level, view_cos},
                      {in_view_r, proj_x_r, proj_y_r, track_depth_r, scale_level_r, view_cos_r}};
  for (int sd = 0; sd < (two ? 2 : 1); sd++)
    for (int k = 0; k < 6; k++) memcpy(outs[sd][k], Hh + off[sd][k], k == 0 ? n : 4 * n);
  return ORBG_OK;
}

// One camera's window searches of the rig form of SearchByProjection(Frame, MapPoints): the kernel of the single-camera form on that
// camera's frame; `amp` / `aob`: the occupancy the kernel filters by (that camera's part of mvpMapPoints), or NULL for none.  Stages
// the inputs and returns the launch.
struct MpsRigSide { MpsDev mp; FrameDev F; };
static int stage_mps_rig_side(orbm_frame* f, const orbm_mappoints_view* pt, const uint8_t* in_view, const float* px, const float* py,
                              const int32_t* level, const float* view_cos, const int32_t* amp, const int32_t* aob, MpsRigSide* out) {
  const int m = pt->m, n = f->fp.n;
  int rc;
  if ((rc = stage_begin(f, (size_t)n * 8 + (size_t)m * (2 + 32 + 6 * 4)))) return rc;
  std::vector<int32_t> none;
  if (!amp) { none.assign((size_t)std::max(n, 1), -1); amp = none.data(); aob = nullptr; }
  stage_occupancy(f, amp, aob, n);
  MpsDev& mp = out->mp;
  mp.m = m;
  mp.in_view = stage_add(f, in_view, m); mp.bad = stage_add(f, pt->bad, m);
  mp.desc = stage_add(f, pt->desc, (size_t)m * 32);
  mp.px = stage_add(f, px, m); mp.py = stage_add(f, py, m); mp.pxr = mp.px;   // (not read: the frame has no uRight)
  mp.depth = stage_add(f, pt->track_depth, m); mp.view_cos = stage_add(f, view_cos, m);
  mp.level = stage_add(f, level, m);
  if ((rc = stage_commit(f))) return rc;
  out->F = frame_dev(f);
  out->F.uright = nullptr;                                // S/ORBmatcher.cc:93: the mvuRight test is for Nleft == -1 only
  return ORBG_OK;
}

extern "C" int orbm_search_by_projection_mps_rig(orbm_frame* L, orbm_frame* R, const orbm_mappoints_view* mps, const orbm_mappoints_view* mps_r,
                                                 const int32_t* l2r, const int32_t* r2l, float th, int far_points, float th_far_points,
                                                 float nnratio, int32_t* amp, int32_t* aob, int* nmatches_out) {
  if (!L || !R || L == R || !mps || !mps_r || !amp || !aob || mps->m < 0 || mps_r->m != mps->m || L->device != R->device) return ORBG_BAD_ARG;
  const int m = mps->m, nl = L->fp.n, nr = R->fp.n;
  if ((nl > 0 && !l2r) || (nr > 0 && !r2l)) return ORBG_BAD_ARG;
  for (int i = 0; i < nl; i++) if (l2r[i] < -1 || l2r[i] >= nr) return ORBG_BAD_ARG;
  for (int i = 0; i < nr; i++) if (r2l[i] < -1 || r2l[i] >= nl) return ORBG_BAD_ARG;
  int rc = select_device(L->device);
  if (rc) return rc;
  if (nmatches_out) *nmatches_out = 0;
  if (m == 0) return ORBG_OK;
  // the right camera's block is entered only with a predicted level (:146-147)
  std::vector<uint8_t> in_view_r((size_t)m);
  for (int i = 0; i < m; i++) in_view_r[i] = mps_r->track_in_view[i] && mps_r->scale_level[i] != -1;
  const std::vector<int32_t> amp0(amp, amp + nl + nr), aob0(aob, aob + nl + nr);
  cache_keypoint_fields(L); cache_keypoint_fields(R);
  // Pass 0: the kernels leave out the features that hold a point with observations at entry, the commit those taken since.  A
  // stereo partner is written whatever it held (:133,200): should that replace a point with observations by one without, the
  // feature has become free and the lists lack it -- pass 1 repeats the call on lists nothing was left out of.
  for (int pass = 0; pass < 2; pass++) {
    const bool filtered = pass == 0;
    std::copy(amp0.begin(), amp0.end(), amp); std::copy(aob0.begin(), aob0.end(), aob);
    MpsRigSide sl, sr;
    if ((rc = stage_mps_rig_side(L, mps, mps->track_in_view, mps->proj_x, mps->proj_y, mps->scale_level, mps->view_cos, filtered ? amp : nullptr,
                                 filtered ? aob : nullptr, &sl)) ||
        (rc = stage_mps_rig_side(R, mps, in_view_r.data(), mps_r->proj_x, mps_r->proj_y, mps_r->scale_level, mps_r->view_cos,
                                 filtered ? amp + nl : nullptr, filtered ? aob + nl : nullptr, &sr)))
      return rc;
    rc = run_search_pair(L, R, m,
      [&](int list_cap, int* cnt, int* cnt_next) {
        hipLaunchKernelGGL(search_mps_kernel, dim3((m + 3) / 4), dim3(256), 0, L->stream, L->fp, sl.F, sl.mp, th, far_points, th_far_points, cnt,
                           cnt_next, L->list.d, list_cap, L->results.d);
      },
      [&](int list_cap, int* cnt, int* cnt_next) {                 // (th is not applied to the right camera's window, :148-151)
        hipLaunchKernelGGL(search_mps_kernel, dim3((m + 3) / 4), dim3(256), 0, R->stream, R->fp, sr.F, sr.mp, 1.0f, far_points, th_far_points, cnt,
                           cnt_next, R->list.d, list_cap, R->results.d);
      });
    if (rc) return rc;
    const QResult* RL = L->results.h;
    const QResult* RR = R->results.h;
    bool freed = false;
    int nmatches = 0;
    auto assign = [&](int g, int i) {
      if (filtered && amp0[g] >= 0 && aob0[g] > 0 && mps->n_obs[i] <= 0) freed = true;
      amp[g] = i; aob[g] = mps->n_obs[i];
    };
    for (int i = 0; i < m && !freed; i++) {
      if (mps->track_in_view[i] && RL[i].n_top) {                                            // :62-143
        Pick pk;
        if ((rc = pick_unclaimed(L, RL[i], 2, [&](int idx) { return amp[idx] >= 0 && aob[idx] > 0; }, &pk))) return rc;
        if (pk.idx1 >= 0 && pk.dist1 <= TH_HIGH) {
          const int bestLevel = L->hk_oct[pk.idx1], bestLevel2 = pk.idx2 >= 0 ? L->hk_oct[pk.idx2] : -1;
          if (bestLevel == bestLevel2 && pk.dist1 > nnratio * pk.dist2) continue;            // :126-127 leaves the point
          if (bestLevel != bestLevel2 || pk.dist1 <= nnratio * pk.dist2) {
            assign(pk.idx1, i);
            if (l2r[pk.idx1] != -1) { assign(l2r[pk.idx1] + nl, i); nmatches++; }
            nmatches++;
          }
        }
      }
      if (in_view_r[i] && RR[i].n_top) {                                                      // :145-211
        Pick pk;
        if ((rc = pick_unclaimed(R, RR[i], 2, [&](int idx) { return amp[idx + nl] >= 0 && aob[idx + nl] > 0; }, &pk))) return rc;
        if (pk.idx1 >= 0 && pk.dist1 <= TH_HIGH) {
          const int bestLevel = R->hk_oct[pk.idx1], bestLevel2 = pk.idx2 >= 0 ? R->hk_oct[pk.idx2] : -1;
          if (bestLevel == bestLevel2 && pk.dist1 > nnratio * pk.dist2) continue;
          if (r2l[pk.idx1] != -1) { assign(r2l[pk.idx1], i); nmatches++; }
          assign(pk.idx1 + nl, i);
          nmatches++;
        }
      }
    }
    if (!freed) {
      if (nmatches_out) *nmatches_out = nmatches;
      return ORBG_OK;
    }
  }
  return ORBG_INTERNAL;
}

extern "C" int orbm_search_by_projection_frame_rig(orbm_frame* FL, orbm_frame* FR, const float* Tcw_cur, const orbg_camera_rig* rig,
                                                   const orbm_lastframe_view* last, float th, int mono, int check_orientation,
                                                   int32_t* amp, int32_t* aob, int* nmatches_out) {
  if (!FL || !Tcw_cur || !rig || !last || !amp || !aob || last->n < 0) return ORBG_BAD_ARG;
  // rig->has_right == 0 with right == NULL: ONE camera behind a model (CurrentFrame.Nleft == -1 and mpCamera a fisheye): the left
  // camera's search alone (:2001-2091 with mpCamera->project; no mvuRight on a monocular frame)
  const bool two = rig->has_right != 0;
  if (two ? (!FR || FL == FR || FL->device != FR->device) : FR != nullptr) return ORBG_BAD_ARG;
  if (rig->left.model != ORBG_CAM_PINHOLE && rig->left.model != ORBG_CAM_KANNALA_BRANDT8) return ORBG_BAD_ARG;
  int rc = select_device(FL->device);
  if (rc) return rc;
  const int m = last->n, nl = FL->fp.n;
  if (nmatches_out) *nmatches_out = 0;
  if (m == 0) return ORBG_OK;
  PoseF Pc, Pl;
  make_pose(Tcw_cur, &Pc);
  make_pose(last->Tcw, &Pl);
  float tlc[3];
  for (int i = 0; i < 3; i++) {
    const float t0 = Pl.R[3 * i] * Pc.Ow[0] + Pl.R[3 * i + 1] * Pc.Ow[1] + Pl.R[3 * i + 2] * Pc.Ow[2];
    tlc[i] = t0 + Pl.t[i];
  }
  const int forward = tlc[2] > FL->fp.b && !mono;
  const int backward = -tlc[2] > FL->fp.b && !mono;
  const RigCamF cam = rig_cam_of(rig->left);
  TrlF Trl;
  memcpy(Trl.m, rig->Trl, sizeof(Trl.m));
  // The kernels list every candidate of a window, taken or not: "the left camera's window is empty" ends a point's turn (:2033-2034)
  // and must be told from "every candidate in it is taken"; the commit below tests mvpMapPoints as it stands at that moment.
  LastDev Ls[2]; FrameDev Fs[2];
  for (int side = 0; side < (two ? 2 : 1); side++) {
    orbm_frame* f = side ? FR : FL;
    const int n = f->fp.n;
    if ((rc = stage_begin(f, (size_t)n * 8 + (size_t)m * (2 + 32 + 12 + 4)))) return rc;
    const std::vector<int32_t> none((size_t)std::max(n, 1), -1);
    stage_occupancy(f, none.data(), nullptr, n);
    LastDev& L = Ls[side];
    L.n = m;
    L.mp_valid = stage_add(f, last->mp_valid, m); L.outlier = stage_add(f, last->outlier, m);
    L.desc = stage_add(f, last->desc, (size_t)m * 32);
    L.world_pos = stage_add(f, last->world_pos, (size_t)m * 3);
    L.octave = stage_add(f, last->octave, m);
    if ((rc = stage_commit(f))) return rc;
    Fs[side] = frame_dev(f);
    Fs[side].uright = nullptr;                             // :2049: the mvuRight test is for Nleft == -1 only
  }
  auto launch = [&](int side) {
    return [&, side](int list_cap, int* cnt, int* cnt_next) {
      orbm_frame* f = side ? FR : FL;
      hipLaunchKernelGGL(search_frame_rig_kernel, dim3((m + 3) / 4), dim3(256), 0, f->stream, f->fp, Fs[side], Ls[side], Pc, cam, Trl, side, th, forward,
                         backward, cnt, cnt_next, f->list.d, list_cap, f->results.d);
    };
  };
  if ((rc = two ? run_search_pair(FL, FR, m, launch(0), launch(1)) : run_search(FL, m, launch(0)))) return rc;
  cache_keypoint_fields(FL);
  if (two) cache_keypoint_fields(FR);
  RotHist rotHist(FL->rot_entries);
  int nmatches = 0;
  const QResult* RL = FL->results.h;
  const QResult* RR = two ? FR->results.h : nullptr;
  for (int i = 0; i < m; i++) {
    if (RL[i].n_top == 0) continue;                        // no query, or vIndices2.empty()
    Pick pk;
    if ((rc = pick_unclaimed(FL, RL[i], 1, [&](int idx) { return amp[idx] >= 0 && aob[idx] > 0; }, &pk))) return rc;
    if (pk.idx1 >= 0 && pk.dist1 <= TH_HIGH) {
      amp[pk.idx1] = i; aob[pk.idx1] = last->n_obs[i];
      nmatches++;
      if (check_orientation) rotHist.add(rot_bin(last->angle[i], FL->hk_angle[pk.idx1]), pk.idx1);
    }
    if (!two || RR[i].n_top == 0) continue;
    if ((rc = pick_unclaimed(FR, RR[i], 1, [&](int idx) { return amp[idx + nl] >= 0 && aob[idx + nl] > 0; }, &pk))) return rc;
    if (pk.idx1 >= 0 && pk.dist1 <= TH_HIGH) {
      amp[pk.idx1 + nl] = i; aob[pk.idx1 + nl] = last->n_obs[i];
      nmatches++;
      if (check_orientation) rotHist.add(rot_bin(last->angle[i], FR->hk_angle[pk.idx1]), pk.idx1 + nl);
    }
  }
  if (check_orientation) rotHist.reject_outside_three_maxima([&](int idx) { amp[idx] = -1; aob[idx] = 0; nmatches--; });
  if (nmatches_out) *nmatches_out = nmatches;
  return ORBG_OK;
}

// Common part of the two entry points below.  `stage_view`: the view is packed into the frame's pinned staging block and read there
// by the kernel (one PCIe trip per query); otherwise L already points at a device-resident copy (orbm_lastview_upload).
static int search_frame_common(orbm_frame* f, const float* Tcw_cur, const float* Tcw_last, int m, LastDev L, const int32_t* n_obs_last,
                               const float* angle_last, hipEvent_t wait_ev, float th, int mono, int check_orientation,
                               int32_t* assigned_mp, int32_t* assigned_obs, int* nmatches_out) {
  int rc;
  const int n = f->fp.n;
  hipStream_t st = f->stream;
  PoseF Pc, Pl;
  make_pose(Tcw_cur, &Pc);
  make_pose(Tcw_last, &Pl);
  // tlc = Rlw*twc + tlw (S/ORBmatcher.cc:1983-1991)
  float tlc[3];
  for (int i = 0; i < 3; i++) {
    const float t0 = Pl.R[3 * i] * Pc.Ow[0] + Pl.R[3 * i + 1] * Pc.Ow[1] + Pl.R[3 * i + 2] * Pc.Ow[2];
    tlc[i] = t0 + Pl.t[i];
  }
  const int forward = tlc[2] > f->fp.b && !mono;
  const int backward = -tlc[2] > f->fp.b && !mono;
  if (wait_ev) ORBG_HIP(hipStreamWaitEvent(st, wait_ev, 0));
  rc = run_search(f, m, [&](int list_cap, int* cnt, int* cnt_next) {
    hipLaunchKernelGGL(search_frame_kernel, dim3((m + 3) / 4), dim3(256), 0, st, f->fp, frame_dev(f), L, Pc, th, forward, backward,
                       cnt, cnt_next, f->list.d, list_cap, f->results.d);
  });
  if (rc) return rc;
  // serial commit (S/ORBmatcher.cc:2041-2091) + rotation consistency (:2164-2183)
  cache_keypoint_fields(f);
  f->claimed_buf.assign((size_t)std::max(n, 1), 0);
  uint8_t* claimed = f->claimed_buf.data();
  RotHist rotHist(f->rot_entries);
  int nmatches = 0;
  const QResult* R = f->results.h;
  auto is_claimed = [&](int idx) { return claimed[idx] != 0; };
  for (int i = 0; i < m; i++) {
    __builtin_prefetch(&R[i + 16]);                     // results sit in pinned memory the GPU has just written
    const QResult& r = R[i];
    if (r.n_top == 0) continue;
    Pick pk;
    if ((rc = pick_unclaimed(f, r, 1, is_claimed, &pk))) return rc;
    if (pk.idx1 < 0) continue;
    const int bestDist = pk.dist1, bestIdx = pk.idx1;
    if (bestDist <= TH_HIGH) {
      assigned_mp[bestIdx] = i;
      assigned_obs[bestIdx] = n_obs_last[i];
      if (n_obs_last[i] > 0) claimed[bestIdx] = 1;
      nmatches++;
      if (check_orientation) rotHist.add(rot_bin(angle_last[i], f->hk_angle[bestIdx]), bestIdx);
    }
  }
  if (check_orientation)
    rotHist.reject_outside_three_maxima([&](int idx) { assigned_mp[idx] = -1; assigned_obs[idx] = 0; nmatches--; });
  if (nmatches_out) *nmatches_out = nmatches;
  return ORBG_OK;
}

extern "C" int orbm_search_by_projection_frame(orbm_frame* f, const float* Tcw_cur, const orbm_lastframe_view* last, float th,
                                               int mono, int check_orientation, int32_t* assigned_mp, int32_t* assigned_obs,
                                               int* nmatches_out) {
  if (!f || !Tcw_cur || !last || !assigned_mp || !assigned_obs || last->n < 0) return ORBG_BAD_ARG;
  int rc = select_device(f->device);
  if (rc) return rc;
  const int m = last->n;
  if (nmatches_out) *nmatches_out = 0;
  if (m == 0) return ORBG_OK;
  const int n = f->fp.n;
  if ((rc = stage_begin(f, (size_t)n * 8 + (size_t)m * (2 + 32 + 12 + 4)))) return rc;
  stage_occupancy(f, assigned_mp, assigned_obs, n);
  LastDev L;
  L.n = m;
  L.mp_valid = stage_add(f, last->mp_valid, m); L.outlier = stage_add(f, last->outlier, m);
  L.desc = stage_add(f, last->desc, (size_t)m * 32);
  L.world_pos = stage_add(f, last->world_pos, (size_t)m * 3);
  L.octave = stage_add(f, last->octave, m);
  if ((rc = stage_commit(f))) return rc;
  return search_frame_common(f, Tcw_cur, last->Tcw, m, L, last->n_obs, last->angle, nullptr, th, mono, check_orientation, assigned_mp,
                             assigned_obs, nmatches_out);
}

// ---- the last frame's view resident on the device (round 4).  Tracking knows mLastFrame's map points when it has finished tracking
// that frame (S/Tracking.cc:2086-2090: mLastFrame = Frame(mCurrentFrame)) -- a whole frame time before SearchByProjection(Current,
// Last) reads them.  orbm_lastview_upload takes the view THEN: one packed copy on the library's M stream while nothing else crosses
// PCIe; the search kernel of the next frame reads HBM.  Read in place (the form above) the queries cross PCIe at the start of the
// step, together with the 614 KB image upload of the constructor that runs next to the search: 24 us per call on an idle GPU,
// 44 us in the agent.
struct orbm_lastview {
  int device = 0;
  hipStream_t stream = nullptr;
  bool ext_stream = false;
  DevBuf<uint8_t> arena;
  PinnedBuf<uint8_t> stage;
  hipEvent_t ev = nullptr;
  bool pending = false;
  int n = 0;
  size_t o_valid = 0, o_outl = 0, o_desc = 0, o_pos = 0, o_oct = 0, bytes = 0;
  std::vector<int32_t> n_obs;
  std::vector<float> angle;
  float Tcw[16] = {0};
};

extern "C" int orbm_lastview_create(int device, int cap_features, orbm_lastview** out) {
  if (!out || cap_features < 0) return ORBG_BAD_ARG;
  int rc = select_device(device);
  if (rc) return rc;
  orbm_lastview* v = new orbm_lastview();
  v->device = device;
  if (orbg::create_stream(&v->stream, "map") != hipSuccess) { delete v; return ORBG_HIP_ERROR; }
  if (hipEventCreateWithFlags(&v->ev, hipEventDisableTiming) != hipSuccess) { orbg::release_stream(v->stream); delete v; return ORBG_HIP_ERROR; }
  const size_t need = (size_t)std::max(cap_features, 16) * 64 + 256;
  if ((rc = v->arena.reserve(need)) || (rc = v->stage.reserve(need))) { v->arena.release(); v->stage.release(); (void)hipEventDestroy(v->ev); orbg::release_stream(v->stream); delete v; return rc; }
  *out = v;
  return ORBG_OK;
}

extern "C" int orbm_lastview_destroy(orbm_lastview* v) {
  if (!v) return ORBG_BAD_ARG;
  (void)hipSetDevice(v->device);
  (void)hipStreamSynchronize(v->stream);
  v->arena.release(); v->stage.release();
  if (v->ev) (void)hipEventDestroy(v->ev);
  if (!v->ext_stream) orbg::release_stream(v->stream);
  delete v;
  return ORBG_OK;
}

extern "C" int orbm_lastview_upload(orbm_lastview* v, const orbm_lastframe_view* last) {
  if (!v || !last || last->n < 0) return ORBG_BAD_ARG;
  const size_t m = (size_t)last->n;
  if (m > 0 && (!last->mp_valid || !last->outlier || !last->world_pos || !last->desc || !last->octave || !last->angle || !last->n_obs)) return ORBG_BAD_ARG;
  int rc = select_device(v->device);
  if (rc) return rc;
  if (v->pending) { ORBG_HIP(hipEventSynchronize(v->ev)); v->pending = false; }      // the staging block of the previous upload
  auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
  v->o_valid = 0; v->o_outl = al(m); v->o_desc = v->o_outl + al(m); v->o_pos = v->o_desc + al(32 * m); v->o_oct = v->o_pos + al(12 * m);
  v->bytes = v->o_oct + al(4 * m);
  if ((rc = v->arena.reserve(v->bytes + 64)) || (rc = v->stage.reserve(v->bytes + 64))) return rc;
  v->n = last->n;
  v->n_obs.assign(last->n_obs, last->n_obs + m);
  v->angle.assign(last->angle, last->angle + m);
  memcpy(v->Tcw, last->Tcw, sizeof(v->Tcw));
  if (m > 0) {
    uint8_t* S = v->stage.h;
    memcpy(S + v->o_valid, last->mp_valid, m); memcpy(S + v->o_outl, last->outlier, m); memcpy(S + v->o_desc, last->desc, 32 * m);
    memcpy(S + v->o_pos, last->world_pos, 12 * m); memcpy(S + v->o_oct, last->octave, 4 * m);
    const int n16 = (int)((v->bytes + 15) / 16);
    hipLaunchKernelGGL(lastview_copy_kernel, dim3((n16 + 255) / 256), dim3(256), 0, v->stream, reinterpret_cast<const uint4*>(v->stage.d),
                       reinterpret_cast<uint4*>(v->arena.p), n16);
    ORBG_HIP(hipGetLastError());
    ORBG_HIP(hipEventRecord(v->ev, v->stream));
    v->pending = true;
  }
  return ORBG_OK;
}

extern "C" int orbm_search_by_projection_frame_resident(orbm_frame* f, const float* Tcw_cur, orbm_lastview* v, float th, int mono,
                                                        int check_orientation, int32_t* assigned_mp, int32_t* assigned_obs, int* nmatches_out) {
  if (!f || !Tcw_cur || !v || !assigned_mp || !assigned_obs || f->device != v->device) return ORBG_BAD_ARG;
  int rc = select_device(f->device);
  if (rc) return rc;
  const int m = v->n;
  if (nmatches_out) *nmatches_out = 0;
  if (m == 0) return ORBG_OK;
  const int n = f->fp.n;
  if ((rc = stage_begin(f, (size_t)n * 8))) return rc;
  stage_occupancy(f, assigned_mp, assigned_obs, n);
  if ((rc = stage_commit(f))) return rc;
  const uint8_t* A = v->arena.p;
  LastDev L;
  L.n = m;
  L.mp_valid = A + v->o_valid; L.outlier = A + v->o_outl; L.desc = A + v->o_desc;
  L.world_pos = reinterpret_cast<const float*>(A + v->o_pos); L.octave = reinterpret_cast<const int*>(A + v->o_oct);
  hipEvent_t wait_ev = nullptr;
  if (v->pending) {
    if (hipEventQuery(v->ev) == hipSuccess) v->pending = false;
    else { (void)hipGetLastError(); wait_ev = v->ev; }
  }
  return search_frame_common(f, Tcw_cur, v->Tcw, m, L, v->n_obs.data(), v->angle.data(), wait_ev, th, mono, check_orientation, assigned_mp,
                             assigned_obs, nmatches_out);
}

// Shared body of SearchByBoW(KeyFrame*, Frame&) (S/ORBmatcher.cc:269-471, by_query = false) and
// SearchByBoW(KeyFrame*, KeyFrame*) (:819-959, by_query = true).  Queries = the flattened pKF / pKF1 side; targets = f.
static int bow_common(orbm_frame* f, const orbm_featvec_view* fvF, const uint8_t* t_valid, const uint8_t* kf_desc, int nkf,
                      const uint8_t* kf_mp_valid, const float* kf_angle, const orbm_featvec_view* fvK, float nnratio,
                      int check_orientation, bool by_query, int32_t* matches, int* nmatches_out, int n_left = -1) {
  int rc = select_device(f->device);
  if (rc) return rc;
  const int n = f->fp.n;
  const int n_out = by_query ? nkf : n;
  for (int i = 0; i < n_out; i++) matches[i] = -1;
  if (nmatches_out) *nmatches_out = 0;
  // merge-join of the two sorted feature vectors (S/ORBmatcher.cc:290-448) -> one job per valid KF feature
  std::vector<BowJob> jobs;
  {
    int k = 0, ff = 0;
    while (k < fvK->n_nodes && ff < fvF->n_nodes) {
      if (fvK->node_id[k] == fvF->node_id[ff]) {
        for (uint32_t a = fvK->start[k]; a < fvK->start[k + 1]; a++) {
          const uint32_t kfi = fvK->feat_idx[a];
          if ((int)kfi >= nkf) return ORBG_BAD_ARG;
          if (!kf_mp_valid[kfi]) continue;
          jobs.push_back(BowJob{(int)kfi, (int)fvF->start[ff], (int)fvF->start[ff + 1]});
        }
        k++; ff++;
      } else if (fvK->node_id[k] < fvF->node_id[ff]) {
        k = (int)(std::lower_bound(fvK->node_id, fvK->node_id + fvK->n_nodes, fvF->node_id[ff]) - fvK->node_id);
      } else {
        ff = (int)(std::lower_bound(fvF->node_id, fvF->node_id + fvF->n_nodes, fvK->node_id[k]) - fvF->node_id);
      }
    }
  }
  const int nq = (int)jobs.size();                         // queries: the valid KF features in the merge-join's order
  if (nq == 0) return ORBG_OK;
  if (n_left >= 0) jobs.insert(jobs.end(), jobs.begin(), jobs.begin() + nq);   // (two-camera Frame) ... once per camera
  const int nj = (int)jobs.size();
  const int nfi = (int)fvF->start[fvF->n_nodes];
  for (int i = 0; i < nfi; i++)
    if ((int)fvF->feat_idx[i] >= n) return ORBG_BAD_ARG;
  if ((rc = stage_begin(f, (size_t)nj * sizeof(BowJob) + (size_t)nfi * 4 + (size_t)nkf * 32 + (size_t)n))) return rc;
  hipStream_t st = f->stream;
  const uint32_t* d_fidx = stage_add(f, fvF->feat_idx, nfi, true);
  const uint8_t* d_tvalid = t_valid ? stage_add(f, t_valid, n, true) : nullptr;
  const BowJob* d_jobs = stage_add(f, jobs.data(), nj);
  const uint8_t* d_kfdesc = stage_add(f, kf_desc, (size_t)nkf * 32);
  if ((rc = stage_commit(f))) return rc;
  rc = run_search(f, nj, [&](int list_cap, int* cnt, int* cnt_next) {
    hipLaunchKernelGGL(search_bow_kernel, dim3((nj + 3) / 4), dim3(256), 0, st, f->desc_p, d_fidx, d_tvalid, d_kfdesc,
                       d_jobs, nj, cnt, cnt_next, f->list.d, list_cap, f->results.d, n_left);
  });
  if (rc) return rc;
  cache_keypoint_fields(f);
  RotHist rotHist(f->rot_entries);
  f->claimed_buf.assign((size_t)std::max(n, 1), 0);       // vpMapPointMatches[idx] != NULL (:324) / vbMatched2[idx] (:872)
  uint8_t* taken = f->claimed_buf.data();
  int nmatches = 0;
  const QResult* R = f->results.h;
  auto is_taken = [&](int idx) { return taken[idx] != 0; };
  if (n_left >= 0) {                                       // S/ORBmatcher.cc:342-430: the best two per camera; the right camera's best under the left's gate
    for (int j = 0; j < nq; j++) {
      Pick pl, pr;
      if (R[j].n_top == 0) continue;                       // no left candidate: bestDist1 = 256 > TH_LOW, neither block runs
      if ((rc = pick_unclaimed(f, R[j], 2, is_taken, &pl))) return rc;
      if (pl.idx1 < 0 || pl.dist1 > TH_LOW) continue;
      if (R[nq + j].n_top && (rc = pick_unclaimed(f, R[nq + j], 1, is_taken, &pr))) return rc;
      const int q = jobs[j].kf_idx;
      if (static_cast<float>(pl.dist1) < nnratio * static_cast<float>(pl.dist2)) {
        taken[pl.idx1] = 1; matches[pl.idx1] = q;
        if (check_orientation) rotHist.add(rot_bin(kf_angle[q], f->hk_angle[pl.idx1]), pl.idx1);
        nmatches++;
      }
      if (pr.idx1 >= 0 && pr.dist1 <= TH_LOW) {            // (`|| true`, :401: no ratio test)
        taken[pr.idx1] = 1; matches[pr.idx1] = q;
        if (check_orientation) rotHist.add(rot_bin(kf_angle[q], f->hk_angle[pr.idx1]), pr.idx1);
        nmatches++;
      }
    }
    if (check_orientation) rotHist.reject_outside_three_maxima([&](int idx) { matches[idx] = -1; nmatches--; });
    if (nmatches_out) *nmatches_out = nmatches;
    return ORBG_OK;
  }
  for (int j = 0; j < nj; j++) {
    __builtin_prefetch(&R[j + 16]);
    const QResult& r = R[j];
    if (r.n_top == 0) continue;
    Pick pk;
    if ((rc = pick_unclaimed(f, r, 2, is_taken, &pk))) return rc;
    if (pk.idx1 < 0) continue;
    const int bestDist1 = pk.dist1, bestIdxF = pk.idx1, bestDist2 = pk.dist2;
    const bool low = by_query ? (bestDist1 < TH_LOW) : (bestDist1 <= TH_LOW);      // :898 vs :373
    if (low) {
      if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
        const int q = jobs[j].kf_idx;
        taken[bestIdxF] = 1;
        if (by_query) matches[q] = bestIdxF; else matches[bestIdxF] = q;
        if (check_orientation) rotHist.add(rot_bin(kf_angle[q], f->hk_angle[bestIdxF]), by_query ? q : bestIdxF);
        nmatches++;
      }
    }
  }
  if (check_orientation) rotHist.reject_outside_three_maxima([&](int idx) { matches[idx] = -1; nmatches--; });
  if (nmatches_out) *nmatches_out = nmatches;
  return ORBG_OK;
}

extern "C" int orbm_search_by_bow(orbm_frame* f, const orbm_featvec_view* fvF, const uint8_t* kf_desc, int nkf,
                                  const uint8_t* kf_mp_valid, const float* kf_angle, const orbm_featvec_view* fvK,
                                  float nnratio, int check_orientation, int32_t* matches, int* nmatches_out) {
  if (!f || !fvF || !fvK || !kf_desc || !kf_mp_valid || !matches || nkf < 0 || (check_orientation && !kf_angle)) return ORBG_BAD_ARG;
  return bow_common(f, fvF, nullptr, kf_desc, nkf, kf_mp_valid, kf_angle, fvK, nnratio, check_orientation, false, matches,
                    nmatches_out);
}

extern "C" int orbm_search_by_bow_rig(orbm_frame* f, int n_left, const orbm_featvec_view* fvF, const uint8_t* kf_desc, int nkf,
                                      const uint8_t* kf_mp_valid, const float* kf_angle, const orbm_featvec_view* fvK, float nnratio,
                                      int check_orientation, int32_t* matches, int* nmatches_out) {
  if (!f || !fvF || !fvK || !kf_desc || !kf_mp_valid || !matches || nkf < 0 || (check_orientation && !kf_angle) || n_left < 0 || n_left > f->fp.n)
    return ORBG_BAD_ARG;
  return bow_common(f, fvF, nullptr, kf_desc, nkf, kf_mp_valid, kf_angle, fvK, nnratio, check_orientation, false, matches, nmatches_out, n_left);
}


extern "C" int orbm_search_by_bow_kf(orbm_frame* kf2, const orbm_featvec_view* fv2, const uint8_t* mp_valid2,
                                     const uint8_t* desc1, int n1, const uint8_t* mp_valid1, const float* angle1,
                                     const orbm_featvec_view* fv1, float nnratio, int check_orientation,
                                     int32_t* matches12, int* nmatches_out) {
  if (!kf2 || !fv2 || !fv1 || !mp_valid2 || !desc1 || !mp_valid1 || !matches12 || n1 < 0 || (check_orientation && !angle1))
    return ORBG_BAD_ARG;
  return bow_common(kf2, fv2, mp_valid2, desc1, n1, mp_valid1, angle1, fv1, nnratio, check_orientation, true, matches12,
                    nmatches_out);
}

// SearchByProjection(KeyFrame*, Scw, ...): Sim3 decomposition (S/ORBmatcher.cc:484-488) with the cv::Mat float conventions
// spelled out in the oracle (scale by (float)(1/(double)scw)), then the candidate kernel and the serial commit (:556-583).
static int search_sim3_common(orbm_frame* f, orbm_map* mp, const float* Scw, const uint8_t* already_found, int th, float ratio_hamming, int camera_project,
                              int32_t* matched, int* nmatches_out, const orbg_camera* cam);
extern "C" int orbm_search_by_projection_sim3(orbm_frame* f, orbm_map* mp, const float* Scw, const uint8_t* already_found,
                                              int th, float ratio_hamming, int camera_project, int32_t* matched,
                                              int* nmatches_out) {
  return search_sim3_common(f, mp, Scw, already_found, th, ratio_hamming, camera_project, matched, nmatches_out, nullptr);
}
extern "C" int orbm_search_by_projection_sim3_cam(orbm_frame* f, orbm_map* mp, const float* Scw, const orbg_camera* cam, const uint8_t* already_found,
                                                  int th, float ratio_hamming, int32_t* matched, int* nmatches_out) {
  if (!cam || (cam->model != ORBG_CAM_PINHOLE && cam->model != ORBG_CAM_KANNALA_BRANDT8)) return ORBG_BAD_ARG;
  return search_sim3_common(f, mp, Scw, already_found, th, ratio_hamming, 1, matched, nmatches_out, cam);
}
static int search_sim3_common(orbm_frame* f, orbm_map* mp, const float* Scw, const uint8_t* already_found, int th, float ratio_hamming, int camera_project,
                              int32_t* matched, int* nmatches_out, const orbg_camera* cam) {
  if (!f || !mp || !Scw || !matched || f->device != mp->device) return ORBG_BAD_ARG;
  int rc = select_device(f->device);
  if (rc) return rc;
  const int m = mp->m;
  if (nmatches_out) *nmatches_out = 0;
  if (m == 0) return ORBG_OK;
  const int n = f->fp.n;
  if ((rc = stage_begin(f, (size_t)n * 4 + (size_t)m))) return rc;
  hipStream_t st = f->stream;
  stage_occupancy(f, matched, nullptr, n);
  const uint8_t* d_found = already_found ? stage_add(f, already_found, m) : nullptr;
  if ((rc = stage_commit(f))) return rc;
  float T16[16];
  {
    double d = 0;
    for (int k = 0; k < 3; k++) d += (double)Scw[k] * (double)Scw[k];
    const float scw = (float)std::sqrt(d);
    const float alpha = (float)(1.0 / (double)scw);
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) T16[4 * i + j] = Scw[4 * i + j] * alpha + 0.0f;
      T16[4 * i + 3] = Scw[4 * i + 3] * alpha + 0.0f;
    }
    T16[12] = T16[13] = T16[14] = 0.f; T16[15] = 1.f;
  }
  PoseF P;
  make_pose(T16, &P);
  FrameDev F = frame_dev(f);
  F.uright = nullptr;                                          // no stereo gate in the KeyFrame searches
  if ((rc = map_sync_to(mp, st))) return rc;
  rc = run_search(f, m, [&](int list_cap, int* cnt, int* cnt_next) {
    if (cam) hipLaunchKernelGGL(search_sim3_kernel<true>, dim3((m + 3) / 4), dim3(256), 0, st, f->fp, F, map_dev(mp), d_found, P,
                       camera_project, th, cnt, cnt_next, f->list.d, list_cap, f->results.d, rig_cam_of(*cam));
    else hipLaunchKernelGGL(search_sim3_kernel<false>, dim3((m + 3) / 4), dim3(256), 0, st, f->fp, F, map_dev(mp), d_found, P,
                       camera_project, th, cnt, cnt_next, f->list.d, list_cap, f->results.d, RigCamF{});
  });
  if (rc) return rc;
  std::vector<uint8_t> claimed(std::max(n, 1), 0);
  int nmatches = 0;
  const QResult* R = f->results.h;
  const float low = TH_LOW * ratio_hamming;
  auto is_claimed = [&](int idx) { return claimed[idx] != 0; };
  for (int i = 0; i < m; i++) {
    __builtin_prefetch(&R[i + 16]);                     // results sit in pinned memory the GPU has just written
    const QResult& r = R[i];
    if (r.n_top == 0) continue;
    Pick pk;
    if ((rc = pick_unclaimed(f, r, 1, is_claimed, &pk))) return rc;
    if (pk.idx1 < 0) continue;
    const int bestDist = pk.dist1, bestIdx = pk.idx1;
    if (bestDist <= low) {
      matched[bestIdx] = i;
      claimed[bestIdx] = 1;
      nmatches++;
    }
  }
  if (nmatches_out) *nmatches_out = nmatches;
  return ORBG_OK;
}

// SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist), S/ORBmatcher.cc:2188-2310:
// the candidate kernel above, then the serial commit (:2265-2281: a feature taken by an earlier point of this call is no candidate
// for a later one) and the rotation-consistency vote (:2284-2306).  kf_points: the keyframe's GetMapPointMatches() uploaded feature
// by feature (orbm_map_upload: bad[i] = no point or isBad()); kf_angle[i] = pKF->mvKeysUn[i].angle.
static int search_reloc_common(orbm_frame* f, orbm_map* mp, const float* Tcw, const uint8_t* already_found, const float* kf_angle, float th, int orb_dist,
                               int check_orientation, int32_t* assigned_mp, int* nmatches_out, const orbg_camera* cam);
extern "C" int orbm_search_by_projection_reloc(orbm_frame* f, orbm_map* mp, const float* Tcw, const uint8_t* already_found,
                                               const float* kf_angle, float th, int orb_dist, int check_orientation,
                                               int32_t* assigned_mp, int* nmatches_out) {
  return search_reloc_common(f, mp, Tcw, already_found, kf_angle, th, orb_dist, check_orientation, assigned_mp, nmatches_out, nullptr);
}
extern "C" int orbm_search_by_projection_reloc_cam(orbm_frame* f, orbm_map* mp, const float* Tcw, const orbg_camera* cam, const uint8_t* already_found,
                                                   const float* kf_angle, float th, int orb_dist, int check_orientation,
                                                   int32_t* assigned_mp, int* nmatches_out) {
  if (!cam || (cam->model != ORBG_CAM_PINHOLE && cam->model != ORBG_CAM_KANNALA_BRANDT8)) return ORBG_BAD_ARG;
  return search_reloc_common(f, mp, Tcw, already_found, kf_angle, th, orb_dist, check_orientation, assigned_mp, nmatches_out, cam);
}
static int search_reloc_common(orbm_frame* f, orbm_map* mp, const float* Tcw, const uint8_t* already_found, const float* kf_angle, float th, int orb_dist,
                               int check_orientation, int32_t* assigned_mp, int* nmatches_out, const orbg_camera* cam) {
  if (!f || !mp || !Tcw || !assigned_mp || f->device != mp->device || (check_orientation && !kf_angle)) return ORBG_BAD_ARG;
  int rc = select_device(f->device);
  if (rc) return rc;
  const int m = mp->m;
  if (nmatches_out) *nmatches_out = 0;
  if (m == 0) return ORBG_OK;
  const int n = f->fp.n;
  if ((rc = stage_begin(f, (size_t)n * 4 + (size_t)m))) return rc;
  hipStream_t st = f->stream;
  stage_occupancy(f, assigned_mp, nullptr, n);                   // any map point blocks a feature (:2246-2247)
  const uint8_t* d_found = already_found ? stage_add(f, already_found, m) : nullptr;
  if ((rc = stage_commit(f))) return rc;
  PoseF P;
  make_pose(Tcw, &P);
  FrameDev F = frame_dev(f);
  F.uright = nullptr;                                          // no stereo gate in this overload
  if ((rc = map_sync_to(mp, st))) return rc;
  rc = run_search(f, m, [&](int list_cap, int* cnt, int* cnt_next) {
    if (cam) hipLaunchKernelGGL(search_reloc_kernel<true>, dim3((m + 3) / 4), dim3(256), 0, st, f->fp, F, map_dev(mp), d_found, P, th, cnt, cnt_next,
                                f->list.d, list_cap, f->results.d, rig_cam_of(*cam));
    else hipLaunchKernelGGL(search_reloc_kernel<false>, dim3((m + 3) / 4), dim3(256), 0, st, f->fp, F, map_dev(mp), d_found, P, th, cnt, cnt_next,
                            f->list.d, list_cap, f->results.d, RigCamF{});
  });
  if (rc) return rc;
  cache_keypoint_fields(f);
  f->claimed_buf.assign((size_t)std::max(n, 1), 0);
  uint8_t* claimed = f->claimed_buf.data();
  RotHist rotHist(f->rot_entries);
  int nmatches = 0;
  const QResult* R = f->results.h;
  auto is_claimed = [&](int idx) { return claimed[idx] != 0; };
  for (int i = 0; i < m; i++) {
    __builtin_prefetch(&R[i + 16]);
    const QResult& r = R[i];
    if (r.n_top == 0) continue;
    Pick pk;
    if ((rc = pick_unclaimed(f, r, 1, is_claimed, &pk))) return rc;
    if (pk.idx1 < 0) continue;
    if (pk.dist1 <= orb_dist) {
      assigned_mp[pk.idx1] = i;
      claimed[pk.idx1] = 1;
      nmatches++;
      if (check_orientation) rotHist.add(rot_bin(kf_angle[i], f->hk_angle[pk.idx1]), pk.idx1);
    }
  }
  if (check_orientation)
    rotHist.reject_outside_three_maxima([&](int idx) { assigned_mp[idx] = -1; nmatches--; });
  if (nmatches_out) *nmatches_out = nmatches;
  return ORBG_OK;
}
