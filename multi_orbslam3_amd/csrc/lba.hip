// liborbgpu -- Local Bundle Adjustment for gfx950 (MI355X).  Replaces the numerical core of
// Optimizer::LocalBundleAdjustment (S/Optimizer.cc:1917-2267) and the vendored g2o machinery it drives
// (Levenberg-Marquardt, BlockSolver<6,3> with Schur complement, Huber kernels, SE3 exp-map vertices) behind
// lba_solve() of include/orbgpu.h.
//
// Design (MI355X-first): everything is FP64 and stays in HBM/L2 for the whole solve; the host only runs the LM
// control flow (lambda schedule, accept/reject) on three scalars per trial that the kernels drop into mapped
// pinned memory.  All reductions are ORDER-FIXED (CSR gathers + tree sums, no floating-point atomics) so a solve
// is bit-reproducible run to run:
//   k_errors        thread/edge : residual, chi2, Huber rho        (+ fixed-order block partial sums)
//   k_linearize     thread/edge : analytic Jacobians, weighted J^T W J blocks (Hpl 6x3, pose 21+6, point 6+3)
//   k_reduce_points thread/point: Hll, bl   = ordered sum over the point's edges
//   k_lin_poses     block/pose  : Hpp, bp   rebuilt from the pose's edges on the fly, wave+block tree sum
//   k_schur         block/pose-pair: S_ij = [i==j](Hpp_i + lambda I) - sum_l Hpl_il (Hll_l+lambda I)^-1 Hpl_jl^T,
//                   gathered over the points both poses observe (structure built once per call); diagonal pairs
//                   also produce b_s,i = bp_i - sum_l Hpl_il Dinv_l bl_l
//   ldlt_mfma.hpp   one workgroup: dense LDL^T + solve of the reduced camera system on the FP64 matrix cores
//                   (v_mfma_f64_16x16x4_f64; 16x16 tiles in registers, dataflow between wavefronts) up to 50 free poses;
//   ldlt_xcd.hpp    the same on eight workgroups of one XCD (hand-overs through that XCD's L2) from 9 tile rows on (21+ free poses);
//                   k_wide_panel / k_wide_update / k_wide_back: the blocked many-workgroup LDL^T of larger windows
//   k_update        thread/vertex: x_l = Dinv_l (bl_l - sum_i Hpl_il^T x_i); trial state = exp(x_p) * T  /  X + x_l
//   k_finish        one workgroup: robust chi2, computeScale, max-diagonal -> pinned host record
// The reduced camera system is tiny (6P x 6P, P <= a few tens): the path is latency bound, not FLOP bound.
// Parity: poses/points within 1e-4 of the oracle after float32 write-back, identical outlier sets.

#include <time.h>

#include <condition_variable>
#include <atomic>
#include <unistd.h>
#include <mutex>
#include <thread>

#include "common.hpp"
#include "wave.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <type_traits>

using namespace orbg;

#include "ldlt_mfma.hpp"
#include "ldlt_xcd.hpp"
#include "se3.hpp"

using namespace orbg_se3;

namespace {

// -DLBA_PROFILE (tools/micro/build_variants.sh -> liborbgpu_lbaprof.so, tools/micro/lba_prof.py): wall-clock stamps (100 MHz, common
// to all compute units) of the three per-LM-step kernels next to the LDL^T -- every workgroup's first and last instruction, and the
// phases of one workgroup of each role -- for the floor models of DESIGN.md section 8.  Slots: [kernel 0..2][0..7] phases of the
// sampled workgroup, [8 + 2 b], [9 + 2 b] start / end of workgroup b (b < 1020).
#ifdef LBA_PROFILE
__device__ long long g_lba_prof[3][8 + 2 * 1020];
#define LBA_PROF_BEGIN(kern) const int lbp_k = (kern); const long long lbp_t0 = (long long)wall_clock64(); \
  if (threadIdx.x == 0 && blockIdx.x < 1020) g_lba_prof[lbp_k][8 + 2 * blockIdx.x] = lbp_t0
#define LBA_PROF_END() do { if (threadIdx.x == 0 && blockIdx.x < 1020) g_lba_prof[lbp_k][9 + 2 * blockIdx.x] = (long long)wall_clock64(); } while (0)
#define LBA_PROF_PHASE(block, slot) do { if (threadIdx.x == 0 && (int)blockIdx.x == (block)) g_lba_prof[lbp_k][slot] = (long long)wall_clock64() - lbp_t0; } while (0)
#else
#define LBA_PROF_BEGIN(kern) do { } while (0)
#define LBA_PROF_END() do { } while (0)
#define LBA_PROF_PHASE(block, slot) do { } while (0)
#endif

__device__ inline void edge_error(const PoseQ& T, const double* X, const Cam& c, const lba_edge& e, double* err, double* Xc) {
  double r[3];
  quat_rotate(T.q, X, r);
  Xc[0] = r[0] + T.t[0]; Xc[1] = r[1] + T.t[1]; Xc[2] = r[2] + T.t[2];
  if (e.ur < 0) {
    const double iz = 1.0 / Xc[2];
    err[0] = (double)e.u - (c.fx * Xc[0] * iz + c.cx);
    err[1] = (double)e.v - (c.fy * Xc[1] * iz + c.cy);
    err[2] = 0;
  } else {
    const float invz = (float)(1.0 / Xc[2]);                 // cam_project: float invz (types_six_dof_expmap.cpp:191)
    const double r0 = Xc[0] * invz * c.fx + c.cx;
    const double r1 = Xc[1] * invz * c.fy + c.cy;
    const double r2 = r0 - (double)(c.bf_f * invz);
    err[0] = (double)e.u - r0; err[1] = (double)e.v - r1; err[2] = (double)e.ur - r2;
  }
}

// The same for a problem that carries an orbg_camera_rig: stereo edges as above; monocular edges through mpCamera->project
// (EdgeSE3ProjectXYZ::computeError, I/OptimizableTypes.h:99-104); the right camera's through mpCamera2 after (mTrl * T)
// (EdgeSE3ProjectXYZToBody::computeError, :127-132).  Xc: the point in the frame of the camera that made the observation.
__device__ inline void edge_error(const PoseQ& T, const double* X, const CamRig& g, const lba_edge& e, double* err, double* Xc) {
  if (e.ur >= 0) { edge_error(T, X, g.c, e, err, Xc); return; }
  double r[3], uv[2];
  if (g.has_right && ur_is_right(e.ur)) {
    PoseQ Trw;
    se3_mul(g.Trl, T, &Trw);
    quat_rotate(Trw.q, X, r);
    Xc[0] = r[0] + Trw.t[0]; Xc[1] = r[1] + Trw.t[1]; Xc[2] = r[2] + Trw.t[2];
    cam_project(g.right, Xc, uv);
  } else {
    quat_rotate(T.q, X, r);
    Xc[0] = r[0] + T.t[0]; Xc[1] = r[1] + T.t[1]; Xc[2] = r[2] + T.t[2];
    cam_project(g.left, Xc, uv);
  }
  err[0] = (double)e.u - uv[0]; err[1] = (double)e.v - uv[1]; err[2] = 0;
}

// isDepthPositive() of an edge: z of the point in the observing camera's frame (G/types/types_six_dof_expmap.h:163-167,
// I/OptimizableTypes.h:106-110,134-138)
__device__ inline bool edge_depth_positive(const PoseQ& T, const double* X, const Cam&, const lba_edge&) {
  double rr[3];
  quat_rotate(T.q, X, rr);
  return rr[2] + T.t[2] > 0.0;
}
__device__ inline bool edge_depth_positive(const PoseQ& T, const double* X, const CamRig& g, const lba_edge& e) {
  double rr[3];
  if (g.has_right && ur_is_right(e.ur)) {
    PoseQ Trw;
    se3_mul(g.Trl, T, &Trw);
    quat_rotate(Trw.q, X, rr);
    return rr[2] + Trw.t[2] > 0.0;
  }
  quat_rotate(T.q, X, rr);
  return rr[2] + T.t[2] > 0.0;
}

__device__ inline void huber(double e, double delta, double dsqr, double* rho0, double* rho1) {
  if (e <= dsqr) { *rho0 = e; *rho1 = 1.; }
  else { const double s = sqrt(e); *rho0 = 2 * s * delta - dsqr; *rho1 = delta / s; }
}

struct Huber { double delta_mono, dsqr_mono, delta_stereo, dsqr_stereo; };

// ---------------------------------------------------------------------------------------------- kernels

// residuals + chi2 + robust rho (computeActiveErrors + activeRobustChi2); block partial sums in fixed order
// what the host reads per LM trial (mapped pinned memory).  Round 6: chi2, scale and ok of a trial travel as self-validating pairs
// (value, value ^ tag(seq)) written with system-scope stores -- the host takes the record when seq is the one it waits for AND the
// three pairs check out under that tag.  Before, a system-scope release fence stood between the values and seq: on gfx950 that is
// a write-back of the XCD's whole L2, microseconds while the launch's edge workgroups have megabytes of Jacobian blocks dirty in
// it (the publisher workgroup ended 2.5 us after everybody else; tools/micro/lba_prof.py).  No ordering is needed now.
struct HostRec { double chi2, scale, maxdiag, chi2_init; int ok; unsigned seq; unsigned long long c_chi2, c_scale, c_ok; };
__host__ __device__ inline unsigned long long rec_tag(unsigned seq) { return (0x9E3779B97F4A7C15ull * ((unsigned long long)seq + 1ull)) | 1ull; }

// Results of a solve, written by the GPU straight into the caller-visible pinned block (no copy commands): per edge a flag
// byte (bit 0 = isDepthPositive() with the final estimate, bit 1 = outlier: chi2 > 5.991 / 7.815 or depth <= 0,
// S/Optimizer.cc:2131-2166) and optionally its chi2; the final poses and points.
template <class CamT>
__global__ __launch_bounds__(256) void k_export(int n_edges, int n_poses, int n_points, const lba_edge* __restrict__ edges,
                                               const PoseQ* __restrict__ poses, const double* __restrict__ points, CamT cam,
                                               const double* __restrict__ chi2, uint8_t* __restrict__ out_flags,
                                               double* __restrict__ out_chi2, PoseQ* __restrict__ out_poses,
                                               double* __restrict__ out_points) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < n_edges) {
    const lba_edge e = edges[k];
    const bool depth_pos = edge_depth_positive(poses[e.pose], points + 3 * (size_t)e.point, cam, e);
    const double thr = e.ur < 0 ? 5.991 : 7.815;
    const double c = chi2[k];
    const bool outlier = c > thr || !depth_pos;
    out_flags[k] = (uint8_t)((depth_pos ? 1 : 0) | (outlier ? 2 : 0));
    if (out_chi2) out_chi2[k] = c;
  }
  if (k < n_poses) out_poses[k] = poses[k];
  if (k < 3 * n_points) out_points[k] = points[k];
}

// The workgroup (of n_edge_blocks that call this) that finishes last adds up the partial sums in index order (deterministic
// whoever is last) and publishes robust chi2 / scale / solver flag to the host record: what used to be a separate one-block
// kernel per LM trial.  Every calling workgroup has written partial[its index] before.
// What the LM step needs to know on the device to prepare the NEXT solve before the host has spoken: the chi2 and lambda the
// trial started from (by value, or from device memory at the start of a round) and where to leave lambda for the accepted case.
struct LmIn { double cur_chi, lambda; const double* chi_p; const double* lambda_p; double* lambda_next; };

// the tail: all n_edge_blocks partials are in memory and visible to this workgroup
__device__ __forceinline__ void publish_trial_tail(int n_edge_blocks, const double* __restrict__ partial, unsigned* __restrict__ ticket,
                                                   const double* __restrict__ scale_partial, int n_scale_partial,
                                                   const int* __restrict__ ok_flag, HostRec* __restrict__ rec, unsigned seq, const LmIn lm,
                                                   int ok_override = 1) {
  __shared__ double parts[1024];
  const int np = n_edge_blocks;
  // all partials are fetched in parallel, then summed by one thread in index order
  const int tot = min(np + n_scale_partial, 1024);
  for (int i = threadIdx.x; i < tot; i += 256)
    parts[i] = __hip_atomic_load(i < np ? &partial[i] : &scale_partial[i - np], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (threadIdx.x == 0) {
    // index order, sixteen LDS reads in flight per step: written as one read per addition the loop is a chain of dependent LDS round
    // trips (~130 cycles each: 76 partials 4 us, the 299 of a C4 window 14 us -- the publisher ended that long after every other
    // workgroup of the launch, tools/micro/lba_prof.py)
    auto ordered_sum = [&](int first, int count) {
      double acc = 0;
      int i = 0;
      for (; i + 16 <= count; i += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = parts[first + i + u];
#pragma unroll
        for (int u = 0; u < 16; u++) acc += v[u];
      }
      for (; i < count; i++) acc += parts[first + i];
      return acc;
    };
    const int np_l = min(np, 1024), ns_l = max(min(np + n_scale_partial, 1024) - np_l, 0);
    double chi = ordered_sum(0, np_l), scale = ordered_sum(np_l, ns_l);
    for (int i = np_l; i < np; i++) chi += __hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = ns_l; i < n_scale_partial; i++) scale += __hip_atomic_load(&scale_partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int okv = ok_override != 1 ? ok_override : ok_flag ? *ok_flag : 1;
    {                                                       // maxdiag / chi2_init stay as k_finish left them
      const unsigned long long tg = rec_tag(seq), bc = (unsigned long long)__double_as_longlong(chi), bs = (unsigned long long)__double_as_longlong(scale);
      unsigned long long* r64 = reinterpret_cast<unsigned long long*>(rec);
      __hip_atomic_store(r64 + 0, bc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(r64 + 1, bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&rec->ok, okv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&rec->c_chi2, bc ^ tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&rec->c_scale, bs ^ tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&rec->c_ok, (unsigned long long)(unsigned)okv ^ tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (lm.lambda_next) {
      // lambda of the NEXT iteration if this trial is accepted -- the host's arithmetic (levenberg.cpp:116-141), operation for
      // operation, so that a solve launched speculatively with it is the solve the host would have launched
      const double cur = lm.chi_p ? *lm.chi_p : lm.cur_chi;
      const double lam = lm.lambda_p ? *lm.lambda_p : lm.lambda;
      const double tempChi = okv ? chi : 1.7976931348623157e308;
      double rho = cur - tempChi;
      rho /= scale + 1e-3;
      const double c3 = 2 * rho - 1;
      double alpha = 1. - c3 * c3 * c3;
      alpha = fmin(alpha, 2. / 3.);
      *lm.lambda_next = lam * fmax(1. / 3., alpha);
    }
    *ticket = 0;
    __hip_atomic_store(&rec->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // the host polls this word instead of hipStreamSynchronize
  }
}
// a workgroup has written partial[its index]: make it visible and count it
__device__ __forceinline__ unsigned trial_arrive(unsigned* __restrict__ ticket) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void publish_trial_record(int n_edge_blocks, const double* __restrict__ partial, unsigned* __restrict__ ticket,
                                                     const double* __restrict__ scale_partial, int n_scale_partial,
                                                     const int* __restrict__ ok_flag, HostRec* __restrict__ rec, unsigned seq,
                                                     const LmIn lm = LmIn{0, 0, nullptr, nullptr, nullptr}) {
  __shared__ int s_last;
  if (threadIdx.x == 0) s_last = (trial_arrive(ticket) == (unsigned)n_edge_blocks - 1);
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  publish_trial_tail(n_edge_blocks, partial, ticket, scale_partial, n_scale_partial, ok_flag, rec, seq, lm);
}
// Round 6 -- the fused linearisation kernels (k_errlin, k_errlin_prep) keep ONE workgroup for the record: their edge workgroups only
// arrive (trial_arrive) and go on with their Jacobians; the publisher waits until all have arrived and publishes.  Before, the edge
// workgroup that arrived last published -- the acquire, 76 dependent additions, the system-scope release -- and only then started
// its Jacobians: it ended 5 us after every other workgroup of the launch (tools/micro/lba_prof.py: 11.0 against 5.6 us), and the
// kernels queued behind the launch waited for it.  The wait is bounded (2 s of wall clock: a launch takes 10 us); if it ever
// expires the record says so (ok = kOkPublishTimedOut) and the solve returns ORBG_INTERNAL instead of hanging.
constexpr int kOkPublishTimedOut = -3;
__device__ __forceinline__ void publish_trial_when_all_arrived(int n_edge_blocks, const double* __restrict__ partial, unsigned* __restrict__ ticket,
                                                               const double* __restrict__ scale_partial, int n_scale_partial,
                                                               const int* __restrict__ ok_flag, HostRec* __restrict__ rec, unsigned seq, const LmIn lm) {
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    const long long t0 = (long long)wall_clock64();
    int ok = 1;
    for (unsigned polls = 0;; polls++) {
      if (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)n_edge_blocks) break;
      __builtin_amdgcn_s_sleep(1);
      if ((polls & 1023u) == 1023u && (long long)wall_clock64() - t0 > 200000000ll) { ok = kOkPublishTimedOut; break; }
    }
    s_ok = ok;                              // (no acquire: the tail reads the partials with agent-scope loads, past this XCD's L2)
  }
  __syncthreads();
  publish_trial_tail(n_edge_blocks, partial, ticket, scale_partial, n_scale_partial, ok_flag, rec, seq, lm, s_ok);
}

// (returns this thread's chi2 -- the caller may go on with it: k_errors_export)
template <class CamT>
__device__ __forceinline__ double errors_block(int bid, int n_edge_blocks, int n_edges, const lba_edge* __restrict__ edges,
                                               const PoseQ* __restrict__ poses, const double* __restrict__ points, CamT cam, Huber hb,
                                               double* __restrict__ err, double* __restrict__ chi2, double* __restrict__ partial,
                                               int final_mode, unsigned* __restrict__ ticket, const double* __restrict__ scale_partial,
                                               int n_scale_partial, const int* __restrict__ ok_flag, HostRec* __restrict__ rec,
                                               unsigned seq) {
  __shared__ double red[256];
  const int k = bid * 256 + threadIdx.x;
  double c_out = 0;
  double rho0 = 0;
  if (k < n_edges) {
    const lba_edge e = edges[k];
    double er[3], Xc[3];
    edge_error(poses[e.pose], points + 3 * (size_t)e.point, cam, e, er, Xc);
    const double om = (double)e.inv_sigma2;
    const int D = e.ur < 0 ? 2 : 3;
    double c = 0;
    for (int i = 0; i < D; i++) c += er[i] * (om * er[i]);
    err[3 * (size_t)k] = er[0]; err[3 * (size_t)k + 1] = er[1]; err[3 * (size_t)k + 2] = er[2];
    chi2[k] = c;
    c_out = c;
    double rho1;
    const bool mono = D == 2;
    huber(c, mono ? hb.delta_mono : hb.delta_stereo, mono ? hb.dsqr_mono : hb.dsqr_stereo, &rho0, &rho1);
  }
  red[threadIdx.x] = rho0;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[bid] = red[0];
  if (final_mode) publish_trial_record(n_edge_blocks, partial, ticket, scale_partial, n_scale_partial, ok_flag, rec, seq);
  return c_out;
}
template <class CamT>
__global__ __launch_bounds__(256) void k_errors(int n_edges, const lba_edge* __restrict__ edges, const PoseQ* __restrict__ poses,
                                               const double* __restrict__ points, CamT cam, Huber hb, double* __restrict__ err,
                                               double* __restrict__ chi2, double* __restrict__ partial,
                                               int final_mode, unsigned* __restrict__ ticket, const double* __restrict__ scale_partial,
                                               int n_scale_partial, const int* __restrict__ ok_flag, HostRec* __restrict__ rec,
                                               unsigned seq) {
  errors_block((int)blockIdx.x, (int)gridDim.x, n_edges, edges, poses, points, cam, hb, err, chi2, partial, final_mode, ticket, scale_partial,
               n_scale_partial, ok_flag, rec, seq);
}
// The last evaluation of a solve and the export of the state it evaluated (speculative: dropped if the trial is rejected) in
// one launch: the export needs nothing of the other workgroups (an edge's flags follow from its own chi2).
template <class CamT>
__global__ __launch_bounds__(256) void k_errors_export(int n_edge_blocks, int n_edges, const lba_edge* __restrict__ edges,
                                                      const PoseQ* __restrict__ poses, const double* __restrict__ points, CamT cam, Huber hb,
                                                      double* __restrict__ err, double* __restrict__ chi2, double* __restrict__ partial,
                                                      unsigned* __restrict__ ticket, const double* __restrict__ scale_partial,
                                                      int n_scale_partial, const int* __restrict__ ok_flag, HostRec* __restrict__ rec,
                                                      unsigned seq, int n_poses, int n_points, uint8_t* __restrict__ out_flags,
                                                      double* __restrict__ out_chi2, PoseQ* __restrict__ out_poses,
                                                      double* __restrict__ out_points) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  // export first (the record's publication ends with returns inside errors_block)
  if (k < n_poses) out_poses[k] = poses[k];
  if (k < 3 * n_points) out_points[k] = points[k];
  if ((int)blockIdx.x >= n_edge_blocks) return;
  bool depth_pos = false; double thr = 0;
  if (k < n_edges) {
    const lba_edge e = edges[k];
    depth_pos = edge_depth_positive(poses[e.pose], points + 3 * (size_t)e.point, cam, e);
    thr = e.ur < 0 ? 5.991 : 7.815;
  }
  // the flags need this edge's chi2: computed below, so the residual part runs first for the values and the flags are
  // written from its return value
  const double c = errors_block((int)blockIdx.x, n_edge_blocks, n_edges, edges, poses, points, cam, hb, err, chi2, partial, 1, ticket,
                                scale_partial, n_scale_partial, ok_flag, rec, seq);
  if (k < n_edges) {
    const bool outlier = c > thr || !depth_pos;
    out_flags[k] = (uint8_t)((depth_pos ? 1 : 0) | (outlier ? 2 : 0));
    if (out_chi2) out_chi2[k] = c;
  }
}

// per-edge blocks: EB[k*27 + ...] = Hpl (6x3, 18) | pointH upper (6) | pointB (3)
constexpr int kEB = 27;

// Jacobians of one edge (stereo: G/types/types_six_dof_expmap.cpp:228-274; mono: S/OptimizableTypes.cpp:139-160)
__device__ inline void edge_jacobians(const PoseQ& T, double x, double y, double z, const Cam& c, float ur, double* A, double* B) {
  const bool mono = ur < 0;
  double R[9];
  quat_to_R(T.q, R);
  const double iz = 1.0 / z, iz2 = iz * iz;     // one division per edge; the reference divides term by term (<= 2 ulp apart)
  if (mono) {
    const double J[6] = {-(c.fx * iz), -0.0, c.fx * x * iz2, -0.0, -(c.fy * iz), c.fy * y * iz2};
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) A[3 * i + j] = J[3 * i] * R[j] + J[3 * i + 1] * R[3 + j] + J[3 * i + 2] * R[6 + j];
    const double S[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 6; j++) B[6 * i + j] = J[3 * i] * S[j] + J[3 * i + 1] * S[6 + j] + J[3 * i + 2] * S[12 + j];
#pragma unroll
    for (int j = 0; j < 3; j++) A[6 + j] = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) B[12 + j] = 0;
  } else {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      A[j] = -c.fx * R[j] * iz + c.fx * x * R[6 + j] * iz2;
      A[3 + j] = -c.fy * R[3 + j] * iz + c.fy * y * R[6 + j] * iz2;
      A[6 + j] = A[j] - c.bf * R[6 + j] * iz2;
    }
    B[0] = x * y * iz2 * c.fx; B[1] = -(1 + (x * x * iz2)) * c.fx; B[2] = y * iz * c.fx; B[3] = -iz * c.fx; B[4] = 0; B[5] = x * iz2 * c.fx;
    B[6] = (1 + y * y * iz2) * c.fy; B[7] = -x * y * iz2 * c.fy; B[8] = -x * iz * c.fy; B[9] = 0; B[10] = -iz * c.fy; B[11] = y * iz2 * c.fy;
    B[12] = B[0] - c.bf * y * iz2; B[13] = B[1] + c.bf * x * iz2; B[14] = B[2]; B[15] = B[3]; B[16] = 0; B[17] = B[5] - c.bf * iz2;
  }
}

// With a camera rig: (x, y, z) = T.map(X) is the point in the LEFT camera's (body) frame for every kind of edge.
//   monocular  S/OptimizableTypes.cpp:139-160   Xi = -projectJac(X_l) * R(T),                  Xj = -projectJac(X_l) * SE3deriv(X_l)
//   right cam  S/OptimizableTypes.cpp:192-214   Xi = -projectJac(X_r) * R(mTrl * T),  Xj = -projectJac(X_r) * R(mTrl) * SE3deriv(X_l),
//              X_r = mTrl.map(X_l)
__device__ inline void edge_jacobians(const PoseQ& T, double x, double y, double z, const CamRig& g, float ur, double* A, double* B) {
  if (ur >= 0) { edge_jacobians(T, x, y, z, g.c, ur, A, B); return; }
  const double Xl[3] = {x, y, z};
  double R[9], J[6], M[6];
  if (g.has_right && ur_is_right(ur)) {
    double rr[3], Xr[3], Rrl[9];
    quat_rotate(g.Trl.q, Xl, rr);
    Xr[0] = rr[0] + g.Trl.t[0]; Xr[1] = rr[1] + g.Trl.t[1]; Xr[2] = rr[2] + g.Trl.t[2];
    cam_project_jac(g.right, Xr, J);
#pragma unroll
    for (int i = 0; i < 6; i++) J[i] = -J[i];
    PoseQ Trw;
    se3_mul(g.Trl, T, &Trw);
    quat_to_R(Trw.q, R);
    quat_to_R(g.Trl.q, Rrl);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) M[3 * i + j] = J[3 * i] * Rrl[j] + J[3 * i + 1] * Rrl[3 + j] + J[3 * i + 2] * Rrl[6 + j];
  } else {
    cam_project_jac(g.left, Xl, J);
#pragma unroll
    for (int i = 0; i < 6; i++) { J[i] = -J[i]; M[i] = J[i]; }
    quat_to_R(T.q, R);
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) A[3 * i + j] = J[3 * i] * R[j] + J[3 * i + 1] * R[3 + j] + J[3 * i + 2] * R[6 + j];
  const double S[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 6; j++) B[6 * i + j] = M[3 * i] * S[j] + M[3 * i + 1] * S[6 + j] + M[3 * i + 2] * S[12 + j];
#pragma unroll
  for (int j = 0; j < 3; j++) A[6 + j] = 0;
#pragma unroll
  for (int j = 0; j < 6; j++) B[12 + j] = 0;
}

// thread per edge; the 256 x 27 block of results is staged in LDS and written as ONE contiguous, coalesced chunk
// FUSED: the residuals are computed here (and stored, with their chi2 and the workgroup's robust partial sum) instead of being
// read back from a preceding k_errors launch -- same functions, same inputs, same bits.  The partial sum is written (and counted:
// trial_arrive) BEFORE the Jacobians; the launch's publisher workgroup publishes the trial record as soon as every edge workgroup
// has arrived: the host gets its verdict ~5 us before the kernel ends and its decision latency hides behind the Jacobians.
struct TrialPublish {
  double* partial; unsigned* ticket; const double* scale_partial; int n_scale_partial; const int* ok_flag; HostRec* rec; unsigned seq;
  int n_edge_blocks;
  LmIn lm;
};

// Hpl = B^T (w Omega) A (6 x 3) of edge k2 alone: what linearize_block writes for its own edge, for the edges a primary adds up
// (camera rigs: a keyframe may observe a landmark with both cameras)
template <bool FUSED, class CamT>
__device__ inline void edge_hpl(const PoseQ& T, const double* X, const CamT& c, const Huber& hb, const lba_edge& e, int k2,
                                const double* __restrict__ err, const double* __restrict__ chi2, double* H) {
  const bool mono = e.ur < 0;
  const int D = mono ? 2 : 3;
  const double om = (double)e.inv_sigma2;
  double er[3], chi_k = 0;
  if constexpr (FUSED) {
    double Xc[3];
    edge_error(T, X, c, e, er, Xc);
    for (int i = 0; i < D; i++) chi_k += er[i] * (om * er[i]);
  } else {
    er[0] = err[3 * (size_t)k2]; er[1] = err[3 * (size_t)k2 + 1]; er[2] = err[3 * (size_t)k2 + 2];
    chi_k = chi2[k2];
  }
  double rho0, rho1;
  huber(chi_k, mono ? hb.delta_mono : hb.delta_stereo, mono ? hb.dsqr_mono : hb.dsqr_stereo, &rho0, &rho1);
  double r[3], A[9], B[18];
  quat_rotate(T.q, X, r);
  edge_jacobians(T, r[0] + T.t[0], r[1] + T.t[1], r[2] + T.t[2], c, e.ur, A, B);
  const double wom = rho1 * om;
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int cidx = 0; cidx < 3; cidx++) {
      double h = 0;
#pragma unroll
      for (int i = 0; i < 3; i++) h += B[6 * i + a] * wom * A[3 * i + cidx];
      H[3 * a + cidx] = h;
    }
}

template <bool FUSED, class CamT>
__device__ __forceinline__ void linearize_block(int bid, int n_edges, const lba_edge* __restrict__ edges, const PoseQ* __restrict__ poses,
                                                  const double* __restrict__ points, CamT c, Huber hb, double* __restrict__ err,
                                                  double* __restrict__ chi2, const int* __restrict__ pose_col,
                                                  const int* __restrict__ point_col, double* __restrict__ EB, const TrialPublish pub) {
  __shared__ double stage[256 * kEB];
  __shared__ double red_f[256];
  const int k = bid * 256 + threadIdx.x;
  const bool live = k < n_edges;
  lba_edge e;
  e.pose = 0; e.point = 0; e.u = 0; e.v = 0; e.ur = -1; e.inv_sigma2 = 0;
  if (live) e = edges[k];
  const PoseQ T = poses[e.pose];
  const double* X = points + 3 * (size_t)e.point;
  const bool mono = e.ur < 0;
  const int D = mono ? 2 : 3;
  double er[3] = {0, 0, 0}, chi_k = 0, rho0 = 0, rho1 = 0;
  if (live) {
    if constexpr (FUSED) {
      double Xc[3];
      edge_error(T, X, c, e, er, Xc);
      const double om0 = (double)e.inv_sigma2;
      for (int i = 0; i < D; i++) chi_k += er[i] * (om0 * er[i]);
      err[3 * (size_t)k] = er[0]; err[3 * (size_t)k + 1] = er[1]; err[3 * (size_t)k + 2] = er[2];
      chi2[k] = chi_k;
    } else {
      er[0] = err[3 * (size_t)k]; er[1] = err[3 * (size_t)k + 1]; er[2] = err[3 * (size_t)k + 2];
      chi_k = chi2[k];
    }
    huber(chi_k, mono ? hb.delta_mono : hb.delta_stereo, mono ? hb.dsqr_mono : hb.dsqr_stereo, &rho0, &rho1);
  }
  if constexpr (FUSED) {
    // the workgroup's robust chi2 partial (same tree as k_errors), then the record if this is the last workgroup
    red_f[threadIdx.x] = live ? rho0 : 0.0;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
      if ((int)threadIdx.x < s2) red_f[threadIdx.x] += red_f[threadIdx.x + s2];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      // (the launch's publisher workgroup does the rest.  The partial goes out as ONE agent-scope store -- written through to where
      // every XCD reads it -- and is counted once that store is acknowledged: no release fence, i.e. no write-back of this XCD's L2
      // with the launch's Jacobian blocks in it, in any of the 44 edge workgroups, and no acquire in the publisher, which reads the
      // partials with agent-scope loads)
      __hip_atomic_store(&pub.partial[bid], red_f[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(pub.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (live) {
    double* out = stage + threadIdx.x * kEB;
    double r[3];
    quat_rotate(T.q, X, r);
    double A[9], B[18];
    edge_jacobians(T, r[0] + T.t[0], r[1] + T.t[1], r[2] + T.t[2], c, e.ur, A, B);
    const double om = (double)e.inv_sigma2;
    const double wom = rho1 * om;
    // rows >= D of A/B/omega_r are exact zeros for monocular edges, so every loop runs a constant 3 rows and
    // unrolls completely (no run-time indexed local arrays => no scratch memory)
    double omega_r[3];
#pragma unroll
    for (int i = 0; i < 3; i++) omega_r[i] = i < D ? -(om * er[i]) * rho1 : 0.0;
    const bool pf = pose_col[e.pose] >= 0, lf = point_col[e.point] >= 0;
#pragma unroll
    for (int a = 0; a < 6; a++)          // Hpl = B^T (w Omega) A   (6x3)
#pragma unroll
      for (int cidx = 0; cidx < 3; cidx++) {
        double h = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) h += B[6 * i + a] * wom * A[3 * i + cidx];
        out[3 * a + cidx] = (pf && lf) ? h : 0.0;
      }
    int o = 18;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b2 = a; b2 < 3; b2++) {
        double h = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) h += A[3 * i + a] * wom * A[3 * i + b2];
        out[o++] = lf ? h : 0.0;
      }
#pragma unroll
    for (int a = 0; a < 3; a++) {
      double sacc = 0;
#pragma unroll
      for (int i = 0; i < 3; i++) sacc += A[3 * i + a] * omega_r[i];
      out[o++] = lf ? sacc : 0.0;
    }
  }
  if constexpr (std::is_same<CamT, CamRig>::value) {
    // A keyframe of a two-camera rig observes a landmark up to twice (left and right camera: two edges between the same two vertices,
    // adjacent in the reference's creation order, S/Optimizer.cc:2021-2120).  g2o adds both into the one Hpl block of the vertex
    // pair; here the FIRST edge of such a run carries the sum and the others carry zero, so that the Schur complement and the
    // back-substitution -- which go by the per-landmark lists of first edges (the host leaves the others out) -- see one block per
    // (keyframe, landmark) as they do everywhere else.  Hll / bl / Hpp / bp are sums over all edges and need nothing of this.
    if (live) {
      double* out = stage + threadIdx.x * kEB;
      if (k > 0 && edges[k - 1].pose == e.pose && edges[k - 1].point == e.point) {
#pragma unroll
        for (int i = 0; i < 18; i++) out[i] = 0.0;
      } else if (pose_col[e.pose] >= 0 && point_col[e.point] >= 0) {
        for (int k2 = k + 1; k2 < n_edges; k2++) {
          const lba_edge e2 = edges[k2];
          if (e2.pose != e.pose || e2.point != e.point) break;
          double H2[18];
          edge_hpl<FUSED>(T, X, c, hb, e2, k2, err, chi2, H2);
#pragma unroll
          for (int i = 0; i < 18; i++) out[i] += H2[i];
        }
      }
    }
  }
  __syncthreads();
  const int valid = min(256, n_edges - bid * 256);
  double* dst = EB + (size_t)bid * 256 * kEB;
  for (int i = threadIdx.x; i < valid * kEB; i += 256) dst[i] = stage[i];
}
template <bool FUSED, class CamT>
__device__ __forceinline__ void lin_poses_block(int bid, const int* __restrict__ ps_start, const int* __restrict__ ps_edges,
                                                  const lba_edge* __restrict__ edges, const PoseQ* __restrict__ poses,
                                                  const double* __restrict__ points, CamT c, Huber hb, const double* __restrict__ err,
                                                  const double* __restrict__ chi2, double* __restrict__ Hpp, double* __restrict__ bp) {
  __shared__ double wpart[4][27];
  const int p = bid;
  const int b = ps_start[p], e_end = ps_start[p + 1];
  double acc[27];
#pragma unroll
  for (int i = 0; i < 27; i++) acc[i] = 0;
  // Every edge of the list observes pose p.  A thread's edges (j, j + 256, ...) are fetched in chunks of
  // three -- indices, then edge records, then landmark positions, each level requested for the whole chunk before the first use
  // (three memory round trips per chunk instead of three per edge) -- and accumulated in the order of the plain loop: same bits.
  constexpr int kCh = 3;
  for (int j0 = b + threadIdx.x; j0 < e_end; j0 += 256 * kCh) {
    int kk[kCh];
#pragma unroll
    for (int u = 0; u < kCh; u++) kk[u] = ps_edges[min(j0 + 256 * u, e_end - 1)];
    lba_edge ev[kCh];
#pragma unroll
    for (int u = 0; u < kCh; u++) ev[u] = edges[kk[u]];
    double Xv[kCh][3];
#pragma unroll
    for (int u = 0; u < kCh; u++) {
      const double* Xp = points + 3 * (size_t)ev[u].point;
      Xv[u][0] = Xp[0]; Xv[u][1] = Xp[1]; Xv[u][2] = Xp[2];
    }
    const PoseQ T = poses[ev[0].pose];                     // (the same pose for every edge of the list: requested with the landmarks)
#pragma unroll
    for (int u = 0; u < kCh; u++) {
      if (j0 + 256 * u >= e_end) continue;
      const int k = kk[u];
      const lba_edge e = ev[u];
      const double* X = Xv[u];
      double r[3];
      quat_rotate(T.q, X, r);
      const bool mono = e.ur < 0;
      const int D = mono ? 2 : 3;
      double er[3], chi_k;
      if constexpr (FUSED) {             // the edge workgroups of this launch are computing the same residuals concurrently
        double Xc[3];
        edge_error(T, X, c, e, er, Xc);
        const double om0 = (double)e.inv_sigma2;
        chi_k = 0;
        for (int i = 0; i < D; i++) chi_k += er[i] * (om0 * er[i]);
      } else {
        er[0] = err[3 * (size_t)k]; er[1] = err[3 * (size_t)k + 1]; er[2] = err[3 * (size_t)k + 2];
        chi_k = chi2[k];
      }
      double A[9], B[18];
      edge_jacobians(T, r[0] + T.t[0], r[1] + T.t[1], r[2] + T.t[2], c, e.ur, A, B);
      double rho0, rho1;
      huber(chi_k, mono ? hb.delta_mono : hb.delta_stereo, mono ? hb.dsqr_mono : hb.dsqr_stereo, &rho0, &rho1);
      const double om = (double)e.inv_sigma2;
      const double wom = rho1 * om;
      double omega_r[3];
#pragma unroll
      for (int i = 0; i < 3; i++) omega_r[i] = i < D ? -(om * er[i]) * rho1 : 0.0;
      int o = 0;
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b2 = a; b2 < 6; b2++) {
          double h = 0;
#pragma unroll
          for (int i = 0; i < 3; i++) h += B[6 * i + a] * wom * B[6 * i + b2];
          acc[o++] += h;
        }
#pragma unroll
      for (int a = 0; a < 6; a++) {
        double sacc = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) sacc += B[6 * i + a] * omega_r[i];
        acc[o++] += sacc;
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 27; i++) {
    const double v = wave_sum_f64(acc[i]);            // DPP tree, fixed order
    if (lane == 0) wpart[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    const double v = ((wpart[0][threadIdx.x] + wpart[1][threadIdx.x]) + wpart[2][threadIdx.x]) + wpart[3][threadIdx.x];
    if (threadIdx.x < 21) Hpp[21 * (size_t)p + threadIdx.x] = v;
    else bp[6 * (size_t)p + threadIdx.x - 21] = v;
  }
}


// buildSystem in ONE launch: the first nP workgroups build Hpp / b_p of "their" pose from its edges (the longer job, so it
// starts first), the remaining ones linearise 256 edges each (Hpl and the point parts).  The two jobs are independent.
// Point workgroups of the fused launch: Hll and b_l of "their" landmark straight from its edges (residual, Huber weight and
// the 3x3 point Jacobian recomputed with the functions the edge workgroups use -- same bits as summing their per-edge
// blocks in k_reduce_points, in the same edge order), so that no separate reduction launch is needed.
// Round 6: FOUR lanes per landmark (a quad).  Lane q takes the landmark's edges q, q + 4, q + 8, ... of its list -- a landmark has
// 5.5 observations on average and up to a few dozen, and with one thread per landmark the slowest thread of a workgroup walked its
// list for 9-10 us (two or three dependent memory round trips per chunk of four edges) while the pose and edge workgroups of the
// same launch were done after 6 (tools/micro/lba_prof.py).  The four partial sums are added as ((s0 + s1) + s2) + s3 on lane 0 of
// the quad (DPP quad_perm); k_reduce_points, the same reduction over stored blocks, uses the same decomposition: same bits.
__device__ __forceinline__ double quad_get(double v, int src /* compile-time 1..3 */) {
  int lo, hi;
  if (src == 1) { lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x55, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x55, 0xF, 0xF, true); }
  else if (src == 2) { lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xAA, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xAA, 0xF, 0xF, true); }
  else { lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xFF, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xFF, 0xF, 0xF, true); }
  return __hiloint2double(hi, lo);
}
template <class CamT>
__device__ __forceinline__ void lin_points_block(int bid, int nL, const int* __restrict__ pt_start, const int* __restrict__ pt_edges,
                                                   const lba_edge* __restrict__ edges, const PoseQ* __restrict__ poses,
                                                   const double* __restrict__ points, CamT c, Huber hb, double* __restrict__ Hll,
                                                   double* __restrict__ bl) {
  const int t = bid * 256 + threadIdx.x;
  const int l = min(t >> 2, nL - 1), q = t & 3;            // (threads past the last landmark repeat it: all 64 lanes stay in the DPP moves)
  const bool mine = (t >> 2) < nL;
  double acc[9];
#pragma unroll
  for (int i = 0; i < 9; i++) acc[i] = 0;
  const int jb = pt_start[l], je = pt_start[l + 1];
  for (int j0 = jb + q; j0 < je; j0 += 8) {
    // this lane's next two edges (j0, j0 + 4): indices, then edges, then poses are requested for both before the first use
    int id[2];
#pragma unroll
    for (int u = 0; u < 2; u++) id[u] = pt_edges[min(j0 + 4 * u, je - 1)];
    lba_edge ev[2];
#pragma unroll
    for (int u = 0; u < 2; u++) ev[u] = edges[id[u]];
    PoseQ Tv[2];
#pragma unroll
    for (int u = 0; u < 2; u++) Tv[u] = poses[ev[u].pose];
    const double* X = points + 3 * (size_t)ev[0].point;       // every edge of the list observes this landmark
    const double Xl[3] = {X[0], X[1], X[2]};
#pragma unroll
    for (int u = 0; u < 2; u++)
      if (j0 + 4 * u < je) {
        const lba_edge e = ev[u];
        const PoseQ T = Tv[u];
        const bool mono = e.ur < 0;
        const int D = mono ? 2 : 3;
        double er[3], Xc[3];
        edge_error(T, Xl, c, e, er, Xc);
        const double om = (double)e.inv_sigma2;
        double chi_k = 0;
        for (int i = 0; i < D; i++) chi_k += er[i] * (om * er[i]);
        double rho0, rho1;
        huber(chi_k, mono ? hb.delta_mono : hb.delta_stereo, mono ? hb.dsqr_mono : hb.dsqr_stereo, &rho0, &rho1);
        double r[3];
        quat_rotate(T.q, Xl, r);
        double A[9], B[18];
        edge_jacobians(T, r[0] + T.t[0], r[1] + T.t[1], r[2] + T.t[2], c, e.ur, A, B);
        const double wom = rho1 * om;
        double omega_r[3];
#pragma unroll
        for (int i = 0; i < 3; i++) omega_r[i] = i < D ? -(om * er[i]) * rho1 : 0.0;
        int o = 0;
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
          for (int b2 = a; b2 < 3; b2++) {
            double h = 0;
#pragma unroll
            for (int i = 0; i < 3; i++) h += A[3 * i + a] * wom * A[3 * i + b2];
            acc[o++] += h;
          }
#pragma unroll
        for (int a = 0; a < 3; a++) {
          double sacc = 0;
#pragma unroll
          for (int i = 0; i < 3; i++) sacc += A[3 * i + a] * omega_r[i];
          acc[o++] += sacc;
        }
      }
  }
#pragma unroll
  for (int i = 0; i < 9; i++) acc[i] = ((acc[i] + quad_get(acc[i], 1)) + quad_get(acc[i], 2)) + quad_get(acc[i], 3);     // (meaningful on lane 0 of the quad)
  if (mine && q == 0) {
    for (int i = 0; i < 6; i++) Hll[6 * (size_t)l + i] = acc[i];
    for (int i = 0; i < 3; i++) bl[3 * (size_t)l + i] = acc[6 + i];
  }
}
// workgroups behind the pose and edge workgroups of a fused linearisation launch: the landmark quads, then ONE publisher
__host__ __device__ inline int errlin_point_blocks(int nL) { return (4 * nL + 255) / 256; }
__host__ __device__ inline int errlin_tail_blocks(int nL) { return errlin_point_blocks(nL) + 1; }

template <class CamT>
__global__ __launch_bounds__(256) void k_lin_all(int nP, int n_edges, const lba_edge* __restrict__ edges, const PoseQ* __restrict__ poses,
                                                const double* __restrict__ points, CamT c, Huber hb, const double* __restrict__ err,
                                                const double* __restrict__ chi2, const int* __restrict__ pose_col,
                                                const int* __restrict__ point_col, double* __restrict__ EB,
                                                const int* __restrict__ ps_start, const int* __restrict__ ps_edges,
                                                double* __restrict__ Hpp, double* __restrict__ bp) {
  if ((int)blockIdx.x < nP) lin_poses_block<false>(blockIdx.x, ps_start, ps_edges, edges, poses, points, c, hb, err, chi2, Hpp, bp);
  else linearize_block<false>(blockIdx.x - nP, n_edges, edges, poses, points, c, hb, const_cast<double*>(err), const_cast<double*>(chi2),
                              pose_col, point_col, EB, TrialPublish{nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0u, 0, LmIn{0, 0, nullptr, nullptr, nullptr}});
}

// k_errors (final mode) + k_lin_all in one launch, for the speculative path of the LM driver: residuals, chi2 and the robust
// partial sums of the TRIAL state, its linearisation into the other set of buffers, and -- by the edge workgroup that
// finishes last -- the record the host is waiting for.  One launch floor (~5 us) less per accepted trial.
template <class CamT>
__global__ __launch_bounds__(256) void k_errlin(int nP, int n_edges, const lba_edge* __restrict__ edges, const PoseQ* __restrict__ poses,
                                               const double* __restrict__ points, CamT c, Huber hb, double* __restrict__ err,
                                               double* __restrict__ chi2, const int* __restrict__ pose_col,
                                               const int* __restrict__ point_col, double* __restrict__ EB,
                                               const int* __restrict__ ps_start, const int* __restrict__ ps_edges,
                                               double* __restrict__ Hpp, double* __restrict__ bp, double* __restrict__ partial,
                                               unsigned* __restrict__ ticket, const double* __restrict__ scale_partial, int n_scale_partial,
                                               const int* __restrict__ ok_flag, HostRec* __restrict__ rec, unsigned seq, int n_edge_blocks,
                                               int nL, const int* __restrict__ pt_start, const int* __restrict__ pt_edges,
                                               double* __restrict__ Hll, double* __restrict__ bl, LmIn lm) {
  const int bid = (int)blockIdx.x;
  LBA_PROF_BEGIN(0);
  if (bid < nP) {
    lin_poses_block<true>(bid, ps_start, ps_edges, edges, poses, points, c, hb, err, chi2, Hpp, bp);
  } else if (bid < nP + n_edge_blocks) {
    linearize_block<true>(bid - nP, n_edges, edges, poses, points, c, hb, err, chi2, pose_col, point_col, EB,
                          TrialPublish{partial, ticket, scale_partial, n_scale_partial, ok_flag, rec, seq, n_edge_blocks, lm});
  } else if (bid - nP - n_edge_blocks < errlin_point_blocks(nL)) {
    lin_points_block(bid - nP - n_edge_blocks, nL, pt_start, pt_edges, edges, poses, points, c, hb, Hll, bl);
  } else {
    publish_trial_when_all_arrived(n_edge_blocks, partial, ticket, scale_partial, n_scale_partial, ok_flag, rec, seq, lm);
  }
  LBA_PROF_END();
}


// Hll (6 upper) + bl (3) per active point: ordered sum over the point's edges
__global__ __launch_bounds__(256) void k_reduce_points(int nL, const int* __restrict__ pt_start, const int* __restrict__ pt_edges,
                                                      const double* __restrict__ EB, double* __restrict__ Hll, double* __restrict__ bl) {
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= nL) return;
  double acc[9];
#pragma unroll
  for (int i = 0; i < 9; i++) acc[i] = 0;
  // the edge list is walked in chunks of 4 whose indices, then blocks, are all requested before the first use: two memory
  // round trips per chunk instead of two per edge (the chain of dependent loads is what this kernel costs).  Sum order: the
  // fused launch's landmark quads (lin_points_block) -- edge j of the list goes to partial sum j & 3, then ((s0 + s1) + s2) + s3
  const int jb = pt_start[l], je = pt_start[l + 1];
  double part[4][9];
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int i = 0; i < 9; i++) part[q][i] = 0;
  for (int j0 = jb; j0 < je; j0 += 4) {
    int id[4];
#pragma unroll
    for (int q = 0; q < 4; q++) id[q] = pt_edges[min(j0 + q, je - 1)];
    double v[4][9];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const double* eb = EB + (size_t)id[q] * kEB + 18;
#pragma unroll
      for (int i = 0; i < 9; i++) v[q][i] = eb[i];
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (j0 + q < je) {
#pragma unroll
        for (int i = 0; i < 9; i++) part[q][i] += v[q][i];
      }
  }
#pragma unroll
  for (int i = 0; i < 9; i++) acc[i] = ((part[0][i] + part[1][i]) + part[2][i]) + part[3][i];
  for (int i = 0; i < 6; i++) Hll[6 * (size_t)l + i] = acc[i];
  for (int i = 0; i < 3; i++) bl[3 * (size_t)l + i] = acc[6 + i];
}

__device__ inline void inv3_sym(const double* h6, double lambda, double* o) {
  // h6 = upper (00,01,02,11,12,22); Eigen-style cofactor inverse of the full symmetric matrix + lambda I
  const double m0 = h6[0] + lambda, m1 = h6[1], m2 = h6[2], m4 = h6[3] + lambda, m5 = h6[4], m8 = h6[5] + lambda;
  const double c00 = m4 * m8 - m5 * m5, c01 = m5 * m2 - m1 * m8, c02 = m1 * m5 - m4 * m2;
  const double det = m0 * c00 + m1 * c01 + m2 * c02;
  const double id = 1.0 / det;
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = o[1]; o[4] = (m0 * m8 - m2 * m2) * id; o[5] = (m2 * m1 - m0 * m5) * id;
  o[6] = o[2]; o[7] = o[5]; o[8] = (m0 * m4 - m1 * m1) * id;
}

// Schur complement block (i1 <= i2) and, on diagonal pairs, the reduced rhs.
struct PairItem { int ea, eb, l; };   // edges (pose i1 / pose i2) of landmark l

constexpr int kSchurThreads = 256;

// Per landmark: its free observations sorted by pose column (stable: the order the host's insertion sort produces), their
// columns and the landmark's pose mask -- the inputs of k_build_items and k_update.  One thread per landmark; the list (a
// handful of entries, at most one per free pose) is sorted in LDS.  Replaces ~25 us of host loops per solve during which the
// device sat idle between the first linearisation and the first Schur complement.
constexpr int kSortPfThreads = 64, kSortPfCap = 64;
// BY_EDGE: the input list is in no particular order (atomic fill): order by (column, edge index); otherwise the input is
// in edge order and a stable sort by column gives the same
template <int NT, int CAP, bool BY_EDGE = false>
__device__ __forceinline__ void sort_pf_block(int bid, int nL, const int* __restrict__ pf_start, int* __restrict__ pf_edges,
                                              int* __restrict__ pf_col, unsigned long long* __restrict__ lm_mask,
                                              const lba_edge* __restrict__ edges, const int* __restrict__ pose_col) {
  __shared__ int s_e[NT][CAP + 1];      // (+1: rows start on different banks)
  __shared__ int s_c[NT][CAP + 1];
  const int l = bid * NT + threadIdx.x;
  if (l >= nL) return;
  const int b0 = pf_start[l], cnt = pf_start[l + 1] - b0;
  unsigned long long m = 0;
  if (cnt <= CAP) {
    int* const se = s_e[threadIdx.x];
    int* const sc = s_c[threadIdx.x];
    for (int j = 0; j < cnt; j++) se[j] = pf_edges[b0 + j];
    for (int j = 0; j < cnt; j++) sc[j] = pose_col[edges[se[j]].pose];
    for (int a2 = 1; a2 < cnt; a2++) {
      const int e = se[a2], key = sc[a2];
      int b2 = a2 - 1;
      while (b2 >= 0 && (sc[b2] > key || (BY_EDGE && sc[b2] == key && se[b2] > e))) { se[b2 + 1] = se[b2]; sc[b2 + 1] = sc[b2]; b2--; }
      se[b2 + 1] = e; sc[b2 + 1] = key;
    }
    for (int j = 0; j < cnt; j++) { pf_edges[b0 + j] = se[j]; pf_col[b0 + j] = sc[j]; m |= 1ull << sc[j]; }
  } else {                                                 // (more observations than free poses: repeated edges) in place
    for (int a2 = b0 + 1; a2 < b0 + cnt; a2++) {
      const int e = pf_edges[a2], key = pose_col[edges[e].pose];
      int b2 = a2 - 1;
      while (b2 >= b0 && (pose_col[edges[pf_edges[b2]].pose] > key ||
                          (BY_EDGE && pose_col[edges[pf_edges[b2]].pose] == key && pf_edges[b2] > e))) { pf_edges[b2 + 1] = pf_edges[b2]; b2--; }
      pf_edges[b2 + 1] = e;
    }
    for (int j = b0; j < b0 + cnt; j++) { const int c = pose_col[edges[pf_edges[j]].pose]; pf_col[j] = c; m |= 1ull << c; }
  }
  lm_mask[l] = m;
}

// CSR lists on the device (edges per active point, per free pose, per active point restricted to free poses).  The host
// counts degrees while it copies the edges, so the list STARTS are known; the fill is one atomic cursor per list (k_csr_fill),
// and because the kernels sum over a list in list order -- and that order has to be the edge order of the oracle -- every
// list is then sorted by edge index (k_csr_sort: a point's handful of entries by its thread in LDS, a pose's few hundred to
// few thousand by a workgroup's bitonic network).  Replaces a host pass of 29 us (C2) / 84 us (C4) during which the device
// had nothing to do.
__global__ __launch_bounds__(256) void k_csr_fill(int n_edges, const lba_edge* __restrict__ edges, const int* __restrict__ pose_col,
                                                 const int* __restrict__ point_col, int* __restrict__ cur_pt, int* __restrict__ cur_ps,
                                                 int* __restrict__ cur_pf, int* __restrict__ pt_edges, int* __restrict__ ps_edges,
                                                 int* __restrict__ pf_edges) {
  // (the pose cursors are few and hot: a workgroup first ranks its edges per pose in LDS and takes one range per pose)
  __shared__ int s_cnt[64], s_base[64];
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < 64) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  int lc = -1, pc = -1, my = 0;
  if (k < n_edges) {
    const lba_edge e = edges[k];
    lc = point_col[e.point]; pc = pose_col[e.pose];
    pt_edges[atomicAdd(&cur_pt[lc], 1)] = k;
    if (pc >= 0) { pf_edges[atomicAdd(&cur_pf[lc], 1)] = k; my = atomicAdd(&s_cnt[pc], 1); }
  }
  __syncthreads();
  if (threadIdx.x < 64 && s_cnt[threadIdx.x] > 0) s_base[threadIdx.x] = atomicAdd(&cur_ps[threadIdx.x], s_cnt[threadIdx.x]);
  __syncthreads();
  if (pc >= 0) ps_edges[s_base[pc] + my] = k;
}
constexpr int kPrep256Cap = 16, kPrep256Pad = 8, kPrep256Zero = 4;
constexpr int kCsrPoseCap = 4096, kCsrPtCap = 16;
__global__ __launch_bounds__(256) void k_csr_sort(int nP, int nL, const int* __restrict__ ps_start, int* __restrict__ ps_edges,
                                                 const int* __restrict__ pt_start, int* __restrict__ pt_edges,
                                                 const int* __restrict__ pf_start, int* __restrict__ pf_edges, int* __restrict__ pf_col,
                                                 unsigned long long* __restrict__ lm_mask, const lba_edge* __restrict__ edges,
                                                 const int* __restrict__ pose_col, int n_unknowns, double* __restrict__ St,
                                                 double* __restrict__ xzero, int n_zero) {
  __shared__ int s_pose[kCsrPoseCap];
  __shared__ int s_pt[256][kCsrPtCap + 1];
  int bid = (int)blockIdx.x;
  const int tid = threadIdx.x;
  if (bid < nP) {
    // one free pose: its edge list ascending (bitonic network over the next power of two, padded with INT_MAX)
    const int b0 = ps_start[bid], cnt = ps_start[bid + 1] - b0;
    int m = 1;
    while (m < cnt) m <<= 1;
    for (int i = tid; i < m; i += 256) s_pose[i] = i < cnt ? ps_edges[b0 + i] : 0x7FFFFFFF;
    __syncthreads();
    for (int k2 = 2; k2 <= m; k2 <<= 1)
      for (int j = k2 >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < m; i += 256) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const int a = s_pose[i], c = s_pose[ixj];
            const bool up = (i & k2) == 0;
            if ((a > c) == up) { s_pose[i] = c; s_pose[ixj] = a; }
          }
        }
        __syncthreads();
      }
    for (int i = tid; i < cnt; i += 256) ps_edges[b0 + i] = s_pose[i];
    return;
  }
  bid -= nP;
  const int n_pt_blocks = (nL + 255) / 256;
  if (bid < n_pt_blocks) {
    const int l = bid * 256 + tid;
    if (l < nL) {
      // the point's edges ascending
      const int b0 = pt_start[l], cnt = pt_start[l + 1] - b0;
      if (cnt <= kCsrPtCap) {
        int* const se = s_pt[tid];
        for (int j = 0; j < cnt; j++) se[j] = pt_edges[b0 + j];
        for (int a2 = 1; a2 < cnt; a2++) {
          const int e = se[a2];
          int b2 = a2 - 1;
          while (b2 >= 0 && se[b2] > e) { se[b2 + 1] = se[b2]; b2--; }
          se[b2 + 1] = e;
        }
        for (int j = 0; j < cnt; j++) pt_edges[b0 + j] = se[j];
      } else {
        for (int a2 = b0 + 1; a2 < b0 + cnt; a2++) {
          const int e = pt_edges[a2];
          int b2 = a2 - 1;
          while (b2 >= b0 && pt_edges[b2] > e) { pt_edges[b2 + 1] = pt_edges[b2]; b2--; }
          pt_edges[b2 + 1] = e;
        }
      }
    }
    // its free observations by (pose column, edge index), their columns and the pose mask
    sort_pf_block<256, kCsrPtCap, true>(bid, nL, pf_start, pf_edges, pf_col, lm_mask, edges, pose_col);
    return;
  }
  bid -= n_pt_blocks;
  if (bid < kPrep256Pad) { if (St) ldltm::image_pad_range(n_unknowns, St, bid * 256 + tid, kPrep256Pad * 256); return; }
  bid -= kPrep256Pad;
  for (int i = bid * 256 + tid; i < n_zero; i += kPrep256Zero * 256) xzero[i] = 0.0;
}

// First launch of a solve: the linearisation of the initial estimate and, in further workgroups, everything else that needs
// only the first upload -- the per-landmark observation lists (256 landmarks per workgroup, lists of up to 16 entries sorted
// in LDS), the padding of the tile image, the zeroed step vector.  (Two launches: 12 + 6.6 us one after the other.)
template <class CamT>
__global__ __launch_bounds__(256) void k_errlin_prep(int n_errlin_blocks, int n_sort_blocks, int* __restrict__ pf_edges_w, int* __restrict__ pf_col_w,
                                                    const int* __restrict__ pf_start, unsigned long long* __restrict__ lm_mask,
                                                    int n_unknowns, double* __restrict__ St, double* __restrict__ xzero, int n_zero,
                                                    int nP, int n_edges, const lba_edge* __restrict__ edges, const PoseQ* __restrict__ poses,
                                               const double* __restrict__ points, CamT c, Huber hb, double* __restrict__ err,
                                               double* __restrict__ chi2, const int* __restrict__ pose_col,
                                               const int* __restrict__ point_col, double* __restrict__ EB,
                                               const int* __restrict__ ps_start, const int* __restrict__ ps_edges,
                                               double* __restrict__ Hpp, double* __restrict__ bp, double* __restrict__ partial,
                                               unsigned* __restrict__ ticket, const double* __restrict__ scale_partial, int n_scale_partial,
                                               const int* __restrict__ ok_flag, HostRec* __restrict__ rec, unsigned seq, int n_edge_blocks,
                                               int nL, const int* __restrict__ pt_start, const int* __restrict__ pt_edges,
                                               double* __restrict__ Hll, double* __restrict__ bl, LmIn lm) {
  int bid = (int)blockIdx.x;
  if (bid >= n_errlin_blocks) {
    bid -= n_errlin_blocks;
    if (bid < n_sort_blocks) { sort_pf_block<256, kPrep256Cap>(bid, nL, pf_start, pf_edges_w, pf_col_w, lm_mask, edges, pose_col); return; }
    bid -= n_sort_blocks;
    if (bid < kPrep256Pad) { if (St) ldltm::image_pad_range(n_unknowns, St, bid * 256 + (int)threadIdx.x, kPrep256Pad * 256); return; }
    bid -= kPrep256Pad;
    for (int i = bid * 256 + (int)threadIdx.x; i < n_zero; i += kPrep256Zero * 256) xzero[i] = 0.0;
    return;
  }

  
  if (bid < nP) {
    lin_poses_block<true>(bid, ps_start, ps_edges, edges, poses, points, c, hb, err, chi2, Hpp, bp);
  } else if (bid < nP + n_edge_blocks) {
    linearize_block<true>(bid - nP, n_edges, edges, poses, points, c, hb, err, chi2, pose_col, point_col, EB,
                          TrialPublish{partial, ticket, scale_partial, n_scale_partial, ok_flag, rec, seq, n_edge_blocks, lm});
  } else if (bid - nP - n_edge_blocks < errlin_point_blocks(nL)) {
    lin_points_block(bid - nP - n_edge_blocks, nL, pt_start, pt_edges, edges, poses, points, c, hb, Hll, bl);
  } else {
    publish_trial_when_all_arrived(n_edge_blocks, partial, ticket, scale_partial, n_scale_partial, ok_flag, rec, seq, lm);
  }
}

// k_sort_pf, the padding of the bordered tile image (ldltm::k_image_pad) and the zeroing of the step vector in ONE launch:
// at the start of a solve the worker thread issues a dozen launches back to back and the device waits for each of them.
__global__ __launch_bounds__(kSortPfThreads) void k_prep(int n_sort_blocks, int nL, const int* __restrict__ pf_start, int* __restrict__ pf_edges,
                                                        int* __restrict__ pf_col, unsigned long long* __restrict__ lm_mask,
                                                        const lba_edge* __restrict__ edges, const int* __restrict__ pose_col,
                                                        int n, double* __restrict__ St, double* __restrict__ xzero, int n_zero) {
  constexpr int kPadBlocks = 32;
  const int bid = (int)blockIdx.x;
  if (bid < n_sort_blocks) { sort_pf_block<kSortPfThreads, kSortPfCap>(bid, nL, pf_start, pf_edges, pf_col, lm_mask, edges, pose_col); return; }
  if (bid < n_sort_blocks + kPadBlocks) {
    if (St) ldltm::image_pad_range(n, St, (bid - n_sort_blocks) * kSortPfThreads + (int)threadIdx.x, kPadBlocks * kSortPfThreads);
    return;
  }
  const int nz = (int)gridDim.x - n_sort_blocks - kPadBlocks;
  for (int i = (bid - n_sort_blocks - kPadBlocks) * kSortPfThreads + (int)threadIdx.x; i < n_zero; i += nz * kSortPfThreads) xzero[i] = 0.0;
}

// Pose-pair -> shared-landmark items built on the device (windows of up to 64 free poses): one workgroup per pose pair
// walks the landmarks in index order, keeps those whose pose mask has both bits (ballot + popcount scan: the items come out
// in landmark order, exactly as the host's counting sort produces them, so the Schur sums keep their order and bits) and
// looks the two edges up in the landmark's short, sorted observation list.  Pair p owns items[p * cap, p * cap + count[p]).
// Replaces ~30 us of host loops and a 0.5 MB upload per solve that sat between the first kernels and the first k_schur.
__device__ __forceinline__ void build_items_block(int pr, int nP, int nL, const unsigned long long* __restrict__ lm_mask,
                                                  const int* __restrict__ pf_start, const int* __restrict__ pf_edges,
                                                  const int* __restrict__ pf_col, PairItem* __restrict__ items, int cap,
                                                  int* __restrict__ pair_count) {
  __shared__ int wcount[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int i1 = 0, rem = pr;
  while (rem >= nP - i1) { rem -= nP - i1; i1++; }
  const int i2 = i1 + rem;
  const unsigned long long need = (1ull << i1) | (1ull << i2);
  PairItem* out = items + (size_t)pr * cap;
  int running = 0;
  for (int l0 = 0; l0 < nL; l0 += 256) {
    const int l = l0 + tid;
    const unsigned long long m = l < nL ? lm_mask[l] : 0ull;
    const bool has = (m & need) == need;
    const unsigned long long bal = __ballot(has);
    if (lane == 0) wcount[wv] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const int c = wcount[w]; if (w < wv) before += c; total += c; }
    if (has) {
      const int pos = running + before + __popcll(bal & ((1ull << lane) - 1ull));
      // the landmark's observations are sorted by pose column: the k-th set bit of its mask is its k-th list entry
      const int b0 = pf_start[l];
      const int ea = pf_edges[b0 + __popcll(m & ((1ull << i1) - 1ull))];
      const int eb = pf_edges[b0 + __popcll(m & ((1ull << i2) - 1ull))];
      if (pos < cap) out[pos] = PairItem{ea, eb, l};        // (cap = edges of the busiest pose: cannot overflow unless a pose observes a landmark twice)
    }
    running += total;
    __syncthreads();
  }
  if (tid == 0) pair_count[pr] = min(running, cap);
}
__global__ __launch_bounds__(256) void k_build_items(int nP, int nL, const unsigned long long* __restrict__ lm_mask,
                                                    const int* __restrict__ pf_start, const int* __restrict__ pf_edges,
                                                    const int* __restrict__ pf_col, PairItem* __restrict__ items, int cap,
                                                    int* __restrict__ pair_count) {
  build_items_block((int)blockIdx.x, nP, nL, lm_mask, pf_start, pf_edges, pf_col, items, cap, pair_count);
}

// Round 6, measured and dropped (tools/micro/lba_prof.py: the launch is as long as its 20 DIAGONAL pairs -- every landmark of a pose, ~550
// items, 230 KB of Jacobian blocks through one compute unit: item loop 6.3 us, block sums at 7.6, end 8.8; the 190 off-diagonal pairs
// end after 3.5):
//  * 512 threads per pair: the first thread was through its items after 3.7 us, the block sums were complete no earlier (7.96 against
//    7.56 us), with two transposes of 21 values for eight wavefronts or with the upper four folded into the one-pass transpose; 9.4 us;
//  * a thread's item records of all three rounds requested up front: 8.94 against 9.03 us (the loop is not waiting for them);
//  * each diagonal pair on FOUR workgroups (a quarter of the items each, 42 block sums per quarter through agent-scope stores, a ticket,
//    the last one adds the quarters in order): item loop 3.0 us, sums at 4.9 -- and 4-5 us for the store / ticket / load hand-over
//    between workgroups on different XCDs: 10.3 us.
template <int PASSES>      // 1: all 42 values in one transpose (88 KB of LDS: one workgroup per compute unit), 2: two passes of 21 (44 KB: three)
__global__ __launch_bounds__(kSchurThreads) void k_schur(int nP, const int* __restrict__ pair_i1, const int* __restrict__ pair_i2,
                                                        const int* __restrict__ pair_start, const PairItem* __restrict__ items,
                                                        const double* __restrict__ EB, const double* __restrict__ Hll,
                                                        const double* __restrict__ bl, const double* __restrict__ Hpp,
                                                        const double* __restrict__ bp, double lambda_v, double* __restrict__ S,
                                                        double* __restrict__ bs, const double* __restrict__ lambda_p, int item_cap,
                                                        const int* __restrict__ pair_count, double* __restrict__ St) {
  const double lambda = lambda_p ? *lambda_p : lambda_v;      // first trial of a round: lambda was computed on the device
  // one workgroup per pose pair, one thread per shared landmark (the diagonal pairs hold every landmark of the pose:
  // ~550 at C2, so 256 threads keep their item loop at 3 rounds); sums in a fixed order: per thread, then 4 x 64, then 4
  // per value one row of 4 segments of 64 + 1 doubles: in the second phase lane (o, q) walks segment q of row o, and without
  // the per-segment pad the four q-lanes of a row hit the same bank on every read (SQ_LDS_BANK_CONFLICT: 371 k cycles)
  // With more pairs than compute units (C4: 50 poses, 1275 pairs) the 42 values go through the transpose in TWO passes of 21:
  // 44 KB of LDS per workgroup instead of 88 -- three workgroups per compute unit instead of one (the kernel ran five rounds of
  // one four-wavefront workgroup per compute unit, each the latency of its dependent loads, item -> landmark / edge blocks:
  // 35 -> 28 us).  The additions and their order are the same in both forms: the same bits.
  constexpr int kSeg = 65, kRow = 4 * kSeg + 1, kPass = 42 / PASSES;
  __shared__ double red[kPass * kRow];
  __shared__ double part[42][4];
  LBA_PROF_BEGIN(1);
  // XCD-aware workgroup -> pair map: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8), each with its own L2;
  // consecutive pairs share a pose and with it that pose's edge blocks, so every XCD takes a contiguous run of the pair list
  // (the b/8-th workgroup of XCD k gets the k-th run's b/8-th pair)
  int pr;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, k = blockIdx.x & 7;
    pr = k * q + min(k, r) + (blockIdx.x >> 3);
  }
  int i1, i2;
  if (pair_i1) { i1 = pair_i1[pr]; i2 = pair_i2[pr]; }
  else {                                                   // pair id = i1 nP - i1 (i1 - 1) / 2 + (i2 - i1), rows in order
    int rem = pr; i1 = 0;
    while (rem >= nP - i1) { rem -= nP - i1; i1++; }
    i2 = i1 + rem;
  }
  const int tid = threadIdx.x;
  double acc[36], cacc[6];
#pragma unroll
  for (int i = 0; i < 36; i++) acc[i] = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) cacc[i] = 0;
  const bool diag = i1 == i2;
  // items of this pair: CSR offsets from the host build, or a fixed-capacity segment filled by k_build_items
  const int j0 = pair_start ? pair_start[pr] : pr * item_cap;
  const int j1 = pair_start ? pair_start[pr + 1] : j0 + pair_count[pr];
  for (int j = j0 + tid; j < j1; j += kSchurThreads) {
    const PairItem it = items[j];
    double Dinv[9];
    inv3_sym(Hll + 6 * (size_t)it.l, lambda, Dinv);
    const double* Bi = EB + (size_t)it.ea * kEB;
    const double* Bj = EB + (size_t)it.eb * kEB;
    double BD[18];
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int c = 0; c < 3; c++) BD[3 * a + c] = Bi[3 * a] * Dinv[c] + Bi[3 * a + 1] * Dinv[3 + c] + Bi[3 * a + 2] * Dinv[6 + c];
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int c = 0; c < 6; c++) acc[6 * a + c] += BD[3 * a] * Bj[3 * c] + BD[3 * a + 1] * Bj[3 * c + 1] + BD[3 * a + 2] * Bj[3 * c + 2];
    if (diag) {
      const double* b = bl + 3 * (size_t)it.l;
#pragma unroll
      for (int a = 0; a < 6; a++) cacc[a] += BD[3 * a] * b[0] + BD[3 * a + 1] * b[1] + BD[3 * a + 2] * b[2];
    }
  }
  LBA_PROF_PHASE(0, 0);               // workgroup 0 = pair (0, 0), a diagonal pair: the item loop is done
  const int slot = (tid >> 6) * kSeg + (tid & 63);
  const int nval = diag ? 42 : 36;
#pragma unroll
  for (int ps = 0; ps < PASSES; ps++) {
    if (ps == 1) __syncthreads();                                   // the first pass' sums have been read
#pragma unroll
    for (int i = 0; i < kPass; i++) {
      const int v = ps * kPass + i;                                 // compile-time
      red[i * kRow + slot] = v < 36 ? acc[v < 36 ? v : 0] : cacc[v >= 36 ? v - 36 : 0];
    }
    __syncthreads();
    if (tid < 4 * kPass && ps * kPass + (tid >> 2) < nval) {
      const int o = tid >> 2, q = tid & 3;
      const double* r = red + o * kRow + q * kSeg;
      double s = 0;
      // (index order, sixteen LDS reads in flight per step -- see publish_trial_tail)
#pragma unroll
      for (int k0 = 0; k0 < 64; k0 += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = r[k0 + u];
#pragma unroll
        for (int u = 0; u < 16; u++) s += v[u];
      }
      part[ps * kPass + o][q] = s;
    }
  }
  __syncthreads();
  LBA_PROF_PHASE(0, 1);               // ... the 42 block sums are in LDS
  const int n = 6 * nP;
  if (tid < 36) {
    const double s = ((part[tid][0] + part[tid][1]) + part[tid][2]) + part[tid][3];
    const int a = tid / 6, c = tid % 6;
    double v = -s;
    if (diag) {
      const int lo = a < c ? a : c, hi = a < c ? c : a;
      const int u = lo * 6 - lo * (lo - 1) / 2 + (hi - lo);        // index into the 21 upper entries
      v += Hpp[21 * (size_t)i1 + u] + (a == c ? lambda : 0.0);
    }
    if (St) {
      // the matrix-core LDL^T reads the matrix as a tile image (ldlt_mfma.hpp): upper-triangle tiles only, the diagonal
      // tiles with both triangles
      const int r = 6 * i1 + a, cc = 6 * i2 + c;
      const int p0 = ldltm::tile_image_pos(r, cc), p1 = ldltm::tile_image_pos(cc, r);
      if (p0 >= 0) St[p0] = v;
      if (p1 >= 0 && r != cc) St[p1] = v;
    } else {
      S[(size_t)(6 * i1 + a) * n + 6 * i2 + c] = v;
      if (!diag) S[(size_t)(6 * i2 + c) * n + 6 * i1 + a] = v;
    }
  } else if (diag && tid < 42) {
    const int a = tid - 36;
    const double s = ((part[tid][0] + part[tid][1]) + part[tid][2]) + part[tid][3];
    const double v = bp[6 * (size_t)i1 + a] - s;
    bs[6 * i1 + a] = v;
    if (St) ldltm::image_put_rhs(St, n, 6 * i1 + a, v);     // the solver reads the right-hand side as the matrix' border column
  }
  LBA_PROF_END();
}

// ---- Blocked LDL^T + solve over MANY workgroups, for windows beyond the matrix-core kernels (more than 50 free poses:
// the matrix-core kernels hold <= 19 tile rows in one CU's registers, the row-pair kernel <= 1344 blocks).  Right-looking,
// 16-column blocks, dense row-major S in global memory (L2-resident), two launches per block column:
//   k_wide_panel(kb):  every workgroup factors the 16 x 16 diagonal block itself (no hand-over between workgroups).  The
//                      block is READ from its diagonal and upper triangle only and workgroup 0 WRITES the strictly lower
//                      triangle only (L_kk for the back-substitution), so a workgroup that is dispatched after workgroup 0
//                      has finished still factors the unfactored block: no location is input and output of one launch;
//                      workgroup 0 also forward-substitutes the right-hand side's block, workgroup g >= 1 turns
//                      row block kb + g into L = A L_kk^-T D^-1 (in place) and W = L D (kept in the mirrored upper block);
//   k_wide_update(kb): A_ij -= W_ik L_jk^T for every trailing block, and the right-hand side as one more row;
// then k_wide_back: x = L^-T z in one workgroup.  Same failure rule as k_ldlt: a zero / non-finite pivot clears the flag.
__global__ __launch_bounds__(256) void k_wide_panel(int n, int kb, double* S, const double* __restrict__ b, double* yw, double* z,
                                                    int* __restrict__ ok_flag) {
  __shared__ double Lk[16][17];
  __shared__ double Aw[16][17];
  const int tid = threadIdx.x, r = tid >> 4, c = tid & 15;
  const int k0 = 16 * kb;
  {
    const int gr = k0 + r, gc = k0 + c;
    const int lo = gr < gc ? gr : gc, hi = gr < gc ? gc : gr;
    Lk[r][c] = (gr < n && gc < n) ? S[(size_t)lo * n + hi] : (r == c ? 1.0 : 0.0);
  }
  if (blockIdx.x == 0 && kb == 0 && tid == 0) *ok_flag = 1;
  __syncthreads();
  bool bad = false;
  for (int j = 0; j < 16; j++) {
    const double d = Lk[j][j];
    if (d == 0.0 || !(d == d) || fabs(d) == INFINITY) bad = true;
    const bool below = r > j, upd = below && c > j && c <= r;
    double v = 0.0, wc = 0.0;
    if (below) v = Lk[r][j];
    if (upd) wc = Lk[c][j];
    __syncthreads();
    const double l = v / d;
    if (upd) Lk[r][c] -= l * wc;
    if (below && c == j) Lk[r][j] = l;
    __syncthreads();
  }
  if (blockIdx.x == 0) {
    if (bad && tid == 0) *ok_flag = 0;
    const int gr = k0 + r, gc = k0 + c;
    if (gr < n && gc < n && c < r) S[(size_t)gr * n + gc] = Lk[r][c];
    // the right-hand side as a one-row block: w = b_k^T L_kk^-T (kept for the trailing update), z = w / D
    const double* src = kb == 0 ? b : yw;
    Aw[r][c] = (r == 0 && k0 + c < n) ? src[k0 + c] : 0.0;
    __syncthreads();
    for (int m = 0; m < 15; m++) {
      if (r == 0 && c > m) Aw[0][c] -= Aw[0][m] * Lk[c][m];
      __syncthreads();
    }
    if (r == 0 && k0 + c < n) { const double w = Aw[0][c]; yw[k0 + c] = w; z[k0 + c] = w / Lk[c][c]; }
    return;
  }
  const int i0 = 16 * (kb + (int)blockIdx.x);
  Aw[r][c] = i0 + r < n ? S[(size_t)(i0 + r) * n + k0 + c] : 0.0;
  __syncthreads();
  for (int m = 0; m < 15; m++) {
    if (c > m) Aw[r][c] -= Aw[r][m] * Lk[c][m];
    __syncthreads();
  }
  if (i0 + r < n) {
    const double w = Aw[r][c];
    S[(size_t)(i0 + r) * n + k0 + c] = w / Lk[c][c];
    S[(size_t)(k0 + c) * n + i0 + r] = w;
  }
}

// grid (m, m + 1), m = trailing row blocks: block (jj, ii) updates tile (kb+1+ii, kb+1+jj), ii >= jj; row ii == m is the right-hand side
__global__ __launch_bounds__(256) void k_wide_update(int n, int kb, double* S, const double* __restrict__ b, double* yw) {
  const int m_blocks = gridDim.x, jj = blockIdx.x, ii = blockIdx.y;
  if (ii < m_blocks && ii < jj) return;
  __shared__ double Wt[16][17];
  __shared__ double Lj[16][17];
  const int tid = threadIdx.x, r = tid >> 4, c = tid & 15;
  const int k0 = 16 * kb, j0 = 16 * (kb + 1 + jj);
  Lj[r][c] = j0 + r < n ? S[(size_t)(j0 + r) * n + k0 + c] : 0.0;               // L_jk[r][c]
  if (ii == m_blocks) {
    if (r == 0) Wt[0][c] = yw[k0 + c];
    __syncthreads();
    if (r == 0 && j0 + c < n) {
      double acc = 0.0;
      for (int m = 0; m < 16; m++) acc += Wt[0][m] * Lj[c][m];
      yw[j0 + c] = (kb == 0 ? b[j0 + c] : yw[j0 + c]) - acc;
    }
    return;
  }
  const int i0 = 16 * (kb + 1 + ii);
  Wt[c][r] = i0 + c < n ? S[(size_t)(k0 + r) * n + i0 + c] : 0.0;               // W_ik[c][r], read along the mirrored block's rows
  __syncthreads();
  if (i0 + r < n && j0 + c < n) {
    double acc = 0.0;
    for (int m = 0; m < 16; m++) acc += Wt[r][m] * Lj[c][m];
    S[(size_t)(i0 + r) * n + j0 + c] -= acc;
  }
}

__global__ __launch_bounds__(1024) void k_wide_back(int n, const double* __restrict__ S, const double* __restrict__ z, double* __restrict__ x,
                                                    const int* __restrict__ ok_flag) {
  extern __shared__ double xs[];
  if (*ok_flag == 0) return;
  const int tid = threadIdx.x, T = (n + 15) / 16;
  for (int i = tid; i < n; i += 1024) xs[i] = z[i];
  __syncthreads();
  for (int kb = T - 1; kb >= 0; kb--) {
    const int k0 = 16 * kb, w = min(16, n - k0);
    if (tid < 64) {
      // the diagonal block: lane i < 16 holds x_i and column i of L_kk; x_j goes round by a lane read, 15 steps in registers
      const int i = tid & 15;
      double lcol[16];
#pragma unroll
      for (int j = 0; j < 16; j++) lcol[j] = (j > i && j < w) ? S[(size_t)(k0 + j) * n + k0 + i] : 0.0;
      double xv = i < w ? xs[k0 + i] : 0.0;
#pragma unroll
      for (int j = 15; j > 0; j--) {
        const double xj = __shfl(xv, j, 16);
        xv -= lcol[j] * xj;
      }
      if (tid < w) xs[k0 + tid] = xv;
    }
    __syncthreads();
    for (int i = tid; i < k0; i += 1024) {
      double acc = 0.0;
      for (int cc = 0; cc < w; cc++) acc += S[(size_t)(k0 + cc) * n + i] * xs[k0 + cc];
      xs[i] -= acc;
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += 1024) x[i] = xs[i];
}

constexpr int kWideMaxUnknowns = 8192;   // k_wide_back keeps x in LDS: n * 8 bytes <= 64 KB (1365 free poses)
static hipError_t launch_ldlt_wide(int n, double* S, const double* b, double* x, int* ok, double* scratch, hipStream_t st) {
  if (n > kWideMaxUnknowns) return hipErrorInvalidValue;
  const int T = (n + 15) / 16;
  double* yw = scratch;
  double* z = scratch + n;
  for (int kb = 0; kb < T; kb++) {
    hipLaunchKernelGGL(k_wide_panel, dim3(T - kb), dim3(256), 0, st, n, kb, S, b, yw, z, ok);
    const int m = T - kb - 1;
    if (m > 0) hipLaunchKernelGGL(k_wide_update, dim3(m, m + 1), dim3(256), 0, st, n, kb, S, b, yw);
  }
  hipLaunchKernelGGL(k_wide_back, dim3(1), dim3(1024), (size_t)n * sizeof(double), st, n, S, z, x, ok);
  return hipGetLastError();
}

// trial state = oplus(current, x): poses exp(x_p) * T; points X + x_l with the landmark back-substitution
// x_l = Dinv_l (bl_l - sum_i Hpl_il^T x_i) folded in (x_l is also stored for computeScale)
template <int NT>
__global__ __launch_bounds__(NT) void k_update(int n_poses, int n_points, int nP, const int* __restrict__ pose_col,
                                               const int* __restrict__ point_col, const PoseQ* __restrict__ poses,
                                               const double* __restrict__ points, double* __restrict__ x,
                                               const int* __restrict__ pf_start, const int* __restrict__ pf_edges,
                                               const int* __restrict__ pf_col, const double* __restrict__ EB,
                                               const double* __restrict__ Hll, const double* __restrict__ bl, double lambda_v,
                                               PoseQ* __restrict__ poses_out, double* __restrict__ points_out,
                                               const double* __restrict__ bp, double* __restrict__ scale_partial,
                                               const double* __restrict__ lambda_p) {
  LBA_PROF_BEGIN(2);
  const double lambda = lambda_p ? *lambda_p : lambda_v;
  __shared__ double red[NT];
  const int i = blockIdx.x * NT + threadIdx.x;
  double sc = 0;                     // this thread's share of computeScale(): sum x (lambda x + b)  (levenberg.cpp:187-194)
  if (i < n_points) {
    const int l = point_col[i];
    double dx[3] = {0, 0, 0};
    if (l >= 0) {
      double cl[3] = {bl[3 * (size_t)l], bl[3 * (size_t)l + 1], bl[3 * (size_t)l + 2]};
      // the point's observations in chunks of kUpdChunk: indices, then all blocks of the chunk, are requested before the first
      // use (one memory round trip per chunk instead of two per observation; with 8 per chunk a point of up to 8 observations --
      // nearly all of them -- costs two round trips in all); the subtraction order is unchanged
      constexpr int kUpdChunk = 8;
      const int j1 = pf_start[l + 1];
      for (int j0 = pf_start[l]; j0 < j1; j0 += kUpdChunk) {
        int eid[kUpdChunk], col[kUpdChunk];
#pragma unroll
        for (int u = 0; u < kUpdChunk; u++) { const int j = min(j0 + u, j1 - 1); eid[u] = pf_edges[j]; col[u] = pf_col[j]; }
        double Bv[kUpdChunk][18], xv[kUpdChunk][6];
#pragma unroll
        for (int u = 0; u < kUpdChunk; u++) {
          const double* Bi = EB + (size_t)eid[u] * kEB;
          const double* xp = x + 6 * (size_t)col[u];
#pragma unroll
          for (int q = 0; q < 18; q++) Bv[u][q] = Bi[q];
#pragma unroll
          for (int a = 0; a < 6; a++) xv[u][a] = xp[a];
        }
#pragma unroll
        for (int u = 0; u < kUpdChunk; u++)
          if (j0 + u < j1) {
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
              for (int a = 0; a < 6; a++) cl[c] -= Bv[u][3 * a + c] * xv[u][a];
          }
      }
      double Dinv[9];
      inv3_sym(Hll + 6 * (size_t)l, lambda, Dinv);
      for (int a = 0; a < 3; a++) {
        dx[a] = Dinv[3 * a] * cl[0] + Dinv[3 * a + 1] * cl[1] + Dinv[3 * a + 2] * cl[2];
        x[6 * (size_t)nP + 3 * (size_t)l + a] = dx[a];
        sc += dx[a] * (lambda * dx[a] + bl[3 * (size_t)l + a]);
      }
    }
    for (int a = 0; a < 3; a++) points_out[3 * (size_t)i + a] = points[3 * (size_t)i + a] + dx[a];
  } else if (i < n_points + n_poses) {
    const int p = i - n_points;
    const int c = pose_col[p];
    if (c >= 0) {
      pose_oplus(poses[p], x + 6 * (size_t)c, &poses_out[p]);
      for (int a = 0; a < 6; a++) { const double xv = x[6 * (size_t)c + a]; sc += xv * (lambda * xv + bp[6 * (size_t)c + a]); }
    } else {
      poses_out[p] = poses[p];
    }
  }
  LBA_PROF_PHASE(0, 0);               // workgroup 0: its 64 landmarks are updated
  // fixed-order block sum -> one partial per block (the last block of the following k_errors adds them up in index order)
  red[threadIdx.x] = sc;
  __syncthreads();
  for (int s2 = NT / 2; s2 > 0; s2 >>= 1) {
    if ((int)threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x == 0) scale_partial[blockIdx.x] = red[0];
  LBA_PROF_END();
}

// one workgroup: chi2 = sum partial[] (fixed order), scale = sum x (lambda x + b), maxdiag; -> pinned record
__device__ __forceinline__ void finish_block(int n_partial, const double* __restrict__ partial, int nP, int nL,
                                             const double* __restrict__ x, const double* __restrict__ bp, const double* __restrict__ bl,
                                             const double* __restrict__ Hpp, const double* __restrict__ Hll, double lambda,
                                             const int* __restrict__ ok_flag, int want_scale, int want_maxdiag, HostRec* __restrict__ rec,
                                             double lambda_init, double* __restrict__ lambda0_out) {
  __shared__ double red[256];
  __shared__ double parts[1024];
  const int tid = threadIdx.x;
  // all partials are fetched in parallel (one memory latency), then summed by one thread in a fixed order
  for (int i = tid; i < n_partial && i < 1024; i += 256) parts[i] = partial[i];
  __syncthreads();
  double chi = 0;
  if (tid == 0) {
    const int nl = min(n_partial, 1024);
    int i = 0;
    for (; i + 16 <= nl; i += 16) {          // (index order, sixteen LDS reads in flight per step)
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = parts[i + u];
#pragma unroll
      for (int u = 0; u < 16; u++) chi += v[u];
    }
    for (; i < nl; i++) chi += parts[i];
    for (i = 1024; i < n_partial; i++) chi += partial[i];
  }
  double sc = 0;
  if (want_scale) {
    const int n6 = 6 * nP, n3 = 3 * nL;
    for (int j = tid; j < n6; j += 256) sc += x[j] * (lambda * x[j] + bp[j]);
    for (int j = tid; j < n3; j += 256) sc += x[n6 + j] * (lambda * x[n6 + j] + bl[j]);
  }
  red[tid] = sc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double scale = red[0];
  __syncthreads();
  double mx = 0;
  if (want_maxdiag) {
    for (int i = tid; i < nP; i += 256) {
      const double* h = Hpp + 21 * (size_t)i;
      const int di[6] = {0, 6, 11, 15, 18, 20};
      for (int j = 0; j < 6; j++) mx = fmax(mx, fabs(h[di[j]]));
    }
    for (int i = tid; i < nL; i += 256) {
      const double* h = Hll + 6 * (size_t)i;
      mx = fmax(mx, fmax(fabs(h[0]), fmax(fabs(h[3]), fabs(h[5]))));
    }
  }
  red[tid] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
    __syncthreads();
  }
  if (tid == 0) {
    if (lambda0_out) {
      // start of an LM round (computeLambdaInit, levenberg.cpp:177-185): the first trial reads lambda from device memory, the
      // host picks chi2 / max diagonal up together with that trial's record -- no synchronisation in between
      rec->chi2_init = chi; rec->maxdiag = red[0];
      lambda0_out[0] = lambda_init > 0 ? lambda_init : 1e-5 * red[0];
      lambda0_out[2] = chi;                          // [1] = lambda of the next iteration (publish_trial_record), [2] = chi2 the round starts from
    } else {
      rec->chi2 = chi; rec->scale = scale; rec->maxdiag = red[0]; rec->ok = ok_flag ? *ok_flag : 1;
    }
  }
}
__global__ __launch_bounds__(256) void k_finish(int n_partial, const double* __restrict__ partial, int nP, int nL,
                                               const double* __restrict__ x, const double* __restrict__ bp, const double* __restrict__ bl,
                                               const double* __restrict__ Hpp, const double* __restrict__ Hll, double lambda,
                                               const int* __restrict__ ok_flag, int want_scale, int want_maxdiag, HostRec* __restrict__ rec,
                                               double lambda_init, double* __restrict__ lambda0_out) {
  finish_block(n_partial, partial, nP, nL, x, bp, bl, Hpp, Hll, lambda, ok_flag, want_scale, want_maxdiag, rec, lambda_init, lambda0_out);
}
// First iteration of a solve: the lambda init (workgroup 0) and the pose-pair items (one workgroup per pair) wait for different
// things of the launch before -- the linearisation and the sorted observation lists -- and not for each other: one launch.
__global__ __launch_bounds__(256) void k_finish_items(int n_partial, const double* __restrict__ partial, int nP, int nL,
                                                     const double* __restrict__ bp, const double* __restrict__ bl,
                                                     const double* __restrict__ Hpp, const double* __restrict__ Hll, HostRec* __restrict__ rec,
                                                     double lambda_init, double* __restrict__ lambda0_out,
                                                     const unsigned long long* __restrict__ lm_mask, const int* __restrict__ pf_start,
                                                     const int* __restrict__ pf_edges, const int* __restrict__ pf_col,
                                                     PairItem* __restrict__ items, int cap, int* __restrict__ pair_count) {
  if (blockIdx.x == 0)
    finish_block(n_partial, partial, nP, nL, nullptr, bp, bl, Hpp, Hll, 0.0, nullptr, 0, 1, rec, lambda_init, lambda0_out);
  else
    build_items_block((int)blockIdx.x - 1, nP, nL, lm_mask, pf_start, pf_edges, pf_col, items, cap, pair_count);
}

}  // namespace

// ---------------------------------------------------------------------------------------------- handle

// The abort flag as the caller owns it: the C-ABI's int32 (with its deterministic test forms) or the reference's own
// one-byte bool (LocalMapping::mbAbortBA behind Optimizer::LocalBundleAdjustment's bool* pbStopFlag, S/LocalMapping.cc:381-386)
struct StopRef {
  const volatile int32_t* i32 = nullptr;
  const volatile uint8_t* u8 = nullptr;
};
// The one switch of the solve (read ONCE, when the handle is created -- the solve path itself never calls getenv):
// ORBG_LDLT_WIDE=1 sends every window to the many-workgroup blocked LDL^T (k_wide_*: the solver of windows beyond 50 free poses) so
// that tests can run it on small problems.  Round 5 removed the measured-slower variants of rounds 1-4 (vector-ALU LDL^T kernels,
// the fused solve + update launch, the A/B forms of the start of a solve): docs/experiments.md keeps their numbers.
// ORBG_LDLT_XCD=0 keeps windows of 21 .. 50 free poses on the one-workgroup kernels instead of the eight-workgroup one
// (ldlt_xcd.hpp); =safe forces that kernel's agent-scope hand-overs (the path it takes by itself when its workgroups do not
// share an XCD).
struct LbaSwitches {
  bool ldlt_wide = false;
  int ldlt_xcd = 1;                    // 0 off, 1 on (9 .. 19 tile rows), 2 on with forced safe hand-overs
  bool ldlt_xcd_short_once = false;    // test hook (ORBG_LDLT_XCD=short): the handle's first eight-workgroup launch misses a participant
  static LbaSwitches from_env() {
    LbaSwitches w;
    w.ldlt_wide = getenv("ORBG_LDLT_WIDE") != nullptr;
    if (const char* e = getenv("ORBG_LDLT_XCD")) { w.ldlt_xcd = !strcmp(e, "0") ? 0 : !strcmp(e, "safe") ? 2 : 1; w.ldlt_xcd_short_once = !strcmp(e, "short"); }
    return w;
  }
};
struct lba_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  bool ext_stream = false;             // `stream` was handed in through lba_set_stream (never destroyed here)
  LbaSwitches sw;
  ldltm::AttrCache ldlt_attr;          // which kernels of THIS handle's device already allow their dynamic LDS size
  ldltx::Context ldlt_x;               // flags / scratch / launch counter of the eight-workgroup LDL^T
  long long xcd_timeouts = 0;          // launches of it that reported kOkTimedOut (each one: the window re-solved on the one-workgroup kernels)
  DevBuf<double> d_xscr;
  DevBuf<unsigned> d_xflags;
  DevBuf<lba_edge> d_edges;
  PinnedBuf<lba_edge> edges_pin;       // the caller's edge list, copied (and validated, counted) in ONE pass
  DevBuf<PairItem> d_items_dev;        // pair items built by k_build_items (fixed-capacity segment per pose pair)
  DevBuf<int> d_pair_count;
  DevBuf<PoseQ> d_poses[2];
  DevBuf<double> d_points[2];
  DevBuf<double> d_err, d_chi2, d_partial, d_EB, d_Hll, d_bl, d_Hpp, d_bp, d_S, d_bs, d_x;
  DevBuf<double> d_wide;               // running / scaled right-hand side of the many-workgroup LDL^T (k_wide_*)
  DevBuf<double> d_St, d_wfac;         // reduced camera matrix as a tile image / factor scratch of the matrix-core LDL^T (ldlt_mfma.hpp)
  DevBuf<double> d_EB2, d_Hll2, d_bl2, d_Hpp2, d_bp2, d_lambda0;   // second linearisation set (speculative next iteration)
  DevBuf<int> d_pose_col, d_point_col, d_pt_start, d_pt_edges, d_ps_start, d_ps_edges, d_pf_start, d_pf_edges, d_pf_col;
  DevBuf<int> d_pair_i1, d_pair_i2, d_pair_start, d_ok;
  DevBuf<PairItem> d_items;
  PinnedBuf<HostRec> rec;
  unsigned rec_seq = 0;
  PinnedBuf<uint8_t> up_h, dl_h;               // per-call upload block (built in place) / download block
  DevBuf<uint8_t> up_d;
  DevBuf<uint8_t> d_flags;
  DevBuf<double> d_scale_partial;
  DevBuf<unsigned> d_ticket;
  StreamSignal sig;              // completion word behind k_export (polled instead of hipStreamSynchronize)
  std::vector<int> s_pose_deg, s_point_deg, s_pose_col, s_point_col, s_pf_deg, s_f1, s_f2, s_f3, s_fill, s_row_off, s_junk;   // host scratch kept across calls
  std::vector<unsigned> s_cnt4;
  float last_ms = 0;
  // live measurement of the dominant kernel (the LDL^T launch): one HIP event pair per solve on the handle's stream
  int prof_on = 0;
  hipEvent_t prof_ev[2] = {nullptr, nullptr};
  double prof_sum_ms = 0; long long prof_n = 0; int prof_n_unknowns = 0;
  long long prof_solves = 0;       // solves since profiling was switched on / reset: every fourth one carries the event pair
  // lba_solve_async: the library-owned "LocalMapping" thread of this handle
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<bool> quit{false};
  // 0 = idle, 1 = a job is waiting for the worker, 2 = the worker is solving.  Changed under `mu` (the condition variable
  // stays the fallback), but both sides first SPIN on it: a futex wake-up costs 10-60 us (more from a deep C-state), and in
  // steady state the worker gets its next keyframe tens of microseconds after it delivered the last one.
  std::atomic<int> job_state{0};
  const lba_problem* job_p = nullptr; StopRef job_stop; lba_result* job_r = nullptr;
  int job_status = ORBG_OK;
  double job_ms = 0;
};

extern "C" int lba_create(int device, int cap_poses, int cap_points, int cap_edges, lba_handle** out) {
  if (!out || cap_poses < 0 || cap_points < 0 || cap_edges < 0) return ORBG_BAD_ARG;
  int rc = select_device(device);
  if (rc) return rc;
  lba_handle* h = new lba_handle();
  h->device = device;
  h->sw = LbaSwitches::from_env();
  if (orbg::create_stream(&h->stream, "lba") != hipSuccess) { delete h; return ORBG_HIP_ERROR; }
  if ((rc = h->rec.reserve(4)) || (rc = h->d_ok.reserve(4))) { delete h; return rc; }
  memset(h->rec.h, 0, 4 * sizeof(HostRec));
  (void)cap_poses; (void)cap_points; (void)cap_edges;   // buffers grow on first use and are kept
  *out = h;
  return ORBG_OK;
}

extern "C" int lba_set_stream(lba_handle* h, void* hip_stream) {
  if (!h) return ORBG_BAD_ARG;
  if (h->job_state.load() != 0) return ORBG_BAD_ARG;        // a solve is in flight on the worker
  int rc = select_device(h->device);
  if (rc) return rc;
  return orbg::swap_stream(&h->stream, &h->ext_stream, hip_stream, "lba");
}

extern "C" int lba_destroy(lba_handle* h) {
  if (!h) return ORBG_BAD_ARG;
  if (h->worker.joinable()) {
    { std::lock_guard<std::mutex> lk(h->mu); h->quit = true; }
    h->cv.notify_all();
    h->worker.join();
  }
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  h->d_edges.release(); h->edges_pin.release(); h->d_items_dev.release(); h->d_pair_count.release(); h->d_poses[0].release(); h->d_poses[1].release(); h->d_points[0].release(); h->d_points[1].release(); 
  h->d_err.release(); h->d_chi2.release(); h->d_partial.release(); h->d_EB.release(); h->d_Hll.release(); h->d_bl.release();
  h->d_Hpp.release(); h->d_bp.release(); h->d_S.release(); h->d_bs.release(); h->d_x.release(); h->d_St.release(); h->d_wfac.release(); h->d_xscr.release(); h->d_xflags.release(); h->ldlt_x.scr = nullptr; h->ldlt_x.flags = nullptr;
  h->d_EB2.release(); h->d_Hll2.release(); h->d_bl2.release(); h->d_Hpp2.release(); h->d_bp2.release(); h->d_lambda0.release();
  h->d_pose_col.release(); h->d_point_col.release(); h->d_pt_start.release(); h->d_pt_edges.release(); h->d_ps_start.release();
  h->d_ps_edges.release(); h->d_pf_start.release(); h->d_pf_edges.release(); h->d_pf_col.release(); h->d_pair_i1.release();
  h->d_pair_i2.release(); h->d_pair_start.release(); h->d_ok.release(); h->d_items.release(); h->rec.release(); h->up_h.release(); h->dl_h.release(); h->up_d.release(); h->d_flags.release(); h->d_scale_partial.release(); h->d_ticket.release(); h->sig.release();
  for (auto& e : h->prof_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
  if (!h->ext_stream) orbg::release_stream(h->stream);
  delete h;
  return ORBG_OK;
}

// ORBG_TRACE=1: average host-side time of the phases of lba_solve_h, printed when the library unloads
struct TraceAcc {
  double t[8] = {0}; long n = 0; const char* name;
  explicit TraceAcc(const char* nm) : name(nm) {}
  ~TraceAcc() {
    if (n && getenv("ORBG_TRACE")) {
      fprintf(stderr, "[orbgpu trace] %s n=%ld:", name, n);
      for (int i = 0; i < 8; i++) fprintf(stderr, " %.1f", t[i] / n * 1e6);
      fprintf(stderr, " us\n");
    }
  }
};
static inline double now_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

template <typename T>
static int upload(DevBuf<T>& b, const std::vector<T>& v, hipStream_t st) {
  int rc = b.reserve(std::max<size_t>(v.size(), 1));
  if (rc) return rc;
  if (!v.empty()) ORBG_HIP(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
  return ORBG_OK;
}

// Upload of a piece of the pinned arena by a kernel of our own: every thread moves 16 bytes, all reads over PCIe are in flight
// at once (~4 us for 150 KB).  The runtime's hipMemcpyAsync runs a blit kernel that takes ~26 us for the same bytes, three
// times per solve, twice on the critical path of the first LM iteration.
__global__ __launch_bounds__(256) void k_upload16(const uint4* __restrict__ src, uint4* __restrict__ dst, unsigned n16) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}
static int upload_arena(lba_handle* h, size_t off0, size_t off1, hipStream_t st) {
  if (off1 <= off0) return ORBG_OK;
  const unsigned n16 = (unsigned)((off1 - off0 + 15) / 16);           // offsets are multiples of 64, the arena has 64 bytes of slack
  hipLaunchKernelGGL(k_upload16, dim3((n16 + 255) / 256), dim3(256), 0, st, reinterpret_cast<const uint4*>(h->up_h.d + off0),
                     reinterpret_cast<uint4*>(h->up_d.p + off0), n16);
  ORBG_HIP(hipGetLastError());
  return ORBG_OK;
}

// internal: k_ldlt_xcd reported kOkTimedOut during this attempt (never leaves the library)
constexpr int kRcLdltTimedOut = -70001;
static int lba_solve_attempt(lba_handle* h, const lba_problem* p, StopRef stop_ref, lba_result* r);
template <class CamT>
static int lba_solve_attempt_t(lba_handle* h, const lba_problem* p, StopRef stop_ref, lba_result* r, const CamT& cam);
// A launch of the eight-workgroup LDL^T whose participants were not all placed in time says so (ldlt_xcd.hpp: kOkTimedOut) instead of
// posing as a non-positive-definite system, which the LM loop would answer with a rejected step and another trajectory than the
// reference's.  The window is then solved again from the caller's (untouched) problem with that kernel switched off for this handle
// -- the ORBG_LDLT_XCD=0 path, parity-tested like the default -- and the event is counted (lba_get_watchdog_count).
static int lba_solve_impl(lba_handle* h, const lba_problem* p, StopRef stop_ref, lba_result* r) {
  int rc = lba_solve_attempt(h, p, stop_ref, r);
  if (rc != kRcLdltTimedOut) return rc;
  h->xcd_timeouts++;
  h->sw.ldlt_xcd = 0;
  if (hipStreamSynchronize(h->stream) != hipSuccess) return ORBG_HIP_ERROR;
  rc = lba_solve_attempt(h, p, stop_ref, r);
  return rc == kRcLdltTimedOut ? ORBG_HIP_ERROR : rc;
}
// The kernels that evaluate edges exist twice: for the five pinhole scalars (every BASELINE configuration; unchanged code) and for
// a problem that carries a camera rig (fisheye models, the right camera's *ToBody edges).
static int lba_solve_attempt(lba_handle* h, const lba_problem* p, StopRef stop_ref, lba_result* r) {
  if (!h || !p) return ORBG_BAD_ARG;
  const Cam cam{p->fx, p->fy, p->cx, p->cy, p->bf, p->bf};
  if (!p->rig) return lba_solve_attempt_t(h, p, stop_ref, r, cam);
  CamRig g;
  if (!cam_rig_from(*p->rig, cam, &g)) return ORBG_BAD_ARG;
  return lba_solve_attempt_t(h, p, stop_ref, r, g);
}
template <class CamT>
static int lba_solve_attempt_t(lba_handle* h, const lba_problem* p, StopRef stop_ref, lba_result* r, const CamT& cam) {
  if (!h) return ORBG_BAD_ARG;
  const LbaSwitches& sw = h->sw;
  if (!h || !p || !r || p->n_poses < 0 || p->n_points < 0 || p->n_edges < 0) return ORBG_BAD_ARG;
  if (!r->poses || !r->points) return ORBG_BAD_ARG;
  int rc = select_device(h->device);
  if (rc) return rc;
  const int NP = p->n_poses, NX = p->n_points, NE = p->n_edges;
  // > 0: stop (the reference's bool); < 0: stop once that many LM trials have been evaluated (deterministic test hook, orbgpu.h)
  // INT32_MIN: raised right after the check that precedes optimize() (the -k form with k = 0)
  int trials_done = 0;
  bool past_precheck = false;
  auto terminate = [&]() {
    if (stop_ref.u8) return *stop_ref.u8 != 0;
    if (!stop_ref.i32) return false;
    const int v = *stop_ref.i32;
    if (v == INT32_MIN) return past_precheck;
    return v > 0 || (v < 0 && trials_done >= -v);
  };
  r->status = LBA_APPLIED; r->iters_round1 = r->iters_round2 = 0; r->n_outliers = 0; r->trace_len = 0;
  r->chi2_initial = r->chi2_final = 0;
  if (terminate()) {                                   // S/Optimizer.cc:2127-2129
    r->status = LBA_ABORTED_BEFORE_OPT;
    memcpy(r->poses, p->poses, sizeof(float) * 16 * (size_t)NP);
    memcpy(r->points, p->points, sizeof(float) * 3 * (size_t)NX);
    for (int k = 0; k < NE; k++) {
      if (r->edge_chi2) r->edge_chi2[k] = 0;
      if (r->edge_depth_pos) r->edge_depth_pos[k] = 1;
      if (r->edge_outlier) r->edge_outlier[k] = 0;
    }
    return ORBG_OK;
  }
  past_precheck = true;
  static TraceAcc tr("lba_solve_h structure (before the first launch) / upload submit / LM loop incl. pair items / export+wait / write-back / "
                     "of the structure: edge pass + layout / CSR lists / of the LM loop: pair items + symbolic + upload");
  const double t_a = now_s();
  hipStream_t st = h->stream;
  // ---- structure (the analogue of BlockSolver::buildStructure, G/core/block_solver.hpp:143-295), host side.
  // Every array the kernels need is built IN PLACE inside one pinned block and goes to the device with ONE copy.
  std::vector<int>& pose_deg = h->s_pose_deg; std::vector<int>& point_deg = h->s_point_deg;
  std::vector<int>& pf_raw = h->s_f3;                      // per point: observations from poses that are not fixed
  pose_deg.assign(NP, 0); point_deg.assign(NX, 0); pf_raw.assign(NX, 0);
  // ONE pass over the caller's edges (360 KB at C2, cold): copy into pinned memory, validate, count degrees; the copy
  // goes to the device at once and every later pass reads the warm pinned copy
  if ((rc = h->edges_pin.reserve(std::max(NE, 1))) || (rc = h->d_edges.reserve(std::max(NE, 1)))) return rc;
  lba_edge* const edges = h->edges_pin.h;
  // Optimizer::LocalBundleAdjustment adds the edges landmark by landmark (S/Optimizer.cc:2007-2124): consecutive edges increment the
  // SAME landmark's counters, and a read-modify-write of one word per edge is a chain of store-to-load forwards.  The landmark
  // counters therefore rotate over four copies (k & 3: a run of up to four edges of one
  // landmark touches four different words), both counts packed in one word (edges | edges of free poses << 16; fewer than 65536 edges);
  // `runs` counts the changes of landmark along the list: equal to the number of observed landmarks <=> every landmark's edges are
  // consecutive, which the list pass below exploits.
  int runs = 0;
  {
    unsigned prev_pt = ~0u;
    if (NE < 65536) {
      std::vector<unsigned>& cnt4 = h->s_cnt4;
      cnt4.assign(4 * (size_t)NX, 0u);
      unsigned* const c4 = cnt4.data();
      int* const pd = pose_deg.data();
      const uint8_t* const fixed = p->pose_fixed;
      for (int k = 0; k < NE; k++) {
        const lba_edge e = p->edges[k];
        edges[k] = e;
        const unsigned ep = (unsigned)e.pose, ex = (unsigned)e.point;
        if (ep >= (unsigned)NP || ex >= (unsigned)NX) return ORBG_BAD_ARG;
        pd[ep]++;
        c4[(size_t)(k & 3) * NX + ex] += 1u + ((fixed[ep] ? 0u : 1u) << 16);
        runs += ex != prev_pt;
        prev_pt = ex;
      }
      int* const pdeg = point_deg.data(); int* const pfr = pf_raw.data();
      for (int i = 0; i < NX; i++) {
        const unsigned sum = c4[i] + c4[(size_t)NX + i] + c4[2 * (size_t)NX + i] + c4[3 * (size_t)NX + i];
        pdeg[i] = (int)(sum & 0xFFFFu); pfr[i] = (int)(sum >> 16);
      }
    } else {
      for (int k = 0; k < NE; k++) {
        const lba_edge e = p->edges[k];
        edges[k] = e;
        const unsigned ep = (unsigned)e.pose, ex = (unsigned)e.point;
        if (ep >= (unsigned)NP || ex >= (unsigned)NX) return ORBG_BAD_ARG;
        pose_deg[ep]++; point_deg[ex]++;
        pf_raw[ex] += p->pose_fixed[ep] ? 0 : 1;
        runs += ex != prev_pt;
        prev_pt = ex;
      }
    }
  }
  // Camera rigs: the second, third .. edge of a run of edges between the same keyframe and landmark (left and right camera, adjacent
  // in the reference's creation order) is left out of the per-landmark lists of free observations that the Schur complement and the
  // back-substitution walk -- the run's first edge carries the Hpl block of the vertex pair (linearize_block).
  constexpr bool kRig = std::is_same<CamT, CamRig>::value;
  std::vector<uint8_t> rig_secondary;
  int n_right_edges = 0;           // EdgeSE3ProjectXYZToBody edges: counted among the outliers, NOT in the 50 % rule's denominator (:2256)
  if constexpr (kRig) {
    if (cam.has_right)
      for (int k = 0; k < NE; k++) n_right_edges += ur_is_right(edges[k].ur) ? 1 : 0;
    rig_secondary.assign((size_t)std::max(NE, 1), 0);
    for (int k = 1; k < NE; k++)
      if (edges[k].pose == edges[k - 1].pose && edges[k].point == edges[k - 1].point) {
        rig_secondary[k] = 1;
        if (!p->pose_fixed[edges[k].pose]) pf_raw[edges[k].point]--;
      }
  }
  if (NE > 0) {
    // (k_upload16: the runtime's blit kernel takes ~50 us for these 190 KB)
    const unsigned n16 = (unsigned)((sizeof(lba_edge) * (size_t)NE + 15) / 16);
    hipLaunchKernelGGL(k_upload16, dim3((n16 + 255) / 256), dim3(256), 0, st, reinterpret_cast<const uint4*>(h->edges_pin.d),
                       reinterpret_cast<uint4*>(h->d_edges.p), n16);
  }
  std::vector<int>& pose_col_v = h->s_pose_col; std::vector<int>& point_col_v = h->s_point_col;
  pose_col_v.assign(NP, -1); point_col_v.assign(NX, -1);
  int nP = 0, nL = 0;
  for (int i = 0; i < NP; i++) if (!p->pose_fixed[i] && pose_deg[i] > 0) pose_col_v[i] = nP++;
  for (int i = 0; i < NX; i++) if (point_deg[i] > 0) point_col_v[i] = nL++;
  // free-pose degree of every active point (-> number of (pose pair, landmark) items), edges per active point, edges per
  // free pose: all of them follow from the degrees counted in the pass above (a pose with an edge has pose_deg > 0, so
  // "not fixed" is "free" there)
  std::vector<int>& pf_deg = h->s_pf_deg; std::vector<int>& pt_cnt = h->s_f1; std::vector<int>& ps_cnt = h->s_f2;
  pf_deg.assign(nL + 1, 0); pt_cnt.assign(nL + 1, 0); ps_cnt.assign(nP + 1, 0);
  int n_free_edges = 0;
  for (int i = 0; i < NX; i++) { const int lc = point_col_v[i]; if (lc >= 0) { pt_cnt[lc] = point_deg[i]; pf_deg[lc] = pf_raw[i]; } }
  for (int i = 0; i < NP; i++) { const int pc = pose_col_v[i]; if (pc >= 0) { ps_cnt[pc] = pose_deg[i]; n_free_edges += pose_deg[i]; } }
  size_t n_items = 0;
  for (int l = 0; l < nL; l++) n_items += (size_t)pf_deg[l] * (pf_deg[l] + 1) / 2;
  const int n_pairs_all = nP * (nP + 1) / 2;
  // arena layout
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 63) & ~(size_t)63; return o; };
  const size_t o_poses = take(sizeof(PoseQ) * (size_t)NP), o_points = take(24 * (size_t)NX);
  // (what the host fills first; then the lists the device may fill itself: see off_a below)
  const size_t o_pose_col = take(4 * (size_t)NP), o_point_col = take(4 * (size_t)NX), o_pt_start = take(4 * ((size_t)nL + 1));
  const size_t o_ps_start = take(4 * ((size_t)nP + 1)), o_pf_start = take(4 * ((size_t)nL + 1));
  const size_t o_cur_pt = take(4 * (size_t)nL), o_cur_ps = take(4 * (size_t)nP), o_cur_pf = take(4 * (size_t)nL);   // fill cursors (k_csr_fill)
  const size_t o_pt_edges = take(4 * (size_t)NE), o_ps_edges = take(4 * (size_t)n_free_edges);
  const size_t o_pf_edges = take(4 * (size_t)n_free_edges), o_pf_col = take(4 * (size_t)n_free_edges);
  // pair items on the device when the pose masks fit one word and the fixed-capacity segments stay small
  // (a pair's items are landmarks both poses observe: never more than the edges of either pose -- a far smaller segment than
  // one entry per landmark, which kept the 50-keyframe window of C4 on the host path)
  int max_pose_edges = 1;
  for (int i = 0; i < nP; i++) max_pose_edges = std::max(max_pose_edges, ps_cnt[i]);
  const int item_cap = std::min(std::max(nL, 1), max_pose_edges);
  const bool dev_items = nP >= 1 && nP <= 64 && (size_t)n_pairs_all * (size_t)item_cap * sizeof(PairItem) <= ((size_t)64 << 20);
  // (the lists of free observations per landmark, still in edge order, go along when the device sorts them: k_prep / k_errlin_prep)
  const bool dev_lists = dev_items && nP >= 1 && ldltm::supports(6 * nP);
  // ... and the device fills the lists itself (k_csr_fill / k_csr_sort) when a pose's list fits the sorting workgroup
  // (measured: a wash at C2 -- 7.7 + 15.2 us of kernels for a 29 us host pass -- and -20 us at C4: used from 16 k edges on)
  const bool dev_csr = dev_lists && NE > 0 && nL > 0 && max_pose_edges <= kCsrPoseCap && NE >= 16384 && !kRig;
  const size_t o_lm_mask = take(8 * (size_t)nL);
  const size_t o_pair_i1 = take(4 * (size_t)n_pairs_all), o_pair_i2 = take(4 * (size_t)n_pairs_all), o_pair_start = take(4 * ((size_t)n_pairs_all + 1));
  const size_t o_items = take(dev_items ? 0 : sizeof(PairItem) * n_items);
  if ((rc = h->up_h.reserve(off + 64)) || (rc = h->up_d.reserve(off + 64))) return rc;
  uint8_t* H = h->up_h.h;
  PoseQ* poses = reinterpret_cast<PoseQ*>(H + o_poses);
  double* points = reinterpret_cast<double*>(H + o_points);
  int* pose_col = reinterpret_cast<int*>(H + o_pose_col); int* point_col = reinterpret_cast<int*>(H + o_point_col);
  int* pt_start = reinterpret_cast<int*>(H + o_pt_start); int* pt_edges = reinterpret_cast<int*>(H + o_pt_edges);
  int* ps_start = reinterpret_cast<int*>(H + o_ps_start); int* ps_edges = reinterpret_cast<int*>(H + o_ps_edges);
  int* pf_start = reinterpret_cast<int*>(H + o_pf_start); int* pf_edges = reinterpret_cast<int*>(H + o_pf_edges);
  int* pf_col = reinterpret_cast<int*>(H + o_pf_col);
  int* pair_i1 = reinterpret_cast<int*>(H + o_pair_i1); int* pair_i2 = reinterpret_cast<int*>(H + o_pair_i2);
  int* pair_start = reinterpret_cast<int*>(H + o_pair_start);
  unsigned long long* lm_mask = reinterpret_cast<unsigned long long*>(H + o_lm_mask);
  PairItem* items = reinterpret_cast<PairItem*>(H + o_items);
  const double t_s1 = now_s();
  memcpy(pose_col, pose_col_v.data(), 4 * (size_t)NP);
  memcpy(point_col, point_col_v.data(), 4 * (size_t)NX);
  // CSR: edges per active point (creation order); per free pose; per active point restricted to free poses (sorted by col)
  pt_start[0] = 0; pf_start[0] = 0; ps_start[0] = 0;
  for (int i = 0; i < nL; i++) { pt_start[i + 1] = pt_start[i] + pt_cnt[i]; pf_start[i + 1] = pf_start[i] + pf_deg[i]; }
  for (int i = 0; i < nP; i++) ps_start[i + 1] = ps_start[i] + ps_cnt[i];
  if (dev_csr) {
    memcpy(H + o_cur_pt, pt_start, 4 * (size_t)nL); memcpy(H + o_cur_ps, ps_start, 4 * (size_t)nP); memcpy(H + o_cur_pf, pf_start, 4 * (size_t)nL);
  } else {
    std::vector<int>& f1 = h->s_f1; std::vector<int>& f2 = h->s_f2; std::vector<int>& f3 = h->s_f3;
    if (runs == nL && !kRig) {
      // every landmark's edges are consecutive (the reference's order): a landmark's list positions are carried in registers along
      // its run instead of in per-landmark cursors (the same store-to-load chains as above), and the "pose is free" test selects
      // the destination (a junk word for edges of fixed poses) instead of branching on a one-in-three condition
      f2.assign(ps_start, ps_start + nP);
      f2.push_back(0);                                       // [nP]: cursor of the edges of fixed poses, into junk
      std::vector<int>& junk = h->s_junk;
      if ((int)junk.size() < NE + 1) junk.resize((size_t)NE + 1);
      int* const ps_base[2] = {junk.data(), ps_edges};
      int* const f2p = f2.data();
      int prev = -1, pt_pos = 0, pf_pos = 0;
      for (int k = 0; k < NE; k++) {
        const int ex = edges[k].point, lc = point_col[ex], pc = pose_col[edges[k].pose];
        const bool ch = ex != prev;
        prev = ex;
        pt_pos = ch ? pt_start[lc] : pt_pos;
        pf_pos = ch ? pf_start[lc] : pf_pos;
        pt_edges[pt_pos++] = k;
        const int fr = pc >= 0;
        ps_base[fr][f2p[fr ? pc : nP]++] = k;
        int* const pf_dst = fr ? pf_edges + pf_pos : junk.data() + NE;
        *pf_dst = k;
        pf_pos += fr;
      }
    } else {
      f1.assign(pt_start, pt_start + nL); f2.assign(ps_start, ps_start + nP); f3.assign(pf_start, pf_start + nL);
      for (int k = 0; k < NE; k++) {
        const int lc = point_col[edges[k].point], pc = pose_col[edges[k].pose];
        pt_edges[f1[lc]++] = k;
        if (pc >= 0) {
          ps_edges[f2[pc]++] = k;
          if (!kRig || !rig_secondary[k]) pf_edges[f3[lc]++] = k;
        }
      }
    }
  }
  const double t_s2 = now_s();
  // ---- initial state: Converter::toSE3Quat (S/Converter.cc:33-43)
  for (int i = 0; i < NP; i++) {
    const float* T = p->poses + 16 * (size_t)i;
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    quat_from_R(R, poses[i].q);
    quat_normalize(poses[i].q);
    poses[i].t[0] = T[3]; poses[i].t[1] = T[7]; poses[i].t[2] = T[11];
  }
  for (size_t i = 0; i < 3 * (size_t)NX; i++) points[i] = p->points[i];

  const int n = 6 * nP;
  const int n_blocks_e = (NE + 255) / 256;
  const double t_b = now_s();
  // part A of the arena (edges, state, the CSR lists the error / linearisation kernels read) goes up now; the pair items
  // the Schur kernel needs are built while the device already computes the first residuals and Jacobians
  // part A: host-filled arrays; + the point / pose lists when the host fills them; + the unsorted free-observation lists when
  // the device only sorts
  const size_t off_a = dev_csr ? o_pt_edges : dev_lists ? o_pf_col : o_pf_edges;
  if ((rc = upload_arena(h, 0, off_a, st))) return rc;
  struct {
    const lba_edge* edges; const int *pose_col, *point_col, *pt_start, *pt_edges, *ps_start, *ps_edges, *pf_start, *pf_edges, *pf_col,
        *pair_i1, *pair_i2, *pair_start;
    const PairItem* items;
  } D;
  {
    const uint8_t* B = h->up_d.p;
    D.edges = h->d_edges.p;
    D.pose_col = reinterpret_cast<const int*>(B + o_pose_col); D.point_col = reinterpret_cast<const int*>(B + o_point_col);
    D.pt_start = reinterpret_cast<const int*>(B + o_pt_start); D.pt_edges = reinterpret_cast<const int*>(B + o_pt_edges);
    D.ps_start = reinterpret_cast<const int*>(B + o_ps_start); D.ps_edges = reinterpret_cast<const int*>(B + o_ps_edges);
    D.pf_start = reinterpret_cast<const int*>(B + o_pf_start); D.pf_edges = reinterpret_cast<const int*>(B + o_pf_edges);
    D.pf_col = reinterpret_cast<const int*>(B + o_pf_col);
    D.pair_i1 = reinterpret_cast<const int*>(B + o_pair_i1); D.pair_i2 = reinterpret_cast<const int*>(B + o_pair_i2);
    if (dev_lists) { D.pair_i1 = nullptr; D.pair_i2 = nullptr; }
    D.pair_start = dev_items ? (const int*)nullptr : reinterpret_cast<const int*>(B + o_pair_start);
    D.items = dev_items ? (const PairItem*)nullptr : reinterpret_cast<const PairItem*>(B + o_items);
  }
  if (dev_items) {
    if ((rc = h->d_items_dev.reserve((size_t)n_pairs_all * item_cap)) || (rc = h->d_pair_count.reserve(std::max(n_pairs_all, 1)))) return rc;
    D.items = h->d_items_dev.p;
  }
  // the three state buffers (current / trial estimate / the trial after it, written on speculation by the fused solve + update
  // launch): buffer 0 IS the uploaded state inside the arena (part A is not written again during the solve; two device-to-device
  // copies of a few KB cost the stream ~10 us each before the first residuals), buffers 1 and 2 allocations of their own.
  // They rotate: trial = (cur + 1) % 3, and an accepted trial becomes the current estimate.
  if ((rc = h->d_poses[1].reserve(std::max(NP, 1))) || (rc = h->d_points[1].reserve(std::max<size_t>(3 * (size_t)NX, 1))) ||
      (rc = h->d_poses[0].reserve(std::max(NP, 1))) || (rc = h->d_points[0].reserve(std::max<size_t>(3 * (size_t)NX, 1))) ||
      (rc = h->d_err.reserve(std::max<size_t>(3 * (size_t)NE, 1))) || (rc = h->d_chi2.reserve(std::max(NE, 1))) ||
      (rc = h->d_partial.reserve(std::max(n_blocks_e, 1))) || (rc = h->d_EB.reserve(std::max<size_t>((size_t)NE * kEB, 1))) ||
      (rc = h->d_Hll.reserve(std::max<size_t>(6 * (size_t)nL, 1))) || (rc = h->d_bl.reserve(std::max<size_t>(3 * (size_t)nL, 1))) ||
      (rc = h->d_Hpp.reserve(std::max<size_t>(21 * (size_t)nP, 1))) || (rc = h->d_bp.reserve(std::max<size_t>(6 * (size_t)nP, 1))) ||
      (rc = h->d_EB2.reserve(std::max<size_t>((size_t)NE * kEB, 1))) || (rc = h->d_Hll2.reserve(std::max<size_t>(6 * (size_t)nL, 1))) ||
      (rc = h->d_bl2.reserve(std::max<size_t>(3 * (size_t)nL, 1))) || (rc = h->d_Hpp2.reserve(std::max<size_t>(21 * (size_t)nP, 1))) ||
      (rc = h->d_bp2.reserve(std::max<size_t>(6 * (size_t)nP, 1))) || (rc = h->d_lambda0.reserve(4)) ||
      (rc = h->d_S.reserve(std::max<size_t>((size_t)n * n, 1))) || (rc = h->d_bs.reserve(std::max(n, 1))) ||
      (rc = h->d_x.reserve(std::max<size_t>((size_t)n + 3 * (size_t)nL, 1))))
    return rc;
  PoseQ* const posesB[3] = {reinterpret_cast<PoseQ*>(h->up_d.p + o_poses), h->d_poses[1].p, h->d_poses[0].p};
  double* const pointsB[3] = {reinterpret_cast<double*>(h->up_d.p + o_points), h->d_points[1].p, h->d_points[0].p};

  const double t_c = now_s();
  Huber hb;
  hb.delta_mono = (float)std::sqrt(5.991); hb.dsqr_mono = hb.delta_mono * hb.delta_mono;          // S/Optimizer.cc:1991-1992
  hb.delta_stereo = (float)std::sqrt(7.815); hb.dsqr_stereo = hb.delta_stereo * hb.delta_stereo;
  // FP64 matrix-core LDL^T (ldlt_mfma.hpp): up to 50 free poses
  const bool force_wide = sw.ldlt_wide;          // test switch: k_wide_* at any size
  const bool use_mfma = nP >= 1 && ldltm::supports(n) && !force_wide;
  // windows beyond the matrix-core kernels (more than 50 free poses): blocked LDL^T over many workgroups
  const bool use_wide = nP >= 1 && !use_mfma;
  if (use_wide && n > kWideMaxUnknowns) return ORBG_CAP_EXCEEDED;      // (k_wide_back's x lives in LDS)
  if (use_wide && (rc = h->d_wide.reserve(2 * (size_t)n + 32))) return rc;
  if (use_mfma) {
    if ((rc = h->d_St.reserve(ldltm::tile_image_doubles(n))) || (rc = h->d_wfac.reserve(ldltm::wglob_doubles(ldltm::make_geo(n))))) return rc;
  }
  // 21 .. 50 free poses: the same tile image, factored by eight workgroups of one XCD (ldlt_xcd.hpp)
  const bool use_xcd = use_mfma && sw.ldlt_xcd != 0 && ldltx::pays(n);
  if (use_xcd && !h->ldlt_x.scr) {
    if ((rc = h->d_xscr.reserve(ldltx::scratch_doubles())) || (rc = h->d_xflags.reserve(ldltx::kFlagWords))) return rc;
    ORBG_HIP(hipMemsetAsync(h->d_xflags.p, 0, ldltx::kFlagWords * sizeof(unsigned), st));
    // (the G / D^-1 pair region: a fresh nonce per bind already makes stale pairs of a recycled allocation fail their test; zero all the same)
    ORBG_HIP(hipMemsetAsync(h->d_xscr.p + ldltx::kGbOff, 0, (ldltx::kWOff - ldltx::kGbOff) * sizeof(double), st));
    h->ldlt_x.bind(h->d_xscr.p, h->d_xflags.p);
    static std::atomic<int> n_users{0};              // users of one process on different XCDs; processes sharing a GPU differ by pid
    h->ldlt_x.pick = ((int)getpid() + n_users.fetch_add(1)) & 7;
  }
  int cur = 0;   // index of the buffer holding the current estimate
  // k_update's workgroup size: the kernel is a chain of dependent memory round trips per landmark; small workgroups spread the same
  // wavefronts over more compute units (measured at C2: 256 threads 0.499 ms per solve, 128: 0.488, 64: 0.484)
  constexpr int upd_threads = 64;
  const int n_blocks_u = (NP + NX + upd_threads - 1) / upd_threads;
  if ((rc = h->d_scale_partial.reserve(std::max(n_blocks_u, 1)))) return rc;
  if (!h->d_ticket.p) {
    if ((rc = h->d_ticket.reserve(4))) return rc;
    ORBG_HIP(hipMemsetAsync(h->d_ticket.p, 0, 4 * sizeof(unsigned), st));
  }
  // final_mode: the last block also publishes {robust chi2, computeScale(), solver flag} to the host record
  auto launch_errors = [&](int buf, int final_mode) {
    if (NE > 0)
      hipLaunchKernelGGL(k_errors<CamT>, dim3(n_blocks_e), dim3(256), 0, st, NE, D.edges, posesB[buf], pointsB[buf], cam, hb,
                         h->d_err.p, h->d_chi2.p, h->d_partial.p, final_mode, h->d_ticket.p, h->d_scale_partial.p, n_blocks_u,
                         h->d_ok.p, h->rec.d, final_mode ? ++h->rec_seq : 0u);
  };
  // two sets of linearisation outputs: while the host waits for the verdict on a trial, the linearisation of the TRIAL
  // state (= the next iteration's, if the trial is accepted -- the usual case) is already running into the other set
  double* const EBs[2] = {h->d_EB.p, h->d_EB2.p};
  double* const Hlls[2] = {h->d_Hll.p, h->d_Hll2.p};
  double* const bls[2] = {h->d_bl.p, h->d_bl2.p};
  double* const Hpps[2] = {h->d_Hpp.p, h->d_Hpp2.p};
  double* const bps[2] = {h->d_bp.p, h->d_bp2.p};
  int ls = 0;                      // linearisation set of the current iteration
  bool spec_ready = false;         // set ls^1 holds the linearisation of the current estimate
  auto launch_linearise = [&](int buf, int set) {
    if (NE > 0 || nP > 0)
      hipLaunchKernelGGL(k_lin_all<CamT>, dim3(nP + (NE > 0 ? n_blocks_e : 0)), dim3(256), 0, st, nP, NE, D.edges, posesB[buf],
                         pointsB[buf], cam, hb, h->d_err.p, h->d_chi2.p, D.pose_col, D.point_col, EBs[set], D.ps_start,
                         D.ps_edges, Hpps[set], bps[set]);
    if (nL > 0)
      hipLaunchKernelGGL(k_reduce_points, dim3((nL + 255) / 256), dim3(256), 0, st, nL, D.pt_start, D.pt_edges, EBs[set],
                         Hlls[set], bls[set]);
  };
  auto finish = [&](double lambda, int want_scale, int want_maxdiag, bool with_ok) -> int {
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, st, n_blocks_e, h->d_partial.p, nP, nL, h->d_x.p, bps[ls], bls[ls], Hpps[ls],
                       Hlls[ls], lambda, with_ok ? h->d_ok.p : (int*)nullptr, want_scale, want_maxdiag, h->rec.d, 0.0, (double*)nullptr);
    ORBG_HIP(hipGetLastError());
    ORBG_HIP(hipStreamSynchronize(st));
    return with_ok && h->rec.h->ok == ldltx::kOkTimedOut ? kRcLdltTimedOut : ORBG_OK;
  };
  auto poll_record = [&]() -> int {
    // the last workgroup of k_errors publishes the record and then its sequence number: spin on that word (the
    // runtime's completion path costs ~10 us per LM trial); fall back to a stream sync if it does not arrive
    volatile unsigned* w = &h->rec.h->seq;
    const unsigned want = h->rec_seq;
    bool got = false;
    if (orbg::poll_allowed()) {              // (the policy of the thread that runs the solve: caller or local-BA worker)
      timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
      const unsigned long long tg = rec_tag(want);
      const volatile HostRec* hr = h->rec.h;
      for (unsigned spins = 0; !got; spins++) {
        if (*w == want) {
          // ... and the three pairs of THIS record have arrived (they and seq are independent stores)
          const double c = hr->chi2, sc = hr->scale; const int okv = hr->ok;
          unsigned long long bc, bs; memcpy(&bc, &c, 8); memcpy(&bs, &sc, 8);
          if ((bc ^ hr->c_chi2) == tg && (bs ^ hr->c_scale) == tg && (((unsigned long long)(unsigned)okv) ^ hr->c_ok) == tg) { got = true; break; }
        }
        if ((spins & 0xFFFF) == 0xFFFF) {
          timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
          if ((t1.tv_sec - t0.tv_sec) * 1000.0 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 > 50.0) break;
        }
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    if (!got) ORBG_HIP(hipStreamSynchronize(st));
    if (h->rec.h->ok == kOkPublishTimedOut) return ORBG_INTERNAL;      // (a fused linearisation's publisher never saw all edge workgroups arrive)
    // the eight-workgroup LDL^T gave up waiting for a participant: not an LM verdict -- the caller re-solves the window (lba_solve_impl)
    return h->rec.h->ok == ldltx::kOkTimedOut ? kRcLdltTimedOut : ORBG_OK;
  };

  // results block (one pinned allocation the export kernel writes straight into) and its launcher: also used speculatively
  size_t doff = 0;
  auto dtake = [&](size_t bytes) { const size_t o = doff; doff = (doff + bytes + 63) & ~(size_t)63; return o; };
  const size_t d_poses_o = dtake(sizeof(PoseQ) * (size_t)NP), d_points_o = dtake(24 * (size_t)NX), d_flags_o = dtake((size_t)NE);
  const size_t d_chi_o = dtake(r->edge_chi2 ? 8 * (size_t)NE : 0);
  if ((rc = h->dl_h.reserve(doff + 64))) return rc;
  auto launch_export = [&](int buf) {
    const int n_thr = std::max(std::max(NE, NP), 3 * NX);
    if (n_thr > 0)
      hipLaunchKernelGGL(k_export<CamT>, dim3((n_thr + 255) / 256), dim3(256), 0, st, NE, NP, NX, D.edges, posesB[buf], pointsB[buf], cam,
                         h->d_chi2.p, h->dl_h.d + d_flags_o, r->edge_chi2 ? reinterpret_cast<double*>(h->dl_h.d + d_chi_o) : (double*)nullptr,
                         reinterpret_cast<PoseQ*>(h->dl_h.d + d_poses_o), reinterpret_cast<double*>(h->dl_h.d + d_points_o));
  };
  // Speculation beyond the next linearisation: `version` counts LM trials; work launched for "this trial gets accepted and
  // ends the round / the solve" is valid only if no later trial ran and the trial's buffer became the current one.
  int version = 0;
  int fin_version = -1;            // k_finish (lambda init of the NEXT round) already ran on the speculative set
  int exp_version = -1, exp_buf = -1;   // k_export of the trial state already in flight (completion word posted)

  int solve_version = -1;          // Schur complement + LDL^T of the NEXT trial already launched (speculatively) at this trial number
  bool prof_pending = false;       // an event pair brackets one LDL^T launch of this call
  const bool prof_this_solve = h->prof_on && (h->prof_solves++ & 3) == 0;
  // in_buf / out_buf: with the fused launch the trial state posesB[out_buf] = posesB[in_buf] (+) x is written by the same launch
  auto launch_solve = [&](int set_, double lam_, const double* lamp_, int in_buf, int out_buf) -> int {
    if (nP > 0) {
      // (more pose pairs than compute units: the two-pass form, three workgroups per compute unit)
      const auto schur_fn = n_pairs_all > 256 ? k_schur<2> : k_schur<1>;
      hipLaunchKernelGGL(schur_fn, dim3(n_pairs_all), dim3(kSchurThreads), 0, st, nP, D.pair_i1, D.pair_i2, D.pair_start, D.items,
                         EBs[set_], Hlls[set_], bls[set_], Hpps[set_], bps[set_], lam_, h->d_S.p, h->d_bs.p, lamp_, item_cap,
                         dev_items ? h->d_pair_count.p : (const int*)nullptr, use_mfma ? h->d_St.p : (double*)nullptr);
      // (two event records and an elapsed-time query cost the solve ~8 us: one solve in four is enough for an average)
      const bool bracket = h->prof_on && !prof_pending && prof_this_solve;
      if (bracket) ORBG_HIP(hipEventRecord(h->prof_ev[0], st));
      if (use_xcd) {
        const bool one_short = h->sw.ldlt_xcd_short_once;
        h->sw.ldlt_xcd_short_once = false;
        ORBG_HIP(ldltx::launch(h->ldlt_x, n, h->d_St.p, h->d_x.p, h->d_ok.p, st, ldltx::kMaxP, sw.ldlt_xcd == 2, one_short));
      } else if (use_mfma) {
        ORBG_HIP(ldltm::launch(n, h->d_St.p, h->d_x.p, h->d_ok.p, h->d_wfac.p, st, &h->ldlt_attr));
      } else {
        ORBG_HIP(launch_ldlt_wide(n, h->d_S.p, h->d_bs.p, h->d_x.p, h->d_ok.p, h->d_wide.p, st));
      }
      if (bracket) { ORBG_HIP(hipEventRecord(h->prof_ev[1], st)); prof_pending = true; }
    } else {
      ORBG_HIP(hipMemsetAsync(h->d_ok.p, 0xFF, sizeof(int), st));   // nothing to solve: ok
    }
    return ORBG_OK;
  };
  double lambda = -1, ni = 2;
  int nBad = 0;
  bool first_chi = true;
  bool err_valid = false;          // d_err / d_chi2 hold the residuals of the CURRENT estimate
  double currentChi = 0;
  bool last_round = false;
  auto optimize = [&](int iterations, int* done_out) -> int {
    int done = 0;
    bool ok = true;
    for (int it = 0; it < iterations && !terminate() && ok; it++) {
      // computeActiveErrors (skipped when the residuals of the current estimate are already on the device:
      // recomputing them would reproduce the same bits) + buildSystem (skipped when the speculative set holds it)
      const bool used_spec = spec_ready;
      if (spec_ready) {
        ls ^= 1;
        spec_ready = false;
      } else {
        if (!err_valid) { launch_errors(cur, 0); err_valid = true; }
        launch_linearise(cur, ls);
      }
      int rc2;
      bool lambda_on_device = false;
      if (it == 0) {
        if (NE > 0) {
          // computeLambdaInit without a host round trip: k_finish leaves lambda in device memory for the first trial
          // (it may already have run, speculatively, behind the last trial of the previous round)
          if (!(used_spec && fin_version == version))
            hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, st, n_blocks_e, h->d_partial.p, nP, nL, h->d_x.p, bps[ls], bls[ls], Hpps[ls],
                               Hlls[ls], 0.0, (int*)nullptr, 0, 1, h->rec.d, p->lambda_init, h->d_lambda0.p);
          lambda_on_device = true;
        } else {
          if ((rc2 = finish(0.0, 0, 1, false))) return rc2;
          currentChi = h->rec.h->chi2;
          lambda = p->lambda_init > 0 ? p->lambda_init : 1e-5 * h->rec.h->maxdiag;
        }
        ni = 2; nBad = 0;
      }
      if (!lambda_on_device && first_chi) { r->chi2_initial = currentChi; first_chi = false; }
      double tempChi = currentChi;
      double iniChi = currentChi;
      double rho = 0;
      int qmax = 0;
      do {
        const int trial = (cur + 1) % 3;
        version++;
        const double* lam_p = lambda_on_device ? h->d_lambda0.p : (const double*)nullptr;
        // the solve of this trial may already be running: it was launched, with the lambda the device computed for the accepted
        // case, behind the previous trial's residual / linearisation kernel (with the fused launch: its update into `trial` too)
        if (!(solve_version == version - 1 && qmax == 0 && used_spec) && (rc2 = launch_solve(ls, lambda, lam_p, cur, trial))) return rc2;
        hipLaunchKernelGGL(k_update<upd_threads>, dim3(n_blocks_u), dim3(upd_threads), 0, st, NP, NX, nP, D.pose_col, D.point_col,
                           posesB[cur], pointsB[cur], h->d_x.p, D.pf_start, D.pf_edges, D.pf_col, EBs[ls],
                           Hlls[ls], bls[ls], lambda, posesB[trial], pointsB[trial], bps[ls], h->d_scale_partial.p, lam_p);
        bool speculated = false, fused_export = false;
        if (NE > 0) {
          // speculate on acceptance: linearise the trial state into the other set while the host waits for the verdict
          // (not after the very last iteration that can run)
          // ... nor when two iterations in a row barely improved chi2: a third one ends the round (nBad >= 3)
          const bool may_continue = !(last_round && (it + 1 >= iterations || nBad >= 2));
          if (may_continue) {
            // residuals + record + linearisation of the trial state in ONE launch
            const int set = ls ^ 1;
            const int n_blocks_l = errlin_tail_blocks(nL);   // the landmark reduction (quads) and the record's publisher ride in the same launch
            hipLaunchKernelGGL(k_errlin<CamT>, dim3(nP + n_blocks_e + n_blocks_l), dim3(256), 0, st, nP, NE, D.edges, posesB[trial],
                               pointsB[trial], cam, hb, h->d_err.p, h->d_chi2.p, D.pose_col, D.point_col, EBs[set], D.ps_start,
                               D.ps_edges, Hpps[set], bps[set], h->d_partial.p, h->d_ticket.p, h->d_scale_partial.p, n_blocks_u, h->d_ok.p,
                               h->rec.d, ++h->rec_seq, n_blocks_e, nL, D.pt_start, D.pt_edges, Hlls[set], bls[set],
                               LmIn{currentChi, lambda, lambda_on_device ? h->d_lambda0.p + 2 : (const double*)nullptr,
                                    lambda_on_device ? h->d_lambda0.p : (const double*)nullptr, h->d_lambda0.p + 1});
            speculated = true;
            // ... and, when the next trial belongs to the same round, its Schur complement + LDL^T with the lambda the device
            // has just computed for the accepted case: the host's verdict then arrives while they run
            if (it + 1 < iterations && nBad < 2 && nP > 0) {
              // (fused launch: the update of the trial AFTER this one, into the third buffer -- this trial's state stays intact
              // in case it is rejected, the current estimate in case it is not)
              if ((rc2 = launch_solve(set, 0.0, h->d_lambda0.p + 1, trial, (trial + 1) % 3))) return rc2;
              solve_version = version;
            }
          } else {
            // the last evaluation that can run in the last round goes together with the (speculative) export of its state
            fused_export = last_round && (it + 1 >= iterations || nBad >= 2) && !lambda_on_device;
            if (fused_export) {
              const int n_thr = std::max(std::max(NE, NP), 3 * NX);
              hipLaunchKernelGGL(k_errors_export<CamT>, dim3((n_thr + 255) / 256), dim3(256), 0, st, n_blocks_e, NE, D.edges, posesB[trial],
                                 pointsB[trial], cam, hb, h->d_err.p, h->d_chi2.p, h->d_partial.p, h->d_ticket.p, h->d_scale_partial.p,
                                 n_blocks_u, h->d_ok.p, h->rec.d, ++h->rec_seq, NP, NX, h->dl_h.d + d_flags_o,
                                 r->edge_chi2 ? reinterpret_cast<double*>(h->dl_h.d + d_chi_o) : (double*)nullptr,
                                 reinterpret_cast<PoseQ*>(h->dl_h.d + d_poses_o), reinterpret_cast<double*>(h->dl_h.d + d_points_o));
            } else {
              launch_errors(trial, 1);
            }
          }
          const bool round_may_end = it + 1 >= iterations || nBad >= 2;
          if (speculated && !last_round && round_may_end && !lambda_on_device) {
            // ... and if this trial ends the round, the next round's lambda init as well
            hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, st, n_blocks_e, h->d_partial.p, nP, nL, h->d_x.p, bps[ls ^ 1], bls[ls ^ 1],
                               Hpps[ls ^ 1], Hlls[ls ^ 1], 0.0, (int*)nullptr, 0, 1, h->rec.d, p->lambda_init, h->d_lambda0.p);
            fin_version = version;
          }
          if (last_round && round_may_end && !lambda_on_device) {
            // ... or, in the last round, the export of the trial state (dropped if the trial is rejected or the round goes on)
            if (!fused_export) launch_export(trial);
            if ((rc2 = h->sig.post(st))) return rc2;
            exp_version = version; exp_buf = trial;
          }
          ORBG_HIP(hipGetLastError());
          if ((rc2 = poll_record())) return rc2;
        } else if ((rc2 = finish(lambda, 1, 0, true))) {
          return rc2;
        }
        if (lambda_on_device) {
          // the round's initial chi2 / lambda, as the device computed them before this first trial
          currentChi = h->rec.h->chi2_init;
          lambda = p->lambda_init > 0 ? p->lambda_init : 1e-5 * h->rec.h->maxdiag;
          lambda_on_device = false;
          if (first_chi) { r->chi2_initial = currentChi; first_chi = false; }
          tempChi = currentChi; iniChi = currentChi;
        }
        const bool ok2 = h->rec.h->ok != 0;
        tempChi = h->rec.h->chi2;
        if (!ok2) tempChi = std::numeric_limits<double>::max();
        rho = currentChi - tempChi;
        double scale = h->rec.h->scale;
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && std::isfinite(tempChi)) {
          const double c3 = 2 * rho - 1;               // (2 rho - 1)^3 as two multiplications: publish_trial_record does the same
          double alpha = 1. - c3 * c3 * c3;            // arithmetic on the device for the speculative next solve
          alpha = std::min(alpha, 2. / 3.);
          const double scaleFactor = std::max(1. / 3., alpha);
          lambda *= scaleFactor;
          ni = 2;
          currentChi = tempChi;
          cur = trial;                                // discardTop(): keep the trial state
          err_valid = true;
          spec_ready = speculated;
        } else {
          lambda *= ni;
          ni *= 2;                                    // pop(): current buffer untouched
          err_valid = false;                          // d_err now belongs to the rejected trial
          spec_ready = false;
        }
        qmax++;
        trials_done++;
      } while (rho < 0 && qmax < 10 && !terminate());
      done++;
      r->chi2_final = currentChi;
      if (r->trace && r->trace_len < r->trace_cap) {
        r->trace[3 * r->trace_len] = lambda; r->trace[3 * r->trace_len + 1] = currentChi; r->trace[3 * r->trace_len + 2] = qmax;
        r->trace_len++;
      }
      if (qmax == 10 || rho == 0) { ok = false; continue; }
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      if (nBad >= 3) ok = false;
    }
    *done_out = done;
    return ORBG_OK;
  };

  // first residuals + linearisation are launched before the host has finished the structure
  const int n_zero = n + 3 * nL;
  // first iteration in two launches (k_errlin_prep, k_finish_items) where the observation lists are sorted on the device
  const bool first2 = dev_lists && NE > 0 && nL > 0 && !terminate();
  if (first2 && dev_csr) {
    const int set = ls ^ 1;
    const int n_blocks_l = (nL + 255) / 256;              // (k_csr_sort: one thread per landmark)
    const int n_tail = errlin_tail_blocks(nL);
    uint8_t* const B = h->up_d.p;
    hipLaunchKernelGGL(k_csr_fill, dim3(n_blocks_e), dim3(256), 0, st, NE, D.edges, D.pose_col, D.point_col, reinterpret_cast<int*>(B + o_cur_pt),
                       reinterpret_cast<int*>(B + o_cur_ps), reinterpret_cast<int*>(B + o_cur_pf), const_cast<int*>(D.pt_edges),
                       const_cast<int*>(D.ps_edges), const_cast<int*>(D.pf_edges));
    hipLaunchKernelGGL(k_csr_sort, dim3(nP + n_blocks_l + kPrep256Pad + kPrep256Zero), dim3(256), 0, st, nP, nL, D.ps_start,
                       const_cast<int*>(D.ps_edges), D.pt_start, const_cast<int*>(D.pt_edges), D.pf_start, const_cast<int*>(D.pf_edges),
                       const_cast<int*>(D.pf_col), reinterpret_cast<unsigned long long*>(B + o_lm_mask), D.edges, D.pose_col, n,
                       use_mfma ? h->d_St.p : (double*)nullptr, h->d_x.p, n_zero);
    hipLaunchKernelGGL(k_errlin<CamT>, dim3(nP + n_blocks_e + n_tail), dim3(256), 0, st, nP, NE, D.edges, posesB[cur],
                       pointsB[cur], cam, hb, h->d_err.p, h->d_chi2.p, D.pose_col, D.point_col, EBs[set], D.ps_start,
                       D.ps_edges, Hpps[set], bps[set], h->d_partial.p, h->d_ticket.p, h->d_scale_partial.p, 0, (const int*)nullptr,
                       h->rec.d, h->rec_seq, n_blocks_e, nL, D.pt_start, D.pt_edges, Hlls[set], bls[set],
                       LmIn{0.0, 0.0, (const double*)nullptr, (const double*)nullptr, (double*)nullptr});
    hipLaunchKernelGGL(k_finish_items, dim3(1 + n_pairs_all), dim3(256), 0, st, n_blocks_e, h->d_partial.p, nP, nL, bps[set], bls[set],
                       Hpps[set], Hlls[set], h->rec.d, p->lambda_init, h->d_lambda0.p,
                       reinterpret_cast<const unsigned long long*>(B + o_lm_mask), D.pf_start, D.pf_edges, D.pf_col,
                       h->d_items_dev.p, item_cap, h->d_pair_count.p);
    ORBG_HIP(hipGetLastError());
    err_valid = true; spec_ready = true; fin_version = version;
  } else if (first2) {
    const int set = ls ^ 1;
    const int n_blocks_l = errlin_tail_blocks(nL), n_err = nP + n_blocks_e + n_blocks_l, nsb = (nL + 255) / 256;
    hipLaunchKernelGGL(k_errlin_prep<CamT>, dim3(n_err + nsb + kPrep256Pad + kPrep256Zero), dim3(256), 0, st, n_err, nsb,
                       const_cast<int*>(D.pf_edges), const_cast<int*>(D.pf_col), D.pf_start,
                       reinterpret_cast<unsigned long long*>(h->up_d.p + o_lm_mask), n, use_mfma ? h->d_St.p : (double*)nullptr,
                       h->d_x.p, n_zero,
                       nP, NE, D.edges, posesB[cur], pointsB[cur], cam, hb, h->d_err.p, h->d_chi2.p, D.pose_col, D.point_col, EBs[set],
                       D.ps_start, D.ps_edges, Hpps[set], bps[set], h->d_partial.p, h->d_ticket.p, h->d_scale_partial.p, 0,
                       (const int*)nullptr, h->rec.d, h->rec_seq, n_blocks_e, nL, D.pt_start, D.pt_edges, Hlls[set], bls[set],
                       LmIn{0.0, 0.0, (const double*)nullptr, (const double*)nullptr, (double*)nullptr});
    hipLaunchKernelGGL(k_finish_items, dim3(1 + n_pairs_all), dim3(256), 0, st, n_blocks_e, h->d_partial.p, nP, nL, bps[set], bls[set],
                       Hpps[set], Hlls[set], h->rec.d, p->lambda_init, h->d_lambda0.p,
                       reinterpret_cast<const unsigned long long*>(h->up_d.p + o_lm_mask), D.pf_start, D.pf_edges, D.pf_col,
                       h->d_items_dev.p, item_cap, h->d_pair_count.p);
    ORBG_HIP(hipGetLastError());
    err_valid = true; spec_ready = true; fin_version = version;
  } else if (!terminate()) {
    if (NE > 0) {
      // residuals + linearisation of the initial estimate in the fused kernel of the later trials (its record is not waited
      // for: it carries the sequence number the host has already seen)
      const int set = ls ^ 1;
      const int n_blocks_l = errlin_tail_blocks(nL);
      hipLaunchKernelGGL(k_errlin<CamT>, dim3(nP + n_blocks_e + n_blocks_l), dim3(256), 0, st, nP, NE, D.edges, posesB[cur],
                         pointsB[cur], cam, hb, h->d_err.p, h->d_chi2.p, D.pose_col, D.point_col, EBs[set], D.ps_start,
                         D.ps_edges, Hpps[set], bps[set], h->d_partial.p, h->d_ticket.p, h->d_scale_partial.p, 0, (const int*)nullptr,
                         h->rec.d, h->rec_seq, n_blocks_e, nL, D.pt_start, D.pt_edges, Hlls[set], bls[set],
                         LmIn{0.0, 0.0, (const double*)nullptr, (const double*)nullptr, (double*)nullptr});
    }
    err_valid = true;
    spec_ready = true;
    if (NE > 0) {
      // the first round's lambda init needs nothing the host is still building: it goes right behind the linearisation
      hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, st, n_blocks_e, h->d_partial.p, nP, nL, h->d_x.p, bps[ls ^ 1], bls[ls ^ 1],
                         Hpps[ls ^ 1], Hlls[ls ^ 1], 0.0, (int*)nullptr, 0, 1, h->rec.d, p->lambda_init, h->d_lambda0.p);
      fin_version = version;
    }
    ORBG_HIP(hipGetLastError());
  }
  // not needed before the first Schur complement: the zeroed step vector and the padding / zeros of the bordered tile image
  // (they depend on n only, k_schur never touches them -- once per call) go behind the first linearisation
  if (first2) {
    // (done by k_errlin_prep)
  } else if (dev_lists) {
    // ... in one launch with the per-landmark lists (k_prep)
    const int nsb = (nL + kSortPfThreads - 1) / kSortPfThreads;
    hipLaunchKernelGGL(k_prep, dim3(nsb + 32 + 8), dim3(kSortPfThreads), 0, st, nsb, nL, D.pf_start, const_cast<int*>(D.pf_edges),
                       const_cast<int*>(D.pf_col), reinterpret_cast<unsigned long long*>(h->up_d.p + o_lm_mask), D.edges, D.pose_col,
                       n, use_mfma ? h->d_St.p : (double*)nullptr, h->d_x.p, n_zero);
    ORBG_HIP(hipGetLastError());
  } else {
    ORBG_HIP(hipMemsetAsync(h->d_x.p, 0, (size_t)n_zero * sizeof(double), st));
    if (use_mfma) ORBG_HIP(ldltm::launch_image_pad(n, h->d_St.p, st));
  }
  const double t_s2b = now_s();
  if (!dev_lists)
  // while those run: the landmarks' free observations sorted by pose column, their pose masks / the pair counts
  {
    // per landmark: stable insertion sort of its free observations by pose column (a handful each), then count its
    // (pose pair) items; pair id = row_off[i1] + i2
    std::vector<int>& row_off = h->s_row_off;
    row_off.resize(std::max(nP, 1));
    for (int i1 = 0; i1 < nP; i1++) row_off[i1] = i1 * nP - i1 * (i1 - 1) / 2 - i1;
    for (int i = 0; i <= n_pairs_all; i++) pair_start[i] = 0;
    for (int l = 0; l < nL; l++) {
      const int b0 = pf_start[l], e0 = pf_start[l + 1];
      for (int a2 = b0 + 1; a2 < e0; a2++) {
        const int e = pf_edges[a2], key = pose_col[edges[e].pose];
        int b2 = a2 - 1;
        while (b2 >= b0 && pose_col[edges[pf_edges[b2]].pose] > key) { pf_edges[b2 + 1] = pf_edges[b2]; b2--; }
        pf_edges[b2 + 1] = e;
      }
      for (int j = b0; j < e0; j++) pf_col[j] = pose_col[edges[pf_edges[j]].pose];
      if (dev_items) {
        unsigned long long m = 0;
        for (int j = b0; j < e0; j++) m |= 1ull << pf_col[j];
        lm_mask[l] = m;
      } else {
        for (int a2 = b0; a2 < e0; a2++) {
          const int ro = row_off[pf_col[a2]] + 1;
          for (int b2 = a2; b2 < e0; b2++) pair_start[ro + pf_col[b2]]++;
        }
      }
    }
  }
  size_t off_b = off_a;
  auto pair_id = [&](int i1, int i2) { return i1 * nP - i1 * (i1 - 1) / 2 + (i2 - i1); };
  if (dev_items && !terminate()) {
    // the pf lists and pose masks go up next and the device builds the pair items behind the first linearisation; the pair
    // table (a function of nP alone) rides along, so nothing is left for a third upload
    for (int i1 = 0; i1 < nP; i1++)
      for (int i2 = i1; i2 < nP; i2++) { pair_i1[pair_id(i1, i2)] = i1; pair_i2[pair_id(i1, i2)] = i2; }
    off_b = o_pair_start;
    if (!dev_lists && (rc = upload_arena(h, off_a, off_b, st))) return rc;      // (dev_lists: k_schur derives the pair from its index)
    if (nL > 0 && !first2)
      hipLaunchKernelGGL(k_build_items, dim3(n_pairs_all), dim3(256), 0, st, nP, nL,
                         reinterpret_cast<const unsigned long long*>(h->up_d.p + o_lm_mask), D.pf_start, D.pf_edges, D.pf_col,
                         h->d_items_dev.p, item_cap, h->d_pair_count.p);
    ORBG_HIP(hipGetLastError());
  }
  // pose pairs (i1 <= i2) and their landmark items, grouped by pair (counting sort keeps landmark order)
  if (!dev_items) {
    for (int i = 0; i < n_pairs_all; i++) pair_start[i + 1] += pair_start[i];
    std::vector<int>& fill = h->s_fill;
    fill.assign(pair_start, pair_start + n_pairs_all);
    const std::vector<int>& row_off = h->s_row_off;
    for (int l = 0; l < nL; l++) {
      const int b0 = pf_start[l], e0 = pf_start[l + 1];
      for (int a2 = b0; a2 < e0; a2++) {
        const int ro = row_off[pf_col[a2]], ea = pf_edges[a2];
        for (int b2 = a2; b2 < e0; b2++) items[fill[ro + pf_col[b2]]++] = PairItem{ea, pf_edges[b2], l};
      }
    }
  }
  // keep every pair (diagonals always; off-diagonals even if empty so that S is fully written)
  for (int i1 = 0; i1 < nP; i1++)
    for (int i2 = i1; i2 < nP; i2++) { pair_i1[pair_id(i1, i2)] = i1; pair_i2[pair_id(i1, i2)] = i2; }

  if (!dev_items && (rc = upload_arena(h, off_b, off, st))) return rc;
  const double t_s3b = now_s();
  int done = 0;
  if ((rc = optimize(p->its_round1 > 0 ? p->its_round1 : 5, &done))) return rc;
  r->iters_round1 = done;
  if (!terminate()) {
    last_round = true;
    if ((rc = optimize(p->its_round2 > 0 ? p->its_round2 : 10, &done))) return rc;
    r->iters_round2 = done;
  }
  const double t_d = now_s();
  // ---- results: chi2 of the LAST error evaluation (d_chi2), depth test with the current estimate (S/Optimizer.cc:2131-2166):
  // flags computed on the device, everything comes back through one pinned block
  if (exp_version == version && exp_buf == cur) {
    if ((rc = h->sig.wait(st))) return rc;              // the speculative export is the final one
  } else {
    launch_export(cur);
    ORBG_HIP(hipGetLastError());
    if ((rc = h->sig.sync(st))) return rc;
  }
  const double t_e = now_s();
  if (prof_pending) {                                   // the stream is idle here: both events have completed
    float ems = 0;
    if (hipEventElapsedTime(&ems, h->prof_ev[0], h->prof_ev[1]) == hipSuccess) { h->prof_sum_ms += ems; h->prof_n++; h->prof_n_unknowns = n; }
  }
  const PoseQ* rposes = reinterpret_cast<const PoseQ*>(h->dl_h.h + d_poses_o);
  const double* rpoints = reinterpret_cast<const double*>(h->dl_h.h + d_points_o);
  const uint8_t* rflags = h->dl_h.h + d_flags_o;
  int n_out = 0;
  {
    // (restrict-qualified locals and no branch in the bodies: the loops vectorise; as one loop with the two tests inside they
    // cost ~1 ns per edge)
    const uint8_t* __restrict__ rf = rflags;
    uint8_t* __restrict__ odp = reinterpret_cast<uint8_t*>(r->edge_depth_pos);
    uint8_t* __restrict__ oout = reinterpret_cast<uint8_t*>(r->edge_outlier);
    if (version == 0) {
      // the flag was raised between the check that precedes optimize() and the first iteration: g2o never evaluated a
      // residual (e->chi2() reads an edge's never-written _error: pinned as zero, as in the oracle), so only the depth test
      // of the unchanged estimate can make an outlier (S/Optimizer.cc:2219-2253)
      if (odp) for (int k = 0; k < NE; k++) odp[k] = rf[k] & 1;
      if (oout) for (int k = 0; k < NE; k++) oout[k] = (rf[k] & 1) ^ 1;
      for (int k = 0; k < NE; k++) n_out += (rf[k] & 1) ^ 1;
    } else {
      if (odp) for (int k = 0; k < NE; k++) odp[k] = rf[k] & 1;
      if (oout) for (int k = 0; k < NE; k++) oout[k] = (rf[k] >> 1) & 1;
      for (int k = 0; k < NE; k++) n_out += (rf[k] >> 1) & 1;
    }
  }
  if (r->edge_chi2 && NE > 0) {
    if (version == 0) memset(r->edge_chi2, 0, 8 * (size_t)NE);
    else memcpy(r->edge_chi2, h->dl_h.h + d_chi_o, 8 * (size_t)NE);
  }
  r->n_outliers = n_out;
  // vToErase.size() >= (vpMapPointEdgeMono.size() + vpMapPointEdgeStereo.size()) * 0.5  (S/Optimizer.cc:2256: the right camera's edges
  // are in vToErase but not in the sum)
  if (NE > 0 && n_out >= (NE - n_right_edges) * 0.5) r->status = LBA_REJECTED_OUTLIERS;
  for (int i = 0; i < NP; i++) {                       // Converter::toCvMat(SE3Quat)
    double R[9];
    quat_to_R(rposes[i].q, R);
    float* T = r->poses + 16 * (size_t)i;
    for (int a = 0; a < 3; a++) { for (int c = 0; c < 3; c++) T[4 * a + c] = (float)R[3 * a + c]; T[4 * a + 3] = (float)rposes[i].t[a]; }
    T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
  }
  for (size_t i = 0; i < 3 * (size_t)NX; i++) r->points[i] = (float)rpoints[i];
  {
    const double t_f = now_s();
    tr.t[0] += t_b - t_a; tr.t[1] += t_c - t_b; tr.t[2] += t_d - t_c; tr.t[3] += t_e - t_d; tr.t[4] += t_f - t_e; tr.t[5] += t_s1 - t_a; tr.t[6] += t_s2 - t_s1; tr.t[7] += t_s3b - t_s2b; tr.n++;
  }
  return ORBG_OK;
}

// Live measurement for bench.py's roofline: with profiling on, ONE launch of the reduced-camera-system LDL^T per solve is
// bracketed by a HIP event pair on the handle's stream.  lba_get_solver_stats returns the accumulated bracket time, the
// number of brackets, the size of the system and which kernel ran; lba_event_overhead measures an empty pair.
extern "C" int lba_set_profiling(lba_handle* h, int on, int reset) {
  if (!h) return ORBG_BAD_ARG;
  int rc = select_device(h->device);
  if (rc) return rc;
  if (on && !h->prof_ev[0]) { ORBG_HIP(hipEventCreate(&h->prof_ev[0])); ORBG_HIP(hipEventCreate(&h->prof_ev[1])); }
  h->prof_on = on;
  if (reset) { h->prof_sum_ms = 0; h->prof_n = 0; h->prof_solves = 0; }
  return ORBG_OK;
}

extern "C" int lba_get_solver_stats(lba_handle* h, double* sum_ms, int64_t* n_brackets, int32_t* n_unknowns, int32_t* matrix_core) {
  if (!h || !sum_ms || !n_brackets) return ORBG_BAD_ARG;
  *sum_ms = h->prof_sum_ms; *n_brackets = h->prof_n;
  if (n_unknowns) *n_unknowns = h->prof_n_unknowns;
  if (matrix_core) *matrix_core = h->prof_n_unknowns >= 1 && ldltm::supports(h->prof_n_unknowns) && !h->sw.ldlt_wide;
  return ORBG_OK;
}

#ifdef LBA_PROFILE
extern "C" int lba_debug_prof(long long* out /* 3 x (8 + 2040) */, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lba_prof), sizeof(g_lba_prof)) != hipSuccess) return -1;
  if (reset) { static long long z[3][8 + 2 * 1020]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_lba_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

extern "C" int lba_get_watchdog_count(lba_handle* h, int64_t* n_timeouts) {
  if (!h || !n_timeouts) return ORBG_BAD_ARG;
  *n_timeouts = h->xcd_timeouts;
  return ORBG_OK;
}

extern "C" int lba_event_overhead(lba_handle* h, int reps, float* ms) {
  if (!h || !ms || reps < 1) return ORBG_BAD_ARG;
  int rc = select_device(h->device);
  if (rc) return rc;
  if ((rc = lba_set_profiling(h, h->prof_on, 0))) return rc;
  if (!h->prof_ev[0]) { ORBG_HIP(hipEventCreate(&h->prof_ev[0])); ORBG_HIP(hipEventCreate(&h->prof_ev[1])); }
  double acc = 0;
  for (int i = 0; i < reps; i++) {
    ORBG_HIP(hipEventRecord(h->prof_ev[0], h->stream));
    ORBG_HIP(hipEventRecord(h->prof_ev[1], h->stream));
    ORBG_HIP(hipStreamSynchronize(h->stream));
    float e = 0;
    ORBG_HIP(hipEventElapsedTime(&e, h->prof_ev[0], h->prof_ev[1]));
    acc += e;
  }
  *ms = (float)(acc / reps);
  return ORBG_OK;
}

// Optimizer::LocalBundleAdjustment runs on the LocalMapping thread, concurrently with Tracking (S/ClientSystem.cc:105-106,
// S/LocalMapping.cc:114-133).  lba_solve_async hands the problem to a worker thread owned by the handle and returns at once;
// lba_wait blocks until that solve has finished and returns its status.  problem / stop_flag / result must stay valid
// until lba_wait returns; one solve in flight per handle.
static inline bool lba_spin_allowed() { return orbg::poll_allowed(); }
// spins until pred() or `limit_us` have passed; returns pred()
template <typename Pred>
static inline bool lba_spin_until(Pred pred, double limit_us) {
  if (!lba_spin_allowed()) return pred();
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned spins = 0;; spins++) {
    if (pred()) return true;
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
    if ((spins & 0xFF) == 0xFF) {
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3 > limit_us) return pred();
    }
  }
}

extern "C" int lba_solve_h(lba_handle* h, const lba_problem* p, const volatile int32_t* stop_flag, lba_result* r) {
  StopRef s; s.i32 = stop_flag;
  return lba_solve_impl(h, p, s, r);
}
extern "C" int lba_solve_hb(lba_handle* h, const lba_problem* p, const volatile uint8_t* stop_bool, lba_result* r) {
  StopRef s; s.u8 = stop_bool;
  return lba_solve_impl(h, p, s, r);
}

static int lba_solve_async_impl(lba_handle* h, const lba_problem* p, StopRef stop_flag, lba_result* r) {
  if (!h || !p || !r) return ORBG_BAD_ARG;
  std::unique_lock<std::mutex> lk(h->mu);
  if (h->job_state.load(std::memory_order_acquire) != 0) return ORBG_BAD_ARG;
  if (!h->worker.joinable()) {
    h->worker = std::thread([h]() {
      orbg::set_thread_role(orbg::kRoleLbaWorker);
      for (;;) {
        // next job: spin briefly (the tracking thread usually submits within tens of microseconds), then sleep
        if (!lba_spin_until([h]() { return h->quit || h->job_state.load(std::memory_order_acquire) == 1; }, 400.0)) {
          std::unique_lock<std::mutex> lk(h->mu);
          h->cv.wait(lk, [h]() { return h->quit || h->job_state.load(std::memory_order_acquire) == 1; });
        }
        const lba_problem* p; StopRef st; lba_result* r;
        {
          std::unique_lock<std::mutex> lk(h->mu);
          if (h->quit) return;
          h->job_state.store(2, std::memory_order_release);
          p = h->job_p; st = h->job_stop; r = h->job_r;
        }
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        const int rc = lba_solve_impl(h, p, st, r);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        {
          std::unique_lock<std::mutex> lk(h->mu);
          h->job_ms = 1e3 * (double)(t1.tv_sec - t0.tv_sec) + 1e-6 * (double)(t1.tv_nsec - t0.tv_nsec);
          h->job_status = rc;
          h->job_state.store(0, std::memory_order_release);
        }
        h->cv.notify_all();
      }
    });
  }
  h->job_p = p; h->job_stop = stop_flag; h->job_r = r;
  h->job_state.store(1, std::memory_order_release);
  lk.unlock();
  h->cv.notify_all();
  return ORBG_OK;
}
extern "C" int lba_solve_async(lba_handle* h, const lba_problem* p, const volatile int32_t* stop_flag, lba_result* r) {
  StopRef s; s.i32 = stop_flag;
  return lba_solve_async_impl(h, p, s, r);
}
extern "C" int lba_solve_async_b(lba_handle* h, const lba_problem* p, const volatile uint8_t* stop_bool, lba_result* r) {
  StopRef s; s.u8 = stop_bool;
  return lba_solve_async_impl(h, p, s, r);
}
extern "C" int lba_wait(lba_handle* h, double* solve_ms) {
  if (!h) return ORBG_BAD_ARG;
  // the caller is usually a few tens of microseconds early: spin, then sleep on the condition variable
  if (!lba_spin_until([h]() { return h->job_state.load(std::memory_order_acquire) == 0; }, 3000.0)) {
    std::unique_lock<std::mutex> lk(h->mu);
    h->cv.wait(lk, [h]() { return h->job_state.load(std::memory_order_acquire) == 0; });
  }
  std::unique_lock<std::mutex> lk(h->mu);
  if (solve_ms) *solve_ms = h->job_ms;
  return h->job_status;
}

static int lba_solve_once(const lba_problem* p, StopRef stop, lba_result* r) {
  if (!p) return ORBG_BAD_ARG;
  lba_handle* h = nullptr;
  int rc = lba_create(p->device, p->n_poses, p->n_points, p->n_edges, &h);
  if (rc) return rc;
  rc = lba_solve_impl(h, p, stop, r);
  lba_destroy(h);
  return rc;
}
extern "C" int lba_solve(const lba_problem* p, const volatile int32_t* stop_flag, lba_result* r) {
  StopRef s; s.i32 = stop_flag;
  return lba_solve_once(p, s, r);
}
extern "C" int lba_solve_b(const lba_problem* p, const volatile uint8_t* stop_bool, lba_result* r) {
  StopRef s; s.u8 = stop_bool;
  return lba_solve_once(p, s, r);
}
