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
//                   (v_mfma_f64_16x16x4_f64; 16x16 tiles in registers, dataflow between wavefronts); k_ldlt_flow /
//                   k_ldlt_rows / k_ldlt are the vector-ALU kernels for windows it does not cover (> 50 free poses)
//   k_update        thread/vertex: x_l = Dinv_l (bl_l - sum_i Hpl_il^T x_i); trial state = exp(x_p) * T  /  X + x_l
//   k_finish        one workgroup: robust chi2, computeScale, max-diagonal -> pinned host record
// The reduced camera system is tiny (6P x 6P, P <= a few tens): the path is latency bound, not FLOP bound.
// Parity: poses/points within 1e-4 of the oracle after float32 write-back, identical outlier sets.

#include <time.h>

#include <condition_variable>
#include <atomic>
#include <mutex>
#include <thread>

#include "common.hpp"
#include "wave.hpp"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>

using namespace orbg;

#include "ldlt_mfma.hpp"

namespace {

struct Cam { double fx, fy, cx, cy, bf; float bf_f; };
struct PoseQ { double q[4]; double t[3]; };   // quaternion x,y,z,w + translation (SE3Quat)

// ---- SE3 / quaternion helpers shared by host and device (Eigen / g2o semantics, see oracle/lba.cc for citations)
__host__ __device__ inline void quat_rotate(const double* q, const double* v, double* out) {
  const double uv0 = 2 * (q[1] * v[2] - q[2] * v[1]), uv1 = 2 * (q[2] * v[0] - q[0] * v[2]), uv2 = 2 * (q[0] * v[1] - q[1] * v[0]);
  out[0] = v[0] + q[3] * uv0 + (q[1] * uv2 - q[2] * uv1);
  out[1] = v[1] + q[3] * uv1 + (q[2] * uv0 - q[0] * uv2);
  out[2] = v[2] + q[3] * uv2 + (q[0] * uv1 - q[1] * uv0);
}

__host__ __device__ inline void quat_to_R(const double* q, double* R) {
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

__host__ __device__ inline void quat_from_R(const double* m, double* q) {
  // Eigen::Quaterniond(Matrix3d); the three "largest diagonal" cases are spelled out so that no local array is
  // indexed at run time (which would put it in scratch memory on the GPU)
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    return;
  }
  int i = 0;
  if (m[4] > m[0]) i = 1;
  if (m[8] > (i == 0 ? m[0] : m[4])) i = 2;
  if (i == 0) {          // j = 1, k = 2
    t = sqrt(m[0] - m[4] - m[8] + 1.0);
    q[0] = 0.5 * t; t = 0.5 / t;
    q[3] = (m[7] - m[5]) * t; q[1] = (m[3] + m[1]) * t; q[2] = (m[6] + m[2]) * t;
  } else if (i == 1) {   // j = 2, k = 0
    t = sqrt(m[4] - m[8] - m[0] + 1.0);
    q[1] = 0.5 * t; t = 0.5 / t;
    q[3] = (m[2] - m[6]) * t; q[2] = (m[7] + m[5]) * t; q[0] = (m[1] + m[3]) * t;
  } else {               // j = 0, k = 1
    t = sqrt(m[8] - m[0] - m[4] + 1.0);
    q[2] = 0.5 * t; t = 0.5 / t;
    q[3] = (m[3] - m[1]) * t; q[0] = (m[2] + m[6]) * t; q[1] = (m[5] + m[7]) * t;
  }
}

__host__ __device__ inline void quat_normalize(double* q) {   // SE3Quat::normalizeRotation
  if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

// estimate = SE3Quat::exp(update) * estimate   (G/types/se3quat.h:225-260,102-110)
__device__ inline void pose_oplus(const PoseQ& T, const double* u, PoseQ* out) {
  const double om0 = u[0], om1 = u[1], om2 = u[2];
  const double theta = sqrt(om0 * om0 + om1 * om1 + om2 * om2);
  const double O[9] = {0, -om2, om1, om2, 0, -om0, -om1, om0, 0};
  double O2[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
  double R[9], V[9];
  if (theta < 0.00001) {
#pragma unroll
    for (int i = 0; i < 9; i++) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
  } else {
    const double s = sin(theta), c = cos(theta);
    const double a = s / theta, b = (1 - c) / (theta * theta), cc = (theta - s) / (theta * theta * theta);
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const double I = (i % 4 == 0 ? 1.0 : 0.0);
      R[i] = I + a * O[i] + b * O2[i];
      V[i] = I + b * O[i] + cc * O2[i];
    }
  }
  double eq[4], et[3];
  quat_from_R(R, eq);
#pragma unroll
  for (int i = 0; i < 3; i++) et[i] = V[3 * i] * u[3] + V[3 * i + 1] * u[4] + V[3 * i + 2] * u[5];
  quat_normalize(eq);
  double rt[3];
  quat_rotate(eq, T.t, rt);
  out->t[0] = et[0] + rt[0]; out->t[1] = et[1] + rt[1]; out->t[2] = et[2] + rt[2];
  const double* a = eq; const double* b = T.q;
  out->q[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  out->q[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  out->q[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  out->q[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  quat_normalize(out->q);
}

// pose_oplus with the divisions hoisted (one reciprocal of theta, one per normalisation) and sincos(): the same formulas as
// SE3Quat::exp / operator* / normalizeRotation, fewer dependent FP64 divisions.  Used where the update runs on the critical
// path of a single workgroup (PoseOptimization); results differ from pose_oplus in the last bits only.
__device__ inline void pose_oplus_fast(const PoseQ& T, const double* u, PoseQ* out) {
  const double om0 = u[0], om1 = u[1], om2 = u[2];
  const double th2 = om0 * om0 + om1 * om1 + om2 * om2;
  const double theta = sqrt(th2);
  const double O[9] = {0, -om2, om1, om2, 0, -om0, -om1, om0, 0};
  double O2[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
  double R[9], V[9];
  if (theta < 0.00001) {
#pragma unroll
    for (int i = 0; i < 9; i++) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
  } else {
    double sn, cs;
    sincos(theta, &sn, &cs);
    const double it = 1.0 / theta, it2 = it * it;
    const double a = sn * it, b = (1 - cs) * it2, cc = (theta - sn) * (it2 * it);
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const double I = (i % 4 == 0 ? 1.0 : 0.0);
      R[i] = I + a * O[i] + b * O2[i];
      V[i] = I + b * O[i] + cc * O2[i];
    }
  }
  double eq[4], et[3];
  quat_from_R(R, eq);
#pragma unroll
  for (int i = 0; i < 3; i++) et[i] = V[3 * i] * u[3] + V[3 * i + 1] * u[4] + V[3 * i + 2] * u[5];
  {
    if (eq[3] < 0) { eq[0] = -eq[0]; eq[1] = -eq[1]; eq[2] = -eq[2]; eq[3] = -eq[3]; }
    const double in = 1.0 / sqrt(eq[0] * eq[0] + eq[1] * eq[1] + eq[2] * eq[2] + eq[3] * eq[3]);
    eq[0] *= in; eq[1] *= in; eq[2] *= in; eq[3] *= in;
  }
  double rt[3];
  quat_rotate(eq, T.t, rt);
  out->t[0] = et[0] + rt[0]; out->t[1] = et[1] + rt[1]; out->t[2] = et[2] + rt[2];
  const double* a2 = eq; const double* b2 = T.q;
  double q3 = a2[3] * b2[3] - a2[0] * b2[0] - a2[1] * b2[1] - a2[2] * b2[2];
  double q0 = a2[3] * b2[0] + a2[0] * b2[3] + a2[1] * b2[2] - a2[2] * b2[1];
  double q1 = a2[3] * b2[1] + a2[1] * b2[3] + a2[2] * b2[0] - a2[0] * b2[2];
  double q2 = a2[3] * b2[2] + a2[2] * b2[3] + a2[0] * b2[1] - a2[1] * b2[0];
  if (q3 < 0) { q0 = -q0; q1 = -q1; q2 = -q2; q3 = -q3; }
  const double in = 1.0 / sqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  out->q[0] = q0 * in; out->q[1] = q1 * in; out->q[2] = q2 * in; out->q[3] = q3 * in;
}

// 1/sqrt(d): hardware seed + two Newton-Raphson steps
__device__ __forceinline__ double fast_rsqrt(double d) {
  double x = __builtin_amdgcn_rsq(d);
  x = x * __builtin_fma(-0.5 * d * x, x, 1.5);
  x = x * __builtin_fma(-0.5 * d * x, x, 1.5);
  return x;
}

// exp(u) * T for the increments of PoseOptimization's trial loop (|omega| < 0.3 rad; anything larger takes pose_oplus_fast):
// the four functions of theta the exponential needs -- sin(theta/2)/theta, cos(theta/2), (1-cos theta)/theta^2,
// (theta - sin theta)/theta^3 -- are even power series in theta, six terms of each are exact to 1e-17 in that range; no
// sqrt, sincos or division (the library sincos alone is ~300 FP64 instructions, and every instruction of this kernel costs
// the workgroup 8 cycles on the critical path), and no cancellation in (1 - cos theta) for the small angles LM steps have.
__device__ inline void pose_oplus_series(const PoseQ& T, const double* u, PoseQ* out) {
  const double om0 = u[0], om1 = u[1], om2 = u[2];
  const double t = om0 * om0 + om1 * om1 + om2 * om2;
  if (t > 0.09) { pose_oplus_fast(T, u, out); return; }
  const double s = __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, -1.0 / 81749606400.0, 1.0 / 185794560.0), -1.0 / 645120.0), 1.0 / 3840.0), -1.0 / 48.0), 0.5);
  const double c = __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, -1.0 / 3715891200.0, 1.0 / 10321920.0), -1.0 / 46080.0), 1.0 / 384.0), -0.125), 1.0);
  const double b = __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, -1.0 / 479001600.0, 1.0 / 3628800.0), -1.0 / 40320.0), 1.0 / 720.0), -1.0 / 24.0), 0.5);
  const double cc = __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, __builtin_fma(t, -1.0 / 6227020800.0, 1.0 / 39916800.0), -1.0 / 362880.0), 1.0 / 5040.0), -1.0 / 120.0), 1.0 / 6.0);
  // V u_t = u_t + b (omega x u_t) + cc (omega x (omega x u_t))
  const double w0 = om1 * u[5] - om2 * u[4], w1 = om2 * u[3] - om0 * u[5], w2 = om0 * u[4] - om1 * u[3];
  const double z0 = om1 * w2 - om2 * w1, z1 = om2 * w0 - om0 * w2, z2 = om0 * w1 - om1 * w0;
  const double et[3] = {u[3] + b * w0 + cc * z0, u[4] + b * w1 + cc * z1, u[5] + b * w2 + cc * z2};
  const double eq[4] = {om0 * s, om1 * s, om2 * s, c};          // unit up to rounding, w > 0
  double rt[3];
  quat_rotate(eq, T.t, rt);
  out->t[0] = et[0] + rt[0]; out->t[1] = et[1] + rt[1]; out->t[2] = et[2] + rt[2];
  const double* b2 = T.q;
  double q3 = eq[3] * b2[3] - eq[0] * b2[0] - eq[1] * b2[1] - eq[2] * b2[2];
  double q0 = eq[3] * b2[0] + eq[0] * b2[3] + eq[1] * b2[2] - eq[2] * b2[1];
  double q1 = eq[3] * b2[1] + eq[1] * b2[3] + eq[2] * b2[0] - eq[0] * b2[2];
  double q2 = eq[3] * b2[2] + eq[2] * b2[3] + eq[0] * b2[1] - eq[1] * b2[0];
  if (q3 < 0) { q0 = -q0; q1 = -q1; q2 = -q2; q3 = -q3; }
  const double in = fast_rsqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  out->q[0] = q0 * in; out->q[1] = q1 * in; out->q[2] = q2 * in; out->q[3] = q3 * in;
}

// 1/d to ~1 ulp: hardware seed + two Newton-Raphson steps
__device__ __forceinline__ double fast_rcp(double d) {
  double x = __builtin_amdgcn_rcp(d);
  x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
  x = __builtin_fma(x, __builtin_fma(-d, x, 1.0), x);
  return x;
}

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

__device__ inline void huber(double e, double delta, double dsqr, double* rho0, double* rho1) {
  if (e <= dsqr) { *rho0 = e; *rho1 = 1.; }
  else { const double s = sqrt(e); *rho0 = 2 * s * delta - dsqr; *rho1 = delta / s; }
}

struct Huber { double delta_mono, dsqr_mono, delta_stereo, dsqr_stereo; };

// ---------------------------------------------------------------------------------------------- kernels

// residuals + chi2 + robust rho (computeActiveErrors + activeRobustChi2); block partial sums in fixed order
struct HostRec { double chi2, scale, maxdiag, chi2_init; int ok; unsigned seq; };   // seq is written last: the host spins on it   // what the host reads per LM trial (mapped pinned memory)

// Results of a solve, written by the GPU straight into the caller-visible pinned block (no copy commands): per edge a flag
// byte (bit 0 = isDepthPositive() with the final estimate, bit 1 = outlier: chi2 > 5.991 / 7.815 or depth <= 0,
// S/Optimizer.cc:2131-2166) and optionally its chi2; the final poses and points.
__global__ __launch_bounds__(256) void k_export(int n_edges, int n_poses, int n_points, const lba_edge* __restrict__ edges,
                                               const PoseQ* __restrict__ poses, const double* __restrict__ points,
                                               const double* __restrict__ chi2, uint8_t* __restrict__ out_flags,
                                               double* __restrict__ out_chi2, PoseQ* __restrict__ out_poses,
                                               double* __restrict__ out_points) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < n_edges) {
    const lba_edge e = edges[k];
    double rr[3];
    quat_rotate(poses[e.pose].q, points + 3 * (size_t)e.point, rr);
    const bool depth_pos = rr[2] + poses[e.pose].t[2] > 0.0;
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

__device__ __forceinline__ void publish_trial_record(int n_edge_blocks, const double* __restrict__ partial, unsigned* __restrict__ ticket,
                                                     const double* __restrict__ scale_partial, int n_scale_partial,
                                                     const int* __restrict__ ok_flag, HostRec* __restrict__ rec, unsigned seq,
                                                     const LmIn lm = LmIn{0, 0, nullptr, nullptr, nullptr}) {
  __shared__ int s_last;
  __shared__ double parts[1024];
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == (unsigned)n_edge_blocks - 1);
  }
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const int np = n_edge_blocks;
  // all partials are fetched in parallel, then summed by one thread in index order
  const int tot = min(np + n_scale_partial, 1024);
  for (int i = threadIdx.x; i < tot; i += 256)
    parts[i] = __hip_atomic_load(i < np ? &partial[i] : &scale_partial[i - np], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (threadIdx.x == 0) {
    double chi = 0, scale = 0;
    for (int i = 0; i < np; i++) chi += i < 1024 ? parts[i] : __hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = 0; i < n_scale_partial; i++)
      scale += np + i < 1024 ? parts[np + i] : __hip_atomic_load(&scale_partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int okv = ok_flag ? *ok_flag : 1;
    rec->chi2 = chi; rec->scale = scale; rec->ok = okv;     // maxdiag / chi2_init stay as k_finish left them
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                  // system-scope release of the record (no acquire half: no L2 invalidation)
    *reinterpret_cast<volatile unsigned*>(&rec->seq) = seq;       // the host polls this word instead of hipStreamSynchronize
  }
}

// (returns this thread's chi2 -- the caller may go on with it: k_errors_export)
__device__ __forceinline__ double errors_block(int bid, int n_edge_blocks, int n_edges, const lba_edge* __restrict__ edges,
                                               const PoseQ* __restrict__ poses, const double* __restrict__ points, Cam cam, Huber hb,
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
__global__ __launch_bounds__(256) void k_errors(int n_edges, const lba_edge* __restrict__ edges, const PoseQ* __restrict__ poses,
                                               const double* __restrict__ points, Cam cam, Huber hb, double* __restrict__ err,
                                               double* __restrict__ chi2, double* __restrict__ partial,
                                               int final_mode, unsigned* __restrict__ ticket, const double* __restrict__ scale_partial,
                                               int n_scale_partial, const int* __restrict__ ok_flag, HostRec* __restrict__ rec,
                                               unsigned seq) {
  errors_block((int)blockIdx.x, (int)gridDim.x, n_edges, edges, poses, points, cam, hb, err, chi2, partial, final_mode, ticket, scale_partial,
               n_scale_partial, ok_flag, rec, seq);
}
// The last evaluation of a solve and the export of the state it evaluated (speculative: dropped if the trial is rejected) in
// one launch: the export needs nothing of the other workgroups (an edge's flags follow from its own chi2).
__global__ __launch_bounds__(256) void k_errors_export(int n_edge_blocks, int n_edges, const lba_edge* __restrict__ edges,
                                                      const PoseQ* __restrict__ poses, const double* __restrict__ points, Cam cam, Huber hb,
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
    double rr[3];
    quat_rotate(poses[e.pose].q, points + 3 * (size_t)e.point, rr);
    depth_pos = rr[2] + poses[e.pose].t[2] > 0.0;
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
__device__ inline void edge_jacobians(const PoseQ& T, double x, double y, double z, const Cam& c, bool mono, double* A, double* B) {
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

// thread per edge; the 256 x 27 block of results is staged in LDS and written as ONE contiguous, coalesced chunk
// FUSED: the residuals are computed here (and stored, with their chi2 and the workgroup's robust partial sum) instead of being
// read back from a preceding k_errors launch -- same functions, same inputs, same bits.  The partial sum is written, and the
// trial record published by the last workgroup, BEFORE the Jacobians: the host gets its verdict ~5 us earlier and its
// decision latency hides behind the rest of this kernel and k_reduce_points.
struct TrialPublish {
  double* partial; unsigned* ticket; const double* scale_partial; int n_scale_partial; const int* ok_flag; HostRec* rec; unsigned seq;
  int n_edge_blocks;
  LmIn lm;
};

template <bool FUSED>
__device__ __forceinline__ void linearize_block(int bid, int n_edges, const lba_edge* __restrict__ edges, const PoseQ* __restrict__ poses,
                                                  const double* __restrict__ points, Cam c, Huber hb, double* __restrict__ err,
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
    if (threadIdx.x == 0) pub.partial[bid] = red_f[0];
    publish_trial_record(pub.n_edge_blocks, pub.partial, pub.ticket, pub.scale_partial, pub.n_scale_partial, pub.ok_flag, pub.rec, pub.seq,
                         pub.lm);
  }
  if (live) {
    double* out = stage + threadIdx.x * kEB;
    double r[3];
    quat_rotate(T.q, X, r);
    double A[9], B[18];
    edge_jacobians(T, r[0] + T.t[0], r[1] + T.t[1], r[2] + T.t[2], c, mono, A, B);
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
  __syncthreads();
  const int valid = min(256, n_edges - bid * 256);
  double* dst = EB + (size_t)bid * 256 * kEB;
  for (int i = threadIdx.x; i < valid * kEB; i += 256) dst[i] = stage[i];
}
template <bool FUSED>
__device__ __forceinline__ void lin_poses_block(int bid, const int* __restrict__ ps_start, const int* __restrict__ ps_edges,
                                                  const lba_edge* __restrict__ edges, const PoseQ* __restrict__ poses,
                                                  const double* __restrict__ points, Cam c, Huber hb, const double* __restrict__ err,
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
      edge_jacobians(T, r[0] + T.t[0], r[1] + T.t[1], r[2] + T.t[2], c, mono, A, B);
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
__device__ __forceinline__ void lin_points_block(int bid, int nL, const int* __restrict__ pt_start, const int* __restrict__ pt_edges,
                                                   const lba_edge* __restrict__ edges, const PoseQ* __restrict__ poses,
                                                   const double* __restrict__ points, Cam c, Huber hb, double* __restrict__ Hll,
                                                   double* __restrict__ bl) {
  const int l = bid * 256 + threadIdx.x;
  if (l >= nL) return;
  double acc[9];
#pragma unroll
  for (int i = 0; i < 9; i++) acc[i] = 0;
  const int jb = pt_start[l], je = pt_start[l + 1];
  for (int j0 = jb; j0 < je; j0 += 4) {
    // indices, then edges, then poses of a chunk of four are requested before the first use
    int id[4];
#pragma unroll
    for (int q = 0; q < 4; q++) id[q] = pt_edges[min(j0 + q, je - 1)];
    lba_edge ev[4];
#pragma unroll
    for (int q = 0; q < 4; q++) ev[q] = edges[id[q]];
    PoseQ Tv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) Tv[q] = poses[ev[q].pose];
    const double* X = points + 3 * (size_t)ev[0].point;       // every edge of the list observes this landmark
    const double Xl[3] = {X[0], X[1], X[2]};
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (j0 + q < je) {
        const lba_edge e = ev[q];
        const PoseQ T = Tv[q];
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
        edge_jacobians(T, r[0] + T.t[0], r[1] + T.t[1], r[2] + T.t[2], c, mono, A, B);
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
  for (int i = 0; i < 6; i++) Hll[6 * (size_t)l + i] = acc[i];
  for (int i = 0; i < 3; i++) bl[3 * (size_t)l + i] = acc[6 + i];
}

__global__ __launch_bounds__(256) void k_lin_all(int nP, int n_edges, const lba_edge* __restrict__ edges, const PoseQ* __restrict__ poses,
                                                const double* __restrict__ points, Cam c, Huber hb, const double* __restrict__ err,
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
__global__ __launch_bounds__(256) void k_errlin(int nP, int n_edges, const lba_edge* __restrict__ edges, const PoseQ* __restrict__ poses,
                                               const double* __restrict__ points, Cam c, Huber hb, double* __restrict__ err,
                                               double* __restrict__ chi2, const int* __restrict__ pose_col,
                                               const int* __restrict__ point_col, double* __restrict__ EB,
                                               const int* __restrict__ ps_start, const int* __restrict__ ps_edges,
                                               double* __restrict__ Hpp, double* __restrict__ bp, double* __restrict__ partial,
                                               unsigned* __restrict__ ticket, const double* __restrict__ scale_partial, int n_scale_partial,
                                               const int* __restrict__ ok_flag, HostRec* __restrict__ rec, unsigned seq, int n_edge_blocks,
                                               int nL, const int* __restrict__ pt_start, const int* __restrict__ pt_edges,
                                               double* __restrict__ Hll, double* __restrict__ bl, LmIn lm) {
  const int bid = (int)blockIdx.x;
  if (bid < nP) {
    lin_poses_block<true>(bid, ps_start, ps_edges, edges, poses, points, c, hb, err, chi2, Hpp, bp);
  } else if (bid < nP + n_edge_blocks) {
    linearize_block<true>(bid - nP, n_edges, edges, poses, points, c, hb, err, chi2, pose_col, point_col, EB,
                          TrialPublish{partial, ticket, scale_partial, n_scale_partial, ok_flag, rec, seq, n_edge_blocks, lm});
  } else {
    lin_points_block(bid - nP - n_edge_blocks, nL, pt_start, pt_edges, edges, poses, points, c, hb, Hll, bl);
  }
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
  // round trips per chunk instead of two per edge (the chain of dependent loads is what this kernel costs); same sum order
  const int jb = pt_start[l], je = pt_start[l + 1];
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
        for (int i = 0; i < 9; i++) acc[i] += v[q][i];
      }
  }
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
__global__ __launch_bounds__(256) void k_errlin_prep(int n_errlin_blocks, int n_sort_blocks, int* __restrict__ pf_edges_w, int* __restrict__ pf_col_w,
                                                    const int* __restrict__ pf_start, unsigned long long* __restrict__ lm_mask,
                                                    int n_unknowns, double* __restrict__ St, double* __restrict__ xzero, int n_zero,
                                                    int nP, int n_edges, const lba_edge* __restrict__ edges, const PoseQ* __restrict__ poses,
                                               const double* __restrict__ points, Cam c, Huber hb, double* __restrict__ err,
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
  } else {
    lin_points_block(bid - nP - n_edge_blocks, nL, pt_start, pt_edges, edges, poses, points, c, hb, Hll, bl);
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
      for (int k = 0; k < 64; k++) s += r[k];
      part[ps * kPass + o][q] = s;
    }
  }
  __syncthreads();
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
}

// Dense LDL^T (no pivoting) + solve, one workgroup, matrix in global memory (L2-resident), row-major full n x n.
// Fails (flag=0) on an exactly-zero / non-finite pivot, as Eigen::SimplicialLDLT would.
__global__ __launch_bounds__(1024) void k_ldlt(int n, double* S, const double* __restrict__ b, double* __restrict__ x,
                                              int* __restrict__ ok_flag, int use_lds) {
  extern __shared__ double sh[];
  double* D = sh;                 // n
  double* y = sh + n;             // n
  double* A = use_lds ? sh + 2 * (size_t)n : S;   // n*n in LDS when it fits one CU, else in place (L2-resident)
  __shared__ int s_ok;
  const int tid = threadIdx.x, nt = blockDim.x;
  if (use_lds)
    for (int i = tid; i < n * n; i += nt) A[i] = S[i];
  if (tid == 0) s_ok = 1;
  __syncthreads();
  for (int j = 0; j < n; j++) {
    // pivot
    if (tid == 0) {
      const double d = A[(size_t)j * n + j];
      if (d == 0.0 || !(d == d) || fabs(d) == INFINITY) s_ok = 0;
      D[j] = d;
    }
    __syncthreads();
    if (!s_ok) break;
    const double d = D[j];
    // column j of L (stored below the diagonal, unscaled copy kept in y as scratch)
    for (int i = j + 1 + tid; i < n; i += nt) {
      const double v = A[(size_t)i * n + j];
      y[i] = v;                       // L_ij * d
      A[(size_t)i * n + j] = v / d;   // L_ij
    }
    __syncthreads();
    // trailing update (lower triangle): A_ik -= L_ij * (L_kj * d)
    const int m = n - j - 1;
    for (int t = tid; t < m * m; t += nt) {
      const int i = j + 1 + t / m, k = j + 1 + t % m;
      if (k <= i) A[(size_t)i * n + k] -= A[(size_t)i * n + j] * y[k];
    }
    __syncthreads();
  }
  if (s_ok) {
    // forward: L z = b ; z/D ; backward: L^T x = z   (serial over columns, parallel over rows)
    for (int i = tid; i < n; i += nt) y[i] = b[i];
    __syncthreads();
    for (int j = 0; j < n; j++) {
      const double yj = y[j];
      for (int i = j + 1 + tid; i < n; i += nt) y[i] -= A[(size_t)i * n + j] * yj;
      __syncthreads();
    }
    for (int i = tid; i < n; i += nt) y[i] /= D[i];
    __syncthreads();
    for (int j = n - 1; j >= 0; j--) {
      const double xj = y[j];
      for (int i = tid; i < j; i += nt) y[i] -= A[(size_t)j * n + i] * xj;
      __syncthreads();
    }
    for (int i = tid; i < n; i += nt) x[i] = y[i];
  }
  if (tid == 0) *ok_flag = s_ok;
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

// Register-blocked dense LDL^T + solve of the reduced camera system: thread t owns the 6x6 block (i,k), i >= k, of the
// lower triangle in registers for the whole factorisation; per block column j: (1) the diagonal owner factors its block
// and forward-substitutes y_j, (2) panel owners compute L_ij = A_ij L_jj^-T D_j^-1 and fold L_ij y_j into the running
// rhs, (3) every trailing owner applies the rank-6 update A_ik -= L_ij (L_kj D_j)^T from the LDS-staged panel.  Two
// barriers per block column forward, two backward; no pivoting; zero / non-finite pivot => ok = 0.
constexpr int kPanStride = 74;   // doubles per panel row-block: L (36, padded to 37) + W (36, padded to 37): conflict-free b64 reads


// Row-pair variant of the register-blocked LDL^T: a 6x6 block is owned by THREE threads (two rows each), which cuts
// the per-step trailing update (the longest phase) and the panel solve to a third, and every thread of block column j
// factors the 6x6 diagonal block redundantly from an LDS copy, so no barrier is needed between "factor" and "panel":
// two barriers per block column.  L is kept (LDS, or in the storage of S for large systems) for the back-substitution.
// R = row pairs per thread (R = 1: up to 341 blocks = 25 poses with 1024 threads).
// Block sparsity of the factor (symbolic elimination on the host, fill-in included): bit i of m[j] = block L_ij is
// structurally non-zero.  The reference exploits the same sparsity through Eigen::SimplicialLDLT (G/solvers/linear_solver_eigen.h);
// here it prunes the trailing update: a block (i,k) is touched at step j only if both L_ij and L_kj exist.
struct LdltNz { unsigned long long m[64]; };

// L_IN_LDS is a template parameter so that the factor's pointer has ONE address space per instantiation: a run-time
// "LDS or global" select makes every access to L a flat_* instruction (aperture check, both wait counters).
//
// Schedule per block column j (two workgroup barriers):
//   barrier X | panel: the threads of column j read the factored diagonal block F_j = {X = L_jj^-1, 1/d, y} from LDS and
//             | turn their two rows of A_ij into L_ij, W_ij = L_ij D_j and the rhs update
//   barrier Z | trailing update A_ik -= L_ij W_kj^T; then the ONE wavefront that owns block (j+1, j+1) factors it
//             | (look-ahead of the diagonal only: its ~130-instruction dependent chain overlaps the other wavefronts'
//             | trailing updates instead of sitting between two barriers in front of every wavefront)
// A wavefront holds 21 blocks x 3 threads (lane 63 idles) so that the three owners of a block always share a wavefront
// and can exchange its rows through LDS without a workgroup barrier.
constexpr int kBlkPerWave = 21;
constexpr int kPanW = 37;            // offset of W inside a panel row-block (odd: conflict-free 64-bit reads across blocks)
typedef double ldlt_d2 __attribute__((ext_vector_type(2)));
// N doubles (N even) from / to a 16-byte aligned address as 128-bit accesses: an LDS instruction costs ~3 cycles of the
// CU's LDS pipe per wavefront for 8 bytes per lane and ~4 for 16, and the trailing update is bound by exactly that
template <int N>
__device__ __forceinline__ void ld_pairs(const double* __restrict__ p, double* out) {
  const ldlt_d2* q = reinterpret_cast<const ldlt_d2*>(__builtin_assume_aligned(p, 16));
#pragma unroll
  for (int i = 0; i < N / 2; i++) { const ldlt_d2 v = q[i]; out[2 * i] = v.x; out[2 * i + 1] = v.y; }
}
template <int N>
__device__ __forceinline__ void st_pairs(double* __restrict__ p, const double* in) {
  ldlt_d2* q = reinterpret_cast<ldlt_d2*>(__builtin_assume_aligned(p, 16));
#pragma unroll
  for (int i = 0; i < N / 2; i++) { ldlt_d2 v; v.x = in[2 * i]; v.y = in[2 * i + 1]; q[i] = v; }
}

template <int NT, int R, bool L_IN_LDS>
__global__ __launch_bounds__(NT) void k_ldlt_rows(int nb, double* __restrict__ S, const double* __restrict__ b,
                                                  double* __restrict__ x, int* __restrict__ ok_flag, LdltNz nz) {
#pragma clang fp contract(fast)        // the solve is tolerance-checked (1e-4), not bit-compared: fused multiply-adds halve the FP64 chain
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double* rr_ = sh;                    // 6*nb running rhs (forward)
  double* zz = rr_ + 6 * nb;           // 6*nb z = D^-1 L^-1 b, then running rhs of the backward pass
  double* Ajj = zz + 6 * nb;           // 36 next diagonal block (rows exchanged between its three owners)
  double* Fjj = Ajj + 36;              // 32 (12 used): 1/d (6), y (6) of the factored diagonal block
  double* pan = Fjj + 32;              // 2 * nb * kPanStride
  double* Lall;                        // nblk * 36
  if constexpr (L_IN_LDS) Lall = pan + 2 * (size_t)nb * kPanStride; else Lall = S;
  __shared__ int s_ok;
  const int t = threadIdx.x;
  const int n = 6 * nb;
  const int nblk = nb * (nb + 1) / 2;
  int ubi[R], ubk[R], upr[R], ublk[R];
  double a[R][12];
#pragma unroll
  for (int s = 0; s < R; s++) {
    const int u = t + s * NT;
    const int lane = u & 63;
    const int cb = (u >> 6) * kBlkPerWave + lane / 3;        // column-major block number
    ubi[s] = -1; ubk[s] = -1; upr[s] = 0; ublk[s] = 0;
    if (lane < 3 * kBlkPerWave && cb < nblk) {
      // Blocks are numbered COLUMN-major over the lower block triangle: the threads of block column j are consecutive,
      // so the panel of a column occupies one or two wavefronts instead of a few lanes of every wavefront.
      // column k starts at C(k) = k*nb - k(k-1)/2; invert with a float guess + fix-up
      const float fnb = (float)nb + 0.5f;
      int bk = (int)(fnb - sqrtf(fmaxf(fnb * fnb - 2.f * (float)cb, 0.f)));
      if (bk < 0) bk = 0;
      if (bk > nb - 1) bk = nb - 1;
      while (bk > 0 && bk * nb - bk * (bk - 1) / 2 > cb) bk--;
      while (bk + 1 < nb && (bk + 1) * nb - (bk + 1) * bk / 2 <= cb) bk++;
      const int bi = bk + (cb - (bk * nb - bk * (bk - 1) / 2));
      ubi[s] = bi; ubk[s] = bk; upr[s] = lane - 3 * (lane / 3); ublk[s] = bi * (bi + 1) / 2 + bk;   // storage stays row-major
#pragma unroll
      for (int q = 0; q < 2; q++)
#pragma unroll
        for (int c = 0; c < 6; c++) a[s][6 * q + c] = S[(size_t)(6 * bi + 2 * upr[s] + q) * n + 6 * ubk[s] + c];
    }
  }
  for (int i = t; i < n; i += NT) { rr_[i] = b[i]; zz[i] = 0; }
  if (t == 0) s_ok = 1;
  __syncthreads();            // S fully consumed before Lall (which may alias S) is written; rhs in LDS

  // Factor the diagonal block jn, executed by its three owners (one wavefront): rows -> LDS -> everyone reads the lower
  // triangle; right-looking LDL^T arranged for depth, explicit inverse X of the unit-triangular factor, y = X r_jn.
  // Publishes F (for the panel threads of column jn), the diagonal block of the stored factor and z_jn.
  auto factor_diag = [&](int jn, int sd) {
    int pr_d = 0;
#pragma unroll
    for (int s = 0; s < R; s++)
      if (s == sd) {
        pr_d = upr[s];
        st_pairs<12>(Ajj + 12 * upr[s], a[s]);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();       // LDS operations of one wavefront complete in order
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double A[6][6], dinv[6], y[6], rj[6];
    {
      double flat[36];
      ld_pairs<36>(Ajj, flat);
#pragma unroll
      for (int q = 0; q < 6; q++)
#pragma unroll
        for (int c = 0; c <= q; c++) A[q][c] = flat[6 * q + c];
    }
    ld_pairs<6>(rr_ + 6 * jn, rj);
    bool good = true;
#pragma unroll
    for (int c = 0; c < 6; c++) {
      const double d = A[c][c];
      if (d == 0.0 || !(d == d) || fabs(d) == INFINITY) good = false;
      const double id = fast_rcp(d);
      dinv[c] = id;
      double W[6];
#pragma unroll
      for (int q = c + 1; q < 6; q++) { W[q] = A[q][c]; A[q][c] = W[q] * id; }       // A[q][c] now holds L[q][c]
#pragma unroll
      for (int q = c + 1; q < 6; q++)
#pragma unroll
        for (int r = c + 1; r <= q; r++) A[q][r] -= A[q][c] * W[r];
    }
    if (!good) s_ok = 0;
    // X = L^-1 (unit lower triangular), column by column; the six columns are independent chains
    double X[6][6];
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
    // One lane publishes the results (an LDS store costs ~16-28 cycles of the wavefront's issue time whatever the number
    // of active lanes): X row-packed into the diagonal block of the stored factor (read by the panel threads of column
    // jn and by the backward pass), 1/d and y into F, z_jn.  L_jj itself is not needed by anyone.
    if (pr_d == 0) {
      double xs[16], dz[18];
#pragma unroll
      for (int c = 1; c < 6; c++)
#pragma unroll
        for (int m = 0; m < c; m++) xs[c * (c - 1) / 2 + m] = X[c][m];
      xs[15] = 0;
#pragma unroll
      for (int c = 0; c < 6; c++) { dz[c] = dinv[c]; dz[6 + c] = y[c]; dz[12 + c] = y[c] * dinv[c]; }
      st_pairs<16>(Lall + (size_t)(jn * (jn + 1) / 2 + jn) * 36, xs);
      st_pairs<12>(Fjj, dz);
      st_pairs<6>(zz + 6 * jn, dz + 12);
    }
  };

  {
    int sd = -1;
#pragma unroll
    for (int s = 0; s < R; s++) if (ubi[s] == 0 && ubk[s] == 0) sd = s;
    if (sd >= 0) factor_diag(0, sd);
  }
  for (int j = 0; j < nb; j++) {
    __syncthreads();          // barrier X: F_j + running rhs of column j complete
    if (!s_ok) break;         // uniform: s_ok is only written between barrier Z and the next barrier X
    double* P = pan + (size_t)(j & 1) * nb * kPanStride;
#pragma unroll
    for (int s = 0; s < R; s++) {
      if (ubk[s] != j || ubi[s] == j) continue;
      double X[6][6], dinv[6], y[6];
      {
        double f[16], g[12];
        ld_pairs<16>(Lall + (size_t)(j * (j + 1) / 2 + j) * 36, f);
        ld_pairs<12>(Fjj, g);
#pragma unroll
        for (int c = 1; c < 6; c++)
#pragma unroll
          for (int m = 0; m < c; m++) X[c][m] = f[c * (c - 1) / 2 + m];
#pragma unroll
        for (int c = 0; c < 6; c++) { dinv[c] = g[c]; y[c] = g[6 + c]; }
      }
      const int row0 = 2 * upr[s];
      double* Lg = Lall + (size_t)ublk[s] * 36;
      double* Lp = P + (size_t)ubi[s] * kPanStride;
      double* Wp = Lp + kPanW;
      // the running rhs is read before the panel is stored (LDS operations complete in order)
      double rhs[2];
      ld_pairs<2>(rr_ + 6 * ubi[s] + row0, rhs);
      double racc[2] = {0, 0}, wv[12], lv[12];
#pragma unroll
      for (int q = 0; q < 2; q++) {
#pragma unroll
        for (int c = 0; c < 6; c++) {
          double v = a[s][6 * q + c];                    // w = a L^-T : w[c] = a[c] + sum_{m<c} a[m] X[c][m]
#pragma unroll
          for (int m = 0; m < c; m++) v += a[s][6 * q + m] * X[c][m];
          const double l = v * dinv[c];
          wv[6 * q + c] = v;
          lv[6 * q + c] = l;
          racc[q] += l * y[c];
        }
      }
#pragma unroll
      for (int q = 0; q < 12; q++) { Wp[6 * row0 + q] = wv[q]; Lp[6 * row0 + q] = lv[q]; }
      st_pairs<12>(Lg + 6 * row0, lv);
      rhs[0] -= racc[0]; rhs[1] -= racc[1];
      st_pairs<2>(rr_ + 6 * ubi[s] + row0, rhs);
    }
    __syncthreads();          // barrier Z: panel of column j published
    const unsigned long long nzj = nb <= 64 ? nz.m[j] : ~0ull;   // rows with a non-zero block in column j
    int sd = -1;
#pragma unroll
    for (int s = 0; s < R; s++) {
      if (ubk[s] > j && ubi[s] >= ubk[s] && ((nzj >> (ubk[s] & 63)) & 1ull) && ((nzj >> (ubi[s] & 63)) & 1ull)) {
        const double* Lp = P + (size_t)ubi[s] * kPanStride + 6 * (2 * upr[s]);   // two rows of L_ij
        const double* Wp = P + (size_t)ubk[s] * kPanStride + kPanW;              // W_kj = L_kj D_j
        double l01[12];
#pragma unroll
        for (int m = 0; m < 12; m++) l01[m] = Lp[m];
#pragma unroll
        for (int c = 0; c < 6; c++) {
          double w[6];
#pragma unroll
          for (int m = 0; m < 6; m++) w[m] = Wp[6 * c + m];
          double s0 = 0, s1 = 0;
#pragma unroll
          for (int m = 0; m < 6; m++) { s0 += l01[m] * w[m]; s1 += l01[6 + m] * w[m]; }
          a[s][c] -= s0;
          a[s][6 + c] -= s1;
        }
      }
      if (ubi[s] == j + 1 && ubk[s] == j + 1) sd = s;
    }
    if (sd >= 0) factor_diag(j + 1, sd);
  }
  __syncthreads();
  const int ok = s_ok;
  if (ok && t < 64 && nb <= 20) {
    // Backward substitution by ONE wavefront with z in registers: per step the six entries of z_i are broadcast with
    // v_readlane, every lane forms x_i = X_i^T z_i from the stored inverse and updates its own entries
    // z_k -= L_ik^T x_i.  No LDS stores in the loop, so the loads of L (which do not depend on the chain) are issued ahead
    // of it.  The phase is bound by the instruction issue of this single wavefront (~5 cycles per instruction).
    // lane o < 60 holds z[o] (block rows 0..9) and z[60 + o] (block rows 10..19): a block row never straddles the two
    double z0 = t < 60 && t < n ? zz[t] : 0.0, z1 = t < 60 && t + 60 < n ? zz[t + 60] : 0.0;
    const int kk = t / 6, cc = t - 6 * kk;             // lane -> (block row kk or kk + 10, component cc)
    for (int i = nb - 1; i >= 0; i--) {
      const double* Lrow = Lall + (size_t)(i * (i + 1) / 2) * 36;
      const double* Xp = Lrow + (size_t)i * 36;        // X_i row-packed: X[q][c] at q(q-1)/2 + c
      double xp[16], l0[6], l1[6];
      ld_pairs<16>(Xp, xp);
      const bool on0 = t < 60 && kk < i, on1 = t < 60 && kk + 10 < i;
      const double* L0 = Lrow + (size_t)(on0 ? kk : 0) * 36 + cc;
      const double* L1 = Lrow + (size_t)(on1 ? kk + 10 : 0) * 36 + cc;
#pragma unroll
      for (int q = 0; q < 6; q++) { l0[q] = L0[6 * q]; l1[q] = L1[6 * q]; }
      const double src = i < 10 ? z0 : z1;             // uniform select
      const int lb = 6 * (i < 10 ? i : i - 10);
      double zi[6], xv[6];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        const int lo = __builtin_amdgcn_readlane((int)(__double_as_longlong(src) & 0xFFFFFFFFll), lb + c);
        const int hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(src) >> 32), lb + c);
        zi[c] = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
      }
#pragma unroll
      for (int c = 0; c < 6; c++) {                    // x_i = X_i^T z_i: x[c] = z[c] + sum_{q>c} X[q][c] z[q], two partial sums
        double va = zi[c], vb = 0;
#pragma unroll
        for (int q = c + 1; q < 6; q++) { if ((q - c) & 1) va += xp[q * (q - 1) / 2 + c] * zi[q]; else vb += xp[q * (q - 1) / 2 + c] * zi[q]; }
        xv[c] = va + vb;
      }
      const double a0 = (l0[0] * xv[0] + l0[1] * xv[1]) + (l0[2] * xv[2] + l0[3] * xv[3]) + (l0[4] * xv[4] + l0[5] * xv[5]);
      const double a1 = (l1[0] * xv[0] + l1[1] * xv[1]) + (l1[2] * xv[2] + l1[3] * xv[3]) + (l1[4] * xv[4] + l1[5] * xv[5]);
      const double xs_ = cc == 0 ? xv[0] : cc == 1 ? xv[1] : cc == 2 ? xv[2] : cc == 3 ? xv[3] : cc == 4 ? xv[4] : xv[5];
      z0 = on0 ? z0 - a0 : (kk == i ? xs_ : z0);       // the solved block row replaces z in place
      z1 = on1 ? z1 - a1 : (kk + 10 == i ? xs_ : z1);
    }
    if (t < 60) {
      if (t < n) x[t] = z0;
      if (t + 60 < n) x[t + 60] = z1;
    }
  } else if (ok && t < 64) {
    // Backward substitution by ONE wavefront, no workgroup barriers: x_i = L_ii^-T z_i as six dot products with the stored
    // inverse, then z_k -= L_ik^T x_i for all k < i spread over the lanes (lane -> (k, c)); LDS accesses of a wavefront
    // are ordered, so the steps chain without synchronisation.
    for (int i = nb - 1; i >= 0; i--) {
      const double* Lii = Lall + (size_t)(i * (i + 1) / 2 + i) * 36;
      double zi[6], xv[6];
#pragma unroll
      for (int c = 0; c < 6; c++) zi[c] = zz[6 * i + c];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double v = zi[c];
#pragma unroll
        for (int q = c + 1; q < 6; q++) v += Lii[q * (q - 1) / 2 + c] * zi[q];     // X_i row-packed
        xv[c] = v;
      }
      if (t < 6) {
#pragma unroll
        for (int c = 0; c < 6; c++) if (t == c) x[6 * i + c] = xv[c];
      }
      for (int o = t; o < 6 * i; o += 64) {
        const int k = o / 6, c = o - 6 * k;
        const double* Lik = Lall + (size_t)(i * (i + 1) / 2 + k) * 36;
        double acc = 0;
#pragma unroll
        for (int q = 0; q < 6; q++) acc += Lik[6 * q + c] * xv[q];
        zz[o] -= acc;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (t == 0) *ok_flag = ok;
}

// Dataflow variant of the row-pair block LDL^T for systems whose factor AND W = L D fit in LDS (nb <= 20 poses):
// no workgroup barriers inside the factorisation.  Every block column lives in ONE wavefront (FlowMap, packed by the
// host), which applies the updates of the earlier columns to its blocks as their panels appear, then factors its diagonal
// block (every lane of the column redundantly -- free in SIMD, and nothing has to be published for the panel), computes
// its panel and raises the column counter.  The rows of the diagonal block reach the column's lanes through v_readlane
// (their owners are the first three lanes of the column's lane range), not through LDS, and the diagonal lane stores its
// own results (X for the backward pass, z_j) only after the counter has moved.  The other wavefronts apply a column's update whenever they get to it, so the
// critical path per column is one update + factor + panel of a single wavefront (~2.9k cycles) instead of two
// barrier-separated phases of the whole workgroup (~4.4k).
struct FlowMap { unsigned char c0[16], c1[16]; };      // wavefront w owns block columns [c0[w], c1[w])
__device__ __forceinline__ double po_readlane_any(double v, int lane) {   // lane must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(1024) void k_ldlt_flow(int nb, double* __restrict__ S, const double* __restrict__ b,
                                                    double* __restrict__ x, int* __restrict__ ok_flag, LdltNz nz, FlowMap map) {
#pragma clang fp contract(fast)        // the solve is tolerance-checked (1e-4), not bit-compared
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double* rr_ = sh;                    // 6*nb running rhs (forward)
  double* zz = rr_ + 6 * nb;           // 6*nb z = D^-1 L^-1 b
  // factor: one record of kPanStride doubles per block, COLUMN-major block order (consecutive lanes <-> consecutive
  // records: 64-bit accesses of a wavefront spread over the banks): L_ij at +0 (the diagonal records hold X row-packed),
  // W_ij = L_ij D_j at +38 (16-byte aligned: 128-bit accesses; a single wavefront pays 25-40 cycles per LDS INSTRUCTION)
  double* Pan = zz + 6 * nb;
  auto rec = [nb](int i, int k_) { return (size_t)(k_ * nb - k_ * (k_ - 1) / 2 + (i - k_)) * kPanStride; };
  __shared__ int s_ok;
  __shared__ int s_done;               // number of block columns whose panel is complete
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int n = 6 * nb;
  const int c0 = map.c0[wv], c1 = map.c1[wv];
  // lane -> block of this wavefront's columns (column-major), three lanes per block
  int bi = -1, bk = -1;
  const int pr = lane - 3 * (lane / 3);
  {
    int q = lane / 3;
    if (lane < 63)
      for (int c = c0; c < c1; c++) {
        const int len = nb - c;
        if (q < len) { bk = c; bi = c + q; break; }
        q -= len;
      }
  }
  const size_t blk = bi >= 0 ? rec(bi, bk) : 0;
  double a[12];
#pragma unroll
  for (int q = 0; q < 12; q++) a[q] = 0;
  if (bi >= 0) {
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int c = 0; c < 6; c++) a[6 * q + c] = S[(size_t)(6 * bi + 2 * pr + q) * n + 6 * bk + c];
  }
  for (int i = t; i < n; i += 1024) { rr_[i] = b[i]; zz[i] = 0; }
  if (t == 0) { s_ok = 1; s_done = 0; }
  __syncthreads();
  if (c0 == 0 && c1 > 0) __builtin_amdgcn_s_setprio(3);
  // the column counter is accessed through the __shared__ object itself: a generic `volatile int*` turns the poll and the
  // publish into flat_load / flat_store with sc0 sc1 and a vmcnt(0) wait -- a ~1.5 k-cycle round trip per hand-over
  for (int j = 0; j < c1; j++) {       // a wavefront is finished once its last column is factored
    // the wavefront whose column comes next is on the critical path: it must not share its SIMD's issue slots evenly with
    // wavefronts that are merely catching up on trailing updates
    if (j + 1 == c0) __builtin_amdgcn_s_setprio(3);
    if (j >= c0) {
      // ---- this wavefront owns column j: all earlier updates are applied (loop order)
      // the diagonal block's rows sit in the first three lanes of this column's lane range: broadcast the lower triangle
      // with v_readlane (42 scalar-broadcast instructions, no LDS round trip: an LDS exchange costs ~1.5 k cycles here)
      int l0 = 0;
      for (int c = c0; c < j; c++) l0 += 3 * (nb - c);
      double A[6][6], dinv[6], y[6], rj[6];
#pragma unroll
      for (int q = 0; q < 6; q++)
#pragma unroll
        for (int c = 0; c <= q; c++) A[q][c] = po_readlane_any(a[6 * (q & 1) + c], l0 + (q >> 1));
      if (bk == j) {
        ld_pairs<6>(rr_ + 6 * j, rj);
        bool good = true;
#pragma unroll
        for (int c = 0; c < 6; c++) {
          const double d = A[c][c];
          if (d == 0.0 || !(d == d) || fabs(d) == INFINITY) good = false;
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
        if (!good) s_ok = 0;
        double X[6][6];
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
        if (bi != j) {
          double rhs[2];
          ld_pairs<2>(rr_ + 6 * bi + 2 * pr, rhs);
          double racc[2] = {0, 0}, wv_[12], lv[12];
#pragma unroll
          for (int q = 0; q < 2; q++) {
#pragma unroll
            for (int c = 0; c < 6; c++) {
              double v = a[6 * q + c];                    // w = a L^-T : w[c] = a[c] + sum_{m<c} a[m] X[c][m]
#pragma unroll
              for (int m = 0; m < c; m++) v += a[6 * q + m] * X[c][m];
              const double l = v * dinv[c];
              wv_[6 * q + c] = v;
              lv[6 * q + c] = l;
              racc[q] += l * y[c];
            }
          }
          st_pairs<12>(Pan + blk + 38 + 12 * pr, wv_);
          st_pairs<12>(Pan + blk + 12 * pr, lv);
          rhs[0] -= racc[0]; rhs[1] -= racc[1];
          st_pairs<2>(rr_ + 6 * bi + 2 * pr, rhs);
        }
        // publish: every LDS store of this wavefront is complete before the counter moves
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (bi == j && pr == 0) {
          __hip_atomic_store(&s_done, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          // the diagonal lane's own results (X row-packed for the backward pass, z_j) are not needed by the next column:
          // they are stored after the counter has moved
          double xs[16], zs[6];
#pragma unroll
          for (int c = 1; c < 6; c++)
#pragma unroll
            for (int m = 0; m < c; m++) xs[c * (c - 1) / 2 + m] = X[c][m];
          xs[15] = 0;
#pragma unroll
          for (int c = 0; c < 6; c++) zs[c] = y[c] * dinv[c];
          st_pairs<16>(Pan + blk, xs);
          st_pairs<6>(zz + 6 * j, zs);
        }
      }
    } else {
      // ---- wait for the panel of column j (uniform spin on the LDS counter)
      while (__hip_atomic_load(&s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= j) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // ---- trailing update with column j for this wavefront's blocks right of it
    const unsigned long long nzj = nz.m[j];
    if (bk > j && ((nzj >> (bk & 63)) & 1ull) && ((nzj >> (bi & 63)) & 1ull)) {
      const double* Lp = Pan + rec(bi, j) + 12 * pr;     // two rows of L_ij
      const double* Wp = Pan + rec(bk, j) + 38;          // W_kj = L_kj D_j
      double l01[12];
#pragma unroll
      for (int m = 0; m < 12; m++) l01[m] = Lp[m];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double w[6];
#pragma unroll
        for (int m = 0; m < 6; m++) w[m] = Wp[6 * c + m];
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int m = 0; m < 6; m++) { s0 += l01[m] * w[m]; s1 += l01[6 + m] * w[m]; }
        a[c] -= s0;
        a[6 + c] -= s1;
      }
    }
  }
  __builtin_amdgcn_s_setprio(0);
  __syncthreads();
  const int ok = s_ok;
  if (ok && t < 64) {
    // backward substitution by one wavefront with z in registers (see k_ldlt_rows)
    double z0 = t < 60 && t < n ? zz[t] : 0.0, z1 = t < 60 && t + 60 < n ? zz[t + 60] : 0.0;
    const int kk = t / 6, cc = t - 6 * kk;
    for (int i = nb - 1; i >= 0; i--) {
      const double* Xp = Pan + rec(i, i);
      double xp[16], l0[6], l1[6];
      ld_pairs<16>(Xp, xp);
      const bool on0 = t < 60 && kk < i, on1 = t < 60 && kk + 10 < i;
      const double* L0 = Pan + rec(i, on0 ? kk : 0) + cc;
      const double* L1 = Pan + rec(i, on1 ? kk + 10 : 0) + cc;
#pragma unroll
      for (int q = 0; q < 6; q++) { l0[q] = L0[6 * q]; l1[q] = L1[6 * q]; }
      const double src = i < 10 ? z0 : z1;
      const int lb = 6 * (i < 10 ? i : i - 10);
      double zi[6], xv[6];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        const int lo = __builtin_amdgcn_readlane((int)(__double_as_longlong(src) & 0xFFFFFFFFll), lb + c);
        const int hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(src) >> 32), lb + c);
        zi[c] = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
      }
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double va = zi[c], vb = 0;
#pragma unroll
        for (int q = c + 1; q < 6; q++) { if ((q - c) & 1) va += xp[q * (q - 1) / 2 + c] * zi[q]; else vb += xp[q * (q - 1) / 2 + c] * zi[q]; }
        xv[c] = va + vb;
      }
      const double a0 = (l0[0] * xv[0] + l0[1] * xv[1]) + (l0[2] * xv[2] + l0[3] * xv[3]) + (l0[4] * xv[4] + l0[5] * xv[5]);
      const double a1 = (l1[0] * xv[0] + l1[1] * xv[1]) + (l1[2] * xv[2] + l1[3] * xv[3]) + (l1[4] * xv[4] + l1[5] * xv[5]);
      const double xs_ = cc == 0 ? xv[0] : cc == 1 ? xv[1] : cc == 2 ? xv[2] : cc == 3 ? xv[3] : cc == 4 ? xv[4] : xv[5];
      z0 = on0 ? z0 - a0 : (kk == i ? xs_ : z0);
      z1 = on1 ? z1 - a1 : (kk + 10 == i ? xs_ : z1);
    }
    if (t < 60) {
      if (t < n) x[t] = z0;
      if (t + 60 < n) x[t + 60] = z1;
    }
  }
  if (t == 0) *ok_flag = ok;
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
  // fixed-order block sum -> one partial per block (the last block of the following k_errors adds them up in index order)
  red[threadIdx.x] = sc;
  __syncthreads();
  for (int s2 = NT / 2; s2 > 0; s2 >>= 1) {
    if ((int)threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x == 0) scale_partial[blockIdx.x] = red[0];
}

// ---- the solve and the update in ONE launch (windows of <= 20 free poses: the column LDL^T).  Workgroup 0 is ldltm::ldlt_cols_body
// (19 us on one compute unit); the other workgroups are k_update's work for 512 vertices each: they request everything that does
// not depend on the solution -- a landmark's right-hand side, its first four Hpl blocks, Hll; a pose's state -- and then wait for
// workgroup 0 to publish "x is ready" (a sequence number, release / acquire at agent scope: the workgroups sit on different XCDs).
// What is left after the wait is arithmetic and one short round trip for x: the ~8 us of dependent loads of k_update and one
// kernel boundary disappear from every LM iteration.  Same operations and order per vertex as k_update; the per-block partial sums
// of computeScale() are over 512 threads here.
struct UpdArgs {
  int n_poses, n_points, nP;
  const int* pose_col; const int* point_col; const PoseQ* poses; const double* points; double* x;
  const int* pf_start; const int* pf_edges; const int* pf_col; const double* EB; const double* Hll; const double* bl; double lambda_v;
  PoseQ* poses_out; double* points_out; const double* bp; double* scale_partial; const double* lambda_p;
  int ldlt_prio;
};
constexpr int kFusedUpdThreads = ldltm::kThreads;

__device__ __forceinline__ void update_after_solve_block(int ub, const UpdArgs& a, const unsigned* __restrict__ x_ready, unsigned seq) {
  extern __shared__ __attribute__((aligned(16))) double upd_sh[];     // the launch's dynamic LDS (the LDL^T workgroup's store): 4 KB of it
  double* const red = upd_sh;
  const int i = ub * kFusedUpdThreads + (int)threadIdx.x;
  const bool is_point = i < a.n_points, is_pose = !is_point && i < a.n_points + a.n_poses;
  // ---- before the solution exists
  int l = -1, j0 = 0, j1 = 0, c = -1;
  double cl[3] = {0, 0, 0}, h6[6] = {1, 0, 0, 1, 0, 1}, Xin[3] = {0, 0, 0}, blv[3] = {0, 0, 0}, bpv[6] = {0, 0, 0, 0, 0, 0};
  int eid[4] = {0, 0, 0, 0}, col[4] = {0, 0, 0, 0}, eid2[4] = {0, 0, 0, 0}, col2[4] = {0, 0, 0, 0};
  double Bv[4][18];
  PoseQ Tin;
  if (is_point) {
    l = a.point_col[i];
    Xin[0] = a.points[3 * (size_t)i]; Xin[1] = a.points[3 * (size_t)i + 1]; Xin[2] = a.points[3 * (size_t)i + 2];
    if (l >= 0) {
#pragma unroll
      for (int q = 0; q < 3; q++) { blv[q] = a.bl[3 * (size_t)l + q]; cl[q] = blv[q]; }
#pragma unroll
      for (int q = 0; q < 6; q++) h6[q] = a.Hll[6 * (size_t)l + q];
      j0 = a.pf_start[l]; j1 = a.pf_start[l + 1];
      if (j1 > j0) {
#pragma unroll
        for (int u = 0; u < 4; u++) { const int j = min(j0 + u, j1 - 1); eid[u] = a.pf_edges[j]; col[u] = a.pf_col[j]; }
#pragma unroll
        for (int u = 0; u < 4; u++) { const int j = min(j0 + 4 + u, j1 - 1); eid2[u] = a.pf_edges[j]; col2[u] = a.pf_col[j]; }   // (indices only)
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const double* Bi = a.EB + (size_t)eid[u] * kEB;
#pragma unroll
          for (int q = 0; q < 18; q++) Bv[u][q] = Bi[q];
        }
      }
    }
  } else if (is_pose) {
    const int p = i - a.n_points;
    c = a.pose_col[p];
    Tin = a.poses[p];
    if (c >= 0) {
#pragma unroll
      for (int q = 0; q < 6; q++) bpv[q] = a.bp[6 * (size_t)c + q];
    }
  }
  const double lambda = a.lambda_p ? *a.lambda_p : a.lambda_v;
  // ---- wait for workgroup 0
  while (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(x_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != (int)seq) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // pairs with workgroup 0's agent-scope release (x itself is also read with agent-scope loads below)
  auto ldx = [&](size_t k) { return __hip_atomic_load(&a.x[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  // ---- after
  double sc = 0;
  if (is_point) {
    double dx[3] = {0, 0, 0};
    if (l >= 0) {
      if (j1 > j0) {
        double xv[4][6];
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
          for (int q = 0; q < 6; q++) xv[u][q] = ldx(6 * (size_t)col[u] + q);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (j0 + u < j1) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++)
#pragma unroll
              for (int q = 0; q < 6; q++) cl[cc] -= Bv[u][3 * q + cc] * xv[u][q];
          }
      }
      for (int jb = j0 + 4; jb < j1; jb += 4) {                   // landmarks with more than four free observations
        int e2[4], c2[4];
        if (jb == j0 + 4) {
#pragma unroll
          for (int u = 0; u < 4; u++) { e2[u] = eid2[u]; c2[u] = col2[u]; }       // observations 4..7: indices came in before the wait
        } else {
#pragma unroll
          for (int u = 0; u < 4; u++) { const int j = min(jb + u, j1 - 1); e2[u] = a.pf_edges[j]; c2[u] = a.pf_col[j]; }
        }
        double B2[4][18], x2[4][6];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const double* Bi = a.EB + (size_t)e2[u] * kEB;
#pragma unroll
          for (int q = 0; q < 18; q++) B2[u][q] = Bi[q];
#pragma unroll
          for (int q = 0; q < 6; q++) x2[u][q] = ldx(6 * (size_t)c2[u] + q);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (jb + u < j1) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++)
#pragma unroll
              for (int q = 0; q < 6; q++) cl[cc] -= B2[u][3 * q + cc] * x2[u][q];
          }
      }
      double Dinv[9];
      inv3_sym(h6, lambda, Dinv);
      for (int q = 0; q < 3; q++) {
        dx[q] = Dinv[3 * q] * cl[0] + Dinv[3 * q + 1] * cl[1] + Dinv[3 * q + 2] * cl[2];
        a.x[6 * (size_t)a.nP + 3 * (size_t)l + q] = dx[q];
        sc += dx[q] * (lambda * dx[q] + blv[q]);
      }
    }
    for (int q = 0; q < 3; q++) a.points_out[3 * (size_t)i + q] = Xin[q] + dx[q];
  } else if (is_pose) {
    const int p = i - a.n_points;
    if (c >= 0) {
      double xc6[6];
#pragma unroll
      for (int q = 0; q < 6; q++) xc6[q] = ldx(6 * (size_t)c + q);
      pose_oplus(Tin, xc6, &a.poses_out[p]);
      for (int q = 0; q < 6; q++) sc += xc6[q] * (lambda * xc6[q] + bpv[q]);
    } else {
      a.poses_out[p] = Tin;
    }
  }
  // fixed-order block sum: DPP tree per wavefront, the eight wave totals in wave order
  const double wsum = wave_sum_f64(sc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = wsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0;
    for (int w2 = 0; w2 < kFusedUpdThreads / 64; w2++) tot += red[w2];
    a.scale_partial[ub] = tot;
  }
}

__global__ __launch_bounds__(ldltm::kThreads) void k_ldlt_cols_update(int n, const double* __restrict__ St, double* __restrict__ x,
                                                                      int* __restrict__ ok_flag, unsigned* __restrict__ x_ready, unsigned seq,
                                                                      UpdArgs ua) {
  if (blockIdx.x == 0) {
    if (ua.ldlt_prio) __builtin_amdgcn_s_setprio(3);     // the chain of dependent pivots: its wavefronts issue ahead of any neighbour's on the CU
    ldltm::ldlt_cols_body<true>(n, St, x, ok_flag);      // x leaves through agent-scope stores of wavefront 0
    __syncthreads();
    if (threadIdx.x == 0) {
      // thread 0 belongs to the wavefront that stored x.  Agent-scope RELEASE on the publication: the compiler emits the L2
      // write-back and s_waitcnt vmcnt(0) in front of the word's store, so that x is visible to every XCD before the word is
      // (a workgroup-scope fence compiles to lgkmcnt(0) only; x and the word live in different allocations = different channels)
      __hip_atomic_store(x_ready, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  update_after_solve_block((int)blockIdx.x - 1, ua, x_ready, seq);
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
    for (int i = 0; i < n_partial && i < 1024; i++) chi += parts[i];
    for (int i = 1024; i < n_partial; i++) chi += partial[i];
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
// A/B, test and experiment switches of the solve (ORBG_* environment variables): read ONCE, when the handle is created -- the solve
// path itself never calls getenv.  A test that wants another variant creates another handle.
struct LbaSwitches {
  bool blit = false, host_items = false, host_lists = false, no_fuse = false, no_first2 = false, host_csr = false, dev_csr = false;
  bool ldlt_valu = false, ldlt_rows = false, ldlt_wide = false, ldlt_dense = false, fuse_update = false, no_spec = false;
  bool no_export_fuse = false, ldlt_prio = false, old_passes = false;
  int upd_threads = 64;              // k_update's workgroup size (ORBG_UPD_THREADS = 64 / 128 / 256)
  ldltm::Switches ldlt;              // which matrix-core kernel a size gets (ORBG_LDLT_TILES / _T9_4W / _8W)
  static LbaSwitches from_env() {
    LbaSwitches w;
    auto on = [](const char* k) { return getenv(k) != nullptr; };
    w.blit = on("ORBG_LBA_BLIT"); w.host_items = on("ORBG_HOST_ITEMS"); w.host_lists = on("ORBG_HOST_LISTS"); w.no_fuse = on("ORBG_NO_FUSE");
    w.no_first2 = on("ORBG_NO_FIRST2"); w.host_csr = on("ORBG_HOST_CSR"); w.dev_csr = on("ORBG_DEV_CSR"); w.ldlt_valu = on("ORBG_LDLT_VALU");
    w.ldlt_rows = on("ORBG_LDLT_ROWS"); w.ldlt_wide = on("ORBG_LDLT_WIDE"); w.ldlt_dense = on("ORBG_LDLT_DENSE"); w.no_spec = on("ORBG_NO_SPEC");
    w.no_export_fuse = on("ORBG_NO_EXPORT_FUSE"); w.ldlt_prio = on("ORBG_LDLT_PRIO"); w.old_passes = on("ORBG_LBA_OLD_PASSES");
    if (const char* e = getenv("ORBG_FUSE_UPDATE")) w.fuse_update = atoi(e) != 0;
    if (const char* e = getenv("ORBG_UPD_THREADS")) { const int v = atoi(e); w.upd_threads = (v == 256 || v == 128) ? v : 64; }
    w.ldlt = ldltm::Switches::from_env();
    return w;
  }
};
struct lba_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  bool ext_stream = false;             // `stream` was handed in through lba_set_stream (never destroyed here)
  LbaSwitches sw;
  ldltm::AttrCache ldlt_attr;          // which kernels of THIS handle's device already allow their dynamic LDS size
  size_t flow_attr = 0, fused_attr = 0;
  DevBuf<lba_edge> d_edges;
  PinnedBuf<lba_edge> edges_pin;       // the caller's edge list, copied (and validated, counted) in ONE pass
  DevBuf<PairItem> d_items_dev;        // pair items built by k_build_items (fixed-capacity segment per pose pair)
  DevBuf<int> d_pair_count;
  DevBuf<PoseQ> d_poses[2];
  DevBuf<double> d_points[2];
  DevBuf<unsigned> d_xready;           // "x is ready" sequence word of the fused LDL^T + update launch
  unsigned xseq = 0;
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
  h->d_edges.release(); h->edges_pin.release(); h->d_items_dev.release(); h->d_pair_count.release(); h->d_poses[0].release(); h->d_poses[1].release(); h->d_points[0].release(); h->d_points[1].release(); h->d_xready.release();
  h->d_err.release(); h->d_chi2.release(); h->d_partial.release(); h->d_EB.release(); h->d_Hll.release(); h->d_bl.release();
  h->d_Hpp.release(); h->d_bp.release(); h->d_S.release(); h->d_bs.release(); h->d_x.release(); h->d_St.release(); h->d_wfac.release();
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
static int upload_arena(lba_handle* h, size_t off0, size_t off1, hipStream_t st, bool blit) {
  if (off1 <= off0) return ORBG_OK;
  if (blit) { ORBG_HIP(hipMemcpyAsync(h->up_d.p + off0, h->up_h.h + off0, off1 - off0, hipMemcpyHostToDevice, st)); return ORBG_OK; }
  const unsigned n16 = (unsigned)((off1 - off0 + 15) / 16);           // offsets are multiples of 64, the arena has 64 bytes of slack
  hipLaunchKernelGGL(k_upload16, dim3((n16 + 255) / 256), dim3(256), 0, st, reinterpret_cast<const uint4*>(h->up_h.d + off0),
                     reinterpret_cast<uint4*>(h->up_d.p + off0), n16);
  ORBG_HIP(hipGetLastError());
  return ORBG_OK;
}

static int lba_solve_impl(lba_handle* h, const lba_problem* p, StopRef stop_ref, lba_result* r) {
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
    if (NE < 65536 && !sw.old_passes) {
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
  if (NE > 0) {
    if (sw.blit) {
      ORBG_HIP(hipMemcpyAsync(h->d_edges.p, edges, sizeof(lba_edge) * (size_t)NE, hipMemcpyHostToDevice, st));
    } else {                                               // (k_upload16 below: the runtime's blit takes ~50 us for these 190 KB)
      const unsigned n16 = (unsigned)((sizeof(lba_edge) * (size_t)NE + 15) / 16);
      hipLaunchKernelGGL(k_upload16, dim3((n16 + 255) / 256), dim3(256), 0, st, reinterpret_cast<const uint4*>(h->edges_pin.d),
                         reinterpret_cast<uint4*>(h->d_edges.p), n16);
    }
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
  const bool dev_items = nP >= 1 && nP <= 64 && (size_t)n_pairs_all * (size_t)item_cap * sizeof(PairItem) <= ((size_t)64 << 20) &&
                         !sw.host_items;
  // (the lists of free observations per landmark, still in edge order, go along when the device sorts them: k_prep / k_errlin_prep)
  const bool dev_lists = dev_items && nP >= 1 && ldltm::supports(6 * nP) && !sw.ldlt_valu && !sw.host_lists;
  // ... and the device fills the lists itself (k_csr_fill / k_csr_sort) when a pose's list fits the sorting workgroup
  // (measured: a wash at C2 -- 7.7 + 15.2 us of kernels for a 29 us host pass -- and -20 us at C4: used from 16 k edges on;
  // ORBG_DEV_CSR=1 forces it, ORBG_HOST_CSR=1 forbids it)
  const bool dev_csr = dev_lists && NE > 0 && nL > 0 && max_pose_edges <= kCsrPoseCap && !sw.no_fuse && !sw.no_first2 &&
                       !sw.host_csr && (NE >= 16384 || sw.dev_csr);
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
  unsigned long long adj[64];                            // adj[i]: poses sharing a landmark with pose i (dev_items)
  for (int i = 0; i < 64; i++) adj[i] = 0;
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
    if (runs == nL && !sw.old_passes) {
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
        if (pc >= 0) { ps_edges[f2[pc]++] = k; pf_edges[f3[lc]++] = k; }
      }
    }
  }
  const double t_s2 = now_s();
  LdltNz ldlt_nz;
  for (int j = 0; j < 64; j++) ldlt_nz.m[j] = ~0ull;
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
  const bool blit = sw.blit;       // A/B switch: the runtime's copies
  if ((rc = upload_arena(h, 0, off_a, st, blit))) return rc;
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
  if (!h->d_xready.p) {
    // the "x is ready" word starts at zero (a recycled allocation may hold another handle's last sequence number); sequence
    // numbers start at 1
    if ((rc = h->d_xready.reserve(4))) return rc;
    ORBG_HIP(hipMemsetAsync(h->d_xready.p, 0, 4 * sizeof(unsigned), h->stream));
    h->xseq = 0;
  }
  PoseQ* const posesB[3] = {reinterpret_cast<PoseQ*>(h->up_d.p + o_poses), h->d_poses[1].p, h->d_poses[0].p};
  double* const pointsB[3] = {reinterpret_cast<double*>(h->up_d.p + o_points), h->d_points[1].p, h->d_points[0].p};

  const double t_c = now_s();
  Cam cam{p->fx, p->fy, p->cx, p->cy, p->bf, p->bf};
  Huber hb;
  hb.delta_mono = (float)std::sqrt(5.991); hb.dsqr_mono = hb.delta_mono * hb.delta_mono;          // S/Optimizer.cc:1991-1992
  hb.delta_stereo = (float)std::sqrt(7.815); hb.dsqr_stereo = hb.delta_stereo * hb.delta_stereo;
  // k_ldlt keeps the matrix in LDS when it fits (n*n + 2n doubles <= 160 KiB), otherwise works in place in L2
  size_t lds_need = ((size_t)n * n + 2 * (size_t)n) * sizeof(double);
  const bool ldlt_lds = lds_need <= 150 * 1024;
  if (!ldlt_lds) lds_need = 2 * (size_t)n * sizeof(double);
  if (lds_need > 64 * 1024) {
    ORBG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ldlt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));
  }

  // row-pair LDL^T: R row pairs per thread, L kept in LDS when it fits next to the panels
  int rows_R = 0, rows_l_in_lds = 0;
  bool rows_small = false;
  size_t rows_lds = 0;
  {
    const size_t nblk = (size_t)nP * (nP + 1) / 2;            // a wavefront holds kBlkPerWave blocks (3 threads each)
    rows_small = nblk <= 10 * kBlkPerWave;
    if (nblk <= 16 * kBlkPerWave) rows_R = 1; else if (nblk <= 32 * kBlkPerWave) rows_R = 2; else if (nblk <= 64 * kBlkPerWave) rows_R = 4;
    const size_t base = (12 * (size_t)nP + 36 + 32 + 2 * (size_t)nP * kPanStride) * sizeof(double);
    const size_t lall = (size_t)nP * (nP + 1) / 2 * 36 * sizeof(double);
    rows_l_in_lds = base + lall <= 150 * 1024;
    rows_lds = base + (rows_l_in_lds ? lall : 0);
    if (rows_R && rows_lds > 64 * 1024) {
      const void* fn;
      if (rows_l_in_lds)
        fn = (rows_R == 1 && rows_small) ? reinterpret_cast<const void*>(k_ldlt_rows<640, 1, true>)
           : rows_R == 1 ? reinterpret_cast<const void*>(k_ldlt_rows<1024, 1, true>)
           : rows_R == 2 ? reinterpret_cast<const void*>(k_ldlt_rows<1024, 2, true>)
                         : reinterpret_cast<const void*>(k_ldlt_rows<1024, 4, true>);
      else
        fn = (rows_R == 1 && rows_small) ? reinterpret_cast<const void*>(k_ldlt_rows<640, 1, false>)
           : rows_R == 1 ? reinterpret_cast<const void*>(k_ldlt_rows<1024, 1, false>)
           : rows_R == 2 ? reinterpret_cast<const void*>(k_ldlt_rows<1024, 2, false>)
                         : reinterpret_cast<const void*>(k_ldlt_rows<1024, 4, false>);
      ORBG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rows_lds));
    }
  }
  // dataflow LDL^T (k_ldlt_flow): whole block columns per wavefront, packed greedily in column order
  FlowMap flow_map;
  bool use_flow = false;
  size_t flow_lds = 0;
  if (nP >= 1 && nP <= 20 && !sw.ldlt_rows) {
    int w = 0, fill = 0;
    for (int i = 0; i < 16; i++) { flow_map.c0[i] = 0; flow_map.c1[i] = 0; }
    bool fits = true;
    flow_map.c0[0] = 0;
    for (int c = 0; c < nP; c++) {
      const int len = nP - c;
      if (fill + len > kBlkPerWave) {
        flow_map.c1[w] = (unsigned char)c;
        if (++w >= 16) { fits = false; break; }
        flow_map.c0[w] = (unsigned char)c;
        fill = 0;
      }
      fill += len;
    }
    if (fits) {
      flow_map.c1[w] = (unsigned char)nP;
      for (int i = w + 1; i < 16; i++) { flow_map.c0[i] = 0; flow_map.c1[i] = 0; }          // idle wavefronts skip the loop
      const size_t nblk = (size_t)nP * (nP + 1) / 2;
      flow_lds = (12 * (size_t)nP + nblk * kPanStride) * sizeof(double);
      use_flow = flow_lds <= 150 * 1024;
      if (use_flow && flow_lds > 64 * 1024) {
        if (h->flow_attr < flow_lds) {            // (per handle = per device; a handle is used by one thread at a time)
          ORBG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ldlt_flow), hipFuncAttributeMaxDynamicSharedMemorySize, (int)flow_lds));
          h->flow_attr = flow_lds;
        }
      }
    }
  }
  // FP64 matrix-core LDL^T (ldlt_mfma.hpp): up to 50 free poses; ORBG_LDLT_VALU=1 switches back to the vector-ALU kernels
  const bool force_wide = sw.ldlt_wide;          // A/B and test switch: k_wide_* at any size
  const bool use_mfma = nP >= 1 && ldltm::supports(n) && !sw.ldlt_valu && !force_wide;
  // windows beyond the matrix-core kernels (more than 50 free poses): blocked LDL^T over many workgroups
  // (51 free poses still fit the row-pair kernel, which stays reachable through ORBG_LDLT_VALU; the blocked form is faster there: 2.7 vs 3.4 ms)
  const bool use_wide = nP >= 1 && (force_wide || (!use_mfma && !use_flow && (!rows_R || (nP > 50 && !sw.ldlt_valu))));
  if (use_wide && n > kWideMaxUnknowns) return ORBG_CAP_EXCEEDED;      // (k_wide_back's x lives in LDS)
  if (use_wide && (rc = h->d_wide.reserve(2 * (size_t)n + 32))) return rc;
  if (use_mfma) {
    if ((rc = h->d_St.reserve(ldltm::tile_image_doubles(n))) || (rc = h->d_wfac.reserve(ldltm::wglob_doubles(ldltm::make_geo(n))))) return rc;
  }
  int cur = 0;   // index of the buffer holding the current estimate
  // k_update's workgroup size: the kernel is a chain of dependent memory round trips per landmark; smaller workgroups spread the
  // same wavefronts over more compute units (ORBG_UPD_THREADS = 64 / 128 / 256 for experiments)
  // measured (tools/lba_time.py, C2): 256 threads 0.499 ms per solve, 128: 0.488, 64: 0.484
  const int upd_threads = sw.upd_threads;
  // ORBG_FUSE_UPDATE=1: windows the column LDL^T covers run the solve and the update as ONE launch (k_ldlt_cols_update).  Off by
  // default -- measured at C2 (tools/lba_time.py): 0.4865 ms per solve fused, 0.4868-0.4911 as two launches.  The update workgroups
  // do overlap their dependent loads with the LDL^T, but they sit on other XCDs than the LDL^T workgroup: the "x is ready" word and
  // x itself reach them through memory (agent-scope stores / loads, ~2 us each way), which costs what the kernel boundary and
  // k_update's own loads cost.  (A same-XCD placement checked through the XCC_ID register would make the hand-over an L2 round trip.)
  // (On by default from the end of round 3 -- next to the tracking chains the fused launch measured 0.586 vs 0.597 ms -- until the
  // hand-over got the agent-scope RELEASE it needs, round 4: with the L2 write-back in front of the word the fused launch is 0.491
  // vs 0.485 ms alone and 0.590 vs 0.584 ms next to the tracking chains.  Two launches are the default again.  Also measured in
  // round 4: a landmark's dependent round trips are ~0.5 us each, a kernel's fixed cost 4-5 us -- a per-landmark record that halves
  // k_update's round trips bought nothing alone and 3 us per solve in the agent: not kept.)
  const bool fuse_upd = use_mfma && ldltm::pick(n, sw.ldlt).cols && sw.fuse_update;
  const int n_blocks_u = fuse_upd ? (NP + NX + kFusedUpdThreads - 1) / kFusedUpdThreads : (NP + NX + upd_threads - 1) / upd_threads;
  if ((rc = h->d_scale_partial.reserve(std::max(n_blocks_u, 1)))) return rc;
  if (fuse_upd) {
    const size_t lds = ldltm::pick(n, sw.ldlt).lds;
    if (h->fused_attr < lds && lds > 48 * 1024) {
      ORBG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ldlt_cols_update), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8 * 1024));
      h->fused_attr = 160 * 1024 - 8 * 1024;
    }
  }
  if (!h->d_ticket.p) {
    if ((rc = h->d_ticket.reserve(4))) return rc;
    ORBG_HIP(hipMemsetAsync(h->d_ticket.p, 0, 4 * sizeof(unsigned), st));
  }
  // final_mode: the last block also publishes {robust chi2, computeScale(), solver flag} to the host record
  auto launch_errors = [&](int buf, int final_mode) {
    if (NE > 0)
      hipLaunchKernelGGL(k_errors, dim3(n_blocks_e), dim3(256), 0, st, NE, D.edges, posesB[buf], pointsB[buf], cam, hb,
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
  const bool no_spec = sw.no_spec;   // A/B switch: no speculative linearisation
  int ls = 0;                      // linearisation set of the current iteration
  bool spec_ready = false;         // set ls^1 holds the linearisation of the current estimate
  auto launch_linearise = [&](int buf, int set) {
    if (NE > 0 || nP > 0)
      hipLaunchKernelGGL(k_lin_all, dim3(nP + (NE > 0 ? n_blocks_e : 0)), dim3(256), 0, st, nP, NE, D.edges, posesB[buf],
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
    return ORBG_OK;
  };
  auto poll_record = [&]() -> int {
    // the last workgroup of k_errors publishes the record and then its sequence number: spin on that word (the
    // runtime's completion path costs ~10 us per LM trial); fall back to a stream sync if it does not arrive
    volatile unsigned* w = &h->rec.h->seq;
    const unsigned want = h->rec_seq;
    bool got = false;
    if (orbg::poll_allowed()) {              // (the policy of the thread that runs the solve: caller or local-BA worker)
      timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
      for (unsigned spins = 0; !got; spins++) {
        if (*w == want) { got = true; break; }
        if ((spins & 0xFFFF) == 0xFFFF) {
          timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
          if ((t1.tv_sec - t0.tv_sec) * 1000.0 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 > 50.0) break;
        }
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    if (!got) ORBG_HIP(hipStreamSynchronize(st));
    return ORBG_OK;
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
      hipLaunchKernelGGL(k_export, dim3((n_thr + 255) / 256), dim3(256), 0, st, NE, NP, NX, D.edges, posesB[buf], pointsB[buf],
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
      if (fuse_upd) {
        UpdArgs ua{NP, NX, nP, D.pose_col, D.point_col, posesB[in_buf], pointsB[in_buf], h->d_x.p, D.pf_start, D.pf_edges, D.pf_col, EBs[set_],
                   Hlls[set_], bls[set_], lam_, posesB[out_buf], pointsB[out_buf], bps[set_], h->d_scale_partial.p, lamp_, sw.ldlt_prio ? 1 : 0};
        h->xseq = h->xseq == 0x7FFFFFFFu ? 1u : h->xseq + 1u;
        hipLaunchKernelGGL(k_ldlt_cols_update, dim3(1 + n_blocks_u), dim3(ldltm::kThreads), ldltm::pick(n, sw.ldlt).lds, st, n, h->d_St.p, h->d_x.p,
                           h->d_ok.p, h->d_xready.p, h->xseq, ua);
      } else if (use_mfma) {
        ORBG_HIP(ldltm::launch(n, h->d_St.p, h->d_x.p, h->d_ok.p, h->d_wfac.p, st, sw.ldlt, &h->ldlt_attr));
      } else if (use_wide) {
        ORBG_HIP(launch_ldlt_wide(n, h->d_S.p, h->d_bs.p, h->d_x.p, h->d_ok.p, h->d_wide.p, st));
      } else if (use_flow) {
        hipLaunchKernelGGL(k_ldlt_flow, dim3(1), dim3(1024), flow_lds, st, nP, h->d_S.p, h->d_bs.p, h->d_x.p, h->d_ok.p, ldlt_nz, flow_map);
      } else if (rows_R) {
        auto go = [&](auto kern, int nt) {
          hipLaunchKernelGGL(kern, dim3(1), dim3(nt), rows_lds, st, nP, h->d_S.p, h->d_bs.p, h->d_x.p, h->d_ok.p, ldlt_nz);
        };
        if (rows_l_in_lds) {
          if (rows_R == 1 && rows_small) go(k_ldlt_rows<640, 1, true>, 640);
          else if (rows_R == 1) go(k_ldlt_rows<1024, 1, true>, 1024);
          else if (rows_R == 2) go(k_ldlt_rows<1024, 2, true>, 1024);
          else go(k_ldlt_rows<1024, 4, true>, 1024);
        } else {
          if (rows_R == 1 && rows_small) go(k_ldlt_rows<640, 1, false>, 640);
          else if (rows_R == 1) go(k_ldlt_rows<1024, 1, false>, 1024);
          else if (rows_R == 2) go(k_ldlt_rows<1024, 2, false>, 1024);
          else go(k_ldlt_rows<1024, 4, false>, 1024);
        }
      }
      else
        hipLaunchKernelGGL(k_ldlt, dim3(1), dim3(1024), lds_need, st, n, h->d_S.p, h->d_bs.p, h->d_x.p, h->d_ok.p, ldlt_lds ? 1 : 0);
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
        if (!fuse_upd) {
          auto upd = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(n_blocks_u), dim3(upd_threads), 0, st, NP, NX, nP, D.pose_col, D.point_col,
                               posesB[cur], pointsB[cur], h->d_x.p, D.pf_start, D.pf_edges, D.pf_col, EBs[ls],
                               Hlls[ls], bls[ls], lambda, posesB[trial], pointsB[trial], bps[ls], h->d_scale_partial.p, lam_p);
          };
          if (upd_threads == 64) upd(k_update<64>); else if (upd_threads == 128) upd(k_update<128>); else upd(k_update<256>);
        }
        bool speculated = false, fused_export = false;
        if (NE > 0) {
          // speculate on acceptance: linearise the trial state into the other set while the host waits for the verdict
          // (not after the very last iteration that can run)
          // ... nor when two iterations in a row barely improved chi2: a third one ends the round (nBad >= 3)
          const bool may_continue = !(last_round && (it + 1 >= iterations || nBad >= 2)) && !no_spec;
          if (may_continue && !sw.no_fuse) {
            // residuals + record + linearisation of the trial state in ONE launch
            const int set = ls ^ 1;
            const int n_blocks_l = (nL + 255) / 256;       // the landmark reduction rides in the same launch (point workgroups)
            hipLaunchKernelGGL(k_errlin, dim3(nP + n_blocks_e + n_blocks_l), dim3(256), 0, st, nP, NE, D.edges, posesB[trial],
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
            fused_export = !may_continue && last_round && (it + 1 >= iterations || nBad >= 2) && !no_spec && !lambda_on_device &&
                           !sw.no_fuse && !sw.no_export_fuse;
            if (fused_export) {
              const int n_thr = std::max(std::max(NE, NP), 3 * NX);
              hipLaunchKernelGGL(k_errors_export, dim3((n_thr + 255) / 256), dim3(256), 0, st, n_blocks_e, NE, D.edges, posesB[trial],
                                 pointsB[trial], cam, hb, h->d_err.p, h->d_chi2.p, h->d_partial.p, h->d_ticket.p, h->d_scale_partial.p,
                                 n_blocks_u, h->d_ok.p, h->rec.d, ++h->rec_seq, NP, NX, h->dl_h.d + d_flags_o,
                                 r->edge_chi2 ? reinterpret_cast<double*>(h->dl_h.d + d_chi_o) : (double*)nullptr,
                                 reinterpret_cast<PoseQ*>(h->dl_h.d + d_poses_o), reinterpret_cast<double*>(h->dl_h.d + d_points_o));
            } else {
              launch_errors(trial, 1);
            }
            if (may_continue) { launch_linearise(trial, ls ^ 1); speculated = true; }
          }
          const bool round_may_end = it + 1 >= iterations || nBad >= 2;
          if (speculated && !last_round && round_may_end && !lambda_on_device) {
            // ... and if this trial ends the round, the next round's lambda init as well
            hipLaunchKernelGGL(k_finish, dim3(1), dim3(256), 0, st, n_blocks_e, h->d_partial.p, nP, nL, h->d_x.p, bps[ls ^ 1], bls[ls ^ 1],
                               Hpps[ls ^ 1], Hlls[ls ^ 1], 0.0, (int*)nullptr, 0, 1, h->rec.d, p->lambda_init, h->d_lambda0.p);
            fin_version = version;
          }
          if (last_round && round_may_end && !no_spec && !lambda_on_device) {
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
  const bool first2 = dev_lists && NE > 0 && nL > 0 && !sw.no_fuse && !sw.no_first2 && !terminate();
  if (first2 && dev_csr) {
    const int set = ls ^ 1;
    const int n_blocks_l = (nL + 255) / 256;
    uint8_t* const B = h->up_d.p;
    hipLaunchKernelGGL(k_csr_fill, dim3(n_blocks_e), dim3(256), 0, st, NE, D.edges, D.pose_col, D.point_col, reinterpret_cast<int*>(B + o_cur_pt),
                       reinterpret_cast<int*>(B + o_cur_ps), reinterpret_cast<int*>(B + o_cur_pf), const_cast<int*>(D.pt_edges),
                       const_cast<int*>(D.ps_edges), const_cast<int*>(D.pf_edges));
    hipLaunchKernelGGL(k_csr_sort, dim3(nP + n_blocks_l + kPrep256Pad + kPrep256Zero), dim3(256), 0, st, nP, nL, D.ps_start,
                       const_cast<int*>(D.ps_edges), D.pt_start, const_cast<int*>(D.pt_edges), D.pf_start, const_cast<int*>(D.pf_edges),
                       const_cast<int*>(D.pf_col), reinterpret_cast<unsigned long long*>(B + o_lm_mask), D.edges, D.pose_col, n,
                       use_mfma ? h->d_St.p : (double*)nullptr, h->d_x.p, n_zero);
    hipLaunchKernelGGL(k_errlin, dim3(nP + n_blocks_e + n_blocks_l), dim3(256), 0, st, nP, NE, D.edges, posesB[cur],
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
    const int n_blocks_l = (nL + 255) / 256, n_err = nP + n_blocks_e + n_blocks_l, nsb = (nL + 255) / 256;
    hipLaunchKernelGGL(k_errlin_prep, dim3(n_err + nsb + kPrep256Pad + kPrep256Zero), dim3(256), 0, st, n_err, nsb,
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
    if (NE > 0 && !sw.no_fuse) {
      // residuals + linearisation of the initial estimate in the fused kernel of the later trials (its record is not waited
      // for: it carries the sequence number the host has already seen)
      const int set = ls ^ 1;
      const int n_blocks_l = (nL + 255) / 256;
      hipLaunchKernelGGL(k_errlin, dim3(nP + n_blocks_e + n_blocks_l), dim3(256), 0, st, nP, NE, D.edges, posesB[cur],
                         pointsB[cur], cam, hb, h->d_err.p, h->d_chi2.p, D.pose_col, D.point_col, EBs[set], D.ps_start,
                         D.ps_edges, Hpps[set], bps[set], h->d_partial.p, h->d_ticket.p, h->d_scale_partial.p, 0, (const int*)nullptr,
                         h->rec.d, h->rec_seq, n_blocks_e, nL, D.pt_start, D.pt_edges, Hlls[set], bls[set],
                         LmIn{0.0, 0.0, (const double*)nullptr, (const double*)nullptr, (double*)nullptr});
    } else {
      launch_errors(cur, 0);
      launch_linearise(cur, ls ^ 1);
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
        for (int j = b0; j < e0; j++) adj[pf_col[j]] |= m;
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
    if (!dev_lists && (rc = upload_arena(h, off_a, off_b, st, blit))) return rc;      // (dev_lists: k_schur derives the pair from its index)
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
  // symbolic elimination of the reduced camera system: which blocks of L are structurally non-zero (fill-in included)
  if (nP <= 64 && !sw.ldlt_dense && !dev_lists) {     // (only the vector-ALU kernels read it; dev_lists implies the matrix-core solver)
    unsigned long long col[64];                         // col[j]: rows i > j with S_ij != 0, then with fill-in
    for (int j = 0; j < nP; j++) {
      unsigned long long mcol = 0;
      for (int i = j + 1; i < nP; i++)
        if (dev_items ? ((adj[j] >> i) & 1ull) != 0 : pair_start[pair_id(j, i) + 1] > pair_start[pair_id(j, i)]) mcol |= 1ull << i;
      col[j] = mcol;
    }
    for (int j = 0; j < nP; j++) {
      const unsigned long long rows = col[j];
      for (int k = j + 1; k < nP; k++)
        if ((rows >> k) & 1ull) col[k] |= k < 63 ? (rows & ~((2ull << k) - 1ull)) : 0ull;   // rows below k of column j fill column k
      ldlt_nz.m[j] = rows;
    }
  }
  // keep every pair (diagonals always; off-diagonals even if empty so that S is fully written)
  for (int i1 = 0; i1 < nP; i1++)
    for (int i2 = i1; i2 < nP; i2++) { pair_i1[pair_id(i1, i2)] = i1; pair_i2[pair_id(i1, i2)] = i2; }

  if (!dev_items && (rc = upload_arena(h, off_b, off, st, blit))) return rc;
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
  if (NE > 0 && n_out >= NE * 0.5) r->status = LBA_REJECTED_OUTLIERS;
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
  if (matrix_core) *matrix_core = h->prof_n_unknowns >= 1 && ldltm::supports(h->prof_n_unknowns) && !h->sw.ldlt_valu && !h->sw.ldlt_wide;
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


// ------------------------------------------------------------------------------------------------
// Optimizer::PoseOptimization(Frame*) (S/Optimizer.cc:964-1278) -- the WHOLE solve in one kernel launch.
//
// One 256-thread workgroup; thread t owns correspondences t, t+256, ...  The four rounds, the Levenberg-Marquardt
// iterations and their accept/reject trials all run on the device: thread 0 holds the 6x6 system, lambda and the
// control flow, everything else is broadcast through LDS.  Reductions are fixed-order (wavefront shuffle tree, then
// the four wavefront partials in order), so the result is bit-reproducible.  A host-driven version would need one
// synchronisation per LM trial (~40-60 per call); this needs one.
namespace {

constexpr int kPoThreads = 256;
constexpr int kPoMaxPer = 16;      // correspondences per thread (n <= 4096)
constexpr int kPoLdsN = 1024;      // correspondences whose inputs are staged in LDS
constexpr int kPoRow = 8 * 33;     // one reduction row: 8 segments of 32 values, padded against LDS bank conflicts

// Block-wide sums of NV per-thread values in a fixed order: transpose through LDS, 8 threads per value add 32
// entries each, one thread per value adds the 8 partials.  out[0..NV) is valid for every thread afterwards.
template <int NV>
__device__ inline void po_block_reduce(const double* vals, double* s_acc, double* s_part, double* out) {
  const int tid = threadIdx.x;
  const int col = (tid >> 5) * 33 + (tid & 31);
#pragma unroll
  for (int v = 0; v < NV; v++) s_acc[v * kPoRow + col] = vals[v];
  __syncthreads();
  if (tid < NV * 8) {
    const double* p = s_acc + (tid >> 3) * kPoRow + (tid & 7) * 33;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { a0 += p[4 * j]; a1 += p[4 * j + 1]; a2 += p[4 * j + 2]; a3 += p[4 * j + 3]; }
    s_part[tid] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  if (tid < NV) {
    const double* p = s_part + tid * 8;
    out[tid] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  }
  __syncthreads();
}

// Block-wide sum of ONE double per thread: DPP tree inside each wavefront, the four wave totals through LDS, added in wave
// order by every thread (one barrier; `slot` alternates between consecutive calls so that no second barrier is needed).
__device__ __forceinline__ double po_block_sum(double v, double (*wsum)[4], int slot) {
  const double w = wave_sum_f64(v);
  if ((threadIdx.x & 63) == 0) wsum[slot][threadIdx.x >> 6] = w;
  __syncthreads();
  return ((wsum[slot][0] + wsum[slot][1]) + wsum[slot][2]) + wsum[slot][3];
}

__device__ __forceinline__ double po_readlane(double v, int lane) {   // lane must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// (H + lambda I) x = b by LDL^T without pivoting, spread over lanes 0..5 of a wave: lane `li` holds row li.  Every
// subtraction happens in the order of a scalar left-looking factorisation (ascending k), so the factors are the
// same bits a serial solve would produce.  Returns false unless every pivot is positive (Eigen::LDLT::isPositive),
// in which case x is left untouched.  x[] comes out wave-uniform.
__device__ inline bool po_solve6(const double* Hrow, double b_li, int li, double lambda, double* x) {
  // (round 4, measured and dropped: the seven divisions as products with 1/d from the hardware seed + two Newton steps -- no
  // measurable gain, 162 vs 158-164 us at 450 correspondences, and one of the twelve parity cases changed an iteration count)
  double A[6], D[6];
#pragma unroll
  for (int j = 0; j < 6; j++) A[j] = Hrow[j] + (j == li ? lambda : 0.0);
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const double d = po_readlane(A[k], k);
    if (!(d > 0.0) || fabs(d) == INFINITY) ok = false;
    D[k] = d;
    const double Lik = A[k] / d;
#pragma unroll
    for (int j = k + 1; j < 6; j++) { const double Ljk = po_readlane(Lik, j); A[j] -= (Lik * Ljk) * d; }
    A[k] = Lik;
  }
  if (!ok) return false;
  double y = b_li;
#pragma unroll
  for (int k = 0; k < 5; k++) { const double yk = po_readlane(y, k); if (li > k) y -= A[k] * yk; }
  double Di = D[0];
#pragma unroll
  for (int k = 1; k < 6; k++) Di = (li == k) ? D[k] : Di;
  y /= Di;
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double sv = po_readlane(y, i);
#pragma unroll
    for (int k = i + 1; k < 6; k++) sv -= po_readlane(A[i], k) * x[k];
    x[i] = sv;
  }
  return true;
}

// ---- two correspondences side by side.  A lone wavefront issues a DEPENDENT FP64 instruction every ~9 cycles and an independent
// one every ~5.4 (tools/micro/fp64_latency; the pipe itself takes one per 4.2), and hipcc keeps the arithmetic of one
// correspondence together when it is written as a scalar function called twice.  The per-correspondence arithmetic is therefore
// written ONCE over a value type V that is either double (one correspondence) or D2 (the thread's correspondences i and i + 256,
// element-wise): every operation of the pair stands next to its twin in the instruction stream, the operations and their order per
// correspondence are exactly the scalar ones -- the same bits, which the LM loop's accept / reject and termination decisions need
// (tried: rotation matrix instead of the quaternion sandwich, Newton reciprocals instead of divisions: different iteration counts).
struct D2 { double a, b; };
struct B2 { bool a, b; };
__device__ __forceinline__ D2 operator+(D2 x, D2 y) { return D2{x.a + y.a, x.b + y.b}; }
__device__ __forceinline__ D2 operator-(D2 x, D2 y) { return D2{x.a - y.a, x.b - y.b}; }
__device__ __forceinline__ D2 operator*(D2 x, D2 y) { return D2{x.a * y.a, x.b * y.b}; }
__device__ __forceinline__ D2 operator-(D2 x) { return D2{-x.a, -x.b}; }
__device__ __forceinline__ D2 operator+(D2 x, double y) { return D2{x.a + y, x.b + y}; }
__device__ __forceinline__ D2 operator+(double x, D2 y) { return D2{x + y.a, x + y.b}; }
__device__ __forceinline__ D2 operator-(D2 x, double y) { return D2{x.a - y, x.b - y}; }
__device__ __forceinline__ D2 operator-(double x, D2 y) { return D2{x - y.a, x - y.b}; }
__device__ __forceinline__ D2 operator*(D2 x, double y) { return D2{x.a * y, x.b * y}; }
__device__ __forceinline__ D2 operator*(double x, D2 y) { return D2{x * y.a, x * y.b}; }
__device__ __forceinline__ D2 operator/(double x, D2 y) { return D2{x / y.a, x / y.b}; }
__device__ __forceinline__ D2 operator/(D2 x, D2 y) { return D2{x.a / y.a, x.b / y.b}; }
__device__ __forceinline__ double po_f32round(double x) { return (double)(float)x; }
__device__ __forceinline__ D2 po_f32round(D2 x) { return D2{(double)(float)x.a, (double)(float)x.b}; }
__device__ __forceinline__ double po_sel(bool c, double x, double y) { return c ? x : y; }
__device__ __forceinline__ D2 po_sel(B2 c, D2 x, D2 y) { return D2{c.a ? x.a : y.a, c.b ? x.b : y.b}; }
__device__ __forceinline__ bool po_neg(double x) { return x < 0; }
__device__ __forceinline__ B2 po_neg(D2 x) { return B2{x.a < 0, x.b < 0}; }
__device__ __forceinline__ double po_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ D2 po_sqrt(D2 x) { return D2{sqrt(x.a), sqrt(x.b)}; }
__device__ __forceinline__ bool po_any_gt(double e, double d) { return !(e <= d); }
__device__ __forceinline__ bool po_any_gt(D2 e, D2 d) { return !(e.a <= d.a) || !(e.b <= d.b); }
__device__ __forceinline__ bool po_le(double e, double d) { return e <= d; }
__device__ __forceinline__ B2 po_le(D2 e, D2 d) { return B2{e.a <= d.a, e.b <= d.b}; }
template <class V> __device__ __forceinline__ V po_c(double x);
template <> __device__ __forceinline__ double po_c<double>(double x) { return x; }
template <> __device__ __forceinline__ D2 po_c<D2>(double x) { return D2{x, x}; }
template <class V> struct PoMask;
template <> struct PoMask<double> { typedef bool type; };
template <> struct PoMask<D2> { typedef B2 type; };

// quat_rotate (above) over V
template <class V>
__device__ __forceinline__ void po_quat_rotate(const double* q, const V* v, V* out) {
  const V uv0 = 2 * (q[1] * v[2] - q[2] * v[1]), uv1 = 2 * (q[2] * v[0] - q[0] * v[2]), uv2 = 2 * (q[0] * v[1] - q[1] * v[0]);
  out[0] = v[0] + q[3] * uv0 + (q[1] * uv2 - q[2] * uv1);
  out[1] = v[1] + q[3] * uv1 + (q[2] * uv0 - q[0] * uv2);
  out[2] = v[2] + q[3] * uv2 + (q[0] * uv1 - q[1] * uv0);
}
// camera-frame point and 1 / z, then the edge error: mono I/OptimizableTypes.h:44-48 (Pinhole::project in double), stereo
// G/types/types_six_dof_expmap.cpp:339-346 (float invz, double bf * invz).  Both forms are evaluated and one is selected.
template <class V>
__device__ __forceinline__ void po_cam_point(const PoseQ& T, const V* X, V* Xc, V* iz) {
  V r[3];
  po_quat_rotate(T.q, X, r);
  Xc[0] = r[0] + T.t[0]; Xc[1] = r[1] + T.t[1]; Xc[2] = r[2] + T.t[2];
  *iz = 1.0 / Xc[2];
}
template <class V, class M>
__device__ __forceinline__ void po_residual(const V* Xc, V iz, V u, V v, V ur, M mono, const Cam& c, V* err) {
  const V m0 = u - (c.fx * Xc[0] * iz + c.cx);
  const V m1 = v - (c.fy * Xc[1] * iz + c.cy);
  const V invz = po_f32round(iz);
  const V r0 = Xc[0] * invz * c.fx + c.cx;
  const V r1 = Xc[1] * invz * c.fy + c.cy;
  const V s2 = ur - (r0 - c.bf * invz);
  err[0] = po_sel(mono, m0, u - r0);
  err[1] = po_sel(mono, m1, v - r1);
  err[2] = po_sel(mono, po_c<V>(0.0), s2);
}
// rho0 / rho1 of RobustKernelHuber::robustify (G/core/robust_kernel_impl.cpp:78-91); the square root only where e > dsqr
template <class V>
__device__ __forceinline__ void po_huber(bool robust, V e, V delta, V dsqr, V one, V* rho0, V* rho1) {
  *rho0 = e; *rho1 = one;
  if (robust && po_any_gt(e, dsqr)) {
    const V sq = po_sqrt(e);
    const auto in = po_le(e, dsqr);
    *rho0 = po_sel(in, e, 2 * sq * delta - dsqr);
    *rho1 = po_sel(in, one, delta / sq);
  }
}
__device__ inline void po_edge_error(const PoseQ& T, const float* X, float u, float v, float ur, const Cam& c, double* err, double* Xc) {
  const double Xd[3] = {X[0], X[1], X[2]};
  double iz;
  po_cam_point<double>(T, Xd, Xc, &iz);
  po_residual<double, bool>(Xc, iz, (double)u, (double)v, (double)ur, ur < 0, c, err);
}

// ---- the per-correspondence arithmetic of the LM loop, written once over V (double: one correspondence, D2: the thread's pair)
// PoEval: everything an evaluation at a pose produces for a correspondence
template <class V> struct PoEval { V Xc[3], iz, err[3], c2, rho0, rho1; };
template <class V, class M>
__device__ __forceinline__ void po_eval(const PoseQ& T, const V* X, V uu, V vv, V ur, V om, M mono, bool robust, const Cam& cam,
                                        double dM, double dS, double dsqM, double dsqS, PoEval<V>* e) {
  po_cam_point<V>(T, X, e->Xc, &e->iz);
  po_residual<V, M>(e->Xc, e->iz, uu, vv, ur, mono, cam, e->err);
  e->c2 = e->err[0] * (om * e->err[0]) + e->err[1] * (om * e->err[1]) + po_sel(mono, po_c<V>(0.0), e->err[2] * (om * e->err[2]));
  po_huber<V>(robust, e->c2, po_sel(mono, po_c<V>(dM), po_c<V>(dS)), po_sel(mono, po_c<V>(dsqM), po_c<V>(dsqS)), po_c<V>(1.0), &e->rho0, &e->rho1);
}
// J^T (w Omega) J (21 entries, upper triangle row-major) and J^T (w Omega) r (6) of one evaluation -> hh[27]
template <class V, class M>
__device__ __forceinline__ void po_hessian(const PoEval<V>& e, V om, M mono, const Cam& cam, V* hh) {
  // Jacobian (D x 6): mono S/OptimizableTypes.cpp:49-63, stereo types_six_dof_expmap.cpp:375-404
  const V xx = e.Xc[0], yy = e.Xc[1], iz = e.iz, iz2 = iz * iz;
  const V zero = po_c<V>(0.0);
  V J[18];
  J[0] = xx * yy * iz2 * cam.fx; J[1] = -(1 + (xx * xx * iz2)) * cam.fx; J[2] = yy * iz * cam.fx; J[3] = -iz * cam.fx; J[4] = zero; J[5] = xx * iz2 * cam.fx;
  J[6] = (1 + yy * yy * iz2) * cam.fy; J[7] = -xx * yy * iz2 * cam.fy; J[8] = -xx * iz * cam.fy; J[9] = zero; J[10] = -iz * cam.fy; J[11] = yy * iz2 * cam.fy;
  J[12] = po_sel(mono, zero, J[0] - cam.bf * yy * iz2); J[13] = po_sel(mono, zero, J[1] + cam.bf * xx * iz2); J[14] = po_sel(mono, zero, J[2]);
  J[15] = po_sel(mono, zero, J[3]); J[16] = zero; J[17] = po_sel(mono, zero, J[5] - cam.bf * iz2);
  const V wom = e.rho1 * om;
  V orr[3];
#pragma unroll
  for (int k = 0; k < 3; k++) orr[k] = -(om * e.err[k]) * e.rho1;
  // J^T (w Omega) J with the weighted rows formed once and the structural zeros of the Jacobian (column 4 of rows 0 and
  // 2, column 3 of row 1) left out: 15 + 45 + 15 multiply-adds per correspondence instead of 126 + 36
  constexpr int kZeroCol[3] = {4, 3, 4};
  V wJ[18];
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int a2 = 0; a2 < 6; a2++) wJ[6 * k + a2] = a2 == kZeroCol[k] ? zero : wom * J[6 * k + a2];
  int o = 0;
#pragma unroll
  for (int a2 = 0; a2 < 6; a2++)
#pragma unroll
    for (int c3 = a2; c3 < 6; c3++) {
      V h = zero;
#pragma unroll
      for (int k = 0; k < 3; k++)
        if (a2 != kZeroCol[k] && c3 != kZeroCol[k]) h = h + J[6 * k + a2] * wJ[6 * k + c3];
      hh[o++] = h;
    }
#pragma unroll
  for (int a2 = 0; a2 < 6; a2++) {
    V sacc = zero;
#pragma unroll
    for (int k = 0; k < 3; k++)
      if (a2 != kZeroCol[k]) sacc = sacc + J[6 * k + a2] * orr[k];
    hh[o++] = sacc;
  }
}
// Optimizer::PoseOptimization (S/Optimizer.cc:992-1290) in ONE launch of one workgroup: 4 rounds x up to 10
// Levenberg-Marquardt iterations (g2o OptimizationAlgorithmLevenberg semantics), outlier re-classification after
// each round.  The LM state (pose, lambda, gains) is kept identically in every thread -- all of them read the same
// block sums from LDS and run the same arithmetic -- so the control flow needs no broadcast; the 6x6 solve runs on
// lanes 0..5 of each wave.
// -DPO_PROFILE: cycles of thread 0 per phase, summed over the call (build, reduce, solve, trial evaluation, trial sum, rest)
#ifdef PO_PROFILE
__device__ long long g_po_prof[8];
#define PO_T0() long long po_t = clock64()
#define PO_ACC(slot) do { const long long po_n = clock64(); if (threadIdx.x == 0) g_po_prof[slot] += po_n - po_t; po_t = po_n; } while (0)
#else
#define PO_T0() do { } while (0)
#define PO_ACC(slot) do { } while (0)
#endif
template <bool LDS_IN>      // LDS_IN: n <= kPoLdsN, the correspondences are staged in LDS (typed LDS accesses: a pointer that may be LDS or
                            // global at run time turns every load into a flat_load with a full wait behind it)
__global__ __launch_bounds__(kPoThreads) void pose_opt_kernel(int n, const float* g_Xw, const float* g_ou, const float* g_ov, const float* g_our,
                                                             const float* g_oinv, Cam cam, PoseQ T0,
                                                             PoseQ* __restrict__ T_out, uint8_t* __restrict__ outlier_out,
                                                             int* __restrict__ stats /*n_bad, iters[4], .., [7] = seq*/,
                                                             double* __restrict__ chi_out, unsigned seq) {
  __shared__ float s_in[7 * kPoLdsN];                    // correspondences staged once (they are re-read ~36 times)
  __shared__ double s_acc[28 * kPoRow];
  __shared__ double s_part[28 * 8];
  __shared__ double red[28];
  __shared__ double s_wsum[2][4];
  __shared__ double s_cand[4][14];                       // LM trial candidates of the current iteration: x[6], pose q[4] t[3], solve ok
  int sum_slot = 0;
  __shared__ double s_chi2[kPoThreads * kPoMaxPer];      // last evaluated chi2 of every correspondence
  __shared__ uint8_t s_out[kPoThreads * kPoMaxPer];      // mvbOutlier
  const int tid = threadIdx.x;
  const int li = min(tid & 63, 5);
  const double dM = (float)sqrt(5.991), dS = (float)sqrt(7.815);
  const double dsqM = dM * dM, dsqS = dS * dS;
  for (int i = tid; i < n; i += kPoThreads) { s_chi2[i] = 0; s_out[i] = 0; }
  if (LDS_IN) {
    // the inputs may sit in mapped host memory (zero-copy): read them exactly once
    for (int i = tid; i < 3 * n; i += kPoThreads) s_in[i] = g_Xw[i];
    for (int i = tid; i < n; i += kPoThreads) { s_in[3 * n + i] = g_ou[i]; s_in[4 * n + i] = g_ov[i]; s_in[5 * n + i] = g_our[i]; s_in[6 * n + i] = g_oinv[i]; }
  }
  const float* const Xw = LDS_IN ? s_in : g_Xw;
  const float* const ou = LDS_IN ? s_in + 3 * n : g_ou;
  const float* const ov = LDS_IN ? s_in + 4 * n : g_ov;
  const float* const our = LDS_IN ? s_in + 5 * n : g_our;
  const float* const oinv = LDS_IN ? s_in + 6 * n : g_oinv;
  double x[6] = {0, 0, 0, 0, 0, 0};
  double lambda = 0, ni = 2, currentChi = 0;
  int nBadLM = 0;
  bool robust = true;
  PoseQ T = T0;
  int nBad = 0;
  if (tid == 0) { for (int i = 0; i < 7; i++) stats[i] = 0; for (int i = 0; i < 4; i++) chi_out[i] = 0; }
  __syncthreads();
  PO_T0();
  for (int round = 0; round < 4; round++) {
    T = T0;                                                   // setEstimate(toSE3Quat(mTcw)) every round (:1191)
    double cnt = 0;
    for (int i = tid; i < n; i += kPoThreads) cnt += !s_out[i];
    const int n_active = (int)po_block_sum(cnt, s_wsum, sum_slot); sum_slot ^= 1;
    int done = 0;
    bool ok = n_active > 0;
    for (int it = 0; it < 10 && ok; it++) {
      // ---- computeActiveErrors + buildSystem at T
      double acc[28];
#pragma unroll
      for (int i = 0; i < 28; i++) acc[i] = 0;
      // a thread's correspondences i, i + 256 are linearised side by side (D2: one instruction stream per correspondence, the
      // two interleaved) and accumulated in the order i, i + 256, ... as a scalar loop would: same sums, bit for bit.  A pass whose
      // second halves all lie beyond n (n <= 256, 512 < n <= 768: the third correspondence of a thread) runs the one-correspondence
      // form of the same arithmetic.  (Measured and dropped, round 4: keeping the accepted trial's evaluation -- camera point, 1/z,
      // residual, Huber terms -- for the next buildSystem: bit-identical and a quarter of buildSystem's arithmetic less, but the 80
      // registers it holds go to the accumulation registers as spills (36 -> 109): 158 -> 164 us at 450 correspondences; adding every
      // Hessian entry to its accumulator as soon as it exists instead of forming the pair's 27 first: spills 36 -> 12, but the
      // accumulators become two-deep dependency chains: 162 -> 169 us.)
      for (int i0 = tid; i0 < n; i0 += 2 * kPoThreads) {
        if ((i0 - tid) + kPoThreads >= n) {
          // ---- single correspondences (uniform: no thread has a partner in this pass)
          if (s_out[i0]) continue;
          const double X[3] = {(double)Xw[3 * i0], (double)Xw[3 * i0 + 1], (double)Xw[3 * i0 + 2]};
          const double ur1 = (double)our[i0], om1 = (double)oinv[i0];
          const bool mono1 = po_neg(ur1);
          PoEval<double> e1;
          po_eval<double, bool>(T, X, (double)ou[i0], (double)ov[i0], ur1, om1, mono1, robust, cam, dM, dS, dsqM, dsqS, &e1);
          double h1[27];
          po_hessian<double, bool>(e1, om1, mono1, cam, h1);
          s_chi2[i0] = e1.c2;
          acc[27] += e1.rho0;
#pragma unroll
          for (int o = 0; o < 27; o++) acc[o] += h1[o];
          continue;
        }
        const int i1r = i0 + kPoThreads;
        const bool in1 = i1r < n;
        const int i1 = in1 ? i1r : i0;
        const bool act0 = !s_out[i0], act1 = in1 && !s_out[i1];
        if (!(act0 || act1)) continue;                             // (both excluded is rare)
        const D2 om{(double)oinv[i0], (double)oinv[i1]};
        const D2 ur{(double)our[i0], (double)our[i1]};
        const B2 mono = po_neg(ur);
        PoEval<D2> e2;
        {
          const D2 X[3] = {D2{(double)Xw[3 * i0], (double)Xw[3 * i1]}, D2{(double)Xw[3 * i0 + 1], (double)Xw[3 * i1 + 1]},
                           D2{(double)Xw[3 * i0 + 2], (double)Xw[3 * i1 + 2]}};
          const D2 uu{(double)ou[i0], (double)ou[i1]}, vv{(double)ov[i0], (double)ov[i1]};
          po_eval<D2, B2>(T, X, uu, vv, ur, om, mono, robust, cam, dM, dS, dsqM, dsqS, &e2);
        }
        D2 hh[27];
        po_hessian<D2, B2>(e2, om, mono, cam, hh);
        if (act0) {
          s_chi2[i0] = e2.c2.a;
          acc[27] += e2.rho0.a;
#pragma unroll
          for (int o = 0; o < 27; o++) acc[o] += hh[o].a;
        }
        if (act1) {
          s_chi2[i1] = e2.c2.b;
          acc[27] += e2.rho0.b;
#pragma unroll
          for (int o = 0; o < 27; o++) acc[o] += hh[o].b;
        }
      }
      PO_ACC(0);
      po_block_reduce<28>(acc, s_acc, s_part, red);
      PO_ACC(1);
      // every thread takes its own copy of the system: row li of H (upper triangle packed row-major in red[0..21)), b
      double Hrow[6], b[6];
#pragma unroll
      for (int j = 0; j < 6; j++) {
        const int a = min(li, j), c = max(li, j);
        Hrow[j] = red[a * 6 - (a * (a - 1)) / 2 + (c - a)];
        b[j] = red[21 + j];
      }
      const double b_li = red[21 + li];
      currentChi = red[27];
      const double iniChi = currentChi;
      if (it == 0) {
        const double mx = fmax(fmax(fmax(fabs(red[0]), fabs(red[6])), fmax(fabs(red[11]), fabs(red[15]))), fmax(fabs(red[18]), fabs(red[20])));
        lambda = 1e-5 * mx; ni = 2; nBadLM = 0;
      }
      // ---- LM trials.  The damping values a run of REJECTED trials goes through are known in advance (lambda *= ni, ni *= 2 per
      // rejection, levenberg.cpp:139-146), and H, b do not change inside an iteration: the four wavefronts solve
      // (H + lambda_c I) x = b and form the trial pose for candidates c = 0..3 at the same time (each on its own SIMD -- they used
      // to repeat the SAME solve four times), hand them over through LDS, and trial q picks up candidate q.  A rejected trial then
      // costs no solve (23 of the 41 trials of a typical call).  Same operations per candidate as the sequential loop: same bits.
      double rho = 0;
      int qmax = 0;
      for (;;) {
        PO_ACC(5);
        const int cslot = qmax & 3;
        if (cslot == 0) {
          double lam_c = lambda, ni_c = ni;
          const int wv = tid >> 6;
          for (int cc = 0; cc < wv; cc++) { lam_c *= ni_c; ni_c *= 2; }
          double xc[6] = {x[0], x[1], x[2], x[3], x[4], x[5]};
          const bool okc = po_solve6(Hrow, b_li, li, lam_c, xc);
          PoseQ Tc;
          pose_oplus_series(T, xc, &Tc);
          if ((tid & 63) == 0) {
            double* sc = s_cand[wv];
#pragma unroll
            for (int j = 0; j < 6; j++) sc[j] = xc[j];
#pragma unroll
            for (int j = 0; j < 4; j++) sc[6 + j] = Tc.q[j];
#pragma unroll
            for (int j = 0; j < 3; j++) sc[10 + j] = Tc.t[j];
            sc[13] = okc ? 1.0 : 0.0;
          }
          __syncthreads();
        }
        const bool ok2 = s_cand[cslot][13] != 0.0;
        PoseQ Tt;
        if (ok2) {
#pragma unroll
          for (int j = 0; j < 6; j++) x[j] = s_cand[cslot][j];
#pragma unroll
          for (int j = 0; j < 4; j++) Tt.q[j] = s_cand[cslot][6 + j];
#pragma unroll
          for (int j = 0; j < 3; j++) Tt.t[j] = s_cand[cslot][10 + j];
        } else {
          pose_oplus_series(T, x, &Tt);                     // the solve failed: update with whatever x holds, as g2o does
        }
        PO_ACC(2);
        double tchi = 0;
        // the residuals of the LAST evaluation stay with the edges, accepted or not (:1196-1270 read e->chi2())
        for (int i0 = tid; i0 < n; i0 += 2 * kPoThreads) {
          if ((i0 - tid) + kPoThreads >= n) {
            if (s_out[i0]) continue;
            const double X[3] = {(double)Xw[3 * i0], (double)Xw[3 * i0 + 1], (double)Xw[3 * i0 + 2]};
            const double ur1 = (double)our[i0];
            PoEval<double> e1;
            po_eval<double, bool>(Tt, X, (double)ou[i0], (double)ov[i0], ur1, (double)oinv[i0], po_neg(ur1), robust, cam, dM, dS, dsqM, dsqS, &e1);
            s_chi2[i0] = e1.c2; tchi += e1.rho0;
            continue;
          }
          const int i1r = i0 + kPoThreads;
          const bool in1 = i1r < n;
          const int i1 = in1 ? i1r : i0;
          const bool act0 = !s_out[i0], act1 = in1 && !s_out[i1];
          if (!(act0 || act1)) continue;
          const D2 X[3] = {D2{(double)Xw[3 * i0], (double)Xw[3 * i1]}, D2{(double)Xw[3 * i0 + 1], (double)Xw[3 * i1 + 1]},
                           D2{(double)Xw[3 * i0 + 2], (double)Xw[3 * i1 + 2]}};
          const D2 uu{(double)ou[i0], (double)ou[i1]}, vv{(double)ov[i0], (double)ov[i1]}, ur{(double)our[i0], (double)our[i1]};
          const D2 om{(double)oinv[i0], (double)oinv[i1]};
          PoEval<D2> e2;
          po_eval<D2, B2>(Tt, X, uu, vv, ur, om, po_neg(ur), robust, cam, dM, dS, dsqM, dsqS, &e2);
          if (act0) { s_chi2[i0] = e2.c2.a; tchi += e2.rho0.a; }
          if (act1) { s_chi2[i1] = e2.c2.b; tchi += e2.rho0.b; }
        }
        PO_ACC(3);
        double tempChi = po_block_sum(tchi, s_wsum, sum_slot); sum_slot ^= 1;
        PO_ACC(4);
        if (!ok2) tempChi = 1.7976931348623157e308;
        rho = currentChi - tempChi;
        double scale = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && fabs(tempChi) != INFINITY && tempChi == tempChi) {
          double alpha = 1. - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
          alpha = fmin(alpha, 2. / 3.);
          lambda *= fmax(1. / 3., alpha);
          ni = 2;
          currentChi = tempChi;
          T = Tt;
        } else {
          lambda *= ni; ni *= 2;
        }
        qmax++;
        if (!(rho < 0 && qmax < 10)) break;
      }
      done++;
      if (qmax == 10 || rho == 0) ok = false;
      else {
        if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
        if (nBadLM >= 3) ok = false;
      }
    }
    if (tid == 0) { stats[1 + round] = done; chi_out[round] = currentChi; }
    // ---- classification (:1196-1270): excluded edges get a fresh residual at the final pose, active ones keep the last one
    double bl = 0;
    for (int i = tid; i < n; i += kPoThreads) {
      const float ur = our[i];
      const bool mono = ur < 0;
      if (s_out[i]) {
        double err[3], Xc[3];
        po_edge_error(T, Xw + 3 * (size_t)i, ou[i], ov[i], ur, cam, err, Xc);
        const double om = (double)oinv[i];
        s_chi2[i] = err[0] * (om * err[0]) + err[1] * (om * err[1]) + (mono ? 0.0 : err[2] * (om * err[2]));
      }
      const float c2f = (float)s_chi2[i];
      const bool bad = c2f > (mono ? 5.991f : 7.815f);
      s_out[i] = bad;
      bl += bad;
    }
    nBad = (int)po_block_sum(bl, s_wsum, sum_slot); sum_slot ^= 1;
    if (round == 2) robust = false;                          // setRobustKernel(0)
    if (n < 10) break;                                        // optimizer.edges().size() < 10
  }
  PO_ACC(5);
  for (int i = tid; i < n; i += kPoThreads) outlier_out[i] = s_out[i];
  if (tid == 0) { *T_out = T; stats[0] = nBad; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");              // system-scope release of every wavefront's results (no acquire half)
  __syncthreads();
  if (tid == 0) *reinterpret_cast<volatile int*>(&stats[7]) = (int)seq;   // results are complete: the host spins on this word
}

}  // namespace

#ifdef PO_PROFILE
extern "C" int pose_opt_debug_prof(long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_po_prof), sizeof(g_po_prof)) != hipSuccess) return -1;
  if (reset) { long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_po_prof), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

// per-thread scratch of pose_optimize (PoseOptimization has no handle: the reference calls a static member); released when the thread exits
namespace {
struct PoScratch {
  PinnedBuf<uint8_t> stage; DevBuf<uint8_t> dev; int device = -1; hipStream_t stream = nullptr; bool ext_stream = false;
  void drop_stream() { if (stream && !ext_stream) orbg::release_stream(stream); stream = nullptr; ext_stream = false; }
  void drop() { stage.release(); dev.release(); drop_stream(); }
  ~PoScratch() { drop(); }
};
PoScratch& po_scratch() { static thread_local PoScratch sc; return sc; }
}  // namespace

// the calling thread's pose_optimize calls on `device` use the caller's stream from now on (NULL: the library's M stream again)
extern "C" int pose_opt_set_stream(int device, void* hip_stream) {
  int rc = select_device(device);
  if (rc) return rc;
  PoScratch& sc = po_scratch();
  if (sc.device != device) { sc.drop(); sc.device = device; }
  if (sc.stream) ORBG_HIP(hipStreamSynchronize(sc.stream));
  sc.drop_stream();
  if (hip_stream) { sc.stream = (hipStream_t)hip_stream; sc.ext_stream = true; }
  return ORBG_OK;
}

extern "C" int pose_optimize(const pose_opt_problem* p, pose_opt_result* r) {
  if (!p || !r || p->n < 0 || (p->n > 0 && (!p->Xw || !p->u || !p->v || !p->ur || !p->inv_sigma2 || !r->outlier))) return ORBG_BAD_ARG;
  if (p->n > kPoThreads * kPoMaxPer) return ORBG_CAP_EXCEEDED;
  int rc = select_device(p->device);
  if (rc) return rc;
  const int n = p->n;
  memcpy(r->Tcw, p->Tcw, sizeof(float) * 16);
  r->n_inliers = 0; r->n_bad = 0;
  for (int i = 0; i < 4; i++) { r->iters[i] = 0; r->chi2[i] = 0; }
  for (int i = 0; i < n; i++) r->outlier[i] = 0;
  if (n < 3) return ORBG_OK;                                  // S/Optimizer.cc:1180-1181
  // one pinned staging block: inputs in, results out (a per-thread cache keeps the allocation across calls); the stream comes from
  // the library's pool (common.hpp: role "po" = M, non-blocking like all of the library's streams) or from pose_opt_set_stream
  PoScratch& sc = po_scratch();
  if (sc.device != p->device) { sc.drop(); sc.device = p->device; }
  if (!sc.stream) { ORBG_HIP(orbg::create_stream(&sc.stream, "po")); sc.ext_stream = false; }
  const size_t in_bytes = ((size_t)n * 7 * 4 + 15) & ~(size_t)15;
  const size_t out_off = in_bytes;
  const size_t out_bytes = sizeof(PoseQ) + 8 * sizeof(int) + 4 * sizeof(double) + (size_t)n + 64;
  if ((rc = sc.stage.reserve(in_bytes + out_bytes + 64)) || (rc = sc.dev.reserve(in_bytes + out_bytes + 64))) return rc;
  float* hs = reinterpret_cast<float*>(sc.stage.h);
  memcpy(hs, p->Xw, (size_t)n * 12);
  memcpy(hs + 3 * (size_t)n, p->u, (size_t)n * 4);
  memcpy(hs + 4 * (size_t)n, p->v, (size_t)n * 4);
  memcpy(hs + 5 * (size_t)n, p->ur, (size_t)n * 4);
  memcpy(hs + 6 * (size_t)n, p->inv_sigma2, (size_t)n * 4);
  // small problems: the kernel reads its inputs straight from this pinned block (once, into LDS); large ones get a device copy.
  // Results always land in the pinned block, followed by a sequence number the host spins on.
  const float* dX;
  if (n <= kPoLdsN) dX = reinterpret_cast<const float*>(sc.stage.d);
  else {
    ORBG_HIP(hipMemcpyAsync(sc.dev.p, sc.stage.h, in_bytes, hipMemcpyHostToDevice, sc.stream));
    dX = reinterpret_cast<const float*>(sc.dev.p);
  }
  uint8_t* dout = sc.stage.d + out_off;
  PoseQ* dT = reinterpret_cast<PoseQ*>(dout);
  double* dchi = reinterpret_cast<double*>(dout + sizeof(PoseQ));
  int* dstats = reinterpret_cast<int*>(dout + sizeof(PoseQ) + 4 * sizeof(double));
  uint8_t* dflag = dout + sizeof(PoseQ) + 4 * sizeof(double) + 8 * sizeof(int);
  PoseQ T0;
  {
    const float* T = p->Tcw;
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    quat_from_R(R, T0.q);
    quat_normalize(T0.q);
    T0.t[0] = T[3]; T0.t[1] = T[7]; T0.t[2] = T[11];
  }
  Cam cam{p->fx, p->fy, p->cx, p->cy, p->bf, p->bf};
  static thread_local unsigned po_seq = 0;
  po_seq = (po_seq + 1) & 0x7FFFFFFFu;
  if (po_seq == 0) po_seq = 1;
  volatile int* seq_word = reinterpret_cast<volatile int*>(sc.stage.h + out_off + sizeof(PoseQ) + 4 * sizeof(double)) + 7;
  *seq_word = 0;
  if (n <= kPoLdsN)
    hipLaunchKernelGGL(pose_opt_kernel<true>, dim3(1), dim3(kPoThreads), 0, sc.stream, n, dX, dX + 3 * (size_t)n, dX + 4 * (size_t)n,
                       dX + 5 * (size_t)n, dX + 6 * (size_t)n, cam, T0, dT, dflag, dstats, dchi, po_seq);
  else
    hipLaunchKernelGGL(pose_opt_kernel<false>, dim3(1), dim3(kPoThreads), 0, sc.stream, n, dX, dX + 3 * (size_t)n, dX + 4 * (size_t)n,
                       dX + 5 * (size_t)n, dX + 6 * (size_t)n, cam, T0, dT, dflag, dstats, dchi, po_seq);
  ORBG_HIP(hipGetLastError());
  {
    bool got = false;
    if (orbg::poll_allowed()) {
      timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
      for (unsigned spins = 0; !got; spins++) {
        if (*seq_word == (int)po_seq) { got = true; break; }
        if ((spins & 0xFFFF) == 0xFFFF) {
          timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
          if ((t1.tv_sec - t0.tv_sec) * 1000.0 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 > 100.0) break;
        }
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    if (!got) ORBG_HIP(hipStreamSynchronize(sc.stream));
  }
  const uint8_t* ho = sc.stage.h + out_off;
  PoseQ Tf;
  memcpy(&Tf, ho, sizeof(PoseQ));
  memcpy(r->chi2, ho + sizeof(PoseQ), 4 * sizeof(double));
  int stats[8];
  memcpy(stats, ho + sizeof(PoseQ) + 4 * sizeof(double), sizeof(stats));
  memcpy(r->outlier, ho + sizeof(PoseQ) + 4 * sizeof(double) + 8 * sizeof(int), (size_t)n);
  r->n_bad = stats[0];
  for (int i = 0; i < 4; i++) r->iters[i] = stats[1 + i];
  r->n_inliers = n - r->n_bad;
  double R[9];
  quat_to_R(Tf.q, R);
  for (int a = 0; a < 3; a++) { for (int c = 0; c < 3; c++) r->Tcw[4 * a + c] = (float)R[3 * a + c]; r->Tcw[4 * a + 3] = (float)Tf.t[a]; }
  r->Tcw[12] = 0; r->Tcw[13] = 0; r->Tcw[14] = 0; r->Tcw[15] = 1;
  return ORBG_OK;
}
