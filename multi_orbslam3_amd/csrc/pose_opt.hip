// liborbgpu -- Optimizer::PoseOptimization for gfx950 (MI355X): the whole solve in one kernel launch (include/orbgpu.h: pose_optimize).
#include <time.h>

#include <algorithm>
#include <cmath>
#include <limits>

#include "common.hpp"
#include "wave.hpp"
#include "se3.hpp"

using namespace orbg;
using namespace orbg_se3;

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
__device__ __forceinline__ void po_hessian(const PoEval<V>& e, V om, M mono, const Cam& cam, V* hh, V /*ur: the rig form below needs it*/) {
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
// ---- the same three pieces for a Frame with a camera rig (pose_opt_problem::rig, S/Optimizer.cc:1085-1151), one correspondence
// per call: monocular entries through mpCamera->project / projectJac (EdgeSE3ProjectXYZOnlyPose, I/OptimizableTypes.h:40-44,
// S/OptimizableTypes.cpp:46-62), the right camera's after mTrl (EdgeSE3ProjectXYZOnlyPoseToBody, I/OptimizableTypes.h:69-73,
// S/OptimizableTypes.cpp:90-108); stereo entries as above.  PoEval::Xc is the point in the LEFT camera's frame throughout.
__device__ inline void po_rig_residual(const PoseQ& T, const double* X, const double* Xl, double u, double v, double ur, const CamRig& g,
                                       double* err) {
  double uv[2];
  if (g.has_right && ur_is_right((float)ur)) {
    PoseQ Trw;
    se3_mul(g.Trl, T, &Trw);
    double r[3];
    quat_rotate(Trw.q, X, r);
    const double Xr[3] = {r[0] + Trw.t[0], r[1] + Trw.t[1], r[2] + Trw.t[2]};
    cam_project(g.right, Xr, uv);
  } else {
    cam_project(g.left, Xl, uv);
  }
  err[0] = u - uv[0]; err[1] = v - uv[1]; err[2] = 0;
}
__device__ inline void po_edge_error(const PoseQ& T, const float* X, float u, float v, float ur, const CamRig& g, double* err, double* Xc) {
  if (ur >= 0) { po_edge_error(T, X, u, v, ur, g.c, err, Xc); return; }
  const double Xd[3] = {X[0], X[1], X[2]};
  double iz;
  po_cam_point<double>(T, Xd, Xc, &iz);
  po_rig_residual(T, Xd, Xc, (double)u, (double)v, (double)ur, g, err);
}
template <class V, class M>
__device__ __forceinline__ void po_eval(const PoseQ& T, const V* X, V uu, V vv, V ur, V om, M mono, bool robust, const CamRig& g,
                                        double dM, double dS, double dsqM, double dsqS, PoEval<V>* e) {
  static_assert(sizeof(V) == sizeof(double), "the rig form evaluates one correspondence per call");
  if (ur >= 0) { po_eval<V, M>(T, X, uu, vv, ur, om, mono, robust, g.c, dM, dS, dsqM, dsqS, e); return; }
  po_cam_point<double>(T, X, e->Xc, &e->iz);
  po_rig_residual(T, X, e->Xc, uu, vv, ur, g, e->err);
  e->c2 = e->err[0] * (om * e->err[0]) + e->err[1] * (om * e->err[1]);
  po_huber<double>(robust, e->c2, dM, dsqM, 1.0, &e->rho0, &e->rho1);
}
template <class V, class M>
__device__ __forceinline__ void po_hessian(const PoEval<V>& e, V om, M mono, const CamRig& g, V* hh, V ur) {
  static_assert(sizeof(V) == sizeof(double), "the rig form evaluates one correspondence per call");
  if (ur >= 0) { po_hessian<V, M>(e, om, mono, g.c, hh, ur); return; }
  const double x = e.Xc[0], y = e.Xc[1], z = e.Xc[2];
  double P[6], Mx[6];
  if (g.has_right && ur_is_right((float)ur)) {
    double rr[3], Rrl[9];
    quat_rotate(g.Trl.q, e.Xc, rr);
    const double Xr[3] = {rr[0] + g.Trl.t[0], rr[1] + g.Trl.t[1], rr[2] + g.Trl.t[2]};
    cam_project_jac(g.right, Xr, P);
#pragma unroll
    for (int i = 0; i < 6; i++) P[i] = -P[i];
    quat_to_R(g.Trl.q, Rrl);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) Mx[3 * i + j] = P[3 * i] * Rrl[j] + P[3 * i + 1] * Rrl[3 + j] + P[3 * i + 2] * Rrl[6 + j];
  } else {
    cam_project_jac(g.left, e.Xc, P);
#pragma unroll
    for (int i = 0; i < 6; i++) Mx[i] = -P[i];
  }
  const double S[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
  double J[12];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 6; j++) J[6 * i + j] = Mx[3 * i] * S[j] + Mx[3 * i + 1] * S[6 + j] + Mx[3 * i + 2] * S[12 + j];
  const double wom = e.rho1 * om;
  const double orr[2] = {-(om * e.err[0]) * e.rho1, -(om * e.err[1]) * e.rho1};
  int o = 0;
#pragma unroll
  for (int a2 = 0; a2 < 6; a2++)
#pragma unroll
    for (int c3 = a2; c3 < 6; c3++) hh[o++] = J[a2] * (wom * J[c3]) + J[6 + a2] * (wom * J[6 + c3]);
#pragma unroll
  for (int a2 = 0; a2 < 6; a2++) hh[o++] = J[a2] * orr[0] + J[6 + a2] * orr[1];
}

// Optimizer::PoseOptimization (S/Optimizer.cc:992-1290) in ONE launch of one workgroup: 4 rounds x up to 10
// Levenberg-Marquardt iterations (g2o OptimizationAlgorithmLevenberg semantics), outlier re-classification after
// each round.  The LM state (pose, lambda, gains) is kept identically in every thread -- all of them read the same
// block sums from LDS and run the same arithmetic -- so the control flow needs no broadcast; the 6x6 solve runs on
// lanes 0..5 of each wave.
// -DPO_PROFILE: cycles of thread 0 per phase, summed over the call (build, reduce, solve, trial evaluation, trial sum, rest)
#ifdef PO_PROFILE
__device__ long long g_po_prof[16];
#define PO_IN0() po_s = clock64()
#define PO_IN(slot) do { const long long po_m = clock64(); if (threadIdx.x == 0) g_po_prof[slot] += po_m - po_s; po_s = po_m; } while (0)
#define PO_T0() long long po_t = clock64()
#define PO_ACC(slot) do { const long long po_n = clock64(); if (threadIdx.x == 0) g_po_prof[slot] += po_n - po_t; po_t = po_n; } while (0)
#else
#define PO_IN0() do { } while (0)
#define PO_IN(slot) do { } while (0)
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
          po_hessian<double, bool>(e1, om1, mono1, cam, h1, ur1);
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
        po_hessian<D2, B2>(e2, om, mono, cam, hh, ur);
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


// ---- The same solve with EIGHT wavefronts for up to 1024 correspondences (what Tracking hands over: 300-700 per call at 1000-2000
// features): one or two correspondences per thread, kept in registers, two wavefronts per SIMD.  A lone wavefront issues a dependent
// FP64 instruction every 8.3-9 cycles where the pipe takes one per 4.2 (tools/micro/fp64_latency): the per-correspondence phases
// (linearisation, trial evaluation) of the four-wavefront kernel run at ~9 cycles per instruction; with a second wavefront on every
// SIMD they share the issue slots.  The LM trial solves (6 x 6 LDL^T + exponential map, ~1200 FP64 instructions) run on wavefronts
// 0-3 only, one candidate each, while 4-7 wait at the barrier.
// Round 6 -- the block sums.  Round 5 kept the four-wavefront kernel's summation ORDER here (lane-pair exchange, then that kernel's
// LDS transpose: 28 x 512 doubles through LDS and three barriers per linearisation, four barriers per trial) so that both kernels
// gave the same bits; its own cycle model showed a quarter of the kernel in those sums (83 k of 325 k cycles for 18 linearisations,
// 68 k for 41 trials).  The order is free: g2o adds the edges serially, the oracle does the same, and what is pinned against the
// oracle is the outlier set, the inlier count and the pose (tests/test_gpu_parity.py), not the bits of a sum.  Now: the 28 sums of a
// linearisation fold across the lanes of each wavefront by recursive halving -- v_permlane32_swap / v_permlane16_swap (gfx950) hand
// half of a lane's values to its partner while it takes the partner's other half, 14 + 7 pair-adds, then four DPP row steps on the 7
// values left -- and only 8 x 28 wave totals cross LDS (one barrier + one to publish); a trial's sum is a DPP tree per wavefront and
// eight totals.  Fixed order, bit-reproducible run to run; NOT the four-wavefront kernel's bits any more (n > 1024 still runs there).
constexpr int kPoWide = 512;
// (X, Y) -> in lanes 0-31 X summed over the lane pair (l, l + 32), in lanes 32-63 Y summed over it.  v_permlane32_swap exchanges the upper
// 32 lanes of its first operand with the lower 32 of its second: X' = [X.lo | Y.lo], Y' = [X.hi | Y.hi]; X' + Y' is the fold.
__device__ __forceinline__ double po_fold32(double x, double y) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// the same across rows of 16 lanes: v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second:
// X' = [X.r0, Y.r0, X.r2, Y.r2], Y' = [X.r1, Y.r1, X.r3, Y.r3]; even rows of X' + Y' hold X over the row pair, odd rows Y
__device__ __forceinline__ double po_fold16(double x, double y) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// sum over the 16 lanes of a row, valid in the row's last lane (row_shr 1, 2, 4, 8: the scan of wave.hpp without its cross-row steps)
__device__ __forceinline__ double po_row_sum(double v) {
#define PO_DPP64(ctrl)                                                                              \
  {                                                                                                 \
    /* bound_ctrl: a lane without a source reads 0 -- no "old" operand to initialise (one v_mov less per DPP move) */ \
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xF, 0xF, true), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xF, 0xF, true); \
    v += __hiloint2double(hi, lo);                                                                  \
  }
  PO_DPP64(0x111) PO_DPP64(0x112) PO_DPP64(0x114) PO_DPP64(0x118)
#undef PO_DPP64
  return v;
}
// Block-wide sums of 28 per-thread values (8 wavefronts), out[0..28) valid for every thread afterwards.  v is consumed.
// After the two folds a lane of row r holds, in v[j], the value with index (r >> 1) * 14 + (r & 1) * 7 + j summed over the four lanes
// of its column c: 8 wavefronts x 16 columns = 128 partial sums per index go through LDS (7 stores per lane instead of 28), thread
// (index, column) adds the eight wavefronts' partials of its column in wave order, a DPP row sum adds the 16 columns.  (All of it in
// registers -- four more DPP steps on the 7 values in every wavefront -- was 215 instructions per wavefront, eight times over on
// four SIMDs: 3600 cycles per linearisation; this is ~100.)
#ifdef PO_PROFILE
#define PO_SUB(slot) do { const long long po_n = clock64(); if (threadIdx.x == 0) g_po_prof[slot] += po_n - *po_tp; *po_tp = po_n; } while (0)
#else
#define PO_SUB(slot) do { } while (0)
#endif
__device__ inline void po_block_reduce_free(double (&v)[28], double* s_p /* [8][28][16] */, double* out, [[maybe_unused]] long long* po_tp = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, row = lane >> 4, wave = tid >> 6;
#pragma unroll
  for (int j = 0; j < 14; j++) v[j] = po_fold32(v[j], v[j + 14]);
  PO_SUB(8);
#pragma unroll
  for (int j = 0; j < 7; j++) v[j] = po_fold16(v[j], v[j + 7]);
  PO_SUB(9);
  {
    double* dst = s_p + ((wave * 28 + (row >> 1) * 14 + (row & 1) * 7) << 4) + (lane & 15);
#pragma unroll
    for (int j = 0; j < 7; j++) dst[16 * j] = v[j];
  }
  PO_SUB(10);
  __syncthreads();
  PO_SUB(11);
  if (tid < 28 * 16) {                       // thread (index = tid >> 4, column = tid & 15)
    const double* p = s_p + tid;
    constexpr int kW = 28 * 16;
    double t = ((p[0] + p[kW]) + (p[2 * kW] + p[3 * kW])) + ((p[4 * kW] + p[5 * kW]) + (p[6 * kW] + p[7 * kW]));
    t = po_row_sum(t);
    if ((tid & 15) == 15) out[tid >> 4] = t;
  }
  PO_SUB(12);
  __syncthreads();
  PO_SUB(13);
}
// sum of one term per thread: DPP tree per wavefront, the eight wave totals in wave order (`slot` alternates between consecutive
// calls: one barrier per sum)
__device__ __forceinline__ double po_block_sum_free(double v, double (*w8)[8], int slot) {
  const double w = wave_sum_f64(v);
  if ((threadIdx.x & 63) == 0) w8[slot][threadIdx.x >> 6] = w;
  __syncthreads();
  const double* p = w8[slot];
  return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
}
// exact count (every term 0 or 1): any order
__device__ __forceinline__ double po_block_count_wide(double v, double* s_cnt) {
  const double w = wave_sum_f64(v);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = w;
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) s += s_cnt[k];
  __syncthreads();
  return s;
}

// NP = 1: n <= 512, one correspondence per thread; NP = 2: n <= 1024, two (a thread adds its own two terms first).
// CamT: Cam (the five pinhole scalars: every BASELINE configuration) or CamRig (a Frame with mpCamera / mpCamera2 models; NP up to 8).
template <int NP, class CamT = Cam>
__global__ __launch_bounds__(kPoWide) void pose_opt_wide_kernel(int n, const float* g_Xw, const float* g_ou, const float* g_ov, const float* g_our,
                                                                const float* g_oinv, CamT cam, PoseQ T0, PoseQ* __restrict__ T_out,
                                                                uint8_t* __restrict__ outlier_out, int* __restrict__ stats, double* __restrict__ chi_out,
                                                                unsigned seq) {
  __shared__ double s_w[8 * 28 * 16];
  __shared__ double red[28];
  __shared__ double s_w8[2][8];
  __shared__ double s_cnt[8];
  __shared__ double s_cand[4][14];
  int sum_slot = 0;
  const int tid = threadIdx.x;
  const int li = min(tid & 63, 5);
  const double dM = (float)sqrt(5.991), dS = (float)sqrt(7.815);
  const double dsqM = dM * dM, dsqS = dS * dS;
  // this thread's correspondences: read once (the inputs may sit in mapped host memory), kept in registers as the floats they are
  bool have[NP], mono[NP], my_out[NP];
  int idx[NP];
  float Xf[NP][3], uf[NP], vf[NP], urf[NP], omf[NP];
  double my_chi2[NP];                                        // last evaluated chi2 of the correspondence
#pragma unroll
  for (int p = 0; p < NP; p++) {
    const int i_mine = tid + 512 * p;                                // correspondences tid, tid + 512 (coalesced reads)
    have[p] = i_mine < n;
    const int i = have[p] ? i_mine : 0;
    idx[p] = i;
    Xf[p][0] = g_Xw[3 * i]; Xf[p][1] = g_Xw[3 * i + 1]; Xf[p][2] = g_Xw[3 * i + 2];
    uf[p] = g_ou[i]; vf[p] = g_ov[i]; urf[p] = g_our[i]; omf[p] = g_oinv[i];
    mono[p] = urf[p] < 0;
    my_out[p] = false; my_chi2[p] = 0;
  }
  double x[6] = {0, 0, 0, 0, 0, 0};
  double lambda = 0, ni = 2, currentChi = 0;
  int nBadLM = 0;
  bool robust = true;
  PoseQ T = T0;
  int nBad = 0;
  if (tid == 0) { for (int k = 0; k < 7; k++) stats[k] = 0; for (int k = 0; k < 4; k++) chi_out[k] = 0; }
  PO_T0();
#ifdef PO_PROFILE
  long long po_s = 0;
#endif
  for (int round = 0; round < 4; round++) {
    T = T0;                                                   // setEstimate(toSE3Quat(mTcw)) every round (:1191)
    bool active[NP];
    double cnt = 0;
#pragma unroll
    for (int p = 0; p < NP; p++) { active[p] = have[p] && !my_out[p]; cnt += active[p] ? 1.0 : 0.0; }
    const int n_active = (int)po_block_count_wide(cnt, s_cnt);
    int done = 0;
    bool ok = n_active > 0;
    for (int it = 0; it < 10 && ok; it++) {
      // ---- computeActiveErrors + buildSystem at T
      double acc[28];
#pragma unroll
      for (int k = 0; k < 28; k++) acc[k] = 0;
#pragma unroll
      for (int p = 0; p < NP; p++) {
        if (active[p]) {
          const double X[3] = {(double)Xf[p][0], (double)Xf[p][1], (double)Xf[p][2]};
          const double om = (double)omf[p];
          PoEval<double> e1;
          po_eval<double, bool>(T, X, (double)uf[p], (double)vf[p], (double)urf[p], om, mono[p], robust, cam, dM, dS, dsqM, dsqS, &e1);
          double h1[27];
          po_hessian<double, bool>(e1, om, mono[p], cam, h1, (double)urf[p]);
          my_chi2[p] = e1.c2;
          acc[27] += e1.rho0;
#pragma unroll
          for (int o = 0; o < 27; o++) acc[o] += h1[o];
        }
      }
      PO_ACC(0);
#ifdef PO_PROFILE
      po_block_reduce_free(acc, s_w, red, &po_t);
#else
      po_block_reduce_free(acc, s_w, red);
#endif
      PO_ACC(1);
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
      // ---- LM trials: wavefronts 0-3 solve the next four candidates of a run of rejections (see pose_opt_kernel), 4-7 wait
      double rho = 0;
      int qmax = 0;
      for (;;) {
        PO_ACC(5);
        const int cslot = qmax & 3;
        if (cslot == 0) {
          if (tid < 256) {
            double lam_c = lambda, ni_c = ni;
            const int wv = tid >> 6;
            for (int cc = 0; cc < wv; cc++) { lam_c *= ni_c; ni_c *= 2; }
            double xc[6] = {x[0], x[1], x[2], x[3], x[4], x[5]};
            PO_IN0();
            const bool okc = po_solve6(Hrow, b_li, li, lam_c, xc);
            PO_IN(6);
            PoseQ Tc;
            pose_oplus_series(T, xc, &Tc);
            PO_IN(7);
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
          }
          __syncthreads();
          PO_IN(14);
        }
        PO_IN0();
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
        PO_IN(15);
        PO_ACC(2);
        double tchi = 0;
#pragma unroll
        for (int p = 0; p < NP; p++) {
          if (active[p]) {
            const double X[3] = {(double)Xf[p][0], (double)Xf[p][1], (double)Xf[p][2]};
            PoEval<double> e1;
            po_eval<double, bool>(Tt, X, (double)uf[p], (double)vf[p], (double)urf[p], (double)omf[p], mono[p], robust, cam, dM, dS, dsqM, dsqS, &e1);
            my_chi2[p] = e1.c2; tchi += e1.rho0;
          }
        }
        PO_ACC(3);
        double tempChi = po_block_sum_free(tchi, s_w8, sum_slot); sum_slot ^= 1;
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
#pragma unroll
    for (int p = 0; p < NP; p++) {
      if (!have[p]) continue;
      if (my_out[p]) {
        double err[3], Xc[3];
        po_edge_error(T, Xf[p], uf[p], vf[p], urf[p], cam, err, Xc);
        const double om = (double)omf[p];
        my_chi2[p] = err[0] * (om * err[0]) + err[1] * (om * err[1]) + (mono[p] ? 0.0 : err[2] * (om * err[2]));
      }
      const float c2f = (float)my_chi2[p];
      my_out[p] = c2f > (mono[p] ? 5.991f : 7.815f);
      bl += my_out[p] ? 1.0 : 0.0;
    }
    nBad = (int)po_block_count_wide(bl, s_cnt);
    if (round == 2) robust = false;                          // setRobustKernel(0)
    if (n < 10) break;                                        // optimizer.edges().size() < 10
  }
  PO_ACC(5);
#pragma unroll
  for (int p = 0; p < NP; p++) if (have[p]) outlier_out[idx[p]] = my_out[p] ? 1 : 0;
  if (tid == 0) { *T_out = T; stats[0] = nBad; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");              // system-scope release of every wavefront's results (no acquire half)
  __syncthreads();
  if (tid == 0) *reinterpret_cast<volatile int*>(&stats[7]) = (int)seq;   // results are complete: the host spins on this word
}

}  // namespace

#ifdef PO_PROFILE
extern "C" int pose_opt_debug_prof(long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_po_prof), sizeof(g_po_prof)) != hipSuccess) return -1;
  if (reset) { long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_po_prof), z, sizeof(z)) != hipSuccess) return -1; }
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
  if (p->rig && p->n > 8 * kPoWide) return ORBG_CAP_EXCEEDED;           // (the rig form runs on the wide kernel only: 4096 correspondences)
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
  CamRig rig;
  if (p->rig && !cam_rig_from(*p->rig, cam, &rig)) return ORBG_BAD_ARG;
  static thread_local unsigned po_seq = 0;
  po_seq = (po_seq + 1) & 0x7FFFFFFFu;
  if (po_seq == 0) po_seq = 1;
  volatile int* seq_word = reinterpret_cast<volatile int*>(sc.stage.h + out_off + sizeof(PoseQ) + 4 * sizeof(double)) + 7;
  *seq_word = 0;
  if (p->rig) {
    // a Frame with camera models / a second camera: the wide kernel over CamRig, up to eight correspondences per thread
#define ORBG_PO_RIG(NP)                                                                                                              \
  hipLaunchKernelGGL((pose_opt_wide_kernel<NP, CamRig>), dim3(1), dim3(kPoWide), 0, sc.stream, n, dX, dX + 3 * (size_t)n,            \
                     dX + 4 * (size_t)n, dX + 5 * (size_t)n, dX + 6 * (size_t)n, rig, T0, dT, dflag, dstats, dchi, po_seq)
    if (n <= kPoWide) ORBG_PO_RIG(1);
    else if (n <= 2 * kPoWide) ORBG_PO_RIG(2);
    else if (n <= 4 * kPoWide) ORBG_PO_RIG(4);
    else ORBG_PO_RIG(8);
#undef ORBG_PO_RIG
  } else if (n <= kPoWide)
    hipLaunchKernelGGL(pose_opt_wide_kernel<1>, dim3(1), dim3(kPoWide), 0, sc.stream, n, dX, dX + 3 * (size_t)n, dX + 4 * (size_t)n,
                       dX + 5 * (size_t)n, dX + 6 * (size_t)n, cam, T0, dT, dflag, dstats, dchi, po_seq);
  else if (n <= 2 * kPoWide)
    hipLaunchKernelGGL(pose_opt_wide_kernel<2>, dim3(1), dim3(kPoWide), 0, sc.stream, n, dX, dX + 3 * (size_t)n, dX + 4 * (size_t)n,
                       dX + 5 * (size_t)n, dX + 6 * (size_t)n, cam, T0, dT, dflag, dstats, dchi, po_seq);
  else if (n <= kPoLdsN)
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
