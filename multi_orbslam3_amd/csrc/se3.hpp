// SE3 / quaternion helpers shared by the local BA (lba.hip) and PoseOptimization (pose_opt.hip): Eigen / g2o semantics, see
// oracle/lba.cc for the citations.  Everything is inline (two translation units of one shared object).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>

namespace orbg_se3 {

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

}  // namespace orbg_se3
