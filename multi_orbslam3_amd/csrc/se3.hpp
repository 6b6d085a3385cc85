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

// SE3Quat::operator* (G/types/se3quat.h:104-110): t = a.t + a.q * b.t, q = a.q * b.q, normalizeRotation()
__host__ __device__ inline void se3_mul(const PoseQ& a, const PoseQ& b, PoseQ* out) {
  double rt[3];
  quat_rotate(a.q, b.t, rt);
  out->t[0] = a.t[0] + rt[0]; out->t[1] = a.t[1] + rt[1]; out->t[2] = a.t[2] + rt[2];
  const double* x = a.q; const double* y = b.q;
  out->q[3] = x[3] * y[3] - x[0] * y[0] - x[1] * y[1] - x[2] * y[2];
  out->q[0] = x[3] * y[0] + x[0] * y[3] + x[1] * y[2] - x[2] * y[1];
  out->q[1] = x[3] * y[1] + x[1] * y[3] + x[2] * y[0] - x[0] * y[2];
  out->q[2] = x[3] * y[2] + x[2] * y[3] + x[0] * y[1] - x[1] * y[0];
  quat_normalize(out->q);
}

// ---- the cameras behind GeometricCamera::project / projectJac (Eigen forms), for the problems that carry an orbg_camera_rig
// (include/orbgpu.h): Pinhole S/CameraModels/Pinhole.cpp:41-47,81-91; KannalaBrandt8 S/CameraModels/KannalaBrandt8.cpp:52-69,166-196.
// mvParameters are float32 in the reference and meet doubles everywhere they are used: kept here as the doubles they promote to.
struct CamModelD { int model; double fx, fy, cx, cy, k1, k2, k3, k4; };
// Cam c: the five scalars the stereo edges carry; left / right / Trl: mpCamera, mpCamera2, Converter::toSE3Quat(mTrl)
struct CamRig { Cam c; CamModelD left, right; PoseQ Trl; int has_right; };

__device__ inline void cam_project(const CamModelD& m, const double* v, double* uv) {
  if (m.model == 1) {
    // theta and psi are float32 VALUES in the reference (atan2f / sqrtf): the correctly rounded float of the double result here
    const double x2_plus_y2 = v[0] * v[0] + v[1] * v[1];
    const double theta = (double)(float)atan2((double)sqrtf((float)x2_plus_y2), (double)(float)v[2]);
    const double psi = (double)(float)atan2((double)(float)v[1], (double)(float)v[0]);
    const double theta2 = theta * theta;
    const double theta3 = theta * theta2;
    const double theta5 = theta3 * theta2;
    const double theta7 = theta5 * theta2;
    const double theta9 = theta7 * theta2;
    const double r = theta + m.k1 * theta3 + m.k2 * theta5 + m.k3 * theta7 + m.k4 * theta9;
    double sn, cs;
    sincos(psi, &sn, &cs);
    uv[0] = m.fx * r * cs + m.cx;
    uv[1] = m.fy * r * sn + m.cy;
  } else {
    uv[0] = m.fx * v[0] / v[2] + m.cx;
    uv[1] = m.fy * v[1] / v[2] + m.cy;
  }
}

// J = d project / d v, 2 x 3 row-major
__device__ inline void cam_project_jac(const CamModelD& m, const double* v, double* J) {
  if (m.model == 1) {
    const double x2 = v[0] * v[0], y2 = v[1] * v[1], z2 = v[2] * v[2];
    const double r2 = x2 + y2;
    const double r = sqrt(r2);
    const double r3 = r2 * r;
    const double theta = atan2(r, v[2]);
    const double theta2 = theta * theta, theta3 = theta2 * theta;
    const double theta4 = theta2 * theta2, theta5 = theta4 * theta;
    const double theta6 = theta2 * theta4, theta7 = theta6 * theta;
    const double theta8 = theta4 * theta4, theta9 = theta8 * theta;
    const double f = theta + theta3 * m.k1 + theta5 * m.k2 + theta7 * m.k3 + theta9 * m.k4;
    const double fd = 1 + 3 * m.k1 * theta2 + 5 * m.k2 * theta4 + 7 * m.k3 * theta6 + 9 * m.k4 * theta8;
    const double den = r2 * (r2 + z2);
    J[0] = m.fx * (fd * v[2] * x2 / den + f * y2 / r3);
    J[3] = m.fy * (fd * v[2] * v[1] * v[0] / den - f * v[1] * v[0] / r3);
    J[1] = m.fx * (fd * v[2] * v[1] * v[0] / den - f * v[1] * v[0] / r3);
    J[4] = m.fy * (fd * v[2] * y2 / den + f * x2 / r3);
    J[2] = -m.fx * fd * v[0] / (r2 + z2);
    J[5] = -m.fy * fd * v[1] / (r2 + z2);
  } else {
    J[0] = m.fx / v[2]; J[1] = 0.0; J[2] = -m.fx * v[0] / (v[2] * v[2]);
    J[3] = 0.0; J[4] = m.fy / v[2]; J[5] = -m.fy * v[1] / (v[2] * v[2]);
  }
}

// orbg_camera / orbg_camera_rig (include/orbgpu.h; templates so that this header needs no other) -> what the kernels take;
// Converter::toSE3Quat(mTrl) (S/Converter.cc:34-44) as for the keyframe poses.  false: a camera model this library does not know.
template <class CameraC>
inline bool cam_model_from(const CameraC& c, CamModelD* m) {
  if (c.model != 0 && c.model != 1) return false;
  *m = CamModelD{c.model, c.fx, c.fy, c.cx, c.cy, c.k[0], c.k[1], c.k[2], c.k[3]};
  return true;
}
template <class RigC>
inline bool cam_rig_from(const RigC& rig, const Cam& scalars, CamRig* g) {
  g->c = scalars;
  g->has_right = rig.has_right ? 1 : 0;
  if (!cam_model_from(rig.left, &g->left)) return false;
  g->right = g->left;
  g->Trl = PoseQ{{0, 0, 0, 1}, {0, 0, 0}};
  if (rig.has_right) {
    if (!cam_model_from(rig.right, &g->right)) return false;
    const float* T = rig.Trl;
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    quat_from_R(R, g->Trl.q);
    quat_normalize(g->Trl.q);
    g->Trl.t[0] = T[3]; g->Trl.t[1] = T[7]; g->Trl.t[2] = T[11];
  }
  return true;
}

// `ur` of an observation made by the rig's right camera (LBA_UR_RIGHT_CAMERA, include/orbgpu.h)
__host__ __device__ inline bool ur_is_right(float ur) { return ur <= -1.5f; }

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
