// ORACLE (test infrastructure, not product code) -- see orb_oracle.h for status: parity unpinned.
//
// CPU restatement of the numerical core of Optimizer::LocalBundleAdjustment (S/Optimizer.cc:1917-2267,
// 2321-2396) together with the vendored g2o pieces it drives:
//   OptimizationAlgorithmLevenberg::solve      G/core/optimization_algorithm_levenberg.cpp:61-194
//   SparseOptimizer::optimize/update/push/pop  G/core/sparse_optimizer.cpp:354-439
//   BlockSolver<6,3> buildSystem/setLambda/solve  G/core/block_solver.hpp:354-486,502-604
//   BaseBinaryEdge::constructQuadraticForm     G/core/base_binary_edge.hpp:55-120
//   RobustKernelHuber::robustify               G/core/robust_kernel_impl.cpp:78-91
//   EdgeStereoSE3ProjectXYZ                    G/types/types_six_dof_expmap.{h:146-175,cpp:190-197,228-274}
//   ORB_SLAM3::EdgeSE3ProjectXYZ               I/OptimizableTypes.h:89-115, S/OptimizableTypes.cpp:139-160
//   SE3Quat::exp / operator* / normalizeRotation  G/types/se3quat.h:102-110,225-260,280-285
// Eigen pieces restated from Eigen 3.x: Quaterniond(Matrix3d), toRotationMatrix, q*v, 3x3 inverse,
// SimplicialLDLT (here: dense LDL^T of the reduced camera matrix, no pivoting, fails on a zero pivot).
// All arithmetic is double; inputs/outputs are float32 exactly where the reference converts
// (S/Converter.cc:33-64).  Summation orders follow g2o (edges in creation order).

#include "orb_oracle.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <utility>
#include <vector>

namespace {

struct Quat { double x, y, z, w; };
struct PoseQ { Quat q; double t[3]; };

// Eigen::Quaterniond(const Matrix3d&)  (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl)
Quat quat_from_R(const double m[9]) {
  Quat q;
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m[7] - m[5]) * t;
    q.y = (m[2] - m[6]) * t;
    q.z = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (m[3 * k + j] - m[3 * j + k]) * t;
    v[j] = (m[3 * j + i] + m[3 * i + j]) * t;
    v[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}

// SE3Quat::normalizeRotation -- G/types/se3quat.h:280-285
void normalize_rotation(Quat& q) {
  if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
  double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}

// Eigen QuaternionBase::toRotationMatrix
void quat_to_R(const Quat& q, double R[9]) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

// Eigen QuaternionBase::_transformVector: v + w*uv + q.vec x uv, uv = 2 * q.vec x v
void quat_rotate(const Quat& q, const double v[3], double out[3]) {
  double uv[3] = {2 * (q.y * v[2] - q.z * v[1]), 2 * (q.z * v[0] - q.x * v[2]), 2 * (q.x * v[1] - q.y * v[0])};
  out[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
  out[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
  out[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}

// Eigen quaternion product a*b
Quat quat_mul(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}

// SE3Quat::exp -- G/types/se3quat.h:225-260 (update = [omega, upsilon])
PoseQ se3_exp(const double u[6]) {
  const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
  const double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
  const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double O2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += O[3 * i + k] * O[3 * k + j];
      O2[3 * i + j] = s;
    }
  double R[9], V[9];
  if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
  } else {
    const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta);
    const double c = (theta - std::sin(theta)) / std::pow(theta, 3);
    for (int i = 0; i < 9; i++) {
      const double I = (i % 4 == 0 ? 1.0 : 0.0);
      R[i] = I + a * O[i] + b * O2[i];
      V[i] = I + b * O[i] + c * O2[i];
    }
  }
  PoseQ p;
  p.q = quat_from_R(R);
  for (int i = 0; i < 3; i++) p.t[i] = V[3 * i] * up[0] + V[3 * i + 1] * up[1] + V[3 * i + 2] * up[2];
  normalize_rotation(p.q);   // SE3Quat(const Quaterniond&, const Vector3d&) ctor :62-64
  return p;
}

// VertexSE3Expmap::oplusImpl: estimate = exp(update) * estimate  (G/types/types_six_dof_expmap.h:73-76,
// SE3Quat::operator* se3quat.h:102-110)
void pose_oplus(PoseQ& T, const double u[6]) {
  PoseQ e = se3_exp(u);
  double rt[3];
  quat_rotate(e.q, T.t, rt);
  PoseQ r;
  for (int i = 0; i < 3; i++) r.t[i] = e.t[i] + rt[i];
  r.q = quat_mul(e.q, T.q);
  normalize_rotation(r.q);
  T = r;
}

struct Cam { double fx, fy, cx, cy, bf; float bf_f; };

// Converter::toSE3Quat(cv::Mat) -- S/Converter.cc:34-44: float32 entries into an Eigen double matrix, SE3Quat(R, t)
PoseQ pose_from_3x4(const float* T, int row_stride) {
  PoseQ P;
  const double R[9] = {T[0], T[1], T[2], T[row_stride], T[row_stride + 1], T[row_stride + 2], T[2 * row_stride], T[2 * row_stride + 1],
                       T[2 * row_stride + 2]};
  P.q = quat_from_R(R);
  normalize_rotation(P.q);
  P.t[0] = T[3]; P.t[1] = T[row_stride + 3]; P.t[2] = T[2 * row_stride + 3];
  return P;
}

// GeometricCamera::project(const Eigen::Vector3d&) / projectJac(const Eigen::Vector3d&): what the ORB_SLAM3:: edges call
// (I/OptimizableTypes.h:43,72,103,131; S/OptimizableTypes.cpp:46-62,90-108,139-160,192-214).  mvParameters are float32 and are
// promoted where they meet a double.
//   Pinhole          S/CameraModels/Pinhole.cpp:41-47, 81-91
//   KannalaBrandt8   S/CameraModels/KannalaBrandt8.cpp:52-69 (theta and psi through atan2f / sqrtf: float32 values), 166-196
struct CamModel { int model; float p[8]; };

void cam_project(const CamModel& c, const double v[3], double uv[2]) {
  if (c.model == ORBG_CAM_KANNALA_BRANDT8) {
    const double x2_plus_y2 = v[0] * v[0] + v[1] * v[1];
    const double theta = atan2f(sqrtf(x2_plus_y2), v[2]);
    const double psi = atan2f(v[1], v[0]);
    const double theta2 = theta * theta;
    const double theta3 = theta * theta2;
    const double theta5 = theta3 * theta2;
    const double theta7 = theta5 * theta2;
    const double theta9 = theta7 * theta2;
    const double r = theta + c.p[4] * theta3 + c.p[5] * theta5 + c.p[6] * theta7 + c.p[7] * theta9;
    uv[0] = c.p[0] * r * std::cos(psi) + c.p[2];
    uv[1] = c.p[1] * r * std::sin(psi) + c.p[3];
  } else {
    uv[0] = c.p[0] * v[0] / v[2] + c.p[2];
    uv[1] = c.p[1] * v[1] / v[2] + c.p[3];
  }
}

// J = d project / d v, 2 x 3 row-major
void cam_project_jac(const CamModel& c, const double v[3], double J[6]) {
  if (c.model == ORBG_CAM_KANNALA_BRANDT8) {
    const double x2 = v[0] * v[0], y2 = v[1] * v[1], z2 = v[2] * v[2];
    const double r2 = x2 + y2;
    const double r = std::sqrt(r2);
    const double r3 = r2 * r;
    const double theta = std::atan2(r, v[2]);
    const double theta2 = theta * theta, theta3 = theta2 * theta;
    const double theta4 = theta2 * theta2, theta5 = theta4 * theta;
    const double theta6 = theta2 * theta4, theta7 = theta6 * theta;
    const double theta8 = theta4 * theta4, theta9 = theta8 * theta;
    const double f = theta + theta3 * c.p[4] + theta5 * c.p[5] + theta7 * c.p[6] + theta9 * c.p[7];
    const double fd = 1 + 3 * c.p[4] * theta2 + 5 * c.p[5] * theta4 + 7 * c.p[6] * theta6 + 9 * c.p[7] * theta8;
    J[0] = c.p[0] * (fd * v[2] * x2 / (r2 * (r2 + z2)) + f * y2 / r3);
    J[3] = c.p[1] * (fd * v[2] * v[1] * v[0] / (r2 * (r2 + z2)) - f * v[1] * v[0] / r3);
    J[1] = c.p[0] * (fd * v[2] * v[1] * v[0] / (r2 * (r2 + z2)) - f * v[1] * v[0] / r3);
    J[4] = c.p[1] * (fd * v[2] * y2 / (r2 * (r2 + z2)) + f * x2 / r3);
    J[2] = -c.p[0] * fd * v[0] / (r2 + z2);
    J[5] = -c.p[1] * fd * v[1] / (r2 + z2);
  } else {
    J[0] = c.p[0] / v[2]; J[1] = 0.f; J[2] = -c.p[0] * v[0] / (v[2] * v[2]);
    J[3] = 0.f; J[4] = c.p[1] / v[2]; J[5] = -c.p[1] * v[1] / (v[2] * v[2]);
  }
}

// mpCamera / mpCamera2 / mTrl of the keyframes (orbg_camera_rig).  on == false: the pinhole-only problem every BASELINE
// configuration is; its monocular edge is written out below with the five scalars (the same expressions as Pinhole::project /
// projectJac give).
struct Rig {
  bool on = false, has_right = false;
  CamModel left{}, right{};
  PoseQ Trl{};          // Converter::toSE3Quat(mTrl), S/Converter.cc:34-44
};

CamModel cam_model_of(const orbg_camera& c) {
  CamModel m;
  m.model = c.model;
  m.p[0] = c.fx; m.p[1] = c.fy; m.p[2] = c.cx; m.p[3] = c.cy;
  for (int i = 0; i < 4; i++) m.p[4 + i] = c.k[i];
  return m;
}

Rig rig_of(const orbg_camera_rig* r) {
  Rig g;
  if (!r) return g;
  g.on = true;
  g.left = cam_model_of(r->left);
  g.has_right = r->has_right != 0;
  if (g.has_right) {
    g.right = cam_model_of(r->right);
    g.Trl = pose_from_3x4(r->Trl, 4);
  }
  return g;
}

// SE3Quat::operator* -- G/types/se3quat.h:104-110
PoseQ se3_mul(const PoseQ& a, const PoseQ& b) {
  PoseQ r;
  double rt[3];
  quat_rotate(a.q, b.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] = a.t[i] + rt[i];
  r.q = quat_mul(a.q, b.q);
  normalize_rotation(r.q);
  return r;
}

inline bool edge_is_right(float ur) { return ur <= -1.5f; }      // LBA_UR_RIGHT_CAMERA

// computeError: stereo G/types/types_six_dof_expmap.h:156-161 + cam_project cpp:190-197 (invz is float32);
// mono I/OptimizableTypes.h:99-104 + Pinhole::project S/CameraModels/Pinhole.cpp:41-47.
inline int edge_dim(const lba_edge& e) { return e.ur < 0 ? 2 : 3; }

// Xc: the point in the frame of the camera that made the observation (what isDepthPositive() looks at).
void edge_error(const PoseQ& T, const double X[3], const Cam& c, const Rig& rig, const lba_edge& e, double err[3], double Xc[3]) {
  double r[3];
  if (rig.has_right && edge_is_right(e.ur)) {
    // EdgeSE3ProjectXYZToBody::computeError, I/OptimizableTypes.h:127-132: obs - pCamera->project((mTrl * T).map(X))
    const PoseQ Trw = se3_mul(rig.Trl, T);
    quat_rotate(Trw.q, X, r);
    for (int i = 0; i < 3; i++) Xc[i] = r[i] + Trw.t[i];
    double uv[2];
    cam_project(rig.right, Xc, uv);
    err[0] = (double)e.u - uv[0]; err[1] = (double)e.v - uv[1]; err[2] = 0;
    return;
  }
  quat_rotate(T.q, X, r);
  for (int i = 0; i < 3; i++) Xc[i] = r[i] + T.t[i];
  if (e.ur < 0 && rig.on) {                                   // EdgeSE3ProjectXYZ::computeError through mpCamera, I/OptimizableTypes.h:99-104
    double uv[2];
    cam_project(rig.left, Xc, uv);
    err[0] = (double)e.u - uv[0]; err[1] = (double)e.v - uv[1]; err[2] = 0;
    return;
  }
  if (e.ur < 0) {
    err[0] = (double)e.u - (c.fx * Xc[0] / Xc[2] + c.cx);
    err[1] = (double)e.v - (c.fy * Xc[1] / Xc[2] + c.cy);
    err[2] = 0;
  } else {
    const float invz = (float)(1.0f / Xc[2]);
    const double r0 = Xc[0] * invz * c.fx + c.cx;
    const double r1 = Xc[1] * invz * c.fy + c.cy;
    const double r2 = r0 - (double)(c.bf_f * invz);
    err[0] = (double)e.u - r0; err[1] = (double)e.v - r1; err[2] = (double)e.ur - r2;
  }
}

// linearizeOplus: stereo cpp:228-274, mono S/OptimizableTypes.cpp:139-160 (+ projectJac Pinhole.cpp:81-91).
// A = d err / d point (D x 3), B = d err / d pose (D x 6, [omega, upsilon]).
// Xc = T.map(X), the point in the LEFT camera's (body) frame, for every kind of edge.
void edge_jacobians(const PoseQ& T, const double Xc[3], const Cam& c, const Rig& rig, const lba_edge& e, double A[9], double B[18]) {
  double R[9];
  const double x = Xc[0], y = Xc[1], z = Xc[2];
  if (rig.on && e.ur < 0) {
    const double S[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
    double J[6], M[6];
    if (rig.has_right && edge_is_right(e.ur)) {
      // EdgeSE3ProjectXYZToBody::linearizeOplus, S/OptimizableTypes.cpp:192-214: X_r = mTrl.map(T_lw.map(X_w));
      // Xi = -projectJac(X_r) * (mTrl * T_lw).rotation();  Xj = -projectJac(X_r) * mTrl.rotation() * SE3deriv(X_l)
      double rr[3], Xr[3];
      quat_rotate(rig.Trl.q, Xc, rr);
      for (int i = 0; i < 3; i++) Xr[i] = rr[i] + rig.Trl.t[i];
      cam_project_jac(rig.right, Xr, J);
      for (int i = 0; i < 6; i++) J[i] = -J[i];
      const PoseQ Trw = se3_mul(rig.Trl, T);
      quat_to_R(Trw.q, R);
      double Rrl[9];
      quat_to_R(rig.Trl.q, Rrl);
      for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) M[3 * i + j] = J[3 * i] * Rrl[j] + J[3 * i + 1] * Rrl[3 + j] + J[3 * i + 2] * Rrl[6 + j];
    } else {
      // EdgeSE3ProjectXYZ::linearizeOplus, S/OptimizableTypes.cpp:139-160, through mpCamera->projectJac
      cam_project_jac(rig.left, Xc, J);
      for (int i = 0; i < 6; i++) { J[i] = -J[i]; M[i] = J[i]; }
      quat_to_R(T.q, R);
    }
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 3; j++) A[3 * i + j] = J[3 * i] * R[j] + J[3 * i + 1] * R[3 + j] + J[3 * i + 2] * R[6 + j];
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 6; j++) B[6 * i + j] = M[3 * i] * S[j] + M[3 * i + 1] * S[6 + j] + M[3 * i + 2] * S[12 + j];
    for (int j = 0; j < 3; j++) A[6 + j] = 0;
    for (int j = 0; j < 6; j++) B[12 + j] = 0;
    return;
  }
  quat_to_R(T.q, R);
  if (e.ur < 0) {
    // projectJac = -[fx/z 0 -fx x/z^2; 0 fy/z -fy y/z^2]
    const double J[6] = {-(c.fx / z), -0.0, -(-c.fx * x / (z * z)), -0.0, -(c.fy / z), -(-c.fy * y / (z * z))};
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 3; j++) A[3 * i + j] = J[3 * i] * R[j] + J[3 * i + 1] * R[3 + j] + J[3 * i + 2] * R[6 + j];
    const double S[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 6; j++) B[6 * i + j] = J[3 * i] * S[j] + J[3 * i + 1] * S[6 + j] + J[3 * i + 2] * S[12 + j];
    for (int j = 0; j < 3; j++) A[6 + j] = 0;
    for (int j = 0; j < 6; j++) B[12 + j] = 0;
  } else {
    const double z_2 = z * z;
    for (int j = 0; j < 3; j++) {
      A[j] = -c.fx * R[j] / z + c.fx * x * R[6 + j] / z_2;
      A[3 + j] = -c.fy * R[3 + j] / z + c.fy * y * R[6 + j] / z_2;
      A[6 + j] = A[j] - c.bf * R[6 + j] / z_2;
    }
    B[0] = x * y / z_2 * c.fx;
    B[1] = -(1 + (x * x / z_2)) * c.fx;
    B[2] = y / z * c.fx;
    B[3] = -1. / z * c.fx;
    B[4] = 0;
    B[5] = x / z_2 * c.fx;
    B[6] = (1 + y * y / z_2) * c.fy;
    B[7] = -x * y / z_2 * c.fy;
    B[8] = -x / z * c.fy;
    B[9] = 0;
    B[10] = -1. / z * c.fy;
    B[11] = y / z_2 * c.fy;
    B[12] = B[0] - c.bf * y / z_2;
    B[13] = B[1] + c.bf * x / z_2;
    B[14] = B[2];
    B[15] = B[3];
    B[16] = 0;
    B[17] = B[5] - c.bf / z_2;
  }
}

// RobustKernelHuber::robustify -- G/core/robust_kernel_impl.cpp:78-91
inline void huber(double e, double delta, double dsqr, double rho[2]) {
  if (e <= dsqr) { rho[0] = e; rho[1] = 1.; }
  else {
    const double sqrte = std::sqrt(e);
    rho[0] = 2 * sqrte * delta - dsqr;
    rho[1] = delta / sqrte;
  }
}

// Eigen fixed-size 3x3 inverse (cofactors / determinant)
bool inv3(const double m[9], double o[9]) {
  const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  const double id = 1.0 / det;
  o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return std::isfinite(id);
}

// Dense LDL^T solve of the symmetric system S x = b using the upper triangle of S (n x n, row-major).
// Mirrors Eigen::SimplicialLDLT semantics used by LinearSolverEigen (G/solvers/linear_solver_eigen.h:94-124):
// no pivoting, failure only on an exactly-zero (or non-finite) pivot.
bool ldlt_solve(std::vector<double>& S, int n, const double* b, double* x) {
  // factor in place: S(i,j) for j>=i holds U = L^T, D on the diagonal
  std::vector<double> D(n);
  for (int j = 0; j < n; j++) {
    double d = S[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= S[(size_t)k * n + j] * S[(size_t)k * n + j] * D[k];
    if (d == 0.0 || !std::isfinite(d)) return false;
    D[j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = S[(size_t)j * n + i];
      for (int k = 0; k < j; k++) s -= S[(size_t)k * n + j] * S[(size_t)k * n + i] * D[k];
      S[(size_t)j * n + i] = s / d;      // L(i,j) stored at (j,i)
    }
  }
  std::vector<double> y(b, b + n);
  for (int i = 0; i < n; i++)            // L y = b
    for (int k = 0; k < i; k++) y[i] -= S[(size_t)k * n + i] * y[k];
  for (int i = 0; i < n; i++) y[i] /= D[i];
  for (int i = n - 1; i >= 0; i--) {     // L^T x = y
    double s = y[i];
    for (int k = i + 1; k < n; k++) s -= S[(size_t)i * n + k] * x[k];
    x[i] = s;
  }
  return true;
}

struct Lba {
  const lba_problem* p;
  Cam cam;
  Rig rig;
  std::vector<PoseQ> poses;
  std::vector<double> points;          // 3 per point
  std::vector<int> pose_col, point_col;  // hessian index or -1
  int nP = 0, nL = 0;
  std::vector<int> active_pose, active_point;
  std::vector<double> err;             // 3 per edge (last computeActiveErrors)
  std::vector<double> chi2;            // per edge
  // system
  std::vector<double> Hpp, Hll, Hpl, bp, bl;   // Hpp: nP x 36, Hll: nL x 9, Hpl: per edge 18 (6x3), b
  // g2o keeps ONE Hpl block per (pose, landmark) vertex pair (BlockSolver::buildStructure: _Hpl->block(ind1, ind2, true),
  // G/core/block_solver.hpp:218-240) and every edge between the two adds into it.  With a camera rig a keyframe observes a landmark
  // up to twice (left and right camera): hpl_slot[k] is the first edge of k's vertex pair, whose 18 values hold that block.
  std::vector<int> hpl_slot;
  std::vector<double> x;               // 6nP + 3nL
  double delta_mono, delta_stereo, dsqr_mono, dsqr_stereo;
  const volatile int32_t* stop;
  int trials_done = 0;        // LM trials evaluated so far: a negative flag value -k means "stop once k trials are done" (orbgpu.h)
  bool past_precheck = false; // INT32_MIN: raised right after the check that precedes optimize() (the -k form with k = 0)
  bool terminate() const {
    if (!stop) return false;
    const int v = *stop;
    if (v == INT32_MIN) return past_precheck;
    return v > 0 || (v < 0 && trials_done >= -v);
  }

  void compute_errors() {
    for (int k = 0; k < p->n_edges; k++) {
      const lba_edge& e = p->edges[k];
      double Xc[3];
      edge_error(poses[e.pose], &points[3 * e.point], cam, rig, e, &err[3 * k], Xc);
      const double om = (double)e.inv_sigma2;
      const int D = edge_dim(e);
      double c = 0;
      for (int i = 0; i < D; i++) c += err[3 * k + i] * (om * err[3 * k + i]);
      chi2[k] = c;
    }
  }
  double robust_chi2() const {
    double chi = 0;
    for (int k = 0; k < p->n_edges; k++) {
      double rho[2];
      const bool mono = p->edges[k].ur < 0;
      huber(chi2[k], mono ? delta_mono : delta_stereo, mono ? dsqr_mono : dsqr_stereo, rho);
      chi += rho[0];
    }
    return chi;
  }
  void build_system() {
    std::fill(Hpp.begin(), Hpp.end(), 0.0); std::fill(Hll.begin(), Hll.end(), 0.0);
    std::fill(Hpl.begin(), Hpl.end(), 0.0); std::fill(bp.begin(), bp.end(), 0.0); std::fill(bl.begin(), bl.end(), 0.0);
    for (int k = 0; k < p->n_edges; k++) {
      const lba_edge& e = p->edges[k];
      const int pc = pose_col[e.pose], lc = point_col[e.point];
      if (pc < 0 && lc < 0) continue;
      double Xc[3], r[3];
      quat_rotate(poses[e.pose].q, &points[3 * e.point], r);
      for (int i = 0; i < 3; i++) Xc[i] = r[i] + poses[e.pose].t[i];
      double A[9], B[18];
      edge_jacobians(poses[e.pose], Xc, cam, rig, e, A, B);
      const int D = edge_dim(e);
      const bool mono = D == 2;
      double rho[2];
      huber(chi2[k], mono ? delta_mono : delta_stereo, mono ? dsqr_mono : dsqr_stereo, rho);
      const double om = (double)e.inv_sigma2;
      const double wom = rho[1] * om;
      double omega_r[3];
      for (int i = 0; i < D; i++) omega_r[i] = -(om * err[3 * k + i]) * rho[1];
      if (lc >= 0) {
        for (int a = 0; a < 3; a++) {
          double s = 0;
          for (int i = 0; i < D; i++) s += A[3 * i + a] * omega_r[i];
          bl[3 * lc + a] += s;
          for (int c = 0; c < 3; c++) {
            double h = 0;
            for (int i = 0; i < D; i++) h += A[3 * i + a] * wom * A[3 * i + c];
            Hll[9 * lc + 3 * a + c] += h;
          }
        }
      }
      if (pc >= 0) {
        for (int a = 0; a < 6; a++) {
          double s = 0;
          for (int i = 0; i < D; i++) s += B[6 * i + a] * omega_r[i];
          bp[6 * pc + a] += s;
          for (int c = 0; c < 6; c++) {
            double h = 0;
            for (int i = 0; i < D; i++) h += B[6 * i + a] * wom * B[6 * i + c];
            Hpp[36 * pc + 6 * a + c] += h;
          }
        }
        if (lc >= 0)
          for (int a = 0; a < 6; a++)
            for (int c = 0; c < 3; c++) {
              double h = 0;
              for (int i = 0; i < D; i++) h += B[6 * i + a] * wom * A[3 * i + c];
              Hpl[18 * (size_t)hpl_slot[k] + 3 * a + c] += h;
            }
      }
    }
  }
  double lambda_init() const {   // computeLambdaInit, levenberg.cpp:171-185
    if (p->lambda_init > 0) return p->lambda_init;
    double mx = 0;
    for (int i = 0; i < nP; i++)
      for (int j = 0; j < 6; j++) mx = std::max(std::fabs(Hpp[36 * i + 7 * j]), mx);
    for (int i = 0; i < nL; i++)
      for (int j = 0; j < 3; j++) mx = std::max(std::fabs(Hll[9 * i + 4 * j]), mx);
    return 1e-5 * mx;
  }
  // BlockSolver::solve with lambda added to every diagonal (setLambda/restoreDiagonal are folded in)
  bool solve(double lambda, const std::vector<std::vector<int>>& edges_of_point) {
    const int n = 6 * nP;
    std::vector<double> S((size_t)n * n, 0.0), coeff(n, 0.0), Dinv(9 * (size_t)nL), db(3 * (size_t)nL);
    for (int i = 0; i < nP; i++)
      for (int a = 0; a < 6; a++)
        for (int c = 0; c < 6; c++) S[(size_t)(6 * i + a) * n + 6 * i + c] = Hpp[36 * i + 6 * a + c] + (a == c ? lambda : 0.0);
    for (int l = 0; l < nL; l++) {
      double Dm[9];
      for (int i = 0; i < 9; i++) Dm[i] = Hll[9 * l + i] + (i % 4 == 0 ? lambda : 0.0);
      inv3(Dm, &Dinv[9 * l]);
      for (int a = 0; a < 3; a++)
        db[3 * l + a] = Dinv[9 * l + 3 * a] * bl[3 * l] + Dinv[9 * l + 3 * a + 1] * bl[3 * l + 1] + Dinv[9 * l + 3 * a + 2] * bl[3 * l + 2];
      const std::vector<int>& ed = edges_of_point[l];   // edges with a free pose, sorted by pose column
      for (size_t o = 0; o < ed.size(); o++) {
        const int k1 = ed[o], i1 = pose_col[p->edges[k1].pose];
        const double* Bi = &Hpl[18 * k1];
        double BDinv[18];
        for (int a = 0; a < 6; a++)
          for (int c = 0; c < 3; c++)
            BDinv[3 * a + c] = Bi[3 * a] * Dinv[9 * l + c] + Bi[3 * a + 1] * Dinv[9 * l + 3 + c] + Bi[3 * a + 2] * Dinv[9 * l + 6 + c];
        for (int a = 0; a < 6; a++)
          coeff[6 * i1 + a] += Bi[3 * a] * db[3 * l] + Bi[3 * a + 1] * db[3 * l + 1] + Bi[3 * a + 2] * db[3 * l + 2];
        for (size_t q = o; q < ed.size(); q++) {          // upper-triangular block pairs only
          const int k2 = ed[q], i2 = pose_col[p->edges[k2].pose];
          const double* Bj = &Hpl[18 * k2];
          for (int a = 0; a < 6; a++)
            for (int c = 0; c < 6; c++)
              S[(size_t)(6 * i1 + a) * n + 6 * i2 + c] -=
                  BDinv[3 * a] * Bj[3 * c] + BDinv[3 * a + 1] * Bj[3 * c + 1] + BDinv[3 * a + 2] * Bj[3 * c + 2];
        }
      }
    }
    std::vector<double> bs(n);
    for (int i = 0; i < n; i++) bs[i] = bp[i] - coeff[i];
    if (n > 0 && !ldlt_solve(S, n, bs.data(), x.data())) return false;
    // landmarks: xl = Dinv * (bl - Hpl^T xp)
    for (int l = 0; l < nL; l++) {
      double cl[3] = {bl[3 * l], bl[3 * l + 1], bl[3 * l + 2]};
      for (int k : edges_of_point[l]) {
        const int i1 = pose_col[p->edges[k].pose];
        const double* Bi = &Hpl[18 * k];
        for (int c = 0; c < 3; c++)
          for (int a = 0; a < 6; a++) cl[c] -= Bi[3 * a + c] * x[6 * i1 + a];
      }
      for (int a = 0; a < 3; a++)
        x[n + 3 * l + a] = Dinv[9 * l + 3 * a] * cl[0] + Dinv[9 * l + 3 * a + 1] * cl[1] + Dinv[9 * l + 3 * a + 2] * cl[2];
    }
    return true;
  }
};

}  // namespace

extern "C" void oracle_se3_exp(const double* upd6, double* q4, double* t3) {
  PoseQ p = se3_exp(upd6);
  q4[0] = p.q.x; q4[1] = p.q.y; q4[2] = p.q.z; q4[3] = p.q.w;
  t3[0] = p.t[0]; t3[1] = p.t[1]; t3[2] = p.t[2];
}

extern "C" void oracle_lba_edge_eval(const double* q4, const double* t3, const double* X3, const float* cam5,
                                     const lba_edge* e, double* err3, double* Jpoint9, double* Jpose18) {
  PoseQ T;
  T.q = {q4[0], q4[1], q4[2], q4[3]};
  T.t[0] = t3[0]; T.t[1] = t3[1]; T.t[2] = t3[2];
  Cam c{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4], cam5[4]};
  double Xc[3];
  const Rig none;
  edge_error(T, X3, c, none, *e, err3, Xc);
  edge_jacobians(T, Xc, c, none, *e, Jpoint9, Jpose18);
}

// the same with the cameras of a rig (monocular / right-camera edges through GeometricCamera::project / projectJac)
extern "C" void oracle_lba_edge_eval_rig(const double* q4, const double* t3, const double* X3, const float* cam5, const orbg_camera_rig* rig,
                                         const lba_edge* e, double* err3, double* Jpoint9, double* Jpose18, double* Xcam3) {
  PoseQ T;
  T.q = {q4[0], q4[1], q4[2], q4[3]};
  T.t[0] = t3[0]; T.t[1] = t3[1]; T.t[2] = t3[2];
  Cam c{cam5[0], cam5[1], cam5[2], cam5[3], cam5[4], cam5[4]};
  const Rig g = rig_of(rig);
  double Xobs[3], r[3], Xc[3];
  edge_error(T, X3, c, g, *e, err3, Xobs);
  quat_rotate(T.q, X3, r);
  for (int i = 0; i < 3; i++) Xc[i] = r[i] + T.t[i];
  edge_jacobians(T, Xc, c, g, *e, Jpoint9, Jpose18);
  if (Xcam3) for (int i = 0; i < 3; i++) Xcam3[i] = Xobs[i];
}

// RobustKernelHuber::robustify's rho and rho' for the known-answer tests
extern "C" void oracle_robust_huber(double e, double delta, double* rho2) {
  huber(e, delta, delta * delta, rho2);
}

extern "C" void oracle_camera_project(const orbg_camera* cam, const double* X3, double* uv2, double* J6) {
  const CamModel m = cam_model_of(*cam);
  if (uv2) cam_project(m, X3, uv2);
  if (J6) cam_project_jac(m, X3, J6);
}

extern "C" int oracle_lba_solve(const lba_problem* p, const volatile int32_t* stop_flag, lba_result* r) {
  if (!p || !r || p->n_poses < 0 || p->n_points < 0 || p->n_edges < 0) return ORBG_BAD_ARG;
  Lba s;
  s.p = p;
  s.stop = stop_flag;
  s.cam = Cam{p->fx, p->fy, p->cx, p->cy, p->bf, p->bf};
  s.rig = rig_of(p->rig);
  if (s.rig.on) {
    if ((s.rig.left.model != ORBG_CAM_PINHOLE && s.rig.left.model != ORBG_CAM_KANNALA_BRANDT8) ||
        (s.rig.has_right && s.rig.right.model != ORBG_CAM_PINHOLE && s.rig.right.model != ORBG_CAM_KANNALA_BRANDT8))
      return ORBG_BAD_ARG;
  }
  const float thHuberMono = (float)std::sqrt(5.991), thHuberStereo = (float)std::sqrt(7.815);   // S/Optimizer.cc:1991-1992
  s.delta_mono = thHuberMono; s.dsqr_mono = s.delta_mono * s.delta_mono;
  s.delta_stereo = thHuberStereo; s.dsqr_stereo = s.delta_stereo * s.delta_stereo;
  // vertices: Converter::toSE3Quat (S/Converter.cc:33-43) / toVector3d
  s.poses.resize(p->n_poses);
  for (int i = 0; i < p->n_poses; i++) {
    const float* T = p->poses + 16 * (size_t)i;
    double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    s.poses[i].q = quat_from_R(R);
    normalize_rotation(s.poses[i].q);
    s.poses[i].t[0] = T[3]; s.poses[i].t[1] = T[7]; s.poses[i].t[2] = T[11];
  }
  s.points.resize(3 * (size_t)p->n_points);
  for (int i = 0; i < 3 * p->n_points; i++) s.points[i] = p->points[i];
  for (int k = 0; k < p->n_edges; k++)
    if (p->edges[k].pose < 0 || p->edges[k].pose >= p->n_poses || p->edges[k].point < 0 || p->edges[k].point >= p->n_points)
      return ORBG_BAD_ARG;
  // active set + index mapping (G/core/sparse_optimizer.cpp:166-190,206-267): free vertices with >= 1 edge
  std::vector<int> pose_deg(p->n_poses, 0), point_deg(p->n_points, 0);
  for (int k = 0; k < p->n_edges; k++) { pose_deg[p->edges[k].pose]++; point_deg[p->edges[k].point]++; }
  s.pose_col.assign(p->n_poses, -1); s.point_col.assign(p->n_points, -1);
  for (int i = 0; i < p->n_poses; i++)
    if (!p->pose_fixed[i] && pose_deg[i] > 0) { s.pose_col[i] = s.nP++; s.active_pose.push_back(i); }
  for (int i = 0; i < p->n_points; i++)
    if (point_deg[i] > 0) { s.point_col[i] = s.nL++; s.active_point.push_back(i); }
  s.hpl_slot.resize((size_t)p->n_edges);
  {
    std::map<std::pair<int, int>, int> first_edge;
    for (int k = 0; k < p->n_edges; k++)
      s.hpl_slot[k] = first_edge.emplace(std::make_pair(p->edges[k].pose, p->edges[k].point), k).first->second;
  }
  std::vector<std::vector<int>> edges_of_point(s.nL);      // per landmark: its Hpl blocks (one per observing free pose)
  for (int k = 0; k < p->n_edges; k++) {
    const int lc = s.point_col[p->edges[k].point];
    if (lc >= 0 && s.pose_col[p->edges[k].pose] >= 0 && s.hpl_slot[k] == k) edges_of_point[lc].push_back(k);
  }
  for (auto& v : edges_of_point)
    std::stable_sort(v.begin(), v.end(), [&](int a, int b) { return s.pose_col[p->edges[a].pose] < s.pose_col[p->edges[b].pose]; });
  s.err.assign(3 * (size_t)p->n_edges, 0.0); s.chi2.assign(p->n_edges, 0.0);
  s.Hpp.assign(36 * (size_t)s.nP, 0.0); s.Hll.assign(9 * (size_t)s.nL, 0.0); s.Hpl.assign(18 * (size_t)p->n_edges, 0.0);
  s.bp.assign(6 * (size_t)s.nP, 0.0); s.bl.assign(3 * (size_t)s.nL, 0.0);
  s.x.assign(6 * (size_t)s.nP + 3 * (size_t)s.nL, 0.0);

  r->status = LBA_APPLIED; r->iters_round1 = r->iters_round2 = 0; r->n_outliers = 0; r->trace_len = 0;
  r->chi2_initial = r->chi2_final = 0;
  auto write_back = [&]() {
    for (int i = 0; i < p->n_poses; i++) {     // Converter::toCvMat(SE3Quat) S/Converter.cc:45-64
      double R[9];
      quat_to_R(s.poses[i].q, R);
      float* T = r->poses + 16 * (size_t)i;
      for (int a = 0; a < 3; a++) { for (int c = 0; c < 3; c++) T[4 * a + c] = (float)R[3 * a + c]; T[4 * a + 3] = (float)s.poses[i].t[a]; }
      T[12] = 0; T[13] = 0; T[14] = 0; T[15] = 1;
    }
    for (int i = 0; i < 3 * p->n_points; i++) r->points[i] = (float)s.points[i];
  };
  if (s.terminate()) {                          // S/Optimizer.cc:2127-2129
    r->status = LBA_ABORTED_BEFORE_OPT;
    std::memcpy(r->poses, p->poses, sizeof(float) * 16 * (size_t)p->n_poses);
    std::memcpy(r->points, p->points, sizeof(float) * 3 * (size_t)p->n_points);
    for (int k = 0; k < p->n_edges; k++) { if (r->edge_chi2) r->edge_chi2[k] = 0; if (r->edge_depth_pos) r->edge_depth_pos[k] = 1; if (r->edge_outlier) r->edge_outlier[k] = 0; }
    return ORBG_OK;
  }
  s.past_precheck = true;

  double lambda = -1, ni = 2;
  int nBad = 0;
  bool first_chi = true;
  // SparseOptimizer::optimize(iterations) + OptimizationAlgorithmLevenberg::solve
  auto optimize = [&](int iterations) -> int {
    int done = 0;
    bool ok = true;
    for (int it = 0; it < iterations && !s.terminate() && ok; it++) {
      s.compute_errors();
      double currentChi = s.robust_chi2();
      if (first_chi) { r->chi2_initial = currentChi; first_chi = false; }
      double tempChi = currentChi;
      const double iniChi = currentChi;
      s.build_system();
      if (it == 0) { lambda = s.lambda_init(); ni = 2; nBad = 0; }
      double rho = 0;
      int qmax = 0;
      do {
        std::vector<PoseQ> backup_poses = s.poses;           // push()
        std::vector<double> backup_points = s.points;
        const bool ok2 = s.solve(lambda, edges_of_point);
        // update(x): oplus on every active vertex (with whatever x holds, as g2o does)
        for (int i = 0; i < s.nP; i++) pose_oplus(s.poses[s.active_pose[i]], &s.x[6 * i]);
        for (int l = 0; l < s.nL; l++)
          for (int a = 0; a < 3; a++) s.points[3 * s.active_point[l] + a] += s.x[6 * s.nP + 3 * l + a];
        s.compute_errors();
        tempChi = s.robust_chi2();
        if (!ok2) tempChi = std::numeric_limits<double>::max();
        rho = currentChi - tempChi;
        double scale = 0;                                     // computeScale, levenberg.cpp:187-194
        for (int j = 0; j < 6 * s.nP; j++) scale += s.x[j] * (lambda * s.x[j] + s.bp[j]);
        for (int j = 0; j < 3 * s.nL; j++) scale += s.x[6 * s.nP + j] * (lambda * s.x[6 * s.nP + j] + s.bl[j]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && std::isfinite(tempChi)) {
          double alpha = 1. - std::pow((2 * rho - 1), 3);
          alpha = std::min(alpha, 2. / 3.);
          const double scaleFactor = std::max(1. / 3., alpha);
          lambda *= scaleFactor;
          ni = 2;
          currentChi = tempChi;
        } else {
          lambda *= ni;
          ni *= 2;
          s.poses = backup_poses;                             // pop()
          s.points = backup_points;
        }
        qmax++;
        s.trials_done++;
      } while (rho < 0 && qmax < 10 && !s.terminate());
      done++;
      r->chi2_final = currentChi;
      if (r->trace && r->trace_len < r->trace_cap) {
        r->trace[3 * r->trace_len] = lambda; r->trace[3 * r->trace_len + 1] = currentChi; r->trace[3 * r->trace_len + 2] = qmax;
        r->trace_len++;
      }
      if (qmax == 10 || rho == 0) { ok = false; continue; }   // Terminate
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      if (nBad >= 3) ok = false;
    }
    return done;
  };

  r->iters_round1 = optimize(p->its_round1 > 0 ? p->its_round1 : 5);    // S/Optimizer.cc:2130-2132
  bool bDoMore = !s.terminate();                                          // :2135-2139
  if (bDoMore) r->iters_round2 = optimize(p->its_round2 > 0 ? p->its_round2 : 10);   // :2202-2203 (outliers kept, Appendix C-6)

  // :2207-2261 -- e->chi2() uses the errors of the LAST computeActiveErrors (possibly a rejected trial);
  // isDepthPositive() uses the current estimates.  If the flag was raised before the first iteration no residual was ever
  // evaluated: g2o's BaseEdge leaves _error uninitialised (G/core/base_edge.h:50-58) -- pinned here as zero (s.chi2 starts zeroed).
  int n_out = 0;
  for (int k = 0; k < p->n_edges; k++) {
    const lba_edge& e = p->edges[k];
    double r3[3], Xc[3];
    if (s.rig.has_right && edge_is_right(e.ur)) {              // ((mTrl * T).map(X))(2) > 0, I/OptimizableTypes.h:134-138
      const PoseQ Trw = se3_mul(s.rig.Trl, s.poses[e.pose]);
      quat_rotate(Trw.q, &s.points[3 * e.point], r3);
      for (int i = 0; i < 3; i++) Xc[i] = r3[i] + Trw.t[i];
    } else {
      quat_rotate(s.poses[e.pose].q, &s.points[3 * e.point], r3);
      for (int i = 0; i < 3; i++) Xc[i] = r3[i] + s.poses[e.pose].t[i];
    }
    const bool depth_pos = Xc[2] > 0.0;
    const double thr = e.ur < 0 ? 5.991 : 7.815;
    const bool outlier = s.chi2[k] > thr || !depth_pos;
    if (r->edge_chi2) r->edge_chi2[k] = s.chi2[k];
    if (r->edge_depth_pos) r->edge_depth_pos[k] = depth_pos;
    if (r->edge_outlier) r->edge_outlier[k] = outlier;
    n_out += outlier;
  }
  r->n_outliers = n_out;
  // :2256-2261 -- vToErase.size() >= (vpMapPointEdgeMono.size() + vpMapPointEdgeStereo.size()) * 0.5: the right camera's edges are in
  // vToErase but not in the sum (a window of right-camera edges alone is always refused)
  int n_not_right = 0;
  for (int k = 0; k < p->n_edges; k++) n_not_right += !(s.rig.has_right && edge_is_right(p->edges[k].ur));
  if (n_out >= n_not_right * 0.5 && p->n_edges > 0) r->status = LBA_REJECTED_OUTLIERS;
  write_back();
  return ORBG_OK;
}


// ------------------------------------------------------------------------------------------------
// Optimizer::PoseOptimization(Frame*) -- S/Optimizer.cc:964-1278: one SE3 vertex, unary edges
// EdgeSE3ProjectXYZOnlyPose (I/OptimizableTypes.h:31-57, S/OptimizableTypes.cpp:49-63),
// g2o::EdgeStereoSE3ProjectXYZOnlyPose (G/types/types_six_dof_expmap.{h:208-236,cpp:339-346,375-404}) and -- with the
// two-camera rig, :1085-1151 -- EdgeSE3ProjectXYZOnlyPoseToBody (I/OptimizableTypes.h:59-87, S/OptimizableTypes.cpp:90-108),
// BlockSolver_6_3 over LinearSolverDense (Eigen::LDLT, must be positive: G/solvers/linear_solver_dense.h:105-110),
// Levenberg-Marquardt as in oracle_lba_solve.
namespace {

struct PoEdge { double X[3]; double u, v, ur; double om; bool mono; bool right; };

// error of a pose-only edge; stereo: invz is float32, bf*invz is a DOUBLE product here (bf is a double member,
// unlike the binary stereo edge whose cam_project takes `const float &bf`)
void po_error(const PoseQ& T, const PoEdge& e, const Cam& c, const Rig& rig, double err[3], double Xc[3]) {
  double r[3];
  if (e.right) {      // EdgeSE3ProjectXYZOnlyPoseToBody::computeError, I/OptimizableTypes.h:69-73
    const PoseQ Trw = se3_mul(rig.Trl, T);
    quat_rotate(Trw.q, e.X, r);
    for (int i = 0; i < 3; i++) Xc[i] = r[i] + Trw.t[i];
    double uv[2];
    cam_project(rig.right, Xc, uv);
    err[0] = e.u - uv[0]; err[1] = e.v - uv[1]; err[2] = 0;
    return;
  }
  quat_rotate(T.q, e.X, r);
  for (int i = 0; i < 3; i++) Xc[i] = r[i] + T.t[i];
  if (e.mono && rig.on) {     // EdgeSE3ProjectXYZOnlyPose::computeError through mpCamera, I/OptimizableTypes.h:40-44
    double uv[2];
    cam_project(rig.left, Xc, uv);
    err[0] = e.u - uv[0]; err[1] = e.v - uv[1]; err[2] = 0;
    return;
  }
  if (e.mono) {
    err[0] = e.u - (c.fx * Xc[0] / Xc[2] + c.cx);
    err[1] = e.v - (c.fy * Xc[1] / Xc[2] + c.cy);
    err[2] = 0;
  } else {
    const float invz = (float)(1.0f / Xc[2]);
    const double r0 = Xc[0] * invz * c.fx + c.cx;
    const double r1 = Xc[1] * invz * c.fy + c.cy;
    err[0] = e.u - r0; err[1] = e.v - r1; err[2] = e.ur - (r0 - c.bf * invz);
  }
}

// Xc = T.map(Xw): the point in the left camera's frame for every kind of edge
void po_jacobian(const double Xc[3], const PoEdge& e, const Cam& c, const Rig& rig, double J[18]) {
  const double x = Xc[0], y = Xc[1], z = Xc[2];
  if (rig.on && e.mono) {
    // S/OptimizableTypes.cpp:46-62 (-projectJac(xyz_trans) * SE3deriv) and :90-108 (-projectJac(X_r) * mTrl.rotation() * SE3deriv(X_l))
    const double S[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
    double P[6], M[6];
    if (e.right) {
      double rr[3], Xr[3], Rrl[9];
      quat_rotate(rig.Trl.q, Xc, rr);
      for (int i = 0; i < 3; i++) Xr[i] = rr[i] + rig.Trl.t[i];
      cam_project_jac(rig.right, Xr, P);
      for (int i = 0; i < 6; i++) P[i] = -P[i];
      quat_to_R(rig.Trl.q, Rrl);
      for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) M[3 * i + j] = P[3 * i] * Rrl[j] + P[3 * i + 1] * Rrl[3 + j] + P[3 * i + 2] * Rrl[6 + j];
    } else {
      cam_project_jac(rig.left, Xc, P);
      for (int i = 0; i < 6; i++) M[i] = -P[i];
    }
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 6; j++) J[6 * i + j] = M[3 * i] * S[j] + M[3 * i + 1] * S[6 + j] + M[3 * i + 2] * S[12 + j];
    for (int j = 0; j < 6; j++) J[12 + j] = 0;
    return;
  }
  if (e.mono) {
    const double P[6] = {-(c.fx / z), -0.0, -(-c.fx * x / (z * z)), -0.0, -(c.fy / z), -(-c.fy * y / (z * z))};
    const double S[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 6; j++) J[6 * i + j] = P[3 * i] * S[j] + P[3 * i + 1] * S[6 + j] + P[3 * i + 2] * S[12 + j];
    for (int j = 0; j < 6; j++) J[12 + j] = 0;
  } else {
    const double invz = 1.0 / z, invz_2 = invz * invz;
    J[0] = x * y * invz_2 * c.fx; J[1] = -(1 + (x * x * invz_2)) * c.fx; J[2] = y * invz * c.fx; J[3] = -invz * c.fx; J[4] = 0; J[5] = x * invz_2 * c.fx;
    J[6] = (1 + y * y * invz_2) * c.fy; J[7] = -x * y * invz_2 * c.fy; J[8] = -x * invz * c.fy; J[9] = 0; J[10] = -invz * c.fy; J[11] = y * invz_2 * c.fy;
    J[12] = J[0] - c.bf * y * invz_2; J[13] = J[1] + c.bf * x * invz_2; J[14] = J[2]; J[15] = J[3]; J[16] = 0; J[17] = J[5] - c.bf * invz_2;
  }
}

// 6x6 LDL^T solve; false unless every pivot is positive (Eigen::LDLT::isPositive)
bool solve6_pd(const double H[36], const double b[6], double x[6]) {
  double L[36] = {0}, D[6];
  for (int j = 0; j < 6; j++) {
    double d = H[7 * j];
    for (int k = 0; k < j; k++) d -= L[6 * j + k] * L[6 * j + k] * D[k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    D[j] = d;
    for (int i = j + 1; i < 6; i++) {
      double s = H[6 * i + j];
      for (int k = 0; k < j; k++) s -= L[6 * i + k] * L[6 * j + k] * D[k];
      L[6 * i + j] = s / d;
    }
  }
  double y[6];
  for (int i = 0; i < 6; i++) { y[i] = b[i]; for (int k = 0; k < i; k++) y[i] -= L[6 * i + k] * y[k]; }
  for (int i = 0; i < 6; i++) y[i] /= D[i];
  for (int i = 5; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < 6; k++) s -= L[6 * k + i] * x[k]; x[i] = s; }
  return true;
}

}  // namespace

extern "C" int oracle_pose_optimize(const pose_opt_problem* p, pose_opt_result* r) {
  if (!p || !r || p->n < 0 || (p->n > 0 && (!p->Xw || !p->u || !p->v || !p->ur || !p->inv_sigma2 || !r->outlier))) return ORBG_BAD_ARG;
  const int n = p->n;
  std::memcpy(r->Tcw, p->Tcw, sizeof(float) * 16);
  r->n_inliers = 0; r->n_bad = 0;
  for (int i = 0; i < 4; i++) { r->iters[i] = 0; r->chi2[i] = 0; }
  for (int i = 0; i < n; i++) r->outlier[i] = 0;                              // mvbOutlier[i] = false   (:1015,1046)
  if (n < 3) return ORBG_OK;                                                   // :1180-1181
  Cam cam{p->fx, p->fy, p->cx, p->cy, p->bf, p->bf};
  const Rig rig = rig_of(p->rig);
  if (rig.on && ((rig.left.model != ORBG_CAM_PINHOLE && rig.left.model != ORBG_CAM_KANNALA_BRANDT8) ||
                 (rig.has_right && rig.right.model != ORBG_CAM_PINHOLE && rig.right.model != ORBG_CAM_KANNALA_BRANDT8)))
    return ORBG_BAD_ARG;
  const double deltaMono = (float)std::sqrt(5.991), deltaStereo = (float)std::sqrt(7.815);   // :1001-1002
  std::vector<PoEdge> E(n);
  for (int i = 0; i < n; i++) {
    E[i].X[0] = p->Xw[3 * i]; E[i].X[1] = p->Xw[3 * i + 1]; E[i].X[2] = p->Xw[3 * i + 2];
    E[i].u = p->u[i]; E[i].v = p->v[i]; E[i].ur = p->ur[i]; E[i].om = p->inv_sigma2[i]; E[i].mono = p->ur[i] < 0;
    E[i].right = rig.has_right && edge_is_right(p->ur[i]);      // (without a second camera any negative ur is a monocular entry, as ever)
  }
  PoseQ T0;
  {
    const float* T = p->Tcw;
    double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    T0.q = quat_from_R(R);
    normalize_rotation(T0.q);
    T0.t[0] = T[3]; T0.t[1] = T[7]; T0.t[2] = T[11];
  }
  std::vector<double> err(3 * (size_t)n, 0.0), chi2(n, 0.0);
  std::vector<uint8_t> level(n, 0);          // e->level(): 1 = excluded from the optimisation
  PoseQ T = T0;
  bool robust = true;
  const float chi2Mono = 5.991f, chi2Stereo = 7.815f;                          // :1183-1184 (same value every round)
  int nBad = 0;
  auto compute_errors = [&](const PoseQ& Tc) {                                  // active edges only
    for (int i = 0; i < n; i++) {
      if (level[i]) continue;
      double Xc[3];
      po_error(Tc, E[i], cam, rig, &err[3 * i], Xc);
      double c = 0;
      const int D = E[i].mono ? 2 : 3;
      for (int k = 0; k < D; k++) c += err[3 * i + k] * (E[i].om * err[3 * i + k]);
      chi2[i] = c;
    }
  };
  auto robust_chi2 = [&]() {
    double s = 0;
    for (int i = 0; i < n; i++) {
      if (level[i]) continue;
      if (robust) { double rho[2]; const double d = E[i].mono ? deltaMono : deltaStereo; huber(chi2[i], d, d * d, rho); s += rho[0]; }
      else s += chi2[i];
    }
    return s;
  };
  for (int it4 = 0; it4 < 4; it4++) {
    T = T0;                                                                     // vSE3->setEstimate(toSE3Quat(mTcw)) :1191
    double lambda = 0, ni = 2;
    int nBadLM = 0, done = 0;
    bool ok = true;
    int n_active = 0;
    for (int i = 0; i < n; i++) n_active += !level[i];
    double currentChi = 0;
    for (int it = 0; it < 10 && ok && n_active > 0; it++) {
      compute_errors(T);
      currentChi = robust_chi2();
      double tempChi = currentChi;
      const double iniChi = currentChi;
      double H[36] = {0}, b[6] = {0};
      for (int i = 0; i < n; i++) {
        if (level[i]) continue;
        double Xc[3], r3[3];
        quat_rotate(T.q, E[i].X, r3);
        for (int k = 0; k < 3; k++) Xc[k] = r3[k] + T.t[k];
        double J[18];
        po_jacobian(Xc, E[i], cam, rig, J);
        const int D = E[i].mono ? 2 : 3;
        double w = 1.0;
        if (robust) { double rho[2]; const double d = E[i].mono ? deltaMono : deltaStereo; huber(chi2[i], d, d * d, rho); w = rho[1]; }
        for (int a = 0; a < 6; a++) {
          double s = 0;
          for (int k = 0; k < D; k++) s += J[6 * k + a] * (-(E[i].om * err[3 * i + k]) * w);
          b[a] += s;
          for (int c = 0; c < 6; c++) {
            double h = 0;
            for (int k = 0; k < D; k++) h += J[6 * k + a] * (w * E[i].om) * J[6 * k + c];
            H[6 * a + c] += h;
          }
        }
      }
      if (it == 0) {
        double mx = 0;
        for (int j = 0; j < 6; j++) mx = std::max(std::fabs(H[7 * j]), mx);
        lambda = 1e-5 * mx; ni = 2; nBadLM = 0;
      }
      double rho = 0;
      int qmax = 0;
      double x[6] = {0, 0, 0, 0, 0, 0};
      do {
        const PoseQ backup = T;
        double Hl[36];
        for (int k = 0; k < 36; k++) Hl[k] = H[k] + (k % 7 == 0 ? lambda : 0.0);
        const bool ok2 = solve6_pd(Hl, b, x);
        pose_oplus(T, x);
        compute_errors(T);
        tempChi = robust_chi2();
        if (!ok2) tempChi = std::numeric_limits<double>::max();
        rho = currentChi - tempChi;
        double scale = 0;
        for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && std::isfinite(tempChi)) {
          double alpha = 1. - std::pow((2 * rho - 1), 3);
          alpha = std::min(alpha, 2. / 3.);
          lambda *= std::max(1. / 3., alpha);
          ni = 2;
          currentChi = tempChi;
        } else {
          lambda *= ni; ni *= 2;
          T = backup;
        }
        qmax++;
      } while (rho < 0 && qmax < 10);
      done++;
      if (qmax == 10 || rho == 0) { ok = false; continue; }
      if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
      if (nBadLM >= 3) ok = false;
    }
    r->iters[it4] = done;
    r->chi2[it4] = currentChi;
    // :1196-1270 -- excluded edges get a fresh error, active ones keep the LAST evaluated one
    nBad = 0;
    for (int i = 0; i < n; i++) {
      if (r->outlier[i]) {
        double Xc[3];
        po_error(T, E[i], cam, rig, &err[3 * i], Xc);
        double c = 0;
        const int D = E[i].mono ? 2 : 3;
        for (int k = 0; k < D; k++) c += err[3 * i + k] * (E[i].om * err[3 * i + k]);
        chi2[i] = c;
      }
      const float c2 = (float)chi2[i];
      if (c2 > (E[i].mono ? chi2Mono : chi2Stereo)) { r->outlier[i] = 1; level[i] = 1; nBad++; }
      else { r->outlier[i] = 0; level[i] = 0; }
    }
    if (it4 == 2) robust = false;                                              // e->setRobustKernel(0) :1219,1245,1268
    if (n < 10) break;                                                          // optimizer.edges().size() < 10 :1272
  }
  {
    double R[9];
    quat_to_R(T.q, R);
    for (int a = 0; a < 3; a++) { for (int c = 0; c < 3; c++) r->Tcw[4 * a + c] = (float)R[3 * a + c]; r->Tcw[4 * a + 3] = (float)T.t[a]; }
    r->Tcw[12] = 0; r->Tcw[13] = 0; r->Tcw[14] = 0; r->Tcw[15] = 1;
  }
  r->n_bad = nBad;
  r->n_inliers = n - nBad;
  return ORBG_OK;
}
