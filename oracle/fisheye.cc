// ORACLE (test infrastructure, not product code) -- see orb_oracle.h for status: parity unpinned.
//
// CPU restatement of Frame::ComputeStereoFishEyeMatches (S/Frame.cc:1093-1150) with KannalaBrandt8::TriangulateMatches / Triangulate /
// unproject / project (S/CameraModels/KannalaBrandt8.cpp:28-44,103-133,335-420) over the flattened view of include/orbgpu.h.
//
// Two things of it are OpenCV's and not in the tree: cv::BFMatcher::knnMatch (restated as: the two smallest Hamming distances over
// ALL train descriptors, the smaller index first among equal distances) and cv::SVD::compute on the 4 x 4 system of Triangulate
// (restated as: the eigenvector of A^T A with the smallest eigenvalue, by cyclic Jacobi rotations in float64 -- OpenCV's is a one-sided
// Jacobi in float32: the homogeneous point agrees to float32 rounding, not to the bit).  cv::Mat conventions as in matching.cc.

#include "orb_oracle.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

// KannalaBrandt8::project(cv::Point3f) -- :28-44 (float32; cos / sin of a float: the float overloads)
inline void kb8_project_f(const orbg_camera& c, const float* p, float* uv) {
  const float x2_plus_y2 = p[0] * p[0] + p[1] * p[1];
  const float theta = atan2f(sqrtf(x2_plus_y2), p[2]);
  const float psi = atan2f(p[1], p[0]);
  const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
  const float r = theta + c.k[0] * theta3 + c.k[1] * theta5 + c.k[2] * theta7 + c.k[3] * theta9;
  uv[0] = c.fx * r * cosf(psi) + c.cx;
  uv[1] = c.fy * r * sinf(psi) + c.cy;
}

// KannalaBrandt8::unproject -- :103-133 (Newton on theta, at most ten steps, precision = 1e-6f)
inline void kb8_unproject_f(const orbg_camera& c, float u, float v, float* ray) {
  const float pwx = (u - c.cx) / c.fx, pwy = (v - c.cy) / c.fy;
  float scale = 1.f;
  float theta_d = sqrtf(pwx * pwx + pwy * pwy);
  theta_d = fminf(fmaxf((float)(-3.1415926535897932384626433832795 / 2.f), theta_d), (float)(3.1415926535897932384626433832795 / 2.f));
  if (theta_d > 1e-8) {
    float theta = theta_d;
    const float precision = 1e-6f;
    for (int j = 0; j < 10; j++) {
      const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
      const float k0_theta2 = c.k[0] * theta2, k1_theta4 = c.k[1] * theta4, k2_theta6 = c.k[2] * theta6, k3_theta8 = c.k[3] * theta8;
      const float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                              (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
      theta = theta - theta_fix;
      if (fabsf(theta_fix) < precision) break;
    }
    scale = tanf(theta) / theta_d;
  }
  ray[0] = pwx * scale; ray[1] = pwy * scale; ray[2] = 1.f;
}

// eigenvector of the symmetric 4 x 4 matrix S with the smallest eigenvalue: cyclic Jacobi, float64
void smallest_eigenvector4(double S[4][4], double* v) {
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0, diag = 0;
    for (int p = 0; p < 4; p++) { diag += S[p][p] * S[p][p]; for (int q = p + 1; q < 4; q++) off += S[p][q] * S[p][q]; }
    if (off <= 1e-28 * diag) break;                         // eigenvectors to ~1e-14: far below the float32 the result is rounded to
    for (int p = 0; p < 4; p++)
      for (int q = p + 1; q < 4; q++) {
        if (S[p][q] == 0.0) continue;
        const double tau = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = t * cs;
        for (int k = 0; k < 4; k++) { const double a = S[k][p], b = S[k][q]; S[k][p] = cs * a - sn * b; S[k][q] = sn * a + cs * b; }
        for (int k = 0; k < 4; k++) { const double a = S[p][k], b = S[q][k]; S[p][k] = cs * a - sn * b; S[q][k] = sn * a + cs * b; }
        for (int k = 0; k < 4; k++) { const double a = V[k][p], b = V[k][q]; V[k][p] = cs * a - sn * b; V[k][q] = sn * a + cs * b; }
      }
  }
  int m = 0;
  for (int i = 1; i < 4; i++) if (S[i][i] < S[m][m]) m = i;
  for (int k = 0; k < 4; k++) v[k] = V[k][m];
}

// KannalaBrandt8::TriangulateMatches -- :335-403 with Triangulate :405-420.  R12 / t12 = mRlr / mtlr (the rotation and translation of mTlr).
float triangulate_matches(const orbg_camera& cam1, const orbg_camera& cam2, const orbx_keypoint& kp1, const orbx_keypoint& kp2, const float* Tlr,
                          float sigmaLevel, float unc, float* p3D) {
  float r1[3], r2[3];
  kb8_unproject_f(cam1, kp1.x, kp1.y, r1);
  kb8_unproject_f(cam2, kp2.x, kp2.y, r2);
  float r21[3];
  for (int i = 0; i < 3; i++) r21[i] = Tlr[4 * i] * r2[0] + Tlr[4 * i + 1] * r2[1] + Tlr[4 * i + 2] * r2[2];
  auto dot = [](const float* a, const float* b) { return (double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2]; };
  const float cosParallaxRays = (float)(dot(r1, r21) / (std::sqrt(dot(r1, r1)) * std::sqrt(dot(r21, r21))));
  if (cosParallaxRays > 0.9998) return -1;
  float R21[9], t21[3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R21[3 * i + j] = Tlr[4 * j + i];
  for (int i = 0; i < 3; i++) { const float t0 = R21[3 * i] * Tlr[3] + R21[3 * i + 1] * Tlr[7] + R21[3 * i + 2] * Tlr[11]; t21[i] = -t0; }
  const float Tcw1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  float Tcw2[12];
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Tcw2[4 * i + j] = R21[3 * i + j]; Tcw2[4 * i + 3] = t21[i]; }
  float A[4][4];                                                       // :409-412: a scaled row minus a row, float32
  for (int j = 0; j < 4; j++) {
    A[0][j] = r1[0] * Tcw1[8 + j] - Tcw1[j];
    A[1][j] = r1[1] * Tcw1[8 + j] - Tcw1[4 + j];
    A[2][j] = r2[0] * Tcw2[8 + j] - Tcw2[j];
    A[3][j] = r2[1] * Tcw2[8 + j] - Tcw2[4 + j];
  }
  double S[4][4];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += (double)A[k][i] * (double)A[k][j]; S[i][j] = s; }
  double v[4];
  smallest_eigenvector4(S, v);
  float x3D[3];
  for (int i = 0; i < 3; i++) x3D[i] = (float)(v[i] / v[3]);
  const float z1 = x3D[2];
  if (z1 <= 0) return -1;
  const float z2 = (float)(dot(R21 + 6, x3D) + t21[2]);
  if (z2 <= 0) return -1;
  float uv1[2];
  kb8_project_f(cam1, x3D, uv1);
  const float errX1 = uv1[0] - kp1.x, errY1 = uv1[1] - kp1.y;
  if ((errX1 * errX1 + errY1 * errY1) > 5.991 * sigmaLevel) return -1;
  float x3D2[3];
  for (int i = 0; i < 3; i++) { const float t0 = R21[3 * i] * x3D[0] + R21[3 * i + 1] * x3D[1] + R21[3 * i + 2] * x3D[2]; x3D2[i] = (float)(t0 + t21[i]); }
  float uv2[2];
  kb8_project_f(cam2, x3D2, uv2);
  const float errX2 = uv2[0] - kp2.x, errY2 = uv2[1] - kp2.y;
  if ((errX2 * errX2 + errY2 * errY2) > 5.991 * unc) return -1;
  std::memcpy(p3D, x3D, sizeof(x3D));
  return z1;
}

}  // namespace

// Frame::ComputeStereoFishEyeMatches -- S/Frame.cc:1093-1150.  Outputs: mvLeftToRightMatch[Nleft], mvRightToLeftMatch[Nright] (-1 = none),
// mvDepth[Nleft] (-1), mvStereo3Dpoints (3 floats per left feature, written where a match was accepted); *n_matches = nMatches.
extern "C" int oracle_fisheye_stereo_matches(const orbx_fisheye_stereo_view* v, int32_t* left_to_right, int32_t* right_to_left, float* depth,
                                             float* points3d, int* n_matches) {
  if (!v || v->n_left < 0 || v->n_right < 0 || v->mono_left < 0 || v->mono_left > v->n_left || v->mono_right < 0 || v->mono_right > v->n_right)
    return ORBG_BAD_ARG;
  for (int i = 0; i < v->n_left; i++) { left_to_right[i] = -1; depth[i] = -1.0f; }
  for (int i = 0; i < v->n_right; i++) right_to_left[i] = -1;
  int nMatches = 0;
  const int nq = v->n_left - v->mono_left, nt = v->n_right - v->mono_right;
  for (int q = 0; q < nq; q++) {
    if (nt < 2) break;                                                     // (*it).size() >= 2
    const uint8_t* dq = v->desc_left + 32 * (size_t)(q + v->mono_left);
    int d1 = 1 << 30, i1 = -1, d2 = 1 << 30;
    for (int t = 0; t < nt; t++) {
      const int d = oracle_hamming(dq, v->desc_right + 32 * (size_t)(t + v->mono_right));
      if (d < d1) { d2 = d1; d1 = d; i1 = t; } else if (d < d2) d2 = d;
    }
    if (!((float)d1 < (float)d2 * 0.7)) continue;                          // Lowe's ratio, :1137
    const orbx_keypoint& kl = v->kps_left[q + v->mono_left];
    const orbx_keypoint& kr = v->kps_right[i1 + v->mono_right];
    const float sigma1 = v->level_sigma2[kl.octave], sigma2 = v->level_sigma2[kr.octave];
    float p3D[3];
    const float z = triangulate_matches(v->left, v->right, kl, kr, v->Tlr, sigma1, sigma2, p3D);
    if (z > 0.0001f) {
      left_to_right[q + v->mono_left] = i1 + v->mono_right;
      right_to_left[i1 + v->mono_right] = q + v->mono_left;
      std::memcpy(points3d + 3 * (size_t)(q + v->mono_left), p3D, sizeof(p3D));
      depth[q + v->mono_left] = z;
      nMatches++;
    }
  }
  if (n_matches) *n_matches = nMatches;
  return ORBG_OK;
}

// (for the known-answer tests) KannalaBrandt8::unproject and TriangulateMatches on one pair
extern "C" void oracle_kb8_unproject(const orbg_camera* cam, float u, float v, float* ray3) { kb8_unproject_f(*cam, u, v, ray3); }
extern "C" float oracle_kb8_triangulate_matches(const orbg_camera* cam1, const orbg_camera* cam2, const float* uv1, const float* uv2, const float* Tlr,
                                                float sigma1, float sigma2, float* p3D) {
  orbx_keypoint k1{}, k2{};
  k1.x = uv1[0]; k1.y = uv1[1]; k2.x = uv2[0]; k2.y = uv2[1];
  return triangulate_matches(*cam1, *cam2, k1, k2, Tlr, sigma1, sigma2, p3D);
}
