// ORACLE (test infrastructure, not product code) -- see orb_oracle.h for status: parity unpinned.
//
// CPU restatement of Frame::ComputeStereoMatches / grid / isInFrustum (S/Frame.cc) and of the
// ORBmatcher searches on the hot path (S/ORBmatcher.cc), over the flattened views of include/orbgpu.h.
//
// Float conventions for cv::Mat expressions (OpenCV 3.x gemm small-matrix path, core/src/matmul.cpp;
// norm/dot accumulate in double):
//   R*X + t      -> per row: float t0 = r0*X0 + r1*X1 + r2*X2 (float, left to right); out = (float)(t0 + t)
//   -R^T * t     -> general gemm path: out_i = (float)(-(sum_k (double)R[k][i]*(double)t[k]))
//   cv::norm(v)  -> (float) sqrt(sum (double)v_i^2);   a.dot(b) -> sum (double)a_i*b_i

#include "orb_oracle.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

const uint8_t* oracle_level_ptr(const oracle_extractor* e, int level, int* w, int* h, int* stride);
const std::vector<float>& oracle_scale(const oracle_extractor* e);
const std::vector<float>& oracle_inv_scale(const oracle_extractor* e);

namespace {
constexpr int TH_HIGH = 100;      // S/ORBmatcher.cc:36
constexpr int TH_LOW = 50;        // :37
constexpr int HISTO_LENGTH = 30;  // :38

struct ScaleTables {
  std::vector<float> scale;
  float log_sf;
  explicit ScaleTables(const orbm_frame_view* v) {
    scale.resize(v->n_levels);
    scale[0] = 1.0f;
    for (int i = 1; i < v->n_levels; i++) scale[i] = scale[i - 1] * v->scale_factor;   // S/ORBextractor.cc:413-421
    log_sf = std::log(v->scale_factor);   // mfLogScaleFactor = log(mfScaleFactor), float overload
  }
};

struct Grid {
  std::vector<int32_t> start, items;
  float w_inv, h_inv;
};

// Frame::PosInGrid -- S/Frame.cc:699-709
inline bool pos_in_grid(const orbm_frame_view* v, float w_inv, float h_inv, float x, float y, int* px, int* py) {
  *px = (int)std::round((x - v->min_x) * w_inv);
  *py = (int)std::round((y - v->min_y) * h_inv);
  return !(*px < 0 || *px >= ORBG_GRID_COLS || *py < 0 || *py >= ORBG_GRID_ROWS);
}

// Frame::AssignFeaturesToGrid -- S/Frame.cc:360-391 (Nleft == -1); CSR cell id = ix*48+iy.
Grid build_grid(const orbm_frame_view* v) {
  Grid g;
  g.w_inv = static_cast<float>(ORBG_GRID_COLS) / static_cast<float>(v->max_x - v->min_x);   // S/Frame.cc:127-144
  g.h_inv = static_cast<float>(ORBG_GRID_ROWS) / static_cast<float>(v->max_y - v->min_y);
  const int nc = ORBG_GRID_COLS * ORBG_GRID_ROWS;
  g.start.assign(nc + 1, 0);
  std::vector<int> cell(v->n, -1);
  for (int i = 0; i < v->n; i++) {
    int px, py;
    if (pos_in_grid(v, g.w_inv, g.h_inv, v->kps[i].x, v->kps[i].y, &px, &py)) {
      cell[i] = px * ORBG_GRID_ROWS + py;
      g.start[cell[i] + 1]++;
    }
  }
  for (int c = 0; c < nc; c++) g.start[c + 1] += g.start[c];
  g.items.assign(g.start[nc], 0);
  std::vector<int> fill(g.start.begin(), g.start.end() - 1);
  for (int i = 0; i < v->n; i++)
    if (cell[i] >= 0) g.items[fill[cell[i]]++] = i;
  return g;
}

// Frame::GetFeaturesInArea -- S/Frame.cc:628-697 (bRight = false, Nleft == -1)
void features_in_area(const orbm_frame_view* v, const Grid& g, float x, float y, float r, int minLevel, int maxLevel,
                      std::vector<int>& out) {
  out.clear();
  const float factorX = r, factorY = r;
  const int nMinCellX = std::max(0, (int)std::floor((x - v->min_x - factorX) * g.w_inv));
  if (nMinCellX >= ORBG_GRID_COLS) return;
  const int nMaxCellX = std::min((int)ORBG_GRID_COLS - 1, (int)std::ceil((x - v->min_x + factorX) * g.w_inv));
  if (nMaxCellX < 0) return;
  const int nMinCellY = std::max(0, (int)std::floor((y - v->min_y - factorY) * g.h_inv));
  if (nMinCellY >= ORBG_GRID_ROWS) return;
  const int nMaxCellY = std::min((int)ORBG_GRID_ROWS - 1, (int)std::ceil((y - v->min_y + factorY) * g.h_inv));
  if (nMaxCellY < 0) return;
  const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
    for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
      const int c = ix * ORBG_GRID_ROWS + iy;
      for (int j = g.start[c]; j < g.start[c + 1]; j++) {
        const orbx_keypoint& kp = v->kps[g.items[j]];
        if (bCheckLevels) {
          if (kp.octave < minLevel) continue;
          if (maxLevel >= 0 && kp.octave > maxLevel) continue;
        }
        const float distx = kp.x - x, disty = kp.y - y;
        if (std::fabs(distx) < factorX && std::fabs(disty) < factorY) out.push_back(g.items[j]);
      }
    }
}

struct Pose {
  float R[9], t[3], Ow[3];
  explicit Pose(const float* T) {
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) R[3 * i + j] = T[4 * i + j];
      t[i] = T[4 * i + 3];
    }
    for (int i = 0; i < 3; i++) {   // mOw = -mRcw.t()*mtcw  (S/Frame.cc:439-445)
      double s = 0;
      for (int k = 0; k < 3; k++) s += (double)R[3 * k + i] * (double)t[k];
      Ow[i] = (float)(-s);
    }
  }
  void map(const float* X, float* out) const {   // R*X + t
    for (int i = 0; i < 3; i++) {
      float t0 = R[3 * i] * X[0] + R[3 * i + 1] * X[1] + R[3 * i + 2] * X[2];
      out[i] = (float)(t0 + t[i]);
    }
  }
};

inline float norm3(const float* v) {
  double s = (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2];
  return (float)std::sqrt(s);
}

// Frame::isInFrustum -- S/Frame.cc:466-543 (Nleft == -1) with MapPoint::PredictScale S/MapPoint.cc:646-661
struct TrackFields { bool in_view; float px, py, pxr, depth, view_cos; int level; };
TrackFields is_in_frustum(const orbm_frame_view* v, const ScaleTables& st, const Pose& pose, const float* P,
                          const float* Pn, float min_dist_raw, float max_dist_raw, float limit) {
  TrackFields f{false, -1.f, -1.f, 0.f, 0.f, 0.f, 0};
  float Pc[3];
  pose.map(P, Pc);
  const float Pc_dist = norm3(Pc);
  const float PcZ = Pc[2];
  const float invz = 1.0f / PcZ;
  if (PcZ < 0.0f) return f;
  const float u = v->fx * Pc[0] / Pc[2] + v->cx;    // Pinhole::project S/CameraModels/Pinhole.cpp:41-47
  const float vv = v->fy * Pc[1] / Pc[2] + v->cy;
  if (u < v->min_x || u > v->max_x) return f;
  if (vv < v->min_y || vv > v->max_y) return f;
  f.px = u; f.py = vv;
  const float maxDistance = 1.2f * max_dist_raw;    // S/MapPoint.cc:617-627
  const float minDistance = 0.8f * min_dist_raw;
  const float PO[3] = {P[0] - pose.Ow[0], P[1] - pose.Ow[1], P[2] - pose.Ow[2]};
  const float dist = norm3(PO);
  if (dist < minDistance || dist > maxDistance) return f;
  const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
  const float viewCos = (float)(dot / dist);
  if (viewCos < limit) return f;
  const float ratio = max_dist_raw / dist;          // PredictScale
  int nScale = (int)std::ceil(std::log(ratio) / st.log_sf);
  if (nScale < 0) nScale = 0;
  else if (nScale >= v->n_levels) nScale = v->n_levels - 1;
  f.in_view = true;
  f.pxr = u - v->bf * invz;
  f.depth = Pc_dist;
  f.level = nScale;
  f.view_cos = viewCos;
  return f;
}

// ORBmatcher::ComputeThreeMaxima -- S/ORBmatcher.cc:2312-2353
void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

inline int rot_bin(float a1, float a2) {   // S/ORBmatcher.cc:2082-2087 (factor = 1/HISTO_LENGTH, Appendix C-3)
  const float factor = 1.0f / HISTO_LENGTH;
  float rot = a1 - a2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)std::round(rot * factor);
  if (bin == HISTO_LENGTH) bin = 0;
  return bin;
}
}  // namespace

// the two pieces of the rotation-consistency check, for the known-answer tests
extern "C" void oracle_three_maxima(const int32_t* bin_sizes, int L, int32_t* ind3) {
  std::vector<std::vector<int>> h(L);
  for (int i = 0; i < L; i++) h[i].assign((size_t)bin_sizes[i], 0);
  int a = -1, b = -1, c = -1;
  three_maxima(h.data(), L, a, b, c);
  ind3[0] = a; ind3[1] = b; ind3[2] = c;
}
extern "C" int oracle_rot_bin(float angle_a, float angle_b) { return rot_bin(angle_a, angle_b); }

// ORBmatcher::DescriptorDistance -- S/ORBmatcher.cc:2358-2374
extern "C" int oracle_hamming(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t pa, pb;
    std::memcpy(&pa, a + 4 * i, 4);
    std::memcpy(&pb, b + 4 * i, 4);
    unsigned int v = pa ^ pb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

extern "C" int oracle_hamming_matrix(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* dist) {
  for (int i = 0; i < nq; i++)
    for (int j = 0; j < nt; j++) dist[(size_t)i * nt + j] = oracle_hamming(q + 32 * (size_t)i, t + 32 * (size_t)j);
  return ORBG_OK;
}

// Sequential best / second-best scan exactly as the matcher inner loops do it (S/ORBmatcher.cc:104-120).
extern "C" int oracle_hamming_best2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* out4) {
  for (int i = 0; i < nq; i++) {
    int best = 256, best2 = 256, bi = -1, bi2 = -1;
    for (int j = 0; j < nt; j++) {
      int d = oracle_hamming(q + 32 * (size_t)i, t + 32 * (size_t)j);
      if (d < best) { best2 = best; bi2 = bi; best = d; bi = j; }
      else if (d < best2) { best2 = d; bi2 = j; }
    }
    out4[4 * i] = best; out4[4 * i + 1] = bi; out4[4 * i + 2] = best2; out4[4 * i + 3] = bi2;
  }
  return ORBG_OK;
}

// Frame::UndistortKeyPoints (S/Frame.cc:721-754) / ComputeImageBounds (:756-783): cv::undistortPoints(mat, mat, toK(), mDistCoef,
// cv::Mat(), mK) on CV_32FC2 points.  cv::undistortPoints is NOT in the reference tree (OpenCV, imgproc/src/undistort.cpp); restated
// from the 3.2-era cvUndistortPoints the README's platform ships: camera matrix and coefficients widened float -> double, R = I,
// RR = P * R with P = mK (products with an identity: exact), per point
//     x = (u - cx) * ifx,  y = (v - cy) * ify            (ifx = 1. / fx)
//     5 x { r2 = x*x + y*y;  icdist = (1 + ((k6*r2 + k5)*r2 + k4)*r2) / (1 + ((k3*r2 + k2)*r2 + k1)*r2);   [k4..k6 = 0 here]
//           dX = 2*p1*x*y + p2*(r2 + 2*x*x);  dY = p1*(r2 + 2*y*y) + 2*p2*x*y;  x = (x0 - dX)*icdist;  y = (y0 - dY)*icdist }
//     xx = RR00*x + RR01*y + RR02,  yy = RR10*x + RR11*y + RR12,  ww = 1. / (RR20*x + RR21*y + RR22);  out = (float)(xx*ww), (float)(yy*ww)
// all in double, no contraction (the library is built without FMA for x86-64).  Later OpenCV versions (>= 3.4.2) add a
// termination criterion whose default is the same 5 iterations and an early exit for icdist < 0, which no point inside an image of
// a camera with EuRoC-like coefficients reaches.  PARITY UNPINNED, like every OpenCV restatement in this oracle.
extern "C" int oracle_undistort_points(const float* xy_in, int n, float fx_, float fy_, float cx_, float cy_, const orbx_distortion* dist,
                                       float* xy_out) {
  if (n < 0 || (n > 0 && (!xy_in || !xy_out))) return ORBG_BAD_ARG;
  if (!dist || dist->k1 == 0.0f) {                               // S/Frame.cc:723-727, :776-782
    for (int i = 0; i < 2 * n; i++) xy_out[i] = xy_in[i];
    return ORBG_OK;
  }
  const double fx = fx_, fy = fy_, cx = cx_, cy = cy_;
  const double ifx = 1. / fx, ify = 1. / fy;
  const double k0 = dist->k1, k1 = dist->k2, k2 = dist->p1, k3 = dist->p2, k4 = dist->k3;   // OpenCV's k[0..4] = k1 k2 p1 p2 k3
  const double RR[3][3] = {{fx, 0, cx}, {0, fy, cy}, {0, 0, 1}};
  for (int i = 0; i < n; i++) {
    double x = xy_in[2 * i], y = xy_in[2 * i + 1];
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
      const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x);
      const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
    const double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
    const double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
    xy_out[2 * i] = (float)(xx * ww);
    xy_out[2 * i + 1] = (float)(yy * ww);
  }
  return ORBG_OK;
}

extern "C" int oracle_build_grid(const orbm_frame_view* view, int32_t* cell_start, int32_t* cell_items) {
  Grid g = build_grid(view);
  std::memcpy(cell_start, g.start.data(), g.start.size() * sizeof(int32_t));
  if (!g.items.empty()) std::memcpy(cell_items, g.items.data(), g.items.size() * sizeof(int32_t));
  return ORBG_OK;
}

extern "C" int oracle_features_in_area(const orbm_frame_view* view, float x, float y, float r, int min_level,
                                       int max_level, int32_t* out_idx, int cap) {
  Grid g = build_grid(view);
  std::vector<int> out;
  features_in_area(view, g, x, y, r, min_level, max_level, out);
  for (size_t i = 0; i < out.size() && (int)i < cap; i++) out_idx[i] = out[i];
  return (int)out.size();
}

extern "C" int oracle_is_in_frustum(const orbm_frame_view* view, const float* Tcw, const orbm_worldpoints_view* pts,
                                    float limit, uint8_t* track_in_view, float* proj_x, float* proj_y,
                                    float* proj_xr, float* track_depth, int32_t* scale_level, float* view_cos) {
  ScaleTables st(view);
  Pose pose(Tcw);
  for (int i = 0; i < pts->m; i++) {
    TrackFields f = is_in_frustum(view, st, pose, pts->pos + 3 * i, pts->normal + 3 * i, pts->min_dist[i],
                                  pts->max_dist[i], limit);
    track_in_view[i] = f.in_view;
    proj_x[i] = f.px; proj_y[i] = f.py; proj_xr[i] = f.pxr; track_depth[i] = f.depth;
    scale_level[i] = f.level; view_cos[i] = f.view_cos;
  }
  return ORBG_OK;
}

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)
// -- S/ORBmatcher.cc:44-214, Nleft == -1 branch only.
extern "C" int oracle_search_by_projection_mps(const orbm_frame_view* view, const orbm_mappoints_view* mps, float th,
                                               int far_points, float th_far_points, float nnratio,
                                               int32_t* assigned_mp, int32_t* assigned_obs, int* nmatches_out) {
  ScaleTables st(view);
  Grid g = build_grid(view);
  int nmatches = 0;
  const bool bFactor = th != 1.0;
  std::vector<int> vIndices;
  for (int iMP = 0; iMP < mps->m; iMP++) {
    if (!mps->track_in_view[iMP]) continue;                                   // :53 (mbTrackInViewR false)
    if (far_points && mps->track_depth[iMP] > th_far_points) continue;        // :56
    if (mps->bad[iMP]) continue;                                              // :59
    const int nPredictedLevel = mps->scale_level[iMP];
    float r = (mps->view_cos[iMP] > 0.998) ? 2.5f : 4.0f;                     // RadiusByViewingCos :216-222
    if (bFactor) r *= th;
    features_in_area(view, g, mps->proj_x[iMP], mps->proj_y[iMP], r * st.scale[nPredictedLevel], nPredictedLevel - 1,
                     nPredictedLevel, vIndices);
    if (vIndices.empty()) continue;
    const uint8_t* dMP = mps->desc + 32 * (size_t)iMP;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int idx : vIndices) {
      if (assigned_mp[idx] >= 0 && assigned_obs[idx] > 0) continue;           // :89-91
      if (view->uright && view->uright[idx] > 0) {                             // :93-98
        const float er = std::fabs(mps->proj_xr[iMP] - view->uright[idx]);
        if (er > r * st.scale[nPredictedLevel]) continue;
      }
      const int dist = oracle_hamming(dMP, view->desc + 32 * (size_t)idx);
      if (dist < bestDist) {
        bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel;
        bestLevel = view->kps[idx].octave; bestIdx = idx;
      } else if (dist < bestDist2) {
        bestLevel2 = view->kps[idx].octave; bestDist2 = dist;
      }
    }
    if (bestDist <= TH_HIGH) {                                                 // :124-141
      if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
      if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
        assigned_mp[bestIdx] = iMP;
        assigned_obs[bestIdx] = mps->n_obs[iMP];
        nmatches++;
      }
    }
  }
  if (nmatches_out) *nmatches_out = nmatches;
  return ORBG_OK;
}

// ---- two-camera rig (Frame::Nleft != -1): isInFrustum / SearchByProjection(Frame, MapPoints)
//
// GeometricCamera::project(cv::Mat) as Frame::isInFrustumChecks calls it: Pinhole (S/CameraModels/Pinhole.cpp:41-47) and
// KannalaBrandt8 (S/CameraModels/KannalaBrandt8.cpp:28-49), both on cv::Point3f, float32 throughout (cos / sin of a float: the float
// overloads -- <math.h> comes in through OpenCV's C headers).
inline void rig_cam_project(const orbg_camera& c, const float* p, float* uv) {
  if (c.model == ORBG_CAM_KANNALA_BRANDT8) {
    const float x2_plus_y2 = p[0] * p[0] + p[1] * p[1];
    const float theta = atan2f(sqrtf(x2_plus_y2), p[2]);
    const float psi = atan2f(p[1], p[0]);
    const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
    const float r = theta + c.k[0] * theta3 + c.k[1] * theta5 + c.k[2] * theta7 + c.k[3] * theta9;
    uv[0] = c.fx * r * cosf(psi) + c.cx;
    uv[1] = c.fy * r * sinf(psi) + c.cy;
  } else {
    uv[0] = c.fx * p[0] / p[2] + c.cx;
    uv[1] = c.fy * p[1] / p[2] + c.cy;
  }
}

// Frame::isInFrustumChecks -- S/Frame.cc:1154-1231.  mR / mt / twc of the right camera are cv::Mat products (float32 small-matrix path):
//   mR = Rrl * mRcw;  mt = Rrl * mtcw + trl;  twc = mRwc * tlr + mOw   (:1160-1164)
struct RigSide { float R[9], t[3], twc[3]; };
RigSide rig_side(const Pose& pose, const float* Trl, const float* Tlr, bool right) {
  RigSide s;
  if (!right) {
    std::memcpy(s.R, pose.R, sizeof(s.R)); std::memcpy(s.t, pose.t, sizeof(s.t)); std::memcpy(s.twc, pose.Ow, sizeof(s.twc));
    return s;
  }
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) s.R[3 * i + j] = Trl[4 * i] * pose.R[j] + Trl[4 * i + 1] * pose.R[3 + j] + Trl[4 * i + 2] * pose.R[6 + j];
    const float t0 = Trl[4 * i] * pose.t[0] + Trl[4 * i + 1] * pose.t[1] + Trl[4 * i + 2] * pose.t[2];
    s.t[i] = (float)(t0 + Trl[4 * i + 3]);
    const float w0 = pose.R[i] * Tlr[3] + pose.R[3 + i] * Tlr[7] + pose.R[6 + i] * Tlr[11];   // mRwc = mRcw.t(), a stored matrix
    s.twc[i] = (float)(w0 + pose.Ow[i]);
  }
  return s;
}
TrackFields rig_frustum_checks(const orbm_frame_view* v, const ScaleTables& st, const RigSide& side, const orbg_camera& cam, const float* P,
                               const float* Pn, float min_dist_raw, float max_dist_raw, float limit) {
  TrackFields f{false, 0.f, 0.f, 0.f, 0.f, 0.f, -1};
  float Pc[3];
  for (int i = 0; i < 3; i++) {
    const float t0 = side.R[3 * i] * P[0] + side.R[3 * i + 1] * P[1] + side.R[3 * i + 2] * P[2];
    Pc[i] = (float)(t0 + side.t[i]);
  }
  const float Pc_dist = norm3(Pc);
  if (Pc[2] < 0.0f) return f;
  float uv[2];
  rig_cam_project(cam, Pc, uv);
  if (uv[0] < v->min_x || uv[0] > v->max_x) return f;
  if (uv[1] < v->min_y || uv[1] > v->max_y) return f;
  const float maxDistance = 1.2f * max_dist_raw, minDistance = 0.8f * min_dist_raw;
  const float PO[3] = {P[0] - side.twc[0], P[1] - side.twc[1], P[2] - side.twc[2]};
  const float dist = norm3(PO);
  if (dist < minDistance || dist > maxDistance) return f;
  const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
  const float viewCos = (float)(dot / dist);
  if (viewCos < limit) return f;
  const float ratio = max_dist_raw / dist;
  int nScale = (int)std::ceil(std::log(ratio) / st.log_sf);
  if (nScale < 0) nScale = 0;
  else if (nScale >= v->n_levels) nScale = v->n_levels - 1;
  f.in_view = true; f.px = uv[0]; f.py = uv[1]; f.level = nScale; f.view_cos = viewCos; f.depth = Pc_dist;
  return f;
}

// Frame::isInFrustum, Nleft != -1 -- S/Frame.cc:545-554: both cameras' checks; a point that fails a camera's checks has
// mbTrackInView(R) = false and mnTrackScaleLevel(R) = -1 (the other fields keep stale values in the reference: 0 here).
extern "C" int oracle_is_in_frustum_rig(const orbm_frame_view* view, const float* Tcw, const orbg_camera_rig* rig, const float* Tlr,
                                        const orbm_worldpoints_view* pts, float limit, uint8_t* in_view, float* px, float* py, float* depth,
                                        int32_t* level, float* view_cos, uint8_t* in_view_r, float* px_r, float* py_r, float* depth_r,
                                        int32_t* level_r, float* view_cos_r) {
  ScaleTables st(view);
  Pose pose(Tcw);
  // rig->has_right == 0: ONE camera behind a model -- the Nleft == -1 branch of Frame::isInFrustum (S/Frame.cc:466-543) with
  // mpCamera->project: the same checks in the same order; the left outputs only
  const RigSide L = rig_side(pose, rig->Trl, Tlr, false), R = rig_side(pose, rig->Trl, Tlr, rig->has_right != 0);
  for (int i = 0; i < pts->m; i++) {
    const TrackFields a = rig_frustum_checks(view, st, L, rig->left, pts->pos + 3 * i, pts->normal + 3 * i, pts->min_dist[i], pts->max_dist[i], limit);
    in_view[i] = a.in_view; px[i] = a.px; py[i] = a.py; depth[i] = a.depth; level[i] = a.level; view_cos[i] = a.view_cos;
    if (!rig->has_right) continue;
    const TrackFields b = rig_frustum_checks(view, st, R, rig->right, pts->pos + 3 * i, pts->normal + 3 * i, pts->min_dist[i], pts->max_dist[i], limit);
    in_view_r[i] = b.in_view; px_r[i] = b.px; py_r[i] = b.py; depth_r[i] = b.depth; level_r[i] = b.level; view_cos_r[i] = b.view_cos;
  }
  return ORBG_OK;
}

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints) -- S/ORBmatcher.cc:44-214 with
// Nleft != -1: `left` / `right` are the two cameras' features (mvKeys with mGrid, mvKeysRight with mGridRight; a right feature i is
// entry Nleft + i of mvpMapPoints / mDescriptors), `mps` the left camera's track fields, `mps_r` the right camera's
// (mbTrackInViewR, mTrackProjXR / YR, mnTrackScaleLevelR, mTrackViewCosR; bad / track_depth / desc / n_obs are read from `mps`).
extern "C" int oracle_search_by_projection_mps_rig(const orbm_frame_view* left, const orbm_frame_view* right, const orbm_mappoints_view* mps,
                                                   const orbm_mappoints_view* mps_r, const int32_t* left_to_right, const int32_t* right_to_left,
                                                   float th, int far_points, float th_far_points, float nnratio, int32_t* assigned_mp,
                                                   int32_t* assigned_obs, int* nmatches_out) {
  ScaleTables st(left);
  const Grid gl = build_grid(left), gr = build_grid(right);
  const int Nleft = left->n;
  int nmatches = 0;
  const bool bFactor = th != 1.0;
  std::vector<int> vIndices;
  for (int iMP = 0; iMP < mps->m; iMP++) {
    if (!mps->track_in_view[iMP] && !mps_r->track_in_view[iMP]) continue;      // :53
    if (far_points && mps->track_depth[iMP] > th_far_points) continue;         // :56
    if (mps->bad[iMP]) continue;                                               // :59
    const uint8_t* dMP = mps->desc + 32 * (size_t)iMP;
    if (mps->track_in_view[iMP]) {                                             // :62-143
      const int nPredictedLevel = mps->scale_level[iMP];
      float r = (mps->view_cos[iMP] > 0.998) ? 2.5f : 4.0f;
      if (bFactor) r *= th;
      features_in_area(left, gl, mps->proj_x[iMP], mps->proj_y[iMP], r * st.scale[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel, vIndices);
      if (!vIndices.empty()) {
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : vIndices) {
          if (assigned_mp[idx] >= 0 && assigned_obs[idx] > 0) continue;        // :89-91 (no mvuRight test: Nleft != -1, :93)
          const int dist = oracle_hamming(dMP, left->desc + 32 * (size_t)idx);
          if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = left->kps[idx].octave; bestIdx = idx; }
          else if (dist < bestDist2) { bestLevel2 = left->kps[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
          if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;   // :126-127: leaves the POINT, the right camera's block too
          if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
            assigned_mp[bestIdx] = iMP; assigned_obs[bestIdx] = mps->n_obs[iMP];
            if (left_to_right[bestIdx] != -1) {                                // :132-136
              assigned_mp[left_to_right[bestIdx] + Nleft] = iMP; assigned_obs[left_to_right[bestIdx] + Nleft] = mps->n_obs[iMP];
              nmatches++;
            }
            nmatches++;
          }
        }
      }
    }
    if (mps_r->track_in_view[iMP]) {                                           // :145-211
      const int nPredictedLevel = mps_r->scale_level[iMP];
      if (nPredictedLevel != -1) {
        const float r = (mps_r->view_cos[iMP] > 0.998) ? 2.5f : 4.0f;           // (th is not applied here)
        features_in_area(right, gr, mps_r->proj_x[iMP], mps_r->proj_y[iMP], r * st.scale[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel, vIndices);
        if (vIndices.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : vIndices) {
          if (assigned_mp[idx + Nleft] >= 0 && assigned_obs[idx + Nleft] > 0) continue;
          const int dist = oracle_hamming(dMP, right->desc + 32 * (size_t)idx);
          if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = right->kps[idx].octave; bestIdx = idx; }
          else if (dist < bestDist2) { bestLevel2 = right->kps[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
          if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
          if (right_to_left[bestIdx] != -1) {                                  // :199-203: whatever that left feature held
            assigned_mp[right_to_left[bestIdx]] = iMP; assigned_obs[right_to_left[bestIdx]] = mps->n_obs[iMP];
            nmatches++;
          }
          assigned_mp[bestIdx + Nleft] = iMP; assigned_obs[bestIdx + Nleft] = mps->n_obs[iMP];
          nmatches++;
        }
      }
    }
  }
  if (nmatches_out) *nmatches_out = nmatches;
  return ORBG_OK;
}

// ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) -- S/ORBmatcher.cc:1970-2186 with
// CurrentFrame.Nleft != -1: the left camera's search through mpCamera->project, then (:2092-2160) the point moved into the right
// camera's frame by mTrl and projected -- through mpCamera again, as the text has it -- into mGridRight.  `last` holds the last frame's
// Nleft + Nright entries (octave / angle of mvKeys resp. mvKeysRight).
extern "C" int oracle_search_by_projection_frame_rig(const orbm_frame_view* left, const orbm_frame_view* right, const float* Tcw_cur,
                                                     const orbg_camera_rig* rig, const orbm_lastframe_view* last, float th, int mono,
                                                     int check_ori, int32_t* assigned_mp, int32_t* assigned_obs, int* nmatches_out) {
  ScaleTables st(left);
  const bool two = rig->has_right != 0 && right != nullptr;        // one camera behind a model (Nleft == -1): the left block alone
  const Grid gl = build_grid(left), gr = two ? build_grid(right) : Grid();
  const int Nleft = left->n;
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  Pose pc(Tcw_cur), pl(last->Tcw);
  float tlc[3];
  pl.map(pc.Ow, tlc);
  const bool bForward = tlc[2] > left->b && !mono;
  const bool bBackward = -tlc[2] > left->b && !mono;
  std::vector<int> vIndices2;
  const float* Trl = rig->Trl;
  for (int i = 0; i < last->n; i++) {
    if (!last->mp_valid[i] || last->outlier[i]) continue;
    float x3Dc[3];
    pc.map(last->world_pos + 3 * i, x3Dc);
    const float invzc = (float)(1.0 / x3Dc[2]);
    if (invzc < 0) continue;
    float uv[2];
    rig_cam_project(rig->left, x3Dc, uv);
    if (uv[0] < left->min_x || uv[0] > left->max_x) continue;
    if (uv[1] < left->min_y || uv[1] > left->max_y) continue;
    const int nLastOctave = last->octave[i];
    const float radius = th * st.scale[nLastOctave];
    auto window = [&](const orbm_frame_view* v, const Grid& g, const float* c) {
      if (!std::isfinite(c[0]) || !std::isfinite(c[1])) { vIndices2.clear(); return; }    // (int)floor(NaN) is INT_MIN on x86: no cell
      if (bForward) features_in_area(v, g, c[0], c[1], radius, nLastOctave, -1, vIndices2);
      else if (bBackward) features_in_area(v, g, c[0], c[1], radius, 0, nLastOctave, vIndices2);
      else features_in_area(v, g, c[0], c[1], radius, nLastOctave - 1, nLastOctave + 1, vIndices2);
    };
    window(left, gl, uv);
    if (vIndices2.empty()) continue;                                              // :2033-2034: the right camera is not tried either
    const uint8_t* dMP = last->desc + 32 * (size_t)i;
    {
      int bestDist = 256, bestIdx2 = -1;
      for (int i2 : vIndices2) {
        if (assigned_mp[i2] >= 0 && assigned_obs[i2] > 0) continue;
        const int dist = oracle_hamming(dMP, left->desc + 32 * (size_t)i2);
        if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
      }
      if (bestDist <= TH_HIGH) {
        assigned_mp[bestIdx2] = i; assigned_obs[bestIdx2] = last->n_obs[i];
        nmatches++;
        if (check_ori) rotHist[rot_bin(last->angle[i], left->kps[bestIdx2].angle)].push_back(bestIdx2);
      }
    }
    if (!two) continue;
    float x3Dr[3];
    for (int a = 0; a < 3; a++) {
      const float t0 = Trl[4 * a] * x3Dc[0] + Trl[4 * a + 1] * x3Dc[1] + Trl[4 * a + 2] * x3Dc[2];
      x3Dr[a] = (float)(t0 + Trl[4 * a + 3]);
    }
    float uvr[2];
    rig_cam_project(rig->left, x3Dr, uvr);                                         // :2095: mpCamera, not mpCamera2
    window(right, gr, uvr);
    int bestDist = 256, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      if (assigned_mp[i2 + Nleft] >= 0 && assigned_obs[i2 + Nleft] > 0) continue;
      const int dist = oracle_hamming(dMP, right->desc + 32 * (size_t)i2);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= TH_HIGH) {
      assigned_mp[bestIdx2 + Nleft] = i; assigned_obs[bestIdx2 + Nleft] = last->n_obs[i];
      nmatches++;
      if (check_ori) rotHist[rot_bin(last->angle[i], right->kps[bestIdx2].angle)].push_back(bestIdx2 + Nleft);
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (int idx : rotHist[i]) { assigned_mp[idx] = -1; assigned_obs[idx] = 0; nmatches--; }
  }
  if (nmatches_out) *nmatches_out = nmatches;
  return ORBG_OK;
}

// Tracking::SearchLocalPoints body (S/Tracking.cc:3111-3153): isInFrustum(.,0.5) then SearchByProjection.
extern "C" int oracle_search_local_points(const orbm_frame_view* view, const orbm_worldpoints_view* pts, const float* Tcw,
                                          float th, int far_points, float th_far_points, float nnratio,
                                          int32_t* assigned_mp, int32_t* assigned_obs, int* nmatches) {
  const int m = pts->m;
  std::vector<uint8_t> tiv(m);
  std::vector<float> px(m), py(m), pxr(m), dep(m), vc(m);
  std::vector<int32_t> lvl(m);
  oracle_is_in_frustum(view, Tcw, pts, 0.5f, tiv.data(), px.data(), py.data(), pxr.data(), dep.data(), lvl.data(), vc.data());
  for (int i = 0; i < m; i++)
    if ((pts->skip && pts->skip[i]) || pts->bad[i]) tiv[i] = 0;
  orbm_mappoints_view mv;
  mv.m = m; mv.track_in_view = tiv.data(); mv.bad = pts->bad; mv.proj_x = px.data(); mv.proj_y = py.data();
  mv.proj_xr = pxr.data(); mv.track_depth = dep.data(); mv.scale_level = lvl.data(); mv.view_cos = vc.data();
  mv.desc = pts->desc; mv.n_obs = pts->n_obs;
  return oracle_search_by_projection_mps(view, &mv, th, far_points, th_far_points, nnratio, assigned_mp, assigned_obs, nmatches);
}

// ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)
// -- S/ORBmatcher.cc:1970-2186, Nleft == -1.
extern "C" int oracle_search_by_projection_frame(const orbm_frame_view* cur, const float* Tcw_cur,
                                                 const orbm_lastframe_view* last, float th, int mono, int check_ori,
                                                 int32_t* assigned_mp, int32_t* assigned_obs, int* nmatches_out) {
  ScaleTables st(cur);
  Grid g = build_grid(cur);
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  Pose pc(Tcw_cur), pl(last->Tcw);
  // twc = -Rcw.t()*tcw ; tlc = Rlw*twc+tlw   :1983-1988
  float tlc[3];
  pl.map(pc.Ow, tlc);
  const bool bForward = tlc[2] > cur->b && !mono;
  const bool bBackward = -tlc[2] > cur->b && !mono;
  std::vector<int> vIndices2;
  for (int i = 0; i < last->n; i++) {
    if (!last->mp_valid[i] || last->outlier[i]) continue;
    float x3Dc[3];
    pc.map(last->world_pos + 3 * i, x3Dc);
    const float xc = x3Dc[0], yc = x3Dc[1];
    const float invzc = (float)(1.0 / x3Dc[2]);
    if (invzc < 0) continue;
    const float u = cur->fx * xc / x3Dc[2] + cur->cx;
    const float v = cur->fy * yc / x3Dc[2] + cur->cy;
    if (u < cur->min_x || u > cur->max_x) continue;
    if (v < cur->min_y || v > cur->max_y) continue;
    const int nLastOctave = last->octave[i];
    const float radius = th * st.scale[nLastOctave];
    if (bForward) features_in_area(cur, g, u, v, radius, nLastOctave, -1, vIndices2);
    else if (bBackward) features_in_area(cur, g, u, v, radius, 0, nLastOctave, vIndices2);
    else features_in_area(cur, g, u, v, radius, nLastOctave - 1, nLastOctave + 1, vIndices2);
    if (vIndices2.empty()) continue;
    const uint8_t* dMP = last->desc + 32 * (size_t)i;
    int bestDist = 256, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      if (assigned_mp[i2] >= 0 && assigned_obs[i2] > 0) continue;                // :2045-2047
      if (cur->uright && cur->uright[i2] > 0) {                                   // :2049-2055
        const float ur = u - cur->bf * invzc;
        const float er = std::fabs(ur - cur->uright[i2]);
        if (er > radius) continue;
      }
      const int dist = oracle_hamming(dMP, cur->desc + 32 * (size_t)i2);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= TH_HIGH) {
      assigned_mp[bestIdx2] = i;
      assigned_obs[bestIdx2] = last->n_obs[i];
      nmatches++;
      if (check_ori) rotHist[rot_bin(last->angle[i], cur->kps[bestIdx2].angle)].push_back(bestIdx2);
    }
  }
  if (check_ori) {                                                                 // :2164-2183
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (int idx : rotHist[i]) {
          assigned_mp[idx] = -1;
          assigned_obs[idx] = 0;
          nmatches--;
        }
  }
  if (nmatches_out) *nmatches_out = nmatches;
  return ORBG_OK;
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) -- S/ORBmatcher.cc:269-471, Nleft == -1.
// SearchByBoW(KeyFrame*, Frame&, ...) on a two-camera Frame (F.Nleft != -1) -- S/ORBmatcher.cc:342-430: `view` holds all N = Nleft +
// Nright features (mvKeys then mvKeysRight; descriptor rows as in mDescriptors), the best two of a bucket are kept per camera, the right
// camera's best is taken -- without a ratio test (`|| true`, :401) -- only under the left one's `bestDist1 <= TH_LOW` (:373-399).
extern "C" int oracle_search_by_bow_rig(const orbm_frame_view* view, int n_left, const orbm_featvec_view* fvF, const uint8_t* kf_desc, int nkf,
                                        const uint8_t* kf_mp_valid, const float* kf_angle, const orbm_featvec_view* fvK, float nnratio,
                                        int check_ori, int32_t* matches, int* nmatches_out) {
  (void)nkf;
  for (int i = 0; i < view->n; i++) matches[i] = -1;
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  int k = 0, f = 0;
  while (k < fvK->n_nodes && f < fvF->n_nodes) {
    if (fvK->node_id[k] == fvF->node_id[f]) {
      for (uint32_t a = fvK->start[k]; a < fvK->start[k + 1]; a++) {
        const uint32_t realIdxKF = fvK->feat_idx[a];
        if (!kf_mp_valid[realIdxKF]) continue;
        const uint8_t* dKF = kf_desc + 32 * (size_t)realIdxKF;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256, bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
        for (uint32_t b = fvF->start[f]; b < fvF->start[f + 1]; b++) {
          const int realIdxF = (int)fvF->feat_idx[b];
          if (matches[realIdxF] >= 0) continue;
          const int dist = oracle_hamming(dKF, view->desc + 32 * (size_t)realIdxF);
          if (realIdxF < n_left && dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
          else if (realIdxF < n_left && dist < bestDist2) bestDist2 = dist;
          if (realIdxF >= n_left && dist < bestDist1R) { bestDist2R = bestDist1R; bestDist1R = dist; bestIdxFR = realIdxF; }
          else if (realIdxF >= n_left && dist < bestDist2R) bestDist2R = dist;
        }
        if (bestDist1 <= TH_LOW) {
          if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
            matches[bestIdxF] = (int)realIdxKF;
            if (check_ori) rotHist[rot_bin(kf_angle[realIdxKF], view->kps[bestIdxF].angle)].push_back(bestIdxF);
            nmatches++;
          }
          if (bestDist1R <= TH_LOW) {
            matches[bestIdxFR] = (int)realIdxKF;
            if (check_ori) rotHist[rot_bin(kf_angle[realIdxKF], view->kps[bestIdxFR].angle)].push_back(bestIdxFR);
            nmatches++;
          }
        }
      }
      k++; f++;
    } else if (fvK->node_id[k] < fvF->node_id[f]) {
      k = (int)(std::lower_bound(fvK->node_id, fvK->node_id + fvK->n_nodes, fvF->node_id[f]) - fvK->node_id);
    } else {
      f = (int)(std::lower_bound(fvF->node_id, fvF->node_id + fvF->n_nodes, fvK->node_id[k]) - fvF->node_id);
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx : rotHist[i]) { matches[idx] = -1; nmatches--; }
    }
  }
  if (nmatches_out) *nmatches_out = nmatches;
  return ORBG_OK;
}

extern "C" int oracle_search_by_bow(const orbm_frame_view* view, const orbm_featvec_view* fvF,
                                    const uint8_t* kf_desc, int nkf, const uint8_t* kf_mp_valid, const float* kf_angle,
                                    const orbm_featvec_view* fvK, float nnratio, int check_ori,
                                    int32_t* matches, int* nmatches_out) {
  (void)nkf;
  for (int i = 0; i < view->n; i++) matches[i] = -1;
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  int k = 0, f = 0;
  while (k < fvK->n_nodes && f < fvF->n_nodes) {
    if (fvK->node_id[k] == fvF->node_id[f]) {
      for (uint32_t a = fvK->start[k]; a < fvK->start[k + 1]; a++) {
        const uint32_t realIdxKF = fvK->feat_idx[a];
        if (!kf_mp_valid[realIdxKF]) continue;                       // !pMP || pMP->isBad()
        const uint8_t* dKF = kf_desc + 32 * (size_t)realIdxKF;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (uint32_t b = fvF->start[f]; b < fvF->start[f + 1]; b++) {
          const uint32_t realIdxF = fvF->feat_idx[b];
          if (matches[realIdxF] >= 0) continue;                      // :324-325
          const int dist = oracle_hamming(dKF, view->desc + 32 * (size_t)realIdxF);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
          else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 <= TH_LOW) {
          if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
            matches[bestIdxF] = (int)realIdxKF;
            if (check_ori) rotHist[rot_bin(kf_angle[realIdxKF], view->kps[bestIdxF].angle)].push_back(bestIdxF);
            nmatches++;
          }
        }
      }
      k++; f++;
    } else if (fvK->node_id[k] < fvF->node_id[f]) {
      // KFit = vFeatVecKF.lower_bound(Fit->first)
      k = (int)(std::lower_bound(fvK->node_id, fvK->node_id + fvK->n_nodes, fvF->node_id[f]) - fvK->node_id);
    } else {
      f = (int)(std::lower_bound(fvF->node_id, fvF->node_id + fvF->n_nodes, fvK->node_id[k]) - fvF->node_id);
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx : rotHist[i]) { matches[idx] = -1; nmatches--; }
    }
  }
  if (nmatches_out) *nmatches_out = nmatches;
  return ORBG_OK;
}

// ---- server-side KeyFrame matchers
//
// Sim3 decomposition as S/ORBmatcher.cc:484-488 / :601-605 with cv::Mat float conventions:
//   scw  = (float) sqrt( sum_k (double)s0k*(double)s0k )                      (Mat::dot accumulates in double)
//   Rcw  = sRcw / scw  -> MatExpr scale by alpha = 1.0/(double)scw, evaluated by convertTo 32f->32f as
//          dst = src * (float)alpha + 0.0f                                     (cvtScale_<float,float,float>)
//   tcw  likewise;   Ow = -Rcw.t()*tcw through the gemm path (see file header)
static void sim3_pose(const float* S, float* T16) {
  double d = 0;
  for (int k = 0; k < 3; k++) d += (double)S[k] * (double)S[k];
  const float scw = (float)std::sqrt(d);
  const float alpha = (float)(1.0 / (double)scw);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) T16[4 * i + j] = S[4 * i + j] * alpha + 0.0f;
    T16[4 * i + 3] = S[4 * i + 3] * alpha + 0.0f;
  }
  T16[12] = T16[13] = T16[14] = 0.f; T16[15] = 1.f;
}

// ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th, ratioHamming)      S/ORBmatcher.cc:473-587
// (camera_project = 1) and the overload carrying vpPointsKFs / vpMatchedKF                   :589-700 (camera_project = 0,
// which projects with a float invz instead of Pinhole::project).  matched[idx] >= 0 means vpMatched[idx] != NULL on entry;
// on exit newly matched features hold the index of the candidate point.  already_found[i] = 1 when vpPoints[i] is
// a member of vpMatched on entry (spAlreadyFound, :491-492).
static int search_by_projection_sim3_impl(const orbm_frame_view* kf, const orbm_worldpoints_view* pts, const float* Scw, const uint8_t* already_found, int th,
                                         float ratio_hamming, int camera_project, int32_t* matched, int* nmatches_out, const orbg_camera* cam);
extern "C" int oracle_search_by_projection_sim3(const orbm_frame_view* kf, const orbm_worldpoints_view* pts, const float* Scw,
                                                const uint8_t* already_found, int th, float ratio_hamming,
                                                int camera_project, int32_t* matched, int* nmatches_out) {
  return search_by_projection_sim3_impl(kf, pts, Scw, already_found, th, ratio_hamming, camera_project, matched, nmatches_out, nullptr);
}
// ... :473-587 with pKF->mpCamera a camera model (a fisheye keyframe): :515 projects through it
extern "C" int oracle_search_by_projection_sim3_cam(const orbm_frame_view* kf, const orbm_worldpoints_view* pts, const float* Scw, const orbg_camera* cam,
                                                    const uint8_t* already_found, int th, float ratio_hamming, int32_t* matched, int* nmatches_out) {
  return search_by_projection_sim3_impl(kf, pts, Scw, already_found, th, ratio_hamming, 1, matched, nmatches_out, cam);
}
static int search_by_projection_sim3_impl(const orbm_frame_view* kf, const orbm_worldpoints_view* pts, const float* Scw, const uint8_t* already_found, int th,
                                         float ratio_hamming, int camera_project, int32_t* matched, int* nmatches_out, const orbg_camera* cam) {
  const ScaleTables st(kf);
  const Grid g = build_grid(kf);
  float T16[16];
  sim3_pose(Scw, T16);
  const Pose pose(T16);
  int nmatches = 0;
  std::vector<int> vIndices;
  for (int iMP = 0; iMP < pts->m; iMP++) {
    if (pts->bad[iMP] || (pts->skip && pts->skip[iMP]) || (already_found && already_found[iMP])) continue;
    const float* p3Dw = pts->pos + 3 * (size_t)iMP;
    float p3Dc[3];
    pose.map(p3Dw, p3Dc);
    if (p3Dc[2] < 0.0) continue;
    float u, v;
    if (cam) {
      float uv[2]; rig_cam_project(*cam, p3Dc, uv); u = uv[0]; v = uv[1];
    } else if (camera_project) {              // Pinhole::project(cv::Point3f)  S/CameraModels/Pinhole.cpp:28-31
      u = kf->fx * p3Dc[0] / p3Dc[2] + kf->cx;
      v = kf->fy * p3Dc[1] / p3Dc[2] + kf->cy;
    } else {                                  // :631-635
      const float invz = 1 / p3Dc[2];
      const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
      u = kf->fx * x + kf->cx;
      v = kf->fy * y + kf->cy;
    }
    if (!(u >= kf->min_x && u < kf->max_x && v >= kf->min_y && v < kf->max_y)) continue;   // KeyFrame::IsInImage
    const float maxDistance = 1.2f * pts->max_dist[iMP], minDistance = 0.8f * pts->min_dist[iMP];
    const float PO[3] = {p3Dw[0] - pose.Ow[0], p3Dw[1] - pose.Ow[1], p3Dw[2] - pose.Ow[2]};
    const float dist = norm3(PO);
    if (dist < minDistance || dist > maxDistance) continue;
    const float* Pn = pts->normal + 3 * (size_t)iMP;
    const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
    if (dot < 0.5 * dist) continue;
    const float ratio = pts->max_dist[iMP] / dist;                      // PredictScale(dist, pKF)  S/MapPoint.cc:629-644
    int lvl = (int)std::ceil(std::log(ratio) / st.log_sf);
    if (lvl < 0) lvl = 0;
    else if (lvl >= kf->n_levels) lvl = kf->n_levels - 1;
    const float radius = th * st.scale[lvl];
    features_in_area(kf, g, u, v, radius, -1, -1, vIndices);            // KeyFrame::GetFeaturesInArea  S/KeyFrame.cc:889-940
    if (vIndices.empty()) continue;
    const uint8_t* dMP = pts->desc + 32 * (size_t)iMP;
    int bestDist = 256, bestIdx = -1;
    for (int idx : vIndices) {
      if (matched[idx] >= 0) continue;
      const int kpLevel = kf->kps[idx].octave;
      if (kpLevel < lvl - 1 || kpLevel > lvl) continue;
      const int d = oracle_hamming(dMP, kf->desc + 32 * (size_t)idx);
      if (d < bestDist) { bestDist = d; bestIdx = idx; }
    }
    if (bestIdx >= 0 && bestDist <= TH_LOW * ratio_hamming) {
      matched[bestIdx] = iMP;
      nmatches++;
    }
  }
  if (nmatches_out) *nmatches_out = nmatches;
  return ORBG_OK;
}

// ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist)
// -- S/ORBmatcher.cc:2188-2310, the relocalisation overload (Tracking::Relocalization, S/Tracking.cc:3372-3410).
// pts = pKF->GetMapPointMatches() flattened feature by feature (m = pKF->N): bad[i] = "no map point, or isBad()" (:2209-2211),
// already_found[i] = sAlreadyFound.count(pMP) (:2211), kf_angle[i] = pKF->mvKeysUn[i].angle (:2261).  assigned_mp[idx] >= 0 means
// CurrentFrame.mvpMapPoints[idx] != NULL on entry (ANY map point blocks the feature here, :2246-2247 -- the other projection
// searches only respect points with observations); on exit newly matched features hold the index i of the keyframe's point.
// Quirks kept: no depth-sign test before Pinhole::project (:2214-2217), bounds inclusive on both sides (:2219-2222), levels
// nPredictedLevel-1 .. nPredictedLevel+1 (:2240), no viewing-angle test, no stereo gate.
static int search_by_projection_reloc_impl(const orbm_frame_view* cur, const float* Tcw_cur, const orbm_worldpoints_view* pts, const uint8_t* already_found,
                                          const float* kf_angle, float th, int orb_dist, int check_ori, int32_t* assigned_mp, int* nmatches_out,
                                          const orbg_camera* cam);
extern "C" int oracle_search_by_projection_reloc(const orbm_frame_view* cur, const float* Tcw_cur, const orbm_worldpoints_view* pts,
                                                 const uint8_t* already_found, const float* kf_angle, float th, int orb_dist,
                                                 int check_ori, int32_t* assigned_mp, int* nmatches_out) {
  return search_by_projection_reloc_impl(cur, Tcw_cur, pts, already_found, kf_angle, th, orb_dist, check_ori, assigned_mp, nmatches_out, nullptr);
}
// ... with CurrentFrame.mpCamera a camera model (a monocular fisheye frame): :2217 projects through it
extern "C" int oracle_search_by_projection_reloc_cam(const orbm_frame_view* cur, const float* Tcw_cur, const orbg_camera* cam, const orbm_worldpoints_view* pts,
                                                     const uint8_t* already_found, const float* kf_angle, float th, int orb_dist,
                                                     int check_ori, int32_t* assigned_mp, int* nmatches_out) {
  return search_by_projection_reloc_impl(cur, Tcw_cur, pts, already_found, kf_angle, th, orb_dist, check_ori, assigned_mp, nmatches_out, cam);
}
static int search_by_projection_reloc_impl(const orbm_frame_view* cur, const float* Tcw_cur, const orbm_worldpoints_view* pts, const uint8_t* already_found,
                                          const float* kf_angle, float th, int orb_dist, int check_ori, int32_t* assigned_mp, int* nmatches_out,
                                          const orbg_camera* cam) {
  const ScaleTables st(cur);
  const Grid g = build_grid(cur);
  const Pose pose(Tcw_cur);
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  std::vector<int> vIndices2;
  for (int i = 0; i < pts->m; i++) {
    if (pts->bad[i] || (pts->skip && pts->skip[i]) || (already_found && already_found[i])) continue;
    const float* x3Dw = pts->pos + 3 * (size_t)i;
    float x3Dc[3];
    pose.map(x3Dw, x3Dc);                                               // Rcw*x3Dw+tcw
    float u = cur->fx * x3Dc[0] / x3Dc[2] + cur->cx;                    // Pinhole::project(cv::Point3f)
    float v = cur->fy * x3Dc[1] / x3Dc[2] + cur->cy;
    if (cam) { float uv[2]; rig_cam_project(*cam, x3Dc, uv); u = uv[0]; v = uv[1]; }
    if (u < cur->min_x || u > cur->max_x) continue;
    if (v < cur->min_y || v > cur->max_y) continue;
    const float PO[3] = {x3Dw[0] - pose.Ow[0], x3Dw[1] - pose.Ow[1], x3Dw[2] - pose.Ow[2]};
    const float dist3D = norm3(PO);
    const float maxDistance = 1.2f * pts->max_dist[i], minDistance = 0.8f * pts->min_dist[i];   // Get{Max,Min}DistanceInvariance
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const float ratio = pts->max_dist[i] / dist3D;                      // PredictScale(dist3D, &CurrentFrame)  S/MapPoint.cc:646-661
    int lvl = (int)std::ceil(std::log(ratio) / st.log_sf);
    if (lvl < 0) lvl = 0;
    else if (lvl >= cur->n_levels) lvl = cur->n_levels - 1;
    const float radius = th * st.scale[lvl];
    features_in_area(cur, g, u, v, radius, lvl - 1, lvl + 1, vIndices2);
    if (vIndices2.empty()) continue;
    const uint8_t* dMP = pts->desc + 32 * (size_t)i;
    int bestDist = 256, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      if (assigned_mp[i2] >= 0) continue;                               // :2246-2247
      const int dist = oracle_hamming(dMP, cur->desc + 32 * (size_t)i2);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= orb_dist && bestIdx2 >= 0) {
      assigned_mp[bestIdx2] = i;
      nmatches++;
      if (check_ori) rotHist[rot_bin(kf_angle[i], cur->kps[bestIdx2].angle)].push_back(bestIdx2);
    }
  }
  if (check_ori) {                                                      // :2284-2306
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int b = 0; b < HISTO_LENGTH; b++)
      if (b != ind1 && b != ind2 && b != ind3)
        for (int idx : rotHist[b]) { assigned_mp[idx] = -1; nmatches--; }
  }
  if (nmatches_out) *nmatches_out = nmatches;
  return ORBG_OK;
}

// ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vpMatches12) -- S/ORBmatcher.cc:819-959.
// kf2 = pKF2 (view + fv2 + mp_valid2 = "has a MapPoint that is not bad"); the pKF1 side comes flattened as descriptors,
// validity, angles and fv1.  matches12[idx1] = idx2 (the adapter stores vpMapPoints2[idx2]) or -1.
extern "C" int oracle_search_by_bow_kf(const orbm_frame_view* kf2, const orbm_featvec_view* fv2, const uint8_t* mp_valid2,
                                       const uint8_t* desc1, int n1, const uint8_t* mp_valid1, const float* angle1,
                                       const orbm_featvec_view* fv1, float nnratio, int check_ori,
                                       int32_t* matches12, int* nmatches_out) {
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  std::vector<uint8_t> vbMatched2(std::max(kf2->n, 1), 0);
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  int a = 0, b = 0;
  while (a < fv1->n_nodes && b < fv2->n_nodes) {
    if (fv1->node_id[a] == fv2->node_id[b]) {
      for (uint32_t i1 = fv1->start[a]; i1 < fv1->start[a + 1]; i1++) {
        const uint32_t idx1 = fv1->feat_idx[i1];
        if (!mp_valid1[idx1]) continue;
        const uint8_t* d1 = desc1 + 32 * (size_t)idx1;
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (uint32_t i2 = fv2->start[b]; i2 < fv2->start[b + 1]; i2++) {
          const uint32_t idx2 = fv2->feat_idx[i2];
          if (vbMatched2[idx2] || !mp_valid2[idx2]) continue;
          const int dist = oracle_hamming(d1, kf2->desc + 32 * (size_t)idx2);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = (int)idx2; }
          else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 < TH_LOW) {
          if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
            matches12[idx1] = bestIdx2;
            vbMatched2[bestIdx2] = 1;
            if (check_ori) rotHist[rot_bin(angle1[idx1], kf2->kps[bestIdx2].angle)].push_back((int)idx1);
            nmatches++;
          }
        }
      }
      a++; b++;
    } else if (fv1->node_id[a] < fv2->node_id[b]) {
      a = (int)(std::lower_bound(fv1->node_id, fv1->node_id + fv1->n_nodes, fv2->node_id[b]) - fv1->node_id);
    } else {
      b = (int)(std::lower_bound(fv2->node_id, fv2->node_id + fv2->n_nodes, fv1->node_id[a]) - fv2->node_id);
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx : rotHist[i]) { matches12[idx] = -1; nmatches--; }
    }
  }
  if (nmatches_out) *nmatches_out = nmatches;
  return ORBG_OK;
}

// Frame::ComputeStereoMatches -- S/Frame.cc:785-963.
// Sub-pixel match by parabola fitting and the disparity test of ComputeStereoMatches -- S/Frame.cc:918-946.  vDists: the 2 L + 1 SAD
// values of the sliding window, bestincR the offset of the smallest (not at either end).  false: the reference `continue`s / stores nothing.
static bool stereo_subpixel(const float* vDists, int L, int bestincR, float scaleduR0, float scale_of_level, float uL, float minD, float maxD,
                            float bf, float* uright, float* depth) {
  const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
  const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
  if (deltaR < -1 || deltaR > 1) return false;
  float bestuR = scale_of_level * ((float)scaleduR0 + (float)bestincR + deltaR);
  float disparity = (uL - bestuR);
  if (disparity >= minD && disparity < maxD) {
    if (disparity <= 0) { disparity = 0.01; bestuR = (float)(uL - 0.01); }   // double literals, :937-941
    *depth = bf / disparity;
    *uright = bestuR;
    return true;
  }
  return false;
}
extern "C" int oracle_stereo_subpixel(const float* dists, int L, int bestincR, float scaleduR0, float scale_of_level, float uL, float minD,
                                      float maxD, float bf, float* uright, float* depth) {
  return stereo_subpixel(dists, L, bestincR, scaleduR0, scale_of_level, uL, minD, maxD, bf, uright, depth) ? 1 : 0;
}

extern "C" int oracle_stereo_match(oracle_extractor* left, oracle_extractor* right,
                                   const orbx_keypoint* kps_l, const uint8_t* desc_l, int N,
                                   const orbx_keypoint* kps_r, const uint8_t* desc_r, int Nr,
                                   float bf, float b, float* uright, float* depth) {
  for (int i = 0; i < N; i++) { uright[i] = -1.0f; depth[i] = -1.0f; }
  const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
  int w0, nRows, s0;
  oracle_level_ptr(left, 0, &w0, &nRows, &s0);
  const std::vector<float>& scaleF = oracle_scale(left);
  const std::vector<float>& invScaleF = oracle_inv_scale(left);
  std::vector<std::vector<int>> vRowIndices(nRows);
  for (int iR = 0; iR < Nr; iR++) {                                            // :802-812
    const float kpY = kps_r[iR].y;
    const float r = 2.0f * scaleF[kps_r[iR].octave];
    const int maxr = (int)std::ceil(kpY + r);
    const int minr = (int)std::floor(kpY - r);
    for (int yi = minr; yi <= maxr; yi++)
      if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);                // clamp: Appendix C-8
  }
  const float minZ = b, minD = 0, maxD = bf / minZ;
  std::vector<std::pair<int, int>> vDistIdx;
  for (int iL = 0; iL < N; iL++) {
    const orbx_keypoint& kpL = kps_l[iL];
    const int levelL = kpL.octave;
    const float vL = kpL.y, uL = kpL.x;
    const int row = (int)vL;
    if (row < 0 || row >= nRows) continue;
    const std::vector<int>& vCandidates = vRowIndices[row];
    if (vCandidates.empty()) continue;
    const float minU = uL - maxD, maxU = uL - minD;
    if (maxU < 0) continue;
    int bestDist = TH_HIGH;
    int bestIdxR = 0;
    const uint8_t* dL = desc_l + 32 * (size_t)iL;
    for (int iR : vCandidates) {
      const orbx_keypoint& kpR = kps_r[iR];
      if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
      const float uR = kpR.x;
      if (uR >= minU && uR <= maxU) {
        const int dist = oracle_hamming(dL, desc_r + 32 * (size_t)iR);
        if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
      }
    }
    if (bestDist < thOrbDist) {                                                // :871-946
      const float uR0 = kps_r[bestIdxR].x;
      const float scaleFactor = invScaleF[kpL.octave];
      const float scaleduL = std::round(kpL.x * scaleFactor);
      const float scaledvL = std::round(kpL.y * scaleFactor);
      const float scaleduR0 = std::round(uR0 * scaleFactor);
      const int w = 5;
      int lw, lh, ls, rw, rh, rs;
      const uint8_t* IL = oracle_level_ptr(left, kpL.octave, &lw, &lh, &ls);
      const uint8_t* IR = oracle_level_ptr(right, kpL.octave, &rw, &rh, &rs);
      const int cyL = (int)scaledvL, cxL = (int)scaleduL;
      int bestDistS = INT_MAX, bestincR = 0;
      const int L = 5;
      float vDists[2 * 5 + 1];
      const float iniu = scaleduR0 + L - w;
      const float endu = scaleduR0 + L + w + 1;
      if (iniu < 0 || endu >= rw) continue;
      const int centreL = IL[(size_t)cyL * ls + cxL];
      for (int incR = -L; incR <= +L; incR++) {
        const int cxR = (int)(scaleduR0 + incR);
        const int centreR = IR[(size_t)cyL * rs + cxR];
        int sad = 0;
        for (int dy = -w; dy <= w; dy++)
          for (int dx = -w; dx <= w; dx++) {
            int a = (int)IL[(size_t)(cyL + dy) * ls + cxL + dx] - centreL;     // CV_16S, IL - IL(w,w)
            int c = (int)IR[(size_t)(cyL + dy) * rs + cxR + dx] - centreR;
            sad += std::abs(a - c);
          }
        float dist = (float)sad;                                               // cv::norm(IL,IR,NORM_L1)
        if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
        vDists[L + incR] = dist;
      }
      if (bestincR == -L || bestincR == L) continue;
      float ur_sub, depth_sub;
      if (stereo_subpixel(vDists, L, bestincR, scaleduR0, scaleF[kpL.octave], uL, minD, maxD, bf, &ur_sub, &depth_sub)) {
        depth[iL] = depth_sub;
        uright[iL] = ur_sub;
        vDistIdx.push_back({bestDistS, iL});
      }
    }
  }
  if (vDistIdx.empty()) return ORBG_OK;   // reference indexes an empty vector here (UB); pinned to "no matches"
  std::sort(vDistIdx.begin(), vDistIdx.end());                                 // :949-962
  const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
  const float thDist = 1.5f * 1.4f * median;
  for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
    if (vDistIdx[i].first < thDist) break;
    uright[vDistIdx[i].second] = -1;
    depth[vDistIdx[i].second] = -1;
  }
  return ORBG_OK;
}
