// Synthetic agent shared by tests/cpp/dropin_parity.cpp and tests/cpp/dropin_bench.cpp: a textured plane seen by a moving stereo
// rig, Frames built through the extractor adapter, map points as LocalMapping creates them, and a local-BA window -- all held in
// the header-only mocks of Frame / KeyFrame / MapPoint / Map (mock_orbslam3.hpp).
#pragma once
#include <cmath>
#include <cstdio>
#include <memory>
#include <vector>

#include "mock_orbslam3.hpp"
#include "orbgpu_dropin.hpp"

using namespace mock;

// ------------------------------------------------------------------------------------------------ synthetic agent
static unsigned g_seed = 1;
static unsigned rnd() { g_seed = g_seed * 1664525u + 1013904223u; return g_seed >> 8; }
static double urand() { return (rnd() & 0xFFFFFF) / double(0x1000000); }
static double nrand() { double a = urand() + 1e-12, b = urand(); return std::sqrt(-2 * std::log(a)) * std::cos(6.283185307179586 * b); }

static const int W = 640, H = 480, TW = 1600, TH = 1200;
static const float FX = 458.654f * 640 / 752, CX = 320.f, CY = 240.f, BF = 47.90639384423901f * 640 / 752, BB = BF / FX;

static std::vector<uint8_t> make_texture() {
  std::vector<float> t((size_t)TW * TH, 110.f);
  for (int cell = 128; cell >= 4; cell /= 2) {                    // value noise octaves
    const int gw = TW / cell + 2, gh = TH / cell + 2;
    std::vector<float> g((size_t)gw * gh);
    for (auto& x : g) x = (float)(urand() - 0.5) * cell * 0.9f;
    for (int y = 0; y < TH; y++)
      for (int x = 0; x < TW; x++) {
        const int gx = x / cell, gy = y / cell; const float fx = (x % cell) / (float)cell, fy = (y % cell) / (float)cell;
        t[(size_t)y * TW + x] += (g[gy * gw + gx] * (1 - fx) + g[gy * gw + gx + 1] * fx) * (1 - fy) + (g[(gy + 1) * gw + gx] * (1 - fx) + g[(gy + 1) * gw + gx + 1] * fx) * fy;
      }
  }
  for (int k = 0; k < 700; k++) {                                // dark / bright rectangles: plenty of FAST corners
    const int x0 = rnd() % (TW - 60), y0 = rnd() % (TH - 60), w = 8 + rnd() % 50, h = 8 + rnd() % 50; const float val = (rnd() & 1) ? 225.f : 30.f;
    for (int y = y0; y < y0 + h; y++) for (int x = x0; x < x0 + w; x++) t[(size_t)y * TW + x] = val + (float)((x * 7 + y * 13) % 9);
  }
  std::vector<uint8_t> out(t.size());
  for (size_t i = 0; i < t.size(); i++) out[i] = (uint8_t)std::min(255.f, std::max(0.f, t[i]));
  return out;
}

static void rot(double rx, double ry, double rz, double R[9]) {
  const double cx = std::cos(rx), sx = std::sin(rx), cy = std::cos(ry), sy = std::sin(ry), cz = std::cos(rz), sz = std::sin(rz);
  const double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  double T[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T[3 * i + j] = 0; for (int k = 0; k < 3; k++) T[3 * i + j] += Ry[3 * i + k] * Rx[3 * k + j]; }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { R[3 * i + j] = 0; for (int k = 0; k < 3; k++) R[3 * i + j] += Rz[3 * i + k] * T[3 * k + j]; }
}
static void pose_of(int k, double T[16]) {                       // left camera: drift + tilt over the plane z = 0
  const double t = 0.02 * k; double R[9];
  rot(0.32 + 0.04 * std::sin(0.7 * t + 0.3), -0.22 + 0.05 * std::sin(0.5 * t), 0.03 * std::sin(0.9 * t), R);
  const double C[3] = {3.75 + 0.6 * t, 3.0 + 0.15 * std::sin(0.8 * t), -2.8 - 0.1 * std::sin(0.6 * t)};
  for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0) ? 1 : 0;
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T[4 * i + j] = R[3 * i + j]; T[4 * i + 3] = -(R[3 * i] * C[0] + R[3 * i + 1] * C[1] + R[3 * i + 2] * C[2]); }
}
static std::vector<uint8_t> render(const std::vector<uint8_t>& tex, const double T[16]) {
  std::vector<uint8_t> im((size_t)W * H);
  double Ow[3];
  for (int i = 0; i < 3; i++) Ow[i] = -(T[i] * T[3] + T[4 + i] * T[7] + T[8 + i] * T[11]);
  for (int v = 0; v < H; v++)
    for (int u = 0; u < W; u++) {
      const double d[3] = {(u - CX) / FX, (v - CY) / FX, 1.0};
      double dw[3];
      for (int i = 0; i < 3; i++) dw[i] = T[i] * d[0] + T[4 + i] * d[1] + T[8 + i] * d[2];
      const double s = -Ow[2] / dw[2];
      const double X = (Ow[0] + s * dw[0]) * 200.0, Y = (Ow[1] + s * dw[1]) * 200.0;
      const double tx = std::min(std::max(X, 0.0), TW - 1.001), ty = std::min(std::max(Y, 0.0), TH - 1.001);
      const int x0 = (int)tx, y0 = (int)ty; const double fx = tx - x0, fy = ty - y0;
      const double val = (tex[(size_t)y0 * TW + x0] * (1 - fx) + tex[(size_t)y0 * TW + x0 + 1] * fx) * (1 - fy) +
                         (tex[(size_t)(y0 + 1) * TW + x0] * (1 - fx) + tex[(size_t)(y0 + 1) * TW + x0 + 1] * fx) * fy;
      im[(size_t)v * W + u] = (uint8_t)std::lrint(std::min(255.0, std::max(0.0, val)));
    }
  return im;
}

static Mat mat44(const double T[16]) { Mat m(4, 4, 4); for (int i = 0; i < 16; i++) m.ptr<float>(0)[i] = (float)T[i]; return m; }

struct Agent {          // everything one run (one Ops) owns: frames, map points, keyframes
  std::vector<std::unique_ptr<GeometricCamera>> cameras;           // (rig scenes) mpCamera / mpCamera2 of every keyframe and frame
  std::vector<std::unique_ptr<orbgpu::FrameOnDevice>> dev_frames;
  std::vector<std::unique_ptr<Frame>> frames;
  std::vector<std::unique_ptr<MapPoint>> points;
  std::vector<std::unique_ptr<KeyFrame>> kfs;
  Map map;
};

// Frame::Frame(stereo) through the adapter: extraction L+R, ComputeStereoMatches, grid -- and the host copies the mocks hold
// keep_on_device: the Frame keeps the device-resident copy its constructor produced (mpGpuFrame) -- valid until `rig` extracts again
static void make_frame(Agent& A, orbgpu::ORBextractor& rig, const std::vector<uint8_t>& tex, int k, bool keep_on_device = false) {
  double T[16], Tr[16]; pose_of(k, T);
  for (int i = 0; i < 16; i++) Tr[i] = T[i];
  Tr[3] -= BB;
  const std::vector<uint8_t> L = render(tex, T), R = render(tex, Tr);
  std::unique_ptr<Frame> F(new Frame);
  F->mnMinX = 0; F->mnMaxX = W; F->mnMinY = 0; F->mnMaxY = H; F->fx = FX; F->fy = FX; F->cx = CX; F->cy = CY; F->mbf = BF; F->mb = BB;
  orbm_frame_view v{0, nullptr, nullptr, nullptr, nullptr, 0, (float)W, 0, (float)H, FX, FX, CX, CY, BF, BB, 8, 1.2f};
  std::unique_ptr<orbgpu::FrameOnDevice> devp(new orbgpu::FrameOnDevice(4096));
  orbgpu::FrameOnDevice& dev = *devp;
  std::vector<orbx_keypoint> keys; std::vector<uint8_t> desc; std::vector<float> ur, dp;
  const int N = dev.StereoCtor(rig, v, L.data(), R.data(), W, H, W, &keys, &desc, &ur, &dp);
  if (keep_on_device) { F->mpGpuFrame = devp.get(); A.dev_frames.push_back(std::move(devp)); }
  F->N = N; F->mvKeys.resize(N); F->mDescriptors = Mat(N, 32, 1); F->mvuRight = ur; F->mvDepth = dp;
  for (int i = 0; i < N; i++) F->mvKeys[i] = KeyPoint{{keys[i].x, keys[i].y}, keys[i].size, keys[i].angle, keys[i].response, keys[i].octave};
  F->mvKeysUn = F->mvKeys;                                        // k1 == 0: no undistortion (S/Frame.cc:723-727)
  std::memcpy(F->mDescriptors.ptr<uint8_t>(0), desc.data(), desc.size());
  F->mvpMapPoints.assign(N, nullptr); F->mvbOutlier.assign(N, false);
  F->mvInvLevelSigma2 = rig.GetInverseScaleSigmaSquares();
  F->mTcw = mat44(T);
  { int ns = 0; for (int i = 0; i < N; i++) ns += dp[i] > 0; std::fprintf(stderr, "frame %d: %d keypoints, %d with stereo depth\n", k, N, ns); }
  for (int i = 0; i < N; i++) F->mFeatVec[(desc[32 * (size_t)i] | (desc[32 * (size_t)i + 1] << 8)) & 0x3FF].push_back(i);   // stand-in vocabulary: 10 descriptor bits = node id
  A.frames.push_back(std::move(F));
}

// map points as LocalMapping creates them from a stereo keyframe (S/LocalMapping.cc CreateNewMapPoints / MapPoint::UpdateNormalAndDepth)
static void make_points_from(Agent& A, const Frame& F, std::vector<MapPoint*>& out, std::vector<int>& feat_of) {
  const float* T = F.mTcw.ptr<float>(0);
  float Ow[3];
  for (int i = 0; i < 3; i++) Ow[i] = -(T[i] * T[3] + T[4 + i] * T[7] + T[8 + i] * T[11]);
  float sf[8]; sf[0] = 1.f; for (int l = 1; l < 8; l++) sf[l] = sf[l - 1] * 1.2f;
  for (int i = 0; i < F.N; i++) {
    const float z = F.mvDepth[i];
    if (!(z > 0)) continue;
    const float xc = (F.mvKeysUn[i].pt.x - CX) * z / FX, yc = (F.mvKeysUn[i].pt.y - CY) * z / FX;
    const float Pc[3] = {xc - T[3], yc - T[7], z - T[11]};
    std::unique_ptr<MapPoint> p(new MapPoint);
    float* X = p->mWorldPos.ptr<float>(0);
    for (int a = 0; a < 3; a++) X[a] = T[a] * Pc[0] + T[4 + a] * Pc[1] + T[8 + a] * Pc[2];
    float PO[3] = {X[0] - Ow[0], X[1] - Ow[1], X[2] - Ow[2]};
    const float dist = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
    for (int a = 0; a < 3; a++) p->mNormalVector.ptr<float>(0)[a] = PO[a] / dist;
    p->mfMaxDistance = dist * sf[F.mvKeysUn[i].octave]; p->mfMinDistance = p->mfMaxDistance / sf[7];
    std::memcpy(p->mDescriptor.ptr<uint8_t>(0), F.mDescriptors.ptr<uint8_t>(i), 32);
    p->mnId = A.points.size(); p->nObs = 3; p->mpMap = &A.map;
    out.push_back(p.get()); feat_of.push_back(i);
    A.points.push_back(std::move(p));
  }
}

// The window Optimizer::LocalBundleAdjustment works on: n_local covisible keyframes (the last one is the new keyframe), n_far older
// ones that only observe shared points (they become the fixed cameras), n_pts points seen by runs of 3-7 keyframes, pixel noise,
// gross outliers.  Returns the new keyframe.
static KeyFrame* build_lba_scene(Agent& A, int n_local, int n_far, int n_pts, double outlier_frac, unsigned seed) {
  g_seed = seed;
  const int P = n_local + n_far;
  std::vector<std::vector<double>> Tt(P, std::vector<double>(16));
  float isig[8]; { float s = 1.f; for (int l = 0; l < 8; l++) { isig[l] = 1.f / (s * s); s *= 1.2f; } }
  for (int k = 0; k < P; k++) {
    double R[9]; rot(0.02 * std::sin(0.3 * k), 0.03 * std::sin(0.2 * k + 1.0), 0.01 * std::sin(0.5 * k), R);
    const double C[3] = {0.12 * k, 0.02 * std::sin(0.4 * k), 0.03 * std::cos(0.3 * k)};
    for (int i = 0; i < 16; i++) Tt[k][i] = (i % 5 == 0) ? 1 : 0;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Tt[k][4 * i + j] = R[3 * i + j]; Tt[k][4 * i + 3] = -(R[3 * i] * C[0] + R[3 * i + 1] * C[1] + R[3 * i + 2] * C[2]); }
    std::unique_ptr<KeyFrame> kf(new KeyFrame);
    kf->mnId = 10 + k; kf->fx = FX; kf->fy = FX; kf->cx = CX; kf->cy = CY; kf->mbf = BF; kf->mpMap = &A.map;
    kf->mvInvLevelSigma2.assign(isig, isig + 8);
    double Tn[16]; for (int i = 0; i < 16; i++) Tn[i] = Tt[k][i];
    Tn[3] += 0.01 * nrand(); Tn[7] += 0.01 * nrand(); Tn[11] += 0.01 * nrand();       // initial estimate: truth + 1 cm
    kf->Tcw = mat44(Tn);
    A.kfs.push_back(std::move(kf));
  }
  A.map.mnInitKFid = 0;                                            // the initial keyframe is not in the window
  KeyFrame* cur = A.kfs[P - 1].get();                              // the new keyframe: covisible with the n_local - 1 before it
  // (covisibility order: ascending ids here -- with descending ids the reference's "two lowest ids" scan, S/Optimizer.cc:1886-1899,
  //  never finds a second keyframe and reads an uninitialised pointer; the glue then fixes only one)
  for (int k = n_far; k <= P - 2; k++) cur->mvpOrderedConnectedKeyFrames.push_back(A.kfs[k].get());
  for (int j = 0; j < n_pts; j++) {
    const int nobs = std::min(3 + (int)(rnd() % 5), P), k0 = rnd() % (P - nobs + 1), kc = std::min(k0 + nobs / 2, P - 1);
    const double z = 2.0 + 7.0 * urand(), u = 60 + (W - 120) * urand(), v = 60 + (H - 120) * urand();
    const double Pc[3] = {(u - CX) * z / FX, (v - CY) * z / FX, z};
    const double* T = Tt[kc].data();
    double Xw[3];
    for (int a = 0; a < 3; a++) Xw[a] = T[a] * (Pc[0] - T[3]) + T[4 + a] * (Pc[1] - T[7]) + T[8 + a] * (Pc[2] - T[11]);
    std::unique_ptr<MapPoint> mp(new MapPoint);
    mp->mnId = 100 + j; mp->mpMap = &A.map;
    for (int a = 0; a < 3; a++) mp->mWorldPos.ptr<float>(0)[a] = (float)(Xw[a] + 0.02 * nrand());
    for (int k = k0; k < k0 + nobs; k++) {
      const double* Tk = Tt[k].data();
      double Xc[3];
      for (int a = 0; a < 3; a++) Xc[a] = Tk[4 * a] * Xw[0] + Tk[4 * a + 1] * Xw[1] + Tk[4 * a + 2] * Xw[2] + Tk[4 * a + 3];
      if (Xc[2] < 0.3) continue;
      const int oct = rnd() % 4; const double sig = std::pow(1.2, oct);
      double pu = FX * Xc[0] / Xc[2] + CX + sig * nrand(), pv = FX * Xc[1] / Xc[2] + CY + sig * nrand();
      if (urand() < outlier_frac) { pu += (urand() < 0.5 ? -1 : 1) * (15 + 20 * urand()); pv += (urand() < 0.5 ? -1 : 1) * (15 + 20 * urand()); }
      KeyFrame* kf = A.kfs[k].get();
      const int li = (int)kf->mvKeysUn.size();
      kf->mvKeysUn.push_back(KeyPoint{{(float)pu, (float)pv}, 31.f, 0.f, 20.f, oct});
      kf->mvuRight.push_back((rnd() % 10 == 0) ? -1.f : (float)(pu - BF / Xc[2] + sig * nrand()));     // one in ten observations is monocular
      kf->mvpMapPoints.push_back(mp.get());
      mp->mObservations[kf] = std::make_tuple(li, -1);
      mp->nObs++;
    }
    A.points.push_back(std::move(mp));
  }
  return cur;
}

// ------------------------------------------------------------------------------------------------ two-fisheye rig (mpCamera2 != NULL)
// TUM-VI-like 512 x 512 KannalaBrandt8 cameras, the right one 10 cm to the side and a degree out of line (mTrl)
static const float KB8_L[8] = {190.978f, 190.973f, 254.932f, 256.897f, 0.00348f, 0.000715f, -0.00205f, 0.000203f};
static const float KB8_R[8] = {190.442f, 190.435f, 252.598f, 254.917f, 0.00340f, 0.00177f, -0.00266f, 0.000330f};
static void kb8_project(const float* c, const double X[3], double uv[2]) {
  const double r = std::sqrt(X[0] * X[0] + X[1] * X[1]), th = std::atan2(r, X[2]), psi = std::atan2(X[1], X[0]);
  const double t2 = th * th, d = th * (1 + t2 * (c[4] + t2 * (c[5] + t2 * (c[6] + t2 * c[7]))));
  uv[0] = c[0] * d * std::cos(psi) + c[2]; uv[1] = c[1] * d * std::sin(psi) + c[3];
}
static void rig_Trl(double T[12]) {
  double R[9]; rot(0.012, -0.02, 0.006, R);
  const double t[3] = {-0.101, 0.0012, -0.0009};
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T[4 * i + j] = R[3 * i + j]; T[4 * i + 3] = t[i]; }
}
static void give_rig(Agent& A, GeometricCamera*& c1, GeometricCamera*& c2, Mat& mTrl) {
  if (A.cameras.empty()) {
    A.cameras.emplace_back(new GeometricCamera(1u, std::vector<float>(KB8_L, KB8_L + 8)));
    A.cameras.emplace_back(new GeometricCamera(1u, std::vector<float>(KB8_R, KB8_R + 8)));
  }
  c1 = A.cameras[0].get(); c2 = A.cameras[1].get();
  double T[12]; rig_Trl(T);
  mTrl = Mat(3, 4, 4);
  for (int i = 0; i < 12; i++) mTrl.ptr<float>(0)[i] = (float)T[i];
}
constexpr int kRigNLeft = 4096;           // KeyFrame::NLeft of the rig scenes: right-camera feature r has index NLeft + r in the observations

// build_lba_scene for keyframes of the two-fisheye rig: every observation of a point by a keyframe is made by the left camera
// (mvKeysUn, mvuRight = -1), by the right one (mvKeysRight, index NLeft + r in the point's observation tuple) or by both
// (S/Optimizer.cc:2021-2120 makes one edge per camera).
static KeyFrame* build_lba_rig_scene(Agent& A, int n_local, int n_far, int n_pts, double outlier_frac, unsigned seed) {
  g_seed = seed;
  const int P = n_local + n_far;
  std::vector<std::vector<double>> Tt(P, std::vector<double>(16));
  float isig[8]; { float s = 1.f; for (int l = 0; l < 8; l++) { isig[l] = 1.f / (s * s); s *= 1.2f; } }
  double Trl[12]; rig_Trl(Trl);
  for (int k = 0; k < P; k++) {
    double R[9]; rot(0.03 * std::sin(0.3 * k), 0.05 * std::sin(0.2 * k + 1.0), 0.02 * std::sin(0.5 * k), R);
    const double C[3] = {0.10 * k, 0.02 * std::sin(0.4 * k), 0.03 * std::cos(0.3 * k)};
    for (int i = 0; i < 16; i++) Tt[k][i] = (i % 5 == 0) ? 1 : 0;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Tt[k][4 * i + j] = R[3 * i + j]; Tt[k][4 * i + 3] = -(R[3 * i] * C[0] + R[3 * i + 1] * C[1] + R[3 * i + 2] * C[2]); }
    std::unique_ptr<KeyFrame> kf(new KeyFrame);
    kf->mnId = 10 + k; kf->fx = KB8_L[0]; kf->fy = KB8_L[1]; kf->cx = KB8_L[2]; kf->cy = KB8_L[3]; kf->mbf = 0.f; kf->mpMap = &A.map;
    kf->mvInvLevelSigma2.assign(isig, isig + 8);
    give_rig(A, kf->mpCamera, kf->mpCamera2, kf->mTrl);
    kf->NLeft = kRigNLeft;
    double Tn[16]; for (int i = 0; i < 16; i++) Tn[i] = Tt[k][i];
    Tn[3] += 0.01 * nrand(); Tn[7] += 0.01 * nrand(); Tn[11] += 0.01 * nrand();
    kf->Tcw = mat44(Tn);
    A.kfs.push_back(std::move(kf));
  }
  A.map.mnInitKFid = 0;
  KeyFrame* cur = A.kfs[P - 1].get();
  for (int k = n_far; k <= P - 2; k++) cur->mvpOrderedConnectedKeyFrames.push_back(A.kfs[k].get());
  for (int j = 0; j < n_pts; j++) {
    const int nobs = std::min(3 + (int)(rnd() % 5), P), k0 = rnd() % (P - nobs + 1), kc = std::min(k0 + nobs / 2, P - 1);
    const double depth = 1.5 + 6.5 * urand(), th = 0.9 * urand(), psi = 6.283185307179586 * urand();
    const double Pc[3] = {depth * std::sin(th) * std::cos(psi), depth * std::sin(th) * std::sin(psi), depth * std::cos(th)};
    const double* T = Tt[kc].data();
    double Xw[3];
    for (int a = 0; a < 3; a++) Xw[a] = T[a] * (Pc[0] - T[3]) + T[4 + a] * (Pc[1] - T[7]) + T[8 + a] * (Pc[2] - T[11]);
    std::unique_ptr<MapPoint> mp(new MapPoint);
    mp->mnId = 100 + j; mp->mpMap = &A.map;
    for (int a = 0; a < 3; a++) mp->mWorldPos.ptr<float>(0)[a] = (float)(Xw[a] + 0.02 * nrand());
    for (int k = k0; k < k0 + nobs; k++) {
      const double* Tk = Tt[k].data();
      double Xl[3], Xr[3];
      for (int a = 0; a < 3; a++) Xl[a] = Tk[4 * a] * Xw[0] + Tk[4 * a + 1] * Xw[1] + Tk[4 * a + 2] * Xw[2] + Tk[4 * a + 3];
      for (int a = 0; a < 3; a++) Xr[a] = Trl[4 * a] * Xl[0] + Trl[4 * a + 1] * Xl[1] + Trl[4 * a + 2] * Xl[2] + Trl[4 * a + 3];
      KeyFrame* kf = A.kfs[k].get();
      int li = -1, ri = -1;
      for (int side = 0; side < 2; side++) {
        const double* Xc = side ? Xr : Xl;
        if (Xc[2] < 0.2 || urand() > (side ? 0.6 : 0.9)) continue;
        double uv[2]; kb8_project(side ? KB8_R : KB8_L, Xc, uv);
        if (!(uv[0] >= 5 && uv[0] < 507 && uv[1] >= 5 && uv[1] < 507)) continue;
        const int oct = rnd() % 4; const double sig = std::pow(1.2, oct);
        double pu = uv[0] + sig * nrand(), pv = uv[1] + sig * nrand();
        if (urand() < outlier_frac) { pu += (urand() < 0.5 ? -1 : 1) * (15 + 20 * urand()); pv += (urand() < 0.5 ? -1 : 1) * (15 + 20 * urand()); }
        const KeyPoint kp{{(float)pu, (float)pv}, 31.f, 0.f, 20.f, oct};
        if (!side) { li = (int)kf->mvKeysUn.size(); kf->mvKeysUn.push_back(kp); kf->mvuRight.push_back(-1.f); }
        else { ri = kRigNLeft + (int)kf->mvKeysRight.size(); kf->mvKeysRight.push_back(kp); }
      }
      if (li < 0 && ri < 0) continue;
      kf->mvpMapPoints.push_back(mp.get());
      mp->mObservations[kf] = std::make_tuple(li, ri);
      mp->nObs++;
    }
    A.points.push_back(std::move(mp));
  }
  return cur;
}

// A Frame of the two-fisheye rig with n_left + n_right tracked map points (features i < Nleft: mvKeys, the others: mvKeysRight,
// S/Optimizer.cc:1085-1151), pixel noise, gross outliers, mTcw = truth perturbed.
static Frame* build_rig_frame(Agent& A, int n_left, int n_right, double outlier_frac, unsigned seed) {
  g_seed = seed;
  std::unique_ptr<Frame> F(new Frame);
  double R[9]; rot(0.05, -0.03, 0.02, R);
  double T[16]; for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0) ? 1 : 0;
  const double t[3] = {0.1, -0.05, 0.2};
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T[4 * i + j] = R[3 * i + j]; T[4 * i + 3] = t[i]; }
  double Trl[12]; rig_Trl(Trl);
  give_rig(A, F->mpCamera, F->mpCamera2, F->mTrl);
  F->Nleft = n_left; F->Nright = n_right; F->N = n_left + n_right;
  float isig[8]; { float s = 1.f; for (int l = 0; l < 8; l++) { isig[l] = 1.f / (s * s); s *= 1.2f; } }
  F->mvInvLevelSigma2.assign(isig, isig + 8);
  F->mvpMapPoints.assign(F->N, nullptr); F->mvbOutlier.assign(F->N, false); F->mvuRight.assign(F->N, -1.f);
  for (int i = 0; i < F->N; i++) {
    const bool right = i >= n_left;
    const double depth = 1.5 + 6.5 * urand(), th = 0.9 * urand(), psi = 6.283185307179586 * urand();
    const double Pc[3] = {depth * std::sin(th) * std::cos(psi), depth * std::sin(th) * std::sin(psi), depth * std::cos(th)};   // in the observing camera
    double Xl[3];
    if (right) { for (int a = 0; a < 3; a++) Xl[a] = Trl[a] * (Pc[0] - Trl[3]) + Trl[4 + a] * (Pc[1] - Trl[7]) + Trl[8 + a] * (Pc[2] - Trl[11]); }
    else { for (int a = 0; a < 3; a++) Xl[a] = Pc[a]; }
    double Xw[3];
    for (int a = 0; a < 3; a++) Xw[a] = T[a] * (Xl[0] - T[3]) + T[4 + a] * (Xl[1] - T[7]) + T[8 + a] * (Xl[2] - T[11]);
    double uv[2]; kb8_project(right ? KB8_R : KB8_L, Pc, uv);
    const int oct = rnd() % 8; const double sig = std::pow(1.2, oct);
    double pu = uv[0] + sig * nrand(), pv = uv[1] + sig * nrand();
    if (urand() < outlier_frac) { pu += (urand() < 0.5 ? -25 : 25); pv += (urand() < 0.5 ? -25 : 25); }
    const KeyPoint kp{{(float)pu, (float)pv}, 31.f, 0.f, 20.f, oct};
    if (right) F->mvKeysRight.push_back(kp); else F->mvKeys.push_back(kp);
    if (i % 7 == 3) continue;                                    // (features without a map point)
    std::unique_ptr<MapPoint> mp(new MapPoint);
    mp->mnId = 100 + i; mp->mpMap = &A.map;
    for (int a = 0; a < 3; a++) mp->mWorldPos.ptr<float>(0)[a] = (float)Xw[a];
    F->mvpMapPoints[i] = mp.get();
    A.points.push_back(std::move(mp));
  }
  F->mvKeysUn = F->mvKeys;
  double Tn[16]; for (int i = 0; i < 16; i++) Tn[i] = T[i];
  { double dR[9], Rn[9]; rot(0.008, -0.006, 0.005, dR);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Rn[3 * i + j] = 0; for (int k = 0; k < 3; k++) Rn[3 * i + j] += dR[3 * i + k] * R[3 * k + j]; }
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Tn[4 * i + j] = Rn[3 * i + j]; Tn[4 * i + 3] = t[i] + 0.02 * nrand(); } }
  F->mTcw = mat44(Tn);
  Frame* out = F.get();
  A.frames.push_back(std::move(F));
  return out;
}

// A tracking scene of the two-fisheye rig for the matcher's two-camera forms (Frame::isInFrustum :545-554, SearchByProjection(Frame,
// MapPoints) :44-214, SearchByProjection(CurrentFrame, LastFrame) :1970-2186): a local map, a current Frame whose two cameras see its
// points at noisy positions with descriptors a few bits away (near-twins, distractors, stereo partners, features that already hold a
// point, points seen in this frame already, bad points), and a two-camera last frame holding some of the points.
struct RigTrack { Frame* cur = nullptr; Frame* last = nullptr; std::vector<MapPoint*> local; KeyFrame* ref = nullptr; };
static RigTrack build_rig_track_scene(Agent& A, unsigned seed, int n_points = 1200, int n_distract = 250, double dz_last = 0.02) {
  g_seed = seed;
  RigTrack out;
  const float size = 512.f;
  std::unique_ptr<Frame> F(new Frame);
  F->mnId = 7;
  double R[9]; rot(0.04, -0.06, 0.03, R);
  const double t[3] = {0.2, -0.1, 0.15};
  double T[16]; for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0) ? 1 : 0;
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T[4 * i + j] = R[3 * i + j]; T[4 * i + 3] = t[i]; }
  F->mTcw = mat44(T);
  double Ow[3]; for (int i = 0; i < 3; i++) Ow[i] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);
  double Trl[12]; rig_Trl(Trl);
  give_rig(A, F->mpCamera, F->mpCamera2, F->mTrl);
  // mTlr = mTrl^-1 (in the reference mTlr is the setting and mTrl derived from it, S/Frame.cc:1073-1079)
  F->mTlr = Mat(3, 4, 4);
  for (int i = 0; i < 3; i++) {
    double tt = 0;
    for (int j = 0; j < 3; j++) { F->mTlr.ptr<float>(0)[4 * i + j] = (float)Trl[4 * j + i]; tt -= Trl[4 * j + i] * Trl[4 * j + 3]; }
    F->mTlr.ptr<float>(0)[4 * i + 3] = (float)tt;
  }
  F->mnMinX = 0; F->mnMaxX = size; F->mnMinY = 0; F->mnMaxY = size; F->fx = KB8_L[0]; F->fy = KB8_L[1]; F->cx = KB8_L[2]; F->cy = KB8_L[3]; F->mbf = 0; F->mb = 0.1f;
  struct Feat { float x, y; int oct; float angle; uint8_t d[32]; int point; };
  std::vector<Feat> fl, fr;
  auto noisy = [&](const uint8_t* d, int bits, uint8_t* o) { std::memcpy(o, d, 32); for (int b = 0; b < bits; b++) { const int k = rnd() % 256; o[k >> 3] ^= (uint8_t)(1u << (k & 7)); } };
  std::vector<double> base_angle(n_points); std::vector<int> lvl(n_points);
  for (int i = 0; i < n_points; i++) {
    const bool via_right = urand() < 0.5;
    const float* cam = via_right ? KB8_R : KB8_L;
    const double u = -40 + (size + 80) * urand(), v = -40 + (size + 80) * urand(), depth = 1.5 + 7.5 * urand();
    const double mx = (u - cam[2]) / cam[0], my = (v - cam[3]) / cam[1], th = std::sqrt(mx * mx + my * my), psi = std::atan2(my, mx);
    const double Pc[3] = {depth * std::sin(th) * std::cos(psi), depth * std::sin(th) * std::sin(psi), depth * std::cos(th)};
    double Xl[3];
    if (via_right) { for (int a = 0; a < 3; a++) Xl[a] = Trl[a] * (Pc[0] - Trl[3]) + Trl[4 + a] * (Pc[1] - Trl[7]) + Trl[8 + a] * (Pc[2] - Trl[11]); }
    else { for (int a = 0; a < 3; a++) Xl[a] = Pc[a]; }
    double Xw[3]; for (int a = 0; a < 3; a++) Xw[a] = R[a] * (Xl[0] - t[0]) + R[3 + a] * (Xl[1] - t[1]) + R[6 + a] * (Xl[2] - t[2]);
    std::unique_ptr<MapPoint> mp(new MapPoint);
    mp->mnId = 1000 + i; mp->mpMap = &A.map;
    double d[3] = {Xw[0] - Ow[0], Xw[1] - Ow[1], Xw[2] - Ow[2]};
    const double dist = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const double spread = urand() < 0.08 ? 0.9 : 0.15;
    double nv[3], nn = 0; for (int a = 0; a < 3; a++) { nv[a] = d[a] / dist + spread * nrand(); nn += nv[a] * nv[a]; }
    lvl[i] = rnd() % 8; base_angle[i] = 360 * urand();
    for (int a = 0; a < 3; a++) { mp->mWorldPos.ptr<float>(0)[a] = (float)Xw[a]; mp->mNormalVector.ptr<float>(0)[a] = (float)(nv[a] / std::sqrt(nn)); }
    const double maxd = dist * std::pow(1.2, lvl[i] - (0.25 + 0.5 * urand())) * (urand() < 0.03 ? 3.0 : 1.0);
    mp->mfMaxDistance = (float)maxd; mp->mfMinDistance = (float)(maxd / std::pow(1.2, 7) * (urand() < 0.03 ? 40.0 : 1.0));
    for (int b = 0; b < 32; b++) mp->mDescriptor.ptr<uint8_t>(0)[b] = (uint8_t)rnd();
    mp->nObs = urand() < 0.05 ? 0 : 1 + (int)(rnd() % 8);
    if (urand() < 0.04) mp->mbBad = true;
    // the two cameras' features of the point
    const double Xc[3] = {Xl[0], Xl[1], Xl[2]};
    double Xr[3]; for (int a = 0; a < 3; a++) Xr[a] = Trl[4 * a] * Xc[0] + Trl[4 * a + 1] * Xc[1] + Trl[4 * a + 2] * Xc[2] + Trl[4 * a + 3];
    for (int side = 0; side < 2; side++) {
      const double* X = side ? Xr : Xc;
      if (X[2] <= 0.05 || urand() < 0.25) continue;
      double uv[2]; kb8_project(side ? KB8_R : KB8_L, X, uv);
      const double sig = urand() < 0.85 ? 1.2 : 9.0;
      uv[0] += sig * nrand(); uv[1] += sig * nrand();
      if (!(uv[0] > 2 && uv[0] < size - 2 && uv[1] > 2 && uv[1] < size - 2)) continue;
      Feat f; f.x = (float)uv[0]; f.y = (float)uv[1]; f.oct = std::max(0, lvl[i] - (urand() < 0.3 ? 1 : 0)); f.point = i;
      f.angle = (float)std::fmod(base_angle[i] + 2 * nrand() + (urand() < 0.05 ? 140.0 : 0.0) + 720.0, 360.0);
      noisy(mp->mDescriptor.ptr<uint8_t>(0), (int)(rnd() % 45), f.d);
      (side ? fr : fl).push_back(f);
      if (urand() < 0.15) { Feat g = f; g.x += (float)nrand(); g.y += (float)nrand(); g.point = -1; if (urand() > 0.6) g.oct = std::max(g.oct - 1, 0);
                            noisy(mp->mDescriptor.ptr<uint8_t>(0), 5 + (int)(rnd() % 45), g.d); g.angle = (float)(360 * urand()); (side ? fr : fl).push_back(g); }
    }
    out.local.push_back(mp.get());
    A.points.push_back(std::move(mp));
  }
  for (int side = 0; side < 2; side++) {
    std::vector<Feat>& v = side ? fr : fl;
    for (int k = 0; k < n_distract; k++) { Feat f; f.x = (float)(2 + (size - 4) * urand()); f.y = (float)(2 + (size - 4) * urand()); f.oct = rnd() % 8; f.angle = (float)(360 * urand());
                                           f.point = -1; for (int b = 0; b < 32; b++) f.d[b] = (uint8_t)rnd(); v.push_back(f); }
    for (size_t k = v.size(); k > 1; k--) std::swap(v[k - 1], v[rnd() % k]);
  }
  const int nl = (int)fl.size(), nr = (int)fr.size();
  F->Nleft = nl; F->Nright = nr; F->N = nl + nr;
  F->mDescriptors = Mat(F->N, 32, 1);
  for (int i = 0; i < F->N; i++) {
    const Feat& f = i < nl ? fl[i] : fr[i - nl];
    const KeyPoint kp{{f.x, f.y}, 31.f, f.angle, 20.f, f.oct};
    if (i < nl) F->mvKeys.push_back(kp); else F->mvKeysRight.push_back(kp);
    std::memcpy(F->mDescriptors.ptr<uint8_t>(i), f.d, 32);
  }
  F->mvKeysUn = F->mvKeys;
  F->mvuRight.assign(F->N, -1.f); F->mvDepth.assign(F->N, -1.f); F->mvbOutlier.assign(F->N, false);
  float isig[8]; { float s = 1.f; for (int l = 0; l < 8; l++) { isig[l] = 1.f / (s * s); s *= 1.2f; } }
  F->mvInvLevelSigma2.assign(isig, isig + 8);
  F->mvLeftToRightMatch.assign(nl, -1); F->mvRightToLeftMatch.assign(nr, -1);
  { std::map<int, int> where_r; for (int j = 0; j < nr; j++) if (fr[j].point >= 0) where_r[fr[j].point] = j;
    for (int j = 0; j < nl; j++) { if (fl[j].point < 0 || urand() > 0.4) continue; auto it = where_r.find(fl[j].point);
                                   if (it != where_r.end()) { F->mvLeftToRightMatch[j] = it->second; F->mvRightToLeftMatch[it->second] = j; } } }
  // features that already hold a point: some of them local points (which SearchLocalPoints then marks as seen), some foreign ones
  F->mvpMapPoints.assign(F->N, nullptr);
  for (int i = 0; i < F->N; i++) {
    if (urand() > 0.08) continue;
    if (urand() < 0.5) { F->mvpMapPoints[i] = out.local[rnd() % out.local.size()]; continue; }
    std::unique_ptr<MapPoint> mp(new MapPoint);
    mp->mnId = 50000 + i; mp->mpMap = &A.map; mp->nObs = (int)(rnd() % 3);
    F->mvpMapPoints[i] = mp.get();
    A.points.push_back(std::move(mp));
  }
  out.cur = F.get();
  A.frames.push_back(std::move(F));
  // the last frame: two cameras too; its entries hold some of the points
  std::unique_ptr<Frame> Lf(new Frame(*out.cur));
  Lf->mnId = 6;
  const int NL = 700, NLl = 380;
  Lf->N = NL; Lf->Nleft = NLl; Lf->Nright = NL - NLl;
  Lf->mvKeys.clear(); Lf->mvKeysRight.clear(); Lf->mvpMapPoints.assign(NL, nullptr); Lf->mvbOutlier.assign(NL, false);
  for (int i = 0; i < NL; i++) {
    const int pi = i % n_points;
    const KeyPoint kp{{0.f, 0.f}, 31.f, (float)std::fmod(base_angle[pi] + 2 * nrand() + 720.0, 360.0), 20.f, std::min(7, std::max(0, lvl[pi] + (int)(rnd() % 3) - 1))};
    if (i < NLl) Lf->mvKeys.push_back(kp); else Lf->mvKeysRight.push_back(kp);
    if (urand() < 0.85) Lf->mvpMapPoints[i] = out.local[pi];
    Lf->mvbOutlier[i] = urand() < 0.06;
  }
  Lf->mvKeysUn = Lf->mvKeys;
  double Tl[16]; for (int i = 0; i < 16; i++) Tl[i] = T[i];
  Tl[3] += 0.03; Tl[7] += 0.01; Tl[11] += dz_last;
  Lf->mTcw = mat44(Tl);
  out.last = Lf.get();
  A.frames.push_back(std::move(Lf));
  // the reference keyframe of TrackReferenceKeyFrame: a two-camera keyframe whose features hold the first 900 points (descriptors a few
  // bits off, angles near the current features'); feature vectors of both over a stand-in vocabulary (word = the descriptor's leading bits)
  {
    std::unique_ptr<KeyFrame> kf(new KeyFrame);
    kf->mnId = 3; kf->mpMap = &A.map;
    kf->mpCamera = out.cur->mpCamera; kf->mpCamera2 = out.cur->mpCamera2; kf->mTrl = out.cur->mTrl;
    const int NK = std::min(900, n_points), NKl = (NK * 4) / 7;
    kf->NLeft = NKl; kf->NRight = NK - NKl;
    kf->mDescriptors = Mat(NK, 32, 1);
    kf->mvpMapPoints.assign(NK, nullptr);
    for (int i = 0; i < NK; i++) {
      const KeyPoint kp{{0.f, 0.f}, 31.f, (float)std::fmod(base_angle[i] + 2 * nrand() + 720.0, 360.0), 20.f, lvl[i]};
      if (i < NKl) kf->mvKeys.push_back(kp); else kf->mvKeysRight.push_back(kp);
      uint8_t* d = kf->mDescriptors.ptr<uint8_t>(i);
      std::memcpy(d, out.local[i]->mDescriptor.ptr<uint8_t>(0), 32);
      for (int b = 0, nb = (int)(rnd() % 24); b < nb; b++) { const int k = 8 + rnd() % 248; d[k >> 3] ^= (uint8_t)(1u << (k & 7)); }    // (byte 0 kept: the word)
      if (urand() < 0.85) kf->mvpMapPoints[i] = out.local[i];
      kf->mFeatVec[d[0] >> 3].push_back((unsigned)i);
    }
    kf->mvKeysUn = kf->mvKeys;
    for (int i = 0; i < out.cur->N; i++) out.cur->mFeatVec[out.cur->mDescriptors.ptr<uint8_t>(i)[0] >> 3].push_back((unsigned)i);
    out.ref = kf.get();
    A.kfs.push_back(std::move(kf));
  }
  return out;
}

// A two-fisheye Frame as its constructor has it when it calls ComputeStereoFishEyeMatches (S/Frame.cc:1079): monocular features first,
// then the lapping-area ones -- 3-D points seen by both cameras (pixel noise by level, descriptors a few bits apart; some with a wrong
// partner position, some too far for any parallax), second right features nearly as close (Lowe's ratio), distractors.
static Frame* build_fisheye_ctor_frame(Agent& A, unsigned seed, int n_stereo = 600, int n_mono = 250, int n_distract = 120) {
  g_seed = seed;
  std::unique_ptr<Frame> F(new Frame);
  give_rig(A, F->mpCamera, F->mpCamera2, F->mTrl);
  double Trl[12]; rig_Trl(Trl);
  F->mTlr = Mat(3, 4, 4);
  for (int i = 0; i < 3; i++) { double tt = 0; for (int j = 0; j < 3; j++) { F->mTlr.ptr<float>(0)[4 * i + j] = (float)Trl[4 * j + i]; tt -= Trl[4 * j + i] * Trl[4 * j + 3]; }
                                F->mTlr.ptr<float>(0)[4 * i + 3] = (float)tt; }
  float sc = 1.f; for (int l = 0; l < 8; l++) { F->mvLevelSigma2.push_back(sc * sc); sc *= 1.2f; }
  struct Feat { float x, y; int oct; uint8_t d[32]; };
  auto random_feat = [&]() { Feat f; f.x = (float)(5 + 502 * urand()); f.y = (float)(5 + 502 * urand()); f.oct = rnd() % 8; for (int b = 0; b < 32; b++) f.d[b] = (uint8_t)rnd(); return f; };
  std::vector<Feat> ml, mr, sl, sr;
  for (int i = 0; i < n_mono; i++) { ml.push_back(random_feat()); if (i % 8) mr.push_back(random_feat()); }
  for (int i = 0; i < n_stereo; i++) {
    const double depth = urand() < 0.9 ? 0.6 + 8.4 * urand() : 40 + 360 * urand();
    const double u = 160 + 330 * urand(), v = 40 + 430 * urand();
    const double mx = (u - KB8_L[2]) / KB8_L[0], my = (v - KB8_L[3]) / KB8_L[1], th = std::sqrt(mx * mx + my * my), psi = std::atan2(my, mx);
    const double Pl[3] = {depth * std::sin(th) * std::cos(psi), depth * std::sin(th) * std::sin(psi), depth * std::cos(th)};
    double Pr[3]; for (int a = 0; a < 3; a++) Pr[a] = Trl[4 * a] * Pl[0] + Trl[4 * a + 1] * Pl[1] + Trl[4 * a + 2] * Pl[2] + Trl[4 * a + 3];
    const int o = rnd() % 8; const double sig = 0.6 * std::pow(1.2, o);
    double uvl[2], uvr[2]; kb8_project(KB8_L, Pl, uvl); kb8_project(KB8_R, Pr, uvr);
    Feat a, b; a.x = (float)(uvl[0] + sig * nrand()); a.y = (float)(uvl[1] + sig * nrand()); a.oct = o;
    b.x = (float)(uvr[0] + sig * nrand()); b.y = (float)(uvr[1] + sig * nrand()); b.oct = std::min(7, std::max(0, o + (int)(rnd() % 3) - 1));
    if (urand() < 0.08) { b.x += (float)((urand() < 0.5 ? -1 : 1) * (6 + 24 * urand())); b.y += (float)((urand() < 0.5 ? -1 : 1) * (6 + 24 * urand())); }
    for (int k = 0; k < 32; k++) a.d[k] = (uint8_t)rnd();
    std::memcpy(b.d, a.d, 32);
    for (int k = 0, nb = (int)(rnd() % 40); k < nb; k++) { const int bit = rnd() % 256; b.d[bit >> 3] ^= (uint8_t)(1u << (bit & 7)); }
    sl.push_back(a); sr.push_back(b);
    if (urand() < 0.12) { Feat c = b; c.x += (float)(3 * nrand()); c.y += (float)(3 * nrand()); for (int k = 0, nb = 1 + (int)(rnd() % 11); k < nb; k++) { const int bit = rnd() % 256; c.d[bit >> 3] ^= (uint8_t)(1u << (bit & 7)); } sr.push_back(c); }
  }
  for (int i = 0; i < n_distract; i++) { sl.push_back(random_feat()); sr.push_back(random_feat()); }
  for (size_t k = sl.size(); k > 1; k--) std::swap(sl[k - 1], sl[rnd() % k]);
  for (size_t k = sr.size(); k > 1; k--) std::swap(sr[k - 1], sr[rnd() % k]);
  F->monoLeft = (int)ml.size(); F->monoRight = (int)mr.size();
  ml.insert(ml.end(), sl.begin(), sl.end()); mr.insert(mr.end(), sr.begin(), sr.end());
  F->Nleft = (int)ml.size(); F->Nright = (int)mr.size(); F->N = F->Nleft + F->Nright;
  F->mDescriptors = Mat(F->Nleft, 32, 1); F->mDescriptorsRight = Mat(F->Nright, 32, 1);
  for (int i = 0; i < F->Nleft; i++) { F->mvKeys.push_back(KeyPoint{{ml[i].x, ml[i].y}, 31.f, 0.f, 20.f, ml[i].oct}); std::memcpy(F->mDescriptors.ptr<uint8_t>(i), ml[i].d, 32); }
  for (int i = 0; i < F->Nright; i++) { F->mvKeysRight.push_back(KeyPoint{{mr[i].x, mr[i].y}, 31.f, 0.f, 20.f, mr[i].oct}); std::memcpy(F->mDescriptorsRight.ptr<uint8_t>(i), mr[i].d, 32); }
  Frame* out = F.get();
  A.frames.push_back(std::move(F));
  return out;
}

// The keyframe after `cur` as LocalMapping would insert it: covisible with `cur` and all but the oldest of its neighbours, observing
// a third of cur's points (MapPoint::AddObservation: the points' change counters move, as in the reference with INTEGRATION.md's hook).
static inline KeyFrame* next_keyframe(Agent& B, KeyFrame* cur, int round) {
  std::unique_ptr<KeyFrame> kf(new KeyFrame(*cur));
  kf->mnId = cur->mnId + 1; kf->mnBALocalForKF = ~0ul; kf->mnBAFixedForKF = ~0ul;
  kf->mvKeysUn.clear(); kf->mvuRight.clear(); kf->mvpMapPoints.clear();
  kf->mvpOrderedConnectedKeyFrames.clear();
  for (size_t q = 1; q < cur->mvpOrderedConnectedKeyFrames.size(); q++) kf->mvpOrderedConnectedKeyFrames.push_back(cur->mvpOrderedConnectedKeyFrames[q]);
  kf->mvpOrderedConnectedKeyFrames.push_back(cur);
  for (size_t li0 = 0; li0 < cur->mvpMapPoints.size(); li0++) {
    MapPoint* mp = cur->mvpMapPoints[li0];
    if (!mp || mp->isBad() || (mp->mnId + round) % 3 != 0) continue;
    const int li = (int)kf->mvKeysUn.size();
    kf->mvKeysUn.push_back(cur->mvKeysUn[li0]); kf->mvuRight.push_back(cur->mvuRight[li0]); kf->mvpMapPoints.push_back(mp);
    mp->AddObservation(kf.get(), li);
  }
  KeyFrame* out = kf.get();
  B.kfs.push_back(std::move(kf));
  return out;
}


