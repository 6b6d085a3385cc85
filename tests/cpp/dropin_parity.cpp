// The reference-side glue (include/orbgpu_dropin.hpp: the INTEGRATION.md bodies with the reference's signatures) run through
// header-only mocks of Frame / KeyFrame / MapPoint / Map, once over liborbgpu (GpuOps, the product) and once over the CPU
// oracle (OracleOps), on the same synthetic agent: a textured plane seen by a moving stereo rig.
//   extractor + stereo Frame constructor, isInFrustum, the three tracking matchers, LocalBundleAdjustment (graph collection,
//   vToErase, the 50 %-outlier early return, *pbStopFlag) and PoseOptimization.
// Exit code 0 = every stage agrees; 3 = no GPU (the glue must fail loudly); anything else = a mismatch (printed).
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <set>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "mock_orbslam3.hpp"
#include "orbgpu_dropin.hpp"
#include "../../oracle/orb_oracle.h"

using namespace mock;
namespace od = orbgpu::dropin;

struct OracleOps {       // the same entry points over the CPU oracle: views instead of device handles
  static int is_in_frustum(const void*, const orbm_frame_view& v, const float* Tcw, const orbm_worldpoints_view& pts, float lim, uint8_t* in_view,
                           float* px, float* py, float* pxr, float* depth, int32_t* level, float* vcos) {
    return oracle_is_in_frustum(&v, Tcw, &pts, lim, in_view, px, py, pxr, depth, level, vcos);
  }
  static int search_mps(const void*, const orbm_frame_view& v, const orbm_mappoints_view& mps, float th, int far_points, float th_far, float nnratio,
                        int32_t* amp, int32_t* aob, int* n) {
    return oracle_search_by_projection_mps(&v, &mps, th, far_points, th_far, nnratio, amp, aob, n);
  }
  static int search_frame(const void*, const orbm_frame_view& v, const float* Tcw, const orbm_lastframe_view& last, float th, int mono, int check_ori,
                          int32_t* amp, int32_t* aob, int* n) {
    return oracle_search_by_projection_frame(&v, Tcw, &last, th, mono, check_ori, amp, aob, n);
  }
  static int search_bow(const void*, const orbm_frame_view& v, const orbm_featvec_view& fvF, const uint8_t* kf_desc, int nkf, const uint8_t* kf_valid,
                        const float* kf_angle, const orbm_featvec_view& fvKF, float nnratio, int check_ori, int32_t* matches, int* n) {
    return oracle_search_by_bow(&v, &fvF, kf_desc, nkf, kf_valid, kf_angle, &fvKF, nnratio, check_ori, matches, n);
  }
  // (the oracle polls an int32: the bool is sampled -- the cases that raise it DURING the solve go through OracleAtTrialOps)
  static int lba(const lba_problem& p, const volatile bool* stop, lba_result& r) {
    volatile int32_t s = (stop && *stop) ? 1 : 0;
    return oracle_lba_solve(&p, &s, &r);
  }
  static int pose_opt(const pose_opt_problem& p, pose_opt_result& r) { return oracle_pose_optimize(&p, &r); }
};

// The product's entry points with a tap on the solver's report: LM iterations per round and the number of LM trials that
// had been evaluated when the solve ended (= when it saw the flag, for an aborted one).
struct GpuTapOps : od::GpuOps {
  static int iters[2], trials, status;
  static int lba(const lba_problem& p, const volatile bool* stop, lba_result& r) {
    static double trace[3 * 64];
    r.trace = trace; r.trace_cap = 64; r.trace_len = 0;
    const int rc = od::GpuOps::lba(p, stop, r);
    iters[0] = r.iters_round1; iters[1] = r.iters_round2; status = r.status; trials = 0;
    for (int i = 0; i < r.trace_len; i++) trials += (int)trace[3 * i + 2];
    r.trace = nullptr; r.trace_cap = r.trace_len = 0;
    return rc;
  }
};
int GpuTapOps::iters[2] = {0, 0}; int GpuTapOps::trials = 0; int GpuTapOps::status = 0;
// The oracle stopped at a given poll point: "the flag reads as raised once k LM trials have been evaluated" (orbgpu.h)
struct OracleAtTrialOps : OracleOps {
  static int k, iters[2];
  static int lba(const lba_problem& p, const volatile bool*, lba_result& r) {
    volatile int32_t s = k == 0 ? INT32_MIN : -k;
    const int rc = oracle_lba_solve(&p, &s, &r);
    iters[0] = r.iters_round1; iters[1] = r.iters_round2;
    return rc;
  }
};
int OracleAtTrialOps::k = 0; int OracleAtTrialOps::iters[2] = {0, 0};

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
  std::vector<std::unique_ptr<Frame>> frames;
  std::vector<std::unique_ptr<MapPoint>> points;
  std::vector<std::unique_ptr<KeyFrame>> kfs;
  Map map;
};

// Frame::Frame(stereo) through the adapter: extraction L+R, ComputeStereoMatches, grid -- and the host copies the mocks hold
static void make_frame(Agent& A, orbgpu::ORBextractor& rig, const std::vector<uint8_t>& tex, int k) {
  double T[16], Tr[16]; pose_of(k, T);
  for (int i = 0; i < 16; i++) Tr[i] = T[i];
  Tr[3] -= BB;
  const std::vector<uint8_t> L = render(tex, T), R = render(tex, Tr);
  std::unique_ptr<Frame> F(new Frame);
  F->mnMinX = 0; F->mnMaxX = W; F->mnMinY = 0; F->mnMaxY = H; F->fx = FX; F->fy = FX; F->cx = CX; F->cy = CY; F->mbf = BF; F->mb = BB;
  orbm_frame_view v{0, nullptr, nullptr, nullptr, nullptr, 0, (float)W, 0, (float)H, FX, FX, CX, CY, BF, BB, 8, 1.2f};
  orbgpu::FrameOnDevice dev(4096);
  std::vector<orbx_keypoint> keys; std::vector<uint8_t> desc; std::vector<float> ur, dp;
  const int N = dev.StereoCtor(rig, v, L.data(), R.data(), W, H, W, &keys, &desc, &ur, &dp);
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

static int g_fail = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { std::printf("MISMATCH %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); g_fail++; } } while (0)

// indices of the matched map points (position in `pool`) per feature, -1 = none
static std::vector<int> assignment_ids(const Frame& F, const std::vector<MapPoint*>& pool) {
  std::map<const MapPoint*, int> id;
  for (size_t i = 0; i < pool.size(); i++) id[pool[i]] = (int)i;
  std::vector<int> out(F.N, -1);
  for (int i = 0; i < F.N; i++) if (F.mvpMapPoints[i]) out[i] = id.count(F.mvpMapPoints[i]) ? id[F.mvpMapPoints[i]] : -2;
  return out;
}

struct TrackOut { int n_visible, n_map, n_frame, n_bow, n_pose; std::vector<int> a_map, a_frame, a_bow; std::vector<bool> outl; std::vector<float> pose; };

template <class Ops>
static TrackOut run_tracking(orbgpu::ORBextractor& rig, const std::vector<uint8_t>& tex) {
  Agent A; TrackOut o;
  for (int k : {4, 5, 6}) make_frame(A, rig, tex, k);
  Frame &F0 = *A.frames[0], &F1 = *A.frames[1], &F2 = *A.frames[2];
  // ---- Tracking::SearchLocalPoints: isInFrustum + SearchByProjection(F, local map points)
  std::vector<MapPoint*> local; std::vector<int> src;
  make_points_from(A, F0, local, src);
  o.n_visible = od::isInFrustumAll<Ops>(F2, local, 0.5f);
  o.n_map = od::SearchByProjection<Ops>(F2, local, 3.0f, false, 50.0f, 0.8f);
  o.a_map = assignment_ids(F2, local);
  // ---- TrackWithMotionModel: SearchByProjection(Current, Last) with the last frame's map points, then PoseOptimization
  std::vector<MapPoint*> lastpts; std::vector<int> lastfeat;
  make_points_from(A, F1, lastpts, lastfeat);
  for (size_t j = 0; j < lastpts.size(); j++) F1.mvpMapPoints[lastfeat[j]] = lastpts[j];
  std::fill(F2.mvpMapPoints.begin(), F2.mvpMapPoints.end(), nullptr);
  {  // pose guess: the true pose, slightly off
    float* T = F2.mTcw.ptr<float>(0); T[3] += 0.012f; T[7] -= 0.008f; T[11] += 0.01f;
  }
  o.n_frame = od::SearchByProjection<Ops>(F2, F1, 7.0f, false, true);
  o.a_frame = assignment_ids(F2, lastpts);
  for (int i = 0; i < F2.N && !lastpts.empty(); i += 17) if (F2.mvpMapPoints[i]) F2.mvpMapPoints[i] = lastpts[(i * 7) % lastpts.size()];   // some wrong associations -> outliers
  o.n_pose = od::PoseOptimization<Ops>(&F2);
  o.outl = F2.mvbOutlier; o.pose.assign(F2.mTcw.ptr<float>(0), F2.mTcw.ptr<float>(0) + 16);
  // ---- relocalisation / reference-keyframe tracking: SearchByBoW(KeyFrame, Frame)
  std::unique_ptr<KeyFrame> kf(new KeyFrame);
  kf->mvKeysUn = F1.mvKeysUn; kf->mDescriptors = F1.mDescriptors; kf->mFeatVec = F1.mFeatVec; kf->mvpMapPoints = F1.mvpMapPoints;
  std::vector<MapPoint*> bow;
  o.n_bow = od::SearchByBoW<Ops>(kf.get(), F2, bow, 0.7f, true);
  Frame tmp; tmp.N = F2.N; tmp.mvpMapPoints = bow;
  o.a_bow = assignment_ids(tmp, lastpts);
  return o;
}

// ------------------------------------------------------------------------------------------------ local BA scenario
struct BaOut { int status, num_fixed, erased, change_index, locked_poses, locked_points; std::vector<float> poses, points; std::vector<int> normal_updates; };

// raise_after_us >= 0: a second thread (Tracking calling LocalMapping::InterruptBA, S/LocalMapping.cc:381-386) sets the bool
// that many microseconds after the call starts
template <class Ops>
static BaOut run_lba(int n_local, int n_far, int n_pts, double outlier_frac, bool stop, unsigned seed, double raise_after_us = -1.0) {
  g_seed = seed;
  Agent A; BaOut o;
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
  bool mbAbortBA = stop;                                            // I/LocalMapping.h:155
  std::thread tracking;
  if (raise_after_us >= 0)
    tracking = std::thread([&mbAbortBA, raise_after_us]() {
      const auto t_end = std::chrono::steady_clock::now() + std::chrono::nanoseconds((long long)(raise_after_us * 1e3));
      while (std::chrono::steady_clock::now() < t_end) {}
      *const_cast<volatile bool*>(&mbAbortBA) = true;
    });
  o.status = od::LocalBundleAdjustment<Ops>(cur, &mbAbortBA, &A.map, o.num_fixed, 0);
  if (tracking.joinable()) tracking.join();
  o.erased = 0; o.change_index = A.map.GetMapChangeIndex(); o.locked_poses = o.locked_points = 0;
  for (auto& kf : A.kfs) o.locked_poses += kf->n_locked_pose_writes;
  for (auto& p : A.points) o.locked_points += p->n_locked_pos_writes;
  for (auto& kf : A.kfs) { o.poses.insert(o.poses.end(), kf->Tcw.ptr<float>(0), kf->Tcw.ptr<float>(0) + 16); for (auto* p : kf->mvpMapPoints) o.erased += p == nullptr; }
  for (auto& p : A.points) { o.points.insert(o.points.end(), p->mWorldPos.ptr<float>(0), p->mWorldPos.ptr<float>(0) + 3); o.normal_updates.push_back(p->n_normal_updates); }
  return o;
}

static float max_abs_diff(const std::vector<float>& a, const std::vector<float>& b) {
  float m = a.size() == b.size() ? 0.f : 1e9f;
  for (size_t i = 0; i < std::min(a.size(), b.size()); i++) m = std::max(m, std::fabs(a[i] - b[i]));
  return m;
}

int main() {
  try {
    g_seed = 20251003u;
    const std::vector<uint8_t> tex = make_texture();
    orbgpu::ORBextractor rig(1000, 1.2f, 8, 20, 7, W, H, /*n_cams*/ 2);
    // ---- tracking side: identical mocks through both entry-point sets
    g_seed = 7; const TrackOut g = run_tracking<od::GpuOps>(rig, tex);
    g_seed = 7; const TrackOut c = run_tracking<OracleOps>(rig, tex);
    std::printf("isInFrustum %d visible; SearchByProjection(F, MPs) %d; SearchByProjection(Cur, Last) %d; SearchByBoW %d; PoseOptimization %d inliers\n",
                g.n_visible, g.n_map, g.n_frame, g.n_bow, g.n_pose);
    EXPECT(g.n_visible == c.n_visible && g.n_visible > 300, "isInFrustum %d vs %d", g.n_visible, c.n_visible);
    EXPECT(g.n_map == c.n_map && g.a_map == c.a_map && g.n_map > 100, "SearchByProjection(F, MPs) %d vs %d", g.n_map, c.n_map);
    EXPECT(g.n_frame == c.n_frame && g.a_frame == c.a_frame && g.n_frame > 100, "SearchByProjection(Cur, Last) %d vs %d", g.n_frame, c.n_frame);
    EXPECT(g.n_bow == c.n_bow && g.a_bow == c.a_bow && g.n_bow > 20, "SearchByBoW %d vs %d", g.n_bow, c.n_bow);
    EXPECT(g.n_pose == c.n_pose && g.outl == c.outl && g.n_pose > 50, "PoseOptimization inliers %d vs %d", g.n_pose, c.n_pose);
    EXPECT(max_abs_diff(g.pose, c.pose) <= 1e-4f, "PoseOptimization pose differs by %g", max_abs_diff(g.pose, c.pose));
    int n_out = 0; for (bool b : g.outl) n_out += b;
    EXPECT(n_out >= 5, "PoseOptimization flagged only %d of the planted wrong associations", n_out);
    // ---- LocalMapping side
    struct Case { const char* name; int n_local, n_far, n_pts; double outl; bool stop; int want_status; } cases[] = {
        {"regular window, 3 % gross outliers (vToErase)", 8, 4, 500, 0.03, false, LBA_APPLIED},
        {"fewer than two fixed keyframes: the two lowest ids get fixed", 6, 0, 300, 0.02, false, LBA_APPLIED},
        {"most observations are outliers: early return, nothing written", 6, 3, 300, 0.85, false, LBA_REJECTED_OUTLIERS},
        {"*pbStopFlag raised before the solve", 6, 3, 300, 0.02, true, LBA_ABORTED_BEFORE_OPT}};
    for (const Case& cs : cases) {
      const BaOut bg = run_lba<od::GpuOps>(cs.n_local, cs.n_far, cs.n_pts, cs.outl, cs.stop, 99);
      const BaOut bc = run_lba<OracleOps>(cs.n_local, cs.n_far, cs.n_pts, cs.outl, cs.stop, 99);
      std::printf("LocalBundleAdjustment [%s]: status %d, %d fixed KFs, %d observations erased\n", cs.name, bg.status, bg.num_fixed, bg.erased);
      EXPECT(bg.status == bc.status && bg.status == cs.want_status, "status %d vs %d (want %d)", bg.status, bc.status, cs.want_status);
      EXPECT(bg.num_fixed == bc.num_fixed && bg.num_fixed >= 2, "num_fixedKF %d vs %d", bg.num_fixed, bc.num_fixed);
      EXPECT(bg.erased == bc.erased, "erased observations %d vs %d", bg.erased, bc.erased);
      EXPECT(max_abs_diff(bg.poses, bc.poses) <= 1e-4f && max_abs_diff(bg.points, bc.points) <= 1e-4f, "state differs: poses %g points %g",
             max_abs_diff(bg.poses, bc.poses), max_abs_diff(bg.points, bc.points));
      EXPECT(bg.normal_updates == bc.normal_updates, "UpdateNormalAndDepth calls differ");
      if (cs.want_status == LBA_APPLIED) {
        EXPECT(bg.erased > 0 && bg.normal_updates[0] == 1, "an applied LBA erases outliers and refreshes the points");
        // SetPose(T, true) for every local keyframe, SetWorldPos(X, true) for every local point, then pMap->IncreaseChangeIndex()
        // (S/Optimizer.cc:2327,2386,2397)
        EXPECT(bg.change_index == 1 && bg.locked_poses >= cs.n_local - 2 && bg.locked_poses <= cs.n_local && bg.locked_points > 0,
               "write-back: change index %d, %d locked pose writes, %d locked point writes", bg.change_index, bg.locked_poses, bg.locked_points);
      } else {
        EXPECT(bg.erased == 0 && bg.normal_updates[0] == 0, "a rejected / aborted LBA must not touch the map");
        EXPECT(bg.change_index == 0 && bg.locked_poses == 0 && bg.locked_points == 0, "a rejected / aborted LBA must not bump the change index");
      }
      EXPECT(bg.change_index == bc.change_index && bg.locked_poses == bc.locked_poses && bg.locked_points == bc.locked_points, "write-back calls differ");
    }
    // ---- InterruptBA() DURING the solve: Tracking sets LocalMapping::mbAbortBA -- the very bool behind pbStopFlag -- from a
    // second thread at an arbitrary moment.  Wherever the product saw it (it reports how many LM trials it had evaluated),
    // the oracle stopped at that same poll point must give the same iteration counts, status and written-back state.
    {
      std::set<std::pair<int, int>> seen;
      const BaOut full = run_lba<GpuTapOps>(8, 4, 500, 0.03, false, 99);
      const int full_it[2] = {GpuTapOps::iters[0], GpuTapOps::iters[1]};
      int n_cut = 0;
      for (int rep = 0; rep < 36; rep++) {
        const BaOut bg = run_lba<GpuTapOps>(8, 4, 500, 0.03, false, 99, 25.0 * rep);
        const int it0 = GpuTapOps::iters[0], it1 = GpuTapOps::iters[1], k = GpuTapOps::trials;
        seen.insert({it0, it1});
        BaOut bc;
        if (bg.status == LBA_ABORTED_BEFORE_OPT) {
          bc = run_lba<OracleOps>(8, 4, 500, 0.03, true, 99);
        } else {
          OracleAtTrialOps::k = k;
          bc = run_lba<OracleAtTrialOps>(8, 4, 500, 0.03, false, 99);
          EXPECT(it0 == OracleAtTrialOps::iters[0] && it1 == OracleAtTrialOps::iters[1], "flag after %d us, seen after %d trials: iterations %d+%d vs oracle %d+%d",
                 25 * rep, k, it0, it1, OracleAtTrialOps::iters[0], OracleAtTrialOps::iters[1]);
        }
        n_cut += it0 != full_it[0] || it1 != full_it[1];
        EXPECT(bg.status == bc.status && bg.erased == bc.erased && bg.change_index == bc.change_index, "flag after %d us: status %d vs %d, erased %d vs %d",
               25 * rep, bg.status, bc.status, bg.erased, bc.erased);
        EXPECT(max_abs_diff(bg.poses, bc.poses) <= 1e-4f && max_abs_diff(bg.points, bc.points) <= 1e-4f, "flag after %d us (%d trials): state differs: poses %g points %g",
               25 * rep, k, max_abs_diff(bg.poses, bc.poses), max_abs_diff(bg.points, bc.points));
      }
      std::printf("InterruptBA during the solve: %zu distinct (round 1, round 2) iteration counts over 36 delays, %d solves cut short (uninterrupted: %d+%d)\n",
                  seen.size(), n_cut, full_it[0], full_it[1]);
      EXPECT(seen.size() >= 2 && n_cut >= 1, "the sweep never interrupted a running solve");
      (void)full;
    }
    if (g_fail) { std::printf("dropin parity: %d mismatches\n", g_fail); return 1; }
    std::printf("dropin parity ok\n");
    return 0;
  } catch (const std::runtime_error& e) {
    std::printf("runtime_error: %s\n", e.what());
    return 3;
  }
}
