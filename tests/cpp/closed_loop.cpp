// CLOSED-LOOP parity: a stereo agent tracks a synthetic sequence through the reference-signature glue (include/orbgpu_dropin.hpp)
// over the mocks, once with the product's entry points (GpuOps: liborbgpu, frames built by the device constructor) and once with
// the CPU oracle's (OracleOps: oracle extractor + oracle matchers / optimisers).  Every run feeds on ITS OWN outputs the way
// Tracking::Track and LocalMapping carry state from call to call in the reference:
//   Frame constructor -> SearchByProjection(Cur, Last) with the pose the motion model predicts from the run's previous two poses
//   -> PoseOptimization -> outliers discarded -> UpdateLocalMap from the run's own covisibility graph -> SearchLocalPoints ->
//   PoseOptimization -> IncreaseFound / outliers dropped -> mVelocity, mLastFrame                   (S/Tracking.cc:2572-2811, 3083-3330)
//   every kf_every-th frame a keyframe: new stereo points (S/Tracking.cc:2952-3082), ProcessNewKeyFrame (observations, normals,
//   distinctive descriptors through the entry points), UpdateConnections, MapPointCulling, a handful of fusions, and
//   LocalBundleAdjustment with its erasures and write-back (S/LocalMapping.cc:140-379,396-485; S/Optimizer.cc:1810-2410);
//   the next frame's local map is rebuilt from what that left behind.
// The scaffolding between the hot-path calls (graph bookkeeping, culling rules, the float pose algebra) is shared host code and
// deterministic (std::map<KeyFrame*> walks are replaced by id order: heap addresses differ between the runs), so the two runs can only
// part where an entry point's output differs.  Two checks:
//  (1) SHADOW: in the product runs every entry-point call -- both searches, both PoseOptimization calls, every local BA, every
//      batch of distinctive descriptors: ~1000 calls on inputs the run produced itself -- is repeated on the oracle with the
//      IDENTICAL inputs and compared at once: match arrays, in-frustum flags, outlier flags, inlier counts, LBA status / iteration
//      counts / outlier edges exact; poses <= 1e-6; LBA poses and points <= 1e-4 (north_star's tolerance).  The run goes on with
//      the product's outputs.  Any mismatch fails the run.
//  (2) INDEPENDENT RUNS: the oracle's own closed loop against the product's, frame by frame.  Every DISCRETE digest must be equal
//      on every frame: features, both match arrays, outlier flags, inlier counts, the local map's make-up (keyframe and point ids
//      in order), per keyframe the new / culled / fused points, the local BA's status, iteration counts and fixed keyframes, which
//      points are bad, every point's observation count, every keyframe's remaining matches.  The float state is compared with the
//      tolerances an honest closed loop allows: on identical inputs the product's PoseOptimization (tree-order sums) and the oracle's
//      (serial sums) return poses one float32 ulp apart now and then (7.5e-9 in a rotation entry: the shadow's worst case), and
//      from the first such ulp on (frame 15 here; the frames before it agree bit for bit) the two runs carry slightly different
//      floats: the motion model extrapolates them, and the local BA stops after 5 + 10 LM iterations, short of convergence, and
//      returns weakly observed points (two stereo observations: 5 cm of depth per pixel of disparity) millimetres apart for inputs
//      micrometres apart.  Measured over 200 frames: poses <= 3.2e-5, keyframe poses <= 4.8e-6, point positions <= 3.3e-3,
//      distance ranges <= 1.2e-2 -- bounded, not growing.  Gates: poses and keyframe poses <= 1e-4, point state <= 5e-2; the first
//      divergent frame is reported.
// A third run takes the product's entry points as they compile against an UNMODIFIED MapPoint (no change counter: every call
// reads every point, no window cache); the default product run uses the counter's exact caches -- so the caches see 200 frames of
// real churn against the oracle's fresh reads.
//   closed_loop [n_frames=200] [kf_every=5] [--oracle-only]     exit 0 = all runs agree, 1 = divergence (printed), 3 = no GPU
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <set>
#include <vector>

#include "mock_orbslam3.hpp"
#include "orbgpu_dropin.hpp"
#include "oracle_ops.hpp"

using namespace mock;

// ------------------------------------------------------------------------------------------------ entry-point sets
template <class Base>
struct LoopTap : Base {          // + the solver's report of the last local BA
  static int& status() { static int v = 0; return v; }
  static int& it1() { static int v = 0; return v; }
  static int& it2() { static int v = 0; return v; }
  static int lba(const lba_problem& p, const volatile bool* stop, lba_result& r) {
    const int rc = Base::lba(p, stop, r);
    status() = r.status; it1() = r.iters_round1; it2() = r.iters_round2;
    return rc;
  }
};
// The product's entry points, every call repeated on the oracle with the same inputs and compared on the spot (check (1) above).
// The frame's host-side view (the product works on the device-resident copy) is set by the agent before it calls the glue.
static int& reloc_frames() { static int n = 0; return n; }    // frames that went through Relocalization's guided search
static int& bow_frames() { static int n = 0; return n; }      // frames tracked through TrackReferenceKeyFrame's SearchByBoW (all runs of the process)
struct Shadow {
  static const orbm_frame_view*& view() { static const orbm_frame_view* v = nullptr; return v; }
  static long& calls() { static long n = 0; return n; }
  static long& fails() { static long n = 0; return n; }
  static float& worst_pose() { static float v = 0; return v; }
  static float& worst_lba() { static float v = 0; return v; }
  static int& frame_no() { static int k = 0; return k; }
  static void fail(const char* what, const char* detail) { if (fails()++ < 20) std::printf("SHADOW MISMATCH at frame %d: %s: %s\n", frame_no(), what, detail); }
};
struct GpuShadowBase : od::GpuOps {
  static int search_frame(const od::FrameKey& key, const orbm_frame_view& v, const float* Tcw, const orbm_lastframe_view& last, float th, int mono, int check_ori,
                          int32_t* amp, int32_t* aob, int* n) {
    const int N = Shadow::view()->n;
    std::vector<int32_t> amp2(amp, amp + N), aob2(aob, aob + N); int n2 = 0;
    const int rc = od::GpuOps::search_frame(key, v, Tcw, last, th, mono, check_ori, amp, aob, n);
    oracle_search_by_projection_frame(Shadow::view(), Tcw, &last, th, mono, check_ori, amp2.data(), aob2.data(), &n2);
    Shadow::calls()++;
    if (n2 != *n || std::memcmp(amp, amp2.data(), 4 * (size_t)N) || std::memcmp(aob, aob2.data(), 4 * (size_t)N)) Shadow::fail("SearchByProjection(Cur, Last)", "match arrays differ");
    return rc;
  }
  static int search_reloc(const od::FrameKey& key, const orbm_frame_view& v, const float* Tcw, const orbm_worldpoints_view& kf_pts, const uint8_t* found,
                          const float* kf_angle, float th, int orb_dist, int check_ori, int32_t* amp, int* n) {
    const int N = Shadow::view()->n;
    std::vector<int32_t> amp2(amp, amp + N); int n2 = 0;
    const int rc = od::GpuOps::search_reloc(key, v, Tcw, kf_pts, found, kf_angle, th, orb_dist, check_ori, amp, n);
    oracle_search_by_projection_reloc(Shadow::view(), Tcw, &kf_pts, found, kf_angle, th, orb_dist, check_ori, amp2.data(), &n2);
    Shadow::calls()++;
    if (n2 != *n || std::memcmp(amp, amp2.data(), 4 * (size_t)N)) Shadow::fail("SearchByProjection(F, KF, sAlreadyFound)", "match arrays differ");
    return rc;
  }
  static int search_bow(const od::FrameKey& key, const orbm_frame_view& v, const orbm_featvec_view& fvF, const uint8_t* kf_desc, int nkf, const uint8_t* kf_valid,
                        const float* kf_angle, const orbm_featvec_view& fvKF, float nnratio, int check_ori, int32_t* matches, int* n) {
    const int N = Shadow::view()->n;
    const int rc = od::GpuOps::search_bow(key, v, fvF, kf_desc, nkf, kf_valid, kf_angle, fvKF, nnratio, check_ori, matches, n);
    std::vector<int32_t> m2(N); int n2 = 0;
    oracle_search_by_bow(Shadow::view(), &fvF, kf_desc, nkf, kf_valid, kf_angle, &fvKF, nnratio, check_ori, m2.data(), &n2);
    Shadow::calls()++;
    if (n2 != *n || std::memcmp(matches, m2.data(), 4 * (size_t)N)) Shadow::fail("SearchByBoW(KF, F)", "matches differ");
    return rc;
  }
  static int search_local_resident(const od::FrameKey& key, const orbm_frame_view& v, const float* Tcw, const orbm_worldpoints_view& pts, const uint8_t* excluded,
                                   bool statics_same, float th, int far_points, float th_far, float nnratio, int32_t* amp, int32_t* aob, int* n, uint8_t* in_frustum) {
    const int N = Shadow::view()->n, M = pts.m;
    std::vector<int32_t> amp2(amp, amp + N), aob2(aob, aob + N); int n2 = 0; std::vector<uint8_t> vis2(M, 0);
    const int rc = od::GpuOps::search_local_resident(key, v, Tcw, pts, excluded, statics_same, th, far_points, th_far, nnratio, amp, aob, n, in_frustum);
    OracleOps::search_local(key, *Shadow::view(), Tcw, pts, th, far_points, th_far, nnratio, amp2.data(), aob2.data(), &n2, vis2.data());
    Shadow::calls()++;
    if (n2 != *n || std::memcmp(amp, amp2.data(), 4 * (size_t)N) || std::memcmp(aob, aob2.data(), 4 * (size_t)N)) Shadow::fail("SearchLocalPoints", "match arrays differ");
    if (std::memcmp(in_frustum, vis2.data(), (size_t)M)) Shadow::fail("SearchLocalPoints", "in-frustum flags differ");
    return rc;
  }
  static int pose_opt(const pose_opt_problem& p, pose_opt_result& r) {
    const int rc = od::GpuOps::pose_opt(p, r);
    std::vector<uint8_t> o2(p.n); pose_opt_result r2{}; r2.outlier = o2.data();
    oracle_pose_optimize(&p, &r2);
    Shadow::calls()++;
    float e = 0; for (int i = 0; i < 16; i++) e = std::max(e, std::fabs(r.Tcw[i] - r2.Tcw[i]));
    Shadow::worst_pose() = std::max(Shadow::worst_pose(), e);
    if (r.n_inliers != r2.n_inliers || std::memcmp(r.outlier, o2.data(), (size_t)p.n)) Shadow::fail("PoseOptimization", "outlier flags / inlier count differ");
    // the pose itself: the round-5 sweep's bound (3.6 % of the problems run one LM iteration more or less at a threshold: <= 1e-5)
    if (e > 1e-5f) { char b[96]; std::snprintf(b, sizeof b, "pose differs by %g", e); Shadow::fail("PoseOptimization", b); }
    return rc;
  }
  static int lba(const lba_problem& p, const volatile bool* stop, lba_result& r) {
    const int rc = od::GpuOps::lba(p, stop, r);
    std::vector<float> po(16 * (size_t)p.n_poses), pt(3 * (size_t)p.n_points); std::vector<uint8_t> eo(p.n_edges), ed(p.n_edges); std::vector<double> ec(p.n_edges);
    lba_result r2{}; r2.poses = po.data(); r2.points = pt.data(); r2.edge_outlier = eo.data(); r2.edge_depth_pos = ed.data(); r2.edge_chi2 = ec.data();
    OracleOps::lba(p, stop, r2);
    Shadow::calls()++;
    if (r.status != r2.status || r.iters_round1 != r2.iters_round1 || r.iters_round2 != r2.iters_round2) { char b[128]; std::snprintf(b, sizeof b, "status %d vs %d, iterations %d+%d vs %d+%d", r.status, r2.status, r.iters_round1, r.iters_round2, r2.iters_round1, r2.iters_round2); Shadow::fail("LocalBundleAdjustment", b); }
    else if (r.status == LBA_APPLIED) {
      float e = 0;
      for (size_t i = 0; i < po.size(); i++) e = std::max(e, std::fabs(po[i] - r.poses[i]));
      for (size_t i = 0; i < pt.size(); i++) e = std::max(e, std::fabs(pt[i] - r.points[i]));
      Shadow::worst_lba() = std::max(Shadow::worst_lba(), e);
      if (e > 1e-4f) { char b[96]; std::snprintf(b, sizeof b, "%d poses / %d points: state differs by %g", p.n_poses, p.n_points, e); Shadow::fail("LocalBundleAdjustment", b); }
      if (std::memcmp(eo.data(), r.edge_outlier, (size_t)p.n_edges)) Shadow::fail("LocalBundleAdjustment", "outlier edges differ");
    }
    return rc;
  }
};
struct GpuLoopOps : LoopTap<GpuShadowBase> {
  static int distinctive(const uint8_t* desc, const int32_t* start, int m, int32_t* best) {
    const int rc = orbm_distinctive_descriptors(0, desc, start, m, best);
    std::vector<int32_t> b2(m, -2);
    oracle_distinctive_descriptors(desc, start, m, b2.data());
    Shadow::calls()++;
    if (std::memcmp(best, b2.data(), 4 * (size_t)m)) Shadow::fail("ComputeDistinctiveDescriptors", "medoid indices differ");
    return rc;
  }
};
struct GpuUnmodifiedLoopOps : GpuLoopOps {      // as against the reference's MapPoint + edit E1 alone
  static constexpr bool kNoChangeStamp = true;
  static constexpr bool kNoLbaCache = true;
};
struct OracleLoopOps : LoopTap<OracleOps> {
  static int distinctive(const uint8_t* desc, const int32_t* start, int m, int32_t* best) { return oracle_distinctive_descriptors(desc, start, m, best); }
};

// ------------------------------------------------------------------------------------------------ scene
static unsigned g_seed = 1;
static unsigned rnd() { g_seed = g_seed * 1664525u + 1013904223u; return g_seed >> 8; }
static double urand() { return (rnd() & 0xFFFFFF) / double(0x1000000); }

static const int W = 640, H = 480, TW = 1700, TH = 900;
static const double PPM = 200.0;                                  // texture pixels per metre
static const float FX = 458.654f * 640 / 752, CX = 320.f, CY = 240.f, BF = 47.90639384423901f * 640 / 752, BB = BF / FX;
static const float TH_DEPTH = 40.0f * BB;                         // mThDepth = mbf * ThDepth / fx (S/Tracking.cc:116-121), ThDepth 40

static std::vector<uint8_t> make_texture() {
  std::vector<float> t((size_t)TW * TH, 110.f);
  for (int cell = 128; cell >= 4; cell /= 2) {
    const int gw = TW / cell + 2, gh = TH / cell + 2;
    std::vector<float> g((size_t)gw * gh);
    for (auto& x : g) x = (float)(urand() - 0.5) * cell * 0.9f;
    for (int y = 0; y < TH; y++)
      for (int x = 0; x < TW; x++) {
        const int gx = x / cell, gy = y / cell; const float fx = (x % cell) / (float)cell, fy = (y % cell) / (float)cell;
        t[(size_t)y * TW + x] += (g[gy * gw + gx] * (1 - fx) + g[gy * gw + gx + 1] * fx) * (1 - fy) + (g[(gy + 1) * gw + gx] * (1 - fx) + g[(gy + 1) * gw + gx + 1] * fx) * fy;
      }
  }
  for (int k = 0; k < 1100; k++) {
    const int x0 = rnd() % (TW - 60), y0 = rnd() % (TH - 60), w = 6 + rnd() % 40, h = 6 + rnd() % 40; const float val = (rnd() & 1) ? 225.f : 30.f;
    for (int y = y0; y < y0 + h; y++) for (int x = x0; x < x0 + w; x++) t[(size_t)y * TW + x] = val + (float)((x * 7 + y * 13) % 9);
  }
  std::vector<uint8_t> out(t.size());
  for (size_t i = 0; i < t.size(); i++) out[i] = (uint8_t)std::min(255.f, std::max(0.f, t[i]));
  return out;
}
static void rot_xyz(double rx, double ry, double rz, double R[9]) {
  const double cx = std::cos(rx), sx = std::sin(rx), cy = std::cos(ry), sy = std::sin(ry), cz = std::cos(rz), sz = std::sin(rz);
  const double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy}, Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  double T[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T[3 * i + j] = 0; for (int k = 0; k < 3; k++) T[3 * i + j] += Ry[3 * i + k] * Rx[3 * k + j]; }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { R[3 * i + j] = 0; for (int k = 0; k < 3; k++) R[3 * i + j] += Rz[3 * i + k] * T[3 * k + j]; }
}
// the rig over the plane z = 0 (the camera looks along +z from z < 0): a sweep along x with a slow start, a weave in y, height and
// attitude wobble -- 2.5 cm a frame at cruise, so that a keyframe's view is left behind after ~20 keyframes
static void true_pose(int k, double T[16]) {
  const double s = k < 12 ? 0.5 * k * k / 12.0 : k - 6.0;          // eased frame count
  double R[9]; rot_xyz(0.10 * std::sin(0.031 * k + 0.3), 0.08 * std::sin(0.023 * k), 0.05 * std::sin(0.041 * k), R);
  const double C[3] = {1.55 + 0.025 * s, 2.25 + 0.35 * std::sin(0.02 * s), -1.55 - 0.12 * std::sin(0.035 * k)};
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
      const double X = (Ow[0] + s * dw[0]) * PPM, Y = (Ow[1] + s * dw[1]) * PPM;
      const double tx = std::min(std::max(X, 0.0), TW - 1.001), ty = std::min(std::max(Y, 0.0), TH - 1.001);
      const int x0 = (int)tx, y0 = (int)ty; const double fx = tx - x0, fy = ty - y0;
      const double val = (tex[(size_t)y0 * TW + x0] * (1 - fx) + tex[(size_t)y0 * TW + x0 + 1] * fx) * (1 - fy) +
                         (tex[(size_t)(y0 + 1) * TW + x0] * (1 - fx) + tex[(size_t)(y0 + 1) * TW + x0 + 1] * fx) * fy;
      im[(size_t)v * W + u] = (uint8_t)std::lrint(std::min(255.0, std::max(0.0, val)));
    }
  return im;
}
struct Sequence {            // rendered once, read by every run
  std::vector<std::vector<uint8_t>> L, R; std::vector<std::vector<double>> T;
  Sequence(int n) {
    g_seed = 20261004u;
    const std::vector<uint8_t> tex = make_texture();
    for (int k = 0; k < n; k++) {
      double Tl[16], Tr[16]; true_pose(k, Tl);
      for (int i = 0; i < 16; i++) Tr[i] = Tl[i];
      Tr[3] -= BB;
      L.push_back(render(tex, Tl)); R.push_back(render(tex, Tr)); T.emplace_back(Tl, Tl + 16);
    }
  }
};

// ------------------------------------------------------------------------------------------------ float pose algebra (cv::Mat CV_32F in the reference)
static void mul44(const float* A, const float* B, float* C) {
  float t[16];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { float s = 0.f; for (int k = 0; k < 4; k++) s += A[4 * i + k] * B[4 * k + j]; t[4 * i + j] = s; }
  std::memcpy(C, t, sizeof(t));
}
static void inv_pose(const float* T, float* Twc) {                // [Rcw^T | -Rcw^T tcw] (S/Frame.cc:439-445)
  for (int i = 0; i < 16; i++) Twc[i] = (i % 5 == 0) ? 1.f : 0.f;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Twc[4 * i + j] = T[4 * j + i];
  for (int i = 0; i < 3; i++) Twc[4 * i + 3] = -(T[i] * T[3] + T[4 + i] * T[7] + T[8 + i] * T[11]);
}
static void centre_of(const float* T, float Ow[3]) { for (int i = 0; i < 3; i++) Ow[i] = -(T[i] * T[3] + T[4 + i] * T[7] + T[8 + i] * T[11]); }
static Mat mat44f(const float* T) { Mat m(4, 4, 4); std::memcpy(m.ptr<float>(0), T, 64); return m; }

// ------------------------------------------------------------------------------------------------ map objects with the reference's mutator semantics
struct LKeyFrame;
struct LPoint : MapPoint {
  int mnFound = 1; long mnFirstKFid = 0; LKeyFrame* mpRefKF = nullptr; long unsigned mnTrackReferenceForFrame = ~0ul;
  void AddObservation(KeyFrame* kf, int idx) override;          // S/MapPoint.cc:233-262
  void EraseObservation(KeyFrame* kf) override;                 // S/MapPoint.cc:264-311
  void SetBadFlag() override;                                   // S/MapPoint.cc:328-357
  void UpdateNormalAndDepth() override;                         // S/MapPoint.cc:545-603
  float GetFoundRatio() const { return (float)mnFound / (float)mnVisible; }     // S/MapPoint.cc:439-443
};
struct LKeyFrame : KeyFrame {
  std::vector<float> mvDepth;
  std::map<long unsigned, int> weight;                          // connected keyframe id -> shared points (mConnectedKeyFrameWeights)
  long unsigned mnTrackReferenceForFrame = ~0ul;
  LKeyFrame* parent = nullptr;
};
static float g_scale[8];
static std::map<long unsigned, LKeyFrame*> g_kf_by_id;            // (one run at a time)

void LPoint::AddObservation(KeyFrame* kf, int idx) {
  mObservations[kf] = std::make_tuple(idx, -1);
  nObs += kf->mvuRight[idx] >= 0 ? 2 : 1;                         // a stereo observation counts twice (:250-253)
  mnChangeStamp++;
}
void LPoint::EraseObservation(KeyFrame* kf) {
  bool bad = false;
  auto it = mObservations.find(kf);
  if (it != mObservations.end()) {
    const int li = std::get<0>(it->second);
    if (li != -1) nObs -= kf->mvuRight[li] >= 0 ? 2 : 1;
    mObservations.erase(it);
    if (mpRefKF == kf) {                                          // (:301-302: mObservations.begin()->first, by address there; the lowest id here)
      LKeyFrame* low = nullptr;
      for (auto& ob : mObservations) if (!low || ob.first->mnId < low->mnId) low = static_cast<LKeyFrame*>(ob.first);
      mpRefKF = low;
    }
    if (nObs <= 2) bad = true;                                    // :305-306
  }
  mnChangeStamp++;
  if (bad) SetBadFlag();
}
void LPoint::SetBadFlag() {
  mbBad = true;
  auto obs = mObservations;
  mObservations.clear();
  for (auto& ob : obs) { const int li = std::get<0>(ob.second); if (li != -1 && ob.first->mvpMapPoints[li] == this) ob.first->mvpMapPoints[li] = nullptr; }     // EraseMapPointMatch(idx)
  mnChangeStamp++;
}
void LPoint::UpdateNormalAndDepth() {
  n_normal_updates++;
  if (!mbBad && !mObservations.empty() && mpRefKF) {
    std::vector<KeyFrame*> ks;
    for (auto& ob : mObservations) if (std::get<0>(ob.second) != -1) ks.push_back(ob.first);
    std::sort(ks.begin(), ks.end(), [](KeyFrame* a, KeyFrame* b) { return a->mnId < b->mnId; });
    const float* X = mWorldPos.ptr<float>(0);
    float nsum[3] = {0, 0, 0};
    for (KeyFrame* kf : ks) {
      float Ow[3]; const Mat T = kf->GetPose(); centre_of(T.ptr<float>(0), Ow);
      const float d[3] = {X[0] - Ow[0], X[1] - Ow[1], X[2] - Ow[2]};
      const float nn = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      for (int a = 0; a < 3; a++) nsum[a] += d[a] / nn;
    }
    float Or[3]; const Mat Tr = mpRefKF->GetPose(); centre_of(Tr.ptr<float>(0), Or);
    const float pc[3] = {X[0] - Or[0], X[1] - Or[1], X[2] - Or[2]};
    const float dist = std::sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);
    const int level = mpRefKF->mvKeysUn[std::get<0>(mObservations[mpRefKF])].octave;
    mfMaxDistance = dist * g_scale[level];
    mfMinDistance = mfMaxDistance / g_scale[7];
    for (int a = 0; a < 3; a++) mNormalVector.ptr<float>(0)[a] = nsum[a] / (float)ks.size();
  }
  mnChangeStamp++;
}

// ------------------------------------------------------------------------------------------------ frame constructors
struct Features { std::vector<orbx_keypoint> keys; std::vector<uint8_t> desc; std::vector<float> ur, dp; };
struct GpuFront {
  orbgpu::ORBextractor rig{1000, 1.2f, 8, 20, 7, W, H, /*n_cams*/ 2};
  std::unique_ptr<orbgpu::FrameOnDevice> dev[2]; int turn = 0;
  GpuFront() { dev[0].reset(new orbgpu::FrameOnDevice(4096)); dev[1].reset(new orbgpu::FrameOnDevice(4096)); }
  void* build(const uint8_t* L, const uint8_t* R, Features& f) {
    orbm_frame_view v{0, nullptr, nullptr, nullptr, nullptr, 0, (float)W, 0, (float)H, FX, FX, CX, CY, BF, BB, 8, 1.2f};
    turn ^= 1;
    dev[turn]->StereoCtor(rig, v, L, R, W, H, W, &f.keys, &f.desc, &f.ur, &f.dp);
    return dev[turn].get();                                       // the Frame keeps its device-resident copy (mpGpuFrame)
  }
  std::vector<float> inv_sigma2() { return rig.GetInverseScaleSigmaSquares(); }
};
struct OracleFront {
  oracle_extractor *l = nullptr, *r = nullptr;
  OracleFront() {
    orbx_config cfg{}; cfg.n_features = 1000; cfg.scale_factor = 1.2f; cfg.n_levels = 8; cfg.ini_th_fast = 20; cfg.min_th_fast = 7; cfg.max_width = W; cfg.max_height = H; cfg.n_cams = 1;
    oracle_extractor_create(&cfg, &l); oracle_extractor_create(&cfg, &r);
  }
  ~OracleFront() { oracle_extractor_destroy(l); oracle_extractor_destroy(r); }
  void* build(const uint8_t* L, const uint8_t* R, Features& f) {
    const int cap = 4096; int nl = 0, nr = 0, nm = 0;
    std::vector<orbx_keypoint> kr(cap); std::vector<uint8_t> dr((size_t)cap * 32);
    f.keys.resize(cap); f.desc.resize((size_t)cap * 32);
    oracle_extract(l, L, W, H, W, 0, 0, f.keys.data(), f.desc.data(), cap, &nl, &nm);
    oracle_extract(r, R, W, H, W, 0, 0, kr.data(), dr.data(), cap, &nr, &nm);
    f.keys.resize(nl); f.desc.resize((size_t)nl * 32); f.ur.assign(nl, -1.f); f.dp.assign(nl, -1.f);
    oracle_stereo_match(l, r, f.keys.data(), f.desc.data(), nl, kr.data(), dr.data(), nr, BF, BB, f.ur.data(), f.dp.data());
    return nullptr;
  }
  std::vector<float> inv_sigma2() { std::vector<float> s(8), a(8), b(8), c(8); std::vector<int32_t> n(8); oracle_get_tables(l, a.data(), b.data(), c.data(), s.data(), n.data()); return s; }
};

// ------------------------------------------------------------------------------------------------ digests
static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull) { const uint8_t* b = (const uint8_t*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } return h; }
struct Digest {
  int N = 0; uint64_t h_feat = 0;
  int n_frame = 0; uint64_t h_assign1 = 0; int n_inl1 = 0; uint64_t h_outl1 = 0; float pose1[16] = {0};
  int n_local_kfs = 0, n_local_pts = 0; uint64_t h_local = 0;
  int n_local = 0; uint64_t h_assign2 = 0; int n_inl2 = 0; uint64_t h_outl2 = 0; float pose2[16] = {0};
  int n_tracked = 0;
  int n_reloc = -1; uint64_t h_reloc = 0;                          // (frames 31, 81, ..) what Relocalization's refinement leaves on a copy of the frame
  int is_kf = 0, n_new = 0, n_culled = 0, n_fused = 0, lba_status = -1, it1 = 0, it2 = 0, n_fixed = 0, n_bad_total = 0, n_kf_matches = 0; uint64_t h_obs = 0;
  std::vector<float> state;                                      // after the local BA: every good point's position / normal / range (8 values), then every keyframe's pose
  size_t n_point_values = 0;
};
static uint64_t hash_assign(const Frame& F) { std::vector<long> ids(F.N); for (int i = 0; i < F.N; i++) ids[i] = F.mvpMapPoints[i] ? (long)F.mvpMapPoints[i]->mnId : -1; return fnv(ids.data(), ids.size() * sizeof(long)); }
static uint64_t hash_outl(const Frame& F) { std::vector<uint8_t> o(F.N); for (int i = 0; i < F.N; i++) o[i] = F.mvbOutlier[i]; return fnv(o.data(), o.size()); }

// ------------------------------------------------------------------------------------------------ the agent
template <class Ops, class Front>
struct Tracker {
  Front front;
  Map map;
  std::vector<std::unique_ptr<LPoint>> points;
  std::vector<std::unique_ptr<LKeyFrame>> kfs;
  std::unique_ptr<Frame> cur, last;
  float vel[16];
  std::vector<LKeyFrame*> localKFs; std::vector<MapPoint*> localPts;
  std::list<LPoint*> recent;                                      // mlpRecentAddedMapPoints
  LKeyFrame* lastKF = nullptr;
  std::vector<float> inv_sigma2;
  long unsigned next_point_id = 0;
  bool lost = false;

  Tracker() {
    inv_sigma2 = front.inv_sigma2();
    g_scale[0] = 1.f; for (int l = 1; l < 8; l++) g_scale[l] = g_scale[l - 1] * 1.2f;
    for (int i = 0; i < 16; i++) vel[i] = (i % 5 == 0) ? 1.f : 0.f;
    g_kf_by_id.clear();
  }

  std::unique_ptr<Frame> make_frame(const Sequence& seq, int k, Digest& d) {
    Features f;
    void* devp = front.build(seq.L[k].data(), seq.R[k].data(), f);
    std::unique_ptr<Frame> F(new Frame);
    F->mnId = k; F->mpGpuFrame = devp;
    F->mnMinX = 0; F->mnMaxX = W; F->mnMinY = 0; F->mnMaxY = H; F->fx = FX; F->fy = FX; F->cx = CX; F->cy = CY; F->mbf = BF; F->mb = BB;
    const int N = (int)f.keys.size();
    F->N = N; F->mvKeys.resize(N); F->mDescriptors = Mat(N, 32, 1); F->mvuRight = f.ur; F->mvDepth = f.dp;
    for (int i = 0; i < N; i++) F->mvKeys[i] = KeyPoint{{f.keys[i].x, f.keys[i].y}, f.keys[i].size, f.keys[i].angle, f.keys[i].response, f.keys[i].octave};
    F->mvKeysUn = F->mvKeys;
    if (N) std::memcpy(F->mDescriptors.ptr<uint8_t>(0), f.desc.data(), (size_t)N * 32);
    F->mvpMapPoints.assign(N, nullptr); F->mvbOutlier.assign(N, false);
    F->mvInvLevelSigma2 = inv_sigma2;
    for (int i = 0; i < N; i++) F->mFeatVec[f.desc[(size_t)32 * i] >> 5].push_back((unsigned)i);      // ComputeBoW over a stand-in vocabulary (eight words: the leading bits)
    d.N = N;
    d.h_feat = fnv(f.keys.data(), f.keys.size() * sizeof(orbx_keypoint));
    d.h_feat = fnv(f.desc.data(), f.desc.size(), d.h_feat); d.h_feat = fnv(f.ur.data(), f.ur.size() * 4, d.h_feat); d.h_feat = fnv(f.dp.data(), f.dp.size() * 4, d.h_feat);
    return F;
  }

  LPoint* new_point(const Frame& F, int i, LKeyFrame* kf) {      // MapPoint(x3D, pKF, map) with x3D = F.UnprojectStereo(i) (S/Frame.cc:1043-1057)
    const float z = F.mvDepth[i], u = F.mvKeysUn[i].pt.x, v = F.mvKeysUn[i].pt.y;
    const float xc[3] = {(u - CX) * z * (1.0f / FX), (v - CY) * z * (1.0f / FX), z};
    float Twc[16]; inv_pose(F.mTcw.ptr<float>(0), Twc);
    std::unique_ptr<LPoint> p(new LPoint);
    for (int a = 0; a < 3; a++) p->mWorldPos.ptr<float>(0)[a] = Twc[4 * a] * xc[0] + Twc[4 * a + 1] * xc[1] + Twc[4 * a + 2] * xc[2] + Twc[4 * a + 3];
    p->mnId = next_point_id++; p->mpMap = &map; p->mnFirstKFid = (long)kf->mnId; p->mpRefKF = kf;
    LPoint* out = p.get();
    points.push_back(std::move(p));
    return out;
  }
  // descriptor = medoid of the observations' descriptors (S/MapPoint.cc:448-522), for many points in one call of the entry point
  void distinctive(const std::vector<LPoint*>& pts) {
    std::vector<uint8_t> desc; std::vector<int32_t> start{0}; std::vector<LPoint*> who; std::vector<std::vector<std::pair<KeyFrame*, int>>> lists;
    for (LPoint* p : pts) {
      if (p->isBad() || p->mObservations.empty()) continue;
      std::vector<std::pair<KeyFrame*, int>> obs;
      for (auto& ob : p->mObservations) if (!ob.first->isBad() && std::get<0>(ob.second) != -1) obs.push_back({ob.first, std::get<0>(ob.second)});
      if (obs.empty()) continue;
      std::sort(obs.begin(), obs.end(), [](const auto& a, const auto& b) { return a.first->mnId < b.first->mnId; });
      for (auto& o : obs) { const uint8_t* dsc = o.first->mDescriptors.template ptr<uint8_t>(o.second); desc.insert(desc.end(), dsc, dsc + 32); }
      start.push_back((int32_t)(desc.size() / 32)); who.push_back(p); lists.push_back(std::move(obs));
    }
    if (who.empty()) return;
    std::vector<int32_t> best(who.size(), -1);
    orbgpu::check(Ops::distinctive(desc.data(), start.data(), (int)who.size(), best.data()), "ComputeDistinctiveDescriptors");
    for (size_t q = 0; q < who.size(); q++) {
      if (best[q] < 0) continue;
      std::memcpy(who[q]->mDescriptor.template ptr<uint8_t>(0), &desc[32 * (size_t)(start[q] + best[q])], 32);
      who[q]->Touch();
    }
  }

  // KeyFrame::UpdateConnections (S/KeyFrame.cc:430-520), orders by (weight, id)
  static void sort_connections(LKeyFrame* kf) {
    std::vector<std::pair<int, long unsigned>> v;
    for (auto& w : kf->weight) v.push_back({w.second, w.first});
    std::sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
    kf->mvpOrderedConnectedKeyFrames.clear();
    for (auto& e : v) kf->mvpOrderedConnectedKeyFrames.push_back(g_kf_by_id[e.second]);
  }
  void update_connections(LKeyFrame* kf) {
    std::map<long unsigned, int> counter;
    for (MapPoint* mp : kf->mvpMapPoints) {
      if (!mp || mp->isBad()) continue;
      for (auto& ob : mp->mObservations) if (ob.first->mnId != kf->mnId) counter[ob.first->mnId]++;
    }
    if (counter.empty()) return;
    int nmax = 0; long unsigned kmax = 0; bool any = false;
    kf->weight.clear();
    for (auto& c : counter) {
      if (c.second > nmax) { nmax = c.second; kmax = c.first; }
      if (c.second >= 15) { kf->weight[c.first] = c.second; g_kf_by_id[c.first]->weight[kf->mnId] = c.second; sort_connections(g_kf_by_id[c.first]); any = true; }
    }
    if (!any) { kf->weight[kmax] = nmax; g_kf_by_id[kmax]->weight[kf->mnId] = nmax; sort_connections(g_kf_by_id[kmax]); }
    sort_connections(kf);
    if (!kf->parent && kf->mnId != map.GetInitKFid()) kf->parent = static_cast<LKeyFrame*>(kf->mvpOrderedConnectedKeyFrames.front());
  }

  LKeyFrame* make_keyframe(Frame& F) {                            // KeyFrame(mCurrentFrame, map, db)
    std::unique_ptr<LKeyFrame> kf(new LKeyFrame);
    kf->mnId = kfs.size(); kf->fx = FX; kf->fy = FX; kf->cx = CX; kf->cy = CY; kf->mbf = BF; kf->mpMap = &map;
    kf->mvKeysUn = F.mvKeysUn; kf->mvuRight = F.mvuRight; kf->mvDepth = F.mvDepth; kf->mvInvLevelSigma2 = F.mvInvLevelSigma2; kf->mDescriptors = F.mDescriptors;
    kf->mvpMapPoints = F.mvpMapPoints; kf->Tcw = F.mTcw; kf->mFeatVec = F.mFeatVec;
    LKeyFrame* out = kf.get();
    g_kf_by_id[out->mnId] = out;
    kfs.push_back(std::move(kf));
    return out;
  }
  // new map points for the stereo features without one, closest first (S/Tracking.cc:2986-3068 / StereoInitialization :2290-2330)
  int create_stereo_points(Frame& F, LKeyFrame* kf, bool all) {
    std::vector<std::pair<float, int>> byDepth;
    for (int i = 0; i < F.N; i++) if (F.mvDepth[i] > 0) byDepth.push_back({F.mvDepth[i], i});
    std::sort(byDepth.begin(), byDepth.end());
    int nPoints = 0, created = 0;
    std::vector<LPoint*> fresh;
    for (auto& e : byDepth) {
      const int i = e.second;
      bool create = false;
      MapPoint* mp = F.mvpMapPoints[i];
      if (!mp) create = true;
      else if (mp->Observations() < 1) { create = true; F.mvpMapPoints[i] = nullptr; }
      if (create) {
        LPoint* p = new_point(F, i, kf);
        p->AddObservation(kf, i);
        kf->mvpMapPoints[i] = p;
        fresh.push_back(p);
        F.mvpMapPoints[i] = p;
        created++;
      }
      nPoints++;
      if (!all && e.first > TH_DEPTH && nPoints > 100) break;
    }
    distinctive(fresh);
    for (LPoint* p : fresh) p->UpdateNormalAndDepth();
    for (LPoint* p : fresh) recent.push_back(p);                 // (ProcessNewKeyFrame's else-branch, S/LocalMapping.cc:425-428)
    return created;
  }

  void init(const Sequence& seq, Digest& d) {                     // Tracking::StereoInitialization, S/Tracking.cc:2262-2345
    cur = make_frame(seq, 0, d);
    float T0[16]; for (int i = 0; i < 16; i++) T0[i] = (float)seq.T[0][i];      // (the world frame is the sequence's: poses compare with the truth)
    cur->mTcw = mat44f(T0);
    LKeyFrame* kf = make_keyframe(*cur);
    map.mnInitKFid = kf->mnId;
    d.is_kf = 1;
    d.n_new = create_stereo_points(*cur, kf, true);
    lastKF = kf;
    localKFs = {kf};
    localPts.clear();
    for (auto& p : points) localPts.push_back(p.get());
    std::memcpy(d.pose1, T0, 64); std::memcpy(d.pose2, T0, 64);
    finish_keyframe_digest(d);
    last.reset(new Frame(*cur));
  }

  void update_local_map(Frame& F, Digest& d) {                   // S/Tracking.cc:3157-3330
    std::map<long unsigned, int> votes;
    for (int i = 0; i < F.N; i++) {
      MapPoint* mp = F.mvpMapPoints[i];
      if (!mp) continue;
      if (mp->isBad()) { F.mvpMapPoints[i] = nullptr; continue; }
      for (auto& ob : mp->mObservations) votes[ob.first->mnId]++;
    }
    localKFs.clear();
    for (auto& v : votes) { LKeyFrame* kf = g_kf_by_id[v.first]; if (kf->isBad()) continue; localKFs.push_back(kf); kf->mnTrackReferenceForFrame = F.mnId; }
    const size_t n0 = localKFs.size();
    for (size_t q = 0; q < n0 && localKFs.size() <= 80; q++) {   // one more neighbour / parent per local keyframe (:3273-3318)
      LKeyFrame* kf = localKFs[q];
      int seen = 0;
      for (KeyFrame* nb0 : kf->mvpOrderedConnectedKeyFrames) {
        if (seen++ >= 10) break;
        LKeyFrame* nb = static_cast<LKeyFrame*>(nb0);
        if (!nb->isBad() && nb->mnTrackReferenceForFrame != F.mnId) { localKFs.push_back(nb); nb->mnTrackReferenceForFrame = F.mnId; break; }
      }
      if (kf->parent && kf->parent->mnTrackReferenceForFrame != F.mnId) { localKFs.push_back(kf->parent); kf->parent->mnTrackReferenceForFrame = F.mnId; }
    }
    localPts.clear();
    for (auto it = localKFs.rbegin(); it != localKFs.rend(); ++it)
      for (MapPoint* mp0 : (*it)->mvpMapPoints) {
        LPoint* mp = static_cast<LPoint*>(mp0);
        if (!mp || mp->mnTrackReferenceForFrame == F.mnId || mp->isBad()) continue;
        localPts.push_back(mp); mp->mnTrackReferenceForFrame = F.mnId;
      }
    d.n_local_kfs = (int)localKFs.size(); d.n_local_pts = (int)localPts.size();
    std::vector<long> ids; for (auto* k : localKFs) ids.push_back((long)k->mnId); for (auto* p : localPts) ids.push_back((long)p->mnId);
    d.h_local = fnv(ids.data(), ids.size() * sizeof(long));
  }

  // a handful of fusions per keyframe in the spirit of LocalMapping::SearchInNeighbors (S/LocalMapping.cc:868-995): a point this
  // keyframe has just created duplicates an older local point that projects onto the same feature with a close descriptor -> the
  // older point takes the observation, the new one is retired
  int fuse_duplicates(LKeyFrame* kf) {
    const Mat Tm = kf->GetPose(); const float* T = Tm.ptr<float>(0);
    int fused = 0;
    std::vector<LPoint*> changed;
    for (MapPoint* q0 : localPts) {
      LPoint* q = static_cast<LPoint*>(q0);
      if (fused >= 12) break;
      if (q->isBad() || q->mObservations.count(kf)) continue;
      const float* X = q->mWorldPos.ptr<float>(0);
      const float xc = T[0] * X[0] + T[1] * X[1] + T[2] * X[2] + T[3], yc = T[4] * X[0] + T[5] * X[1] + T[6] * X[2] + T[7], zc = T[8] * X[0] + T[9] * X[1] + T[10] * X[2] + T[11];
      if (!(zc > 0.1f)) continue;
      const float u = FX * xc / zc + CX, v = FX * yc / zc + CY;
      if (u < 20 || u > W - 20 || v < 20 || v > H - 20) continue;
      int bi = -1, bd = 51;
      for (size_t i = 0; i < kf->mvKeysUn.size(); i++) {
        LPoint* p = static_cast<LPoint*>(kf->mvpMapPoints[i]);
        if (!p || p->isBad() || p->mnFirstKFid != (long)kf->mnId) continue;       // only this keyframe's brand-new points are candidates
        const float du = kf->mvKeysUn[i].pt.x - u, dv = kf->mvKeysUn[i].pt.y - v;
        if (du * du + dv * dv > 4.0f) continue;
        const int dist = oracle_hamming(q->mDescriptor.ptr<uint8_t>(0), kf->mDescriptors.ptr<uint8_t>((int)i));   // (scaffolding: a plain bit count)
        if (dist < bd) { bd = dist; bi = (int)i; }
      }
      if (bi < 0) continue;
      LPoint* p = static_cast<LPoint*>(kf->mvpMapPoints[bi]);
      p->SetBadFlag();
      kf->mvpMapPoints[bi] = q;
      q->AddObservation(kf, bi);
      changed.push_back(q);
      fused++;
    }
    distinctive(changed);
    for (LPoint* q : changed) q->UpdateNormalAndDepth();
    return fused;
  }

  void finish_keyframe_digest(Digest& d) {
    d.n_bad_total = 0; d.n_kf_matches = 0;
    std::vector<long> obs;
    for (auto& p : points) {
      d.n_bad_total += p->isBad();
      if (p->isBad()) continue;
      obs.push_back((long)p->mnId); obs.push_back(p->nObs); obs.push_back((long)p->mObservations.size());
      const float* X = p->mWorldPos.template ptr<float>(0); const float* n = p->mNormalVector.template ptr<float>(0);
      d.state.insert(d.state.end(), X, X + 3); d.state.insert(d.state.end(), n, n + 3); d.state.push_back(p->mfMinDistance); d.state.push_back(p->mfMaxDistance);
    }
    d.h_obs = fnv(obs.data(), obs.size() * sizeof(long));
    d.n_point_values = d.state.size();
    for (auto& k : kfs) { d.state.insert(d.state.end(), k->Tcw.template ptr<float>(0), k->Tcw.template ptr<float>(0) + 16); for (auto* m : k->mvpMapPoints) d.n_kf_matches += m != nullptr; }
  }

  void keyframe(Frame& F, Digest& d) {
    d.is_kf = 1;
    LKeyFrame* kf = make_keyframe(F);                             // Tracking::CreateNewKeyFrame, S/Tracking.cc:2952-3082
    d.n_new = create_stereo_points(F, kf, false);
    lastKF = kf;
    // ---- LocalMapping::RunClient for this keyframe (S/LocalMapping.cc:140-379), run here before the next frame arrives
    std::vector<LPoint*> touched;                                 // ProcessNewKeyFrame, :396-437
    for (size_t i = 0; i < kf->mvpMapPoints.size(); i++) {
      LPoint* mp = static_cast<LPoint*>(kf->mvpMapPoints[i]);
      if (!mp || mp->isBad() || mp->mObservations.count(kf)) continue;
      mp->AddObservation(kf, (int)i);
      mp->UpdateNormalAndDepth();
      touched.push_back(mp);
    }
    distinctive(touched);
    update_connections(kf);
    d.n_culled = 0;                                               // MapPointCulling, :446-485 (stereo: cnThObs = 3)
    for (auto it = recent.begin(); it != recent.end();) {
      LPoint* mp = *it;
      const int age = (int)kf->mnId - (int)mp->mnFirstKFid;
      if (mp->isBad()) it = recent.erase(it);
      else if (mp->GetFoundRatio() < 0.25f) { mp->SetBadFlag(); d.n_culled++; it = recent.erase(it); }
      else if (age >= 2 && mp->Observations() <= 3) { mp->SetBadFlag(); d.n_culled++; it = recent.erase(it); }
      else if (age >= 3) it = recent.erase(it);
      else ++it;
    }
    d.n_fused = fuse_duplicates(kf);
    update_connections(kf);                                       // SearchInNeighbors ends with it (:993)
    if (kfs.size() > 2) {                                         // :243-246
      bool mbAbortBA = false; int num_fixed = 0;
      d.lba_status = od::LocalBundleAdjustment<Ops>(static_cast<KeyFrame*>(kf), &mbAbortBA, &map, num_fixed, 0);
      d.it1 = Ops::it1(); d.it2 = Ops::it2(); d.n_fixed = num_fixed;
    }
    finish_keyframe_digest(d);
  }

  void step(const Sequence& seq, int k, int kf_every, Digest& d) {
    cur = make_frame(seq, k, d);
    Frame& F = *cur;
    od::FrameFlat host_view; od::flatten_frame<OracleOps>(F, host_view);      // (for the shadow calls: the product itself works on the device-resident frame)
    Shadow::view() = &host_view.v; Shadow::frame_no() = k;
    // ---- TrackWithMotionModel (S/Tracking.cc:2572-2686)
    float Tpred[16]; mul44(vel, last->mTcw.ptr<float>(0), Tpred);
    F.mTcw = mat44f(Tpred);
    bool by_bow = false;
    if (k % 25 == 12 && !kfs.empty()) {
      // ---- TrackReferenceKeyFrame (S/Tracking.cc:2860-2926): SearchByBoW against the reference keyframe, the last pose as the guess
      std::vector<MapPoint*> vm;
      const int nb = od::SearchByBoW<Ops>(static_cast<KeyFrame*>(kfs.back().get()), F, vm, 0.7f, true);
      if (nb >= 15) { F.mvpMapPoints = vm; F.mTcw = last->mTcw; d.n_frame = nb; by_bow = true; bow_frames()++; }
    }
    if (k % 50 == 31 && !kfs.empty()) {
      // ---- Tracking::Relocalization's refinement (S/Tracking.cc:3385-3500) against the last keyframe as the one candidate, on a COPY of
      // the frame (the agent itself goes on with the motion model: a relocalisation that changes its state makes the independent runs part
      // at the first correspondence whose chi2 sits on PoseOptimization's threshold -- tried: frame 91 of 200): SearchByBoW (0.75), the PnP
      // solver's answer stood in for by the last pose and by "the first 40 matches are its inliers" (the solver is not on the path),
      // PoseOptimization, outliers dropped, and -- fewer than 50 inliers -- SearchByProjection(F, pKF, sFound, 10, 100), PoseOptimization,
      // and between 30 and 50 once more with (3, 64)
      Frame R(F);
      KeyFrame* pKF = static_cast<KeyFrame*>(kfs.back().get());
      std::vector<MapPoint*> vm;
      const int nb = od::SearchByBoW<Ops>(pKF, R, vm, 0.75f, true);
      d.n_reloc = 0;
      if (nb >= 15) {
        R.mTcw = last->mTcw;
        std::set<MapPoint*> sFound;
        int taken = 0;
        for (int j = 0; j < R.N; j++) {
          if (vm[j] && taken < 40) { R.mvpMapPoints[j] = vm[j]; sFound.insert(vm[j]); taken++; } else R.mvpMapPoints[j] = nullptr;
        }
        int nGood = od::PoseOptimization<Ops>(&R);
        if (nGood >= 10) {
          for (int io = 0; io < R.N; io++) if (R.mvbOutlier[io]) R.mvpMapPoints[io] = nullptr;
          if (nGood < 50) {
            int nadditional = od::SearchByProjection<Ops>(R, pKF, sFound, 10.f, 100, true);
            if (nadditional + nGood >= 50) {
              nGood = od::PoseOptimization<Ops>(&R);
              if (nGood > 30 && nGood < 50) {
                sFound.clear();
                for (int ip = 0; ip < R.N; ip++) if (R.mvpMapPoints[ip]) sFound.insert(R.mvpMapPoints[ip]);
                nadditional = od::SearchByProjection<Ops>(R, pKF, sFound, 3.f, 64, true);
                if (nGood + nadditional >= 50) { nGood = od::PoseOptimization<Ops>(&R); for (int io = 0; io < R.N; io++) if (R.mvbOutlier[io]) R.mvpMapPoints[io] = nullptr; }
              }
            }
          }
          d.n_reloc = nGood; d.h_reloc = hash_assign(R);
          reloc_frames()++;
        }
      }
    }
    if (!by_bow) {
    d.n_frame = od::SearchByProjection<Ops>(F, *last, 7.0f, false, true);
    if (d.n_frame < 20) { std::fill(F.mvpMapPoints.begin(), F.mvpMapPoints.end(), nullptr); d.n_frame = od::SearchByProjection<Ops>(F, *last, 14.0f, false, true); }
    }
    d.h_assign1 = hash_assign(F);
    if (d.n_frame < 20) { lost = true; return; }
    d.n_inl1 = od::PoseOptimization<Ops>(&F);
    d.h_outl1 = hash_outl(F);
    std::memcpy(d.pose1, F.mTcw.ptr<float>(0), 64);
    int nmatchesMap = 0;
    for (int i = 0; i < F.N; i++) {
      MapPoint* mp = F.mvpMapPoints[i];
      if (!mp) continue;
      if (F.mvbOutlier[i]) { F.mvpMapPoints[i] = nullptr; F.mvbOutlier[i] = false; mp->mbTrackInView = false; mp->mnLastFrameSeen = F.mnId; }
      else if (mp->Observations() > 0) nmatchesMap++;
    }
    if (nmatchesMap < 10) { lost = true; return; }
    // ---- TrackLocalMap (S/Tracking.cc:2689-2808)
    update_local_map(F, d);
    d.n_local = od::SearchLocalPoints<Ops>(F, localPts, /*th (stereo, :3131-3151)*/ 1.0f, /*mbFarPoints*/ false, 50.0f, 0.8f);
    d.h_assign2 = hash_assign(F);
    d.n_inl2 = od::PoseOptimization<Ops>(&F);
    d.h_outl2 = hash_outl(F);
    std::memcpy(d.pose2, F.mTcw.ptr<float>(0), 64);
    d.n_tracked = 0;
    for (int i = 0; i < F.N; i++) {
      MapPoint* mp = F.mvpMapPoints[i];
      if (!mp) continue;
      if (!F.mvbOutlier[i]) { static_cast<LPoint*>(mp)->mnFound++; if (mp->Observations() > 0) d.n_tracked++; }
      else F.mvpMapPoints[i] = nullptr;                           // stereo: :2780-2781
    }
    if (d.n_tracked < 30) { lost = true; return; }
    // ---- motion model, clean-up, keyframe decision (S/Tracking.cc:2140-2215)
    float LastTwc[16]; inv_pose(last->mTcw.ptr<float>(0), LastTwc);
    mul44(F.mTcw.ptr<float>(0), LastTwc, vel);
    for (int i = 0; i < F.N; i++) { MapPoint* mp = F.mvpMapPoints[i]; if (mp && mp->Observations() < 1) { F.mvbOutlier[i] = false; F.mvpMapPoints[i] = nullptr; } }
    if (k % kf_every == 0) keyframe(F, d);
    for (int i = 0; i < F.N; i++) if (F.mvpMapPoints[i] && F.mvbOutlier[i]) F.mvpMapPoints[i] = nullptr;
    last.reset(new Frame(F));
  }
};

template <class Ops, class Front>
static std::vector<Digest> run(const Sequence& seq, int n_frames, int kf_every, const char* name, double* secs) {
  od::LbaWindowCache<KeyFrame, MapPoint>::instance().clear();
  od::local_map_cache<Ops>().invalidate();
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<Digest> out(n_frames);
  {
    Tracker<Ops, Front> trk;
    trk.init(seq, out[0]);
    for (int k = 1; k < n_frames; k++) {
      trk.step(seq, k, kf_every, out[k]);
      if (trk.lost) { std::printf("[%s] tracking lost at frame %d (%d frame matches, %d map inliers)\n", name, k, out[k].n_frame, out[k].n_tracked); out.resize(k + 1); break; }
    }
    int kf = 0, maxw = 0; long bad = 0; double err = 0;
    for (auto& d : out) { kf += d.is_kf; maxw = std::max(maxw, d.n_local_kfs); }
    for (auto& p : trk.points) bad += p->isBad();
    for (int i = 0; i < 16; i++) err = std::max(err, std::fabs((double)out.back().pose2[i] - seq.T[out.size() - 1][i]));
    std::printf("[%s] %zu frames, %d keyframes, %zu map points (%ld bad), local map up to %d keyframes; last pose vs the sequence's truth: %.4f\n", name, out.size(), kf,
                trk.points.size(), bad, maxw, err);
    od::LbaWindowCache<KeyFrame, MapPoint>::instance().clear();
  }
  *secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return out;
}

static float kPoseTol = 1e-4f, kStateTol = 5e-2f, kKfPoseTol = 1e-4f;
static float max_abs(const float* a, const float* b, size_t n) { float m = 0; for (size_t i = 0; i < n; i++) m = std::max(m, std::fabs(a[i] - b[i])); return m; }
// first divergent frame, or -1
static int compare(const std::vector<Digest>& g, const std::vector<Digest>& c, const char* what, float* worst_pose, float* worst_state) {
  *worst_pose = 0; *worst_state = 0;
  if (g.size() != c.size()) { std::printf("DIVERGED [%s]: runs of %zu and %zu frames\n", what, g.size(), c.size()); return (int)std::min(g.size(), c.size()); }
  for (size_t k = 0; k < g.size(); k++) {
    const Digest &a = g[k], &b = c[k];
    const float p1 = max_abs(a.pose1, b.pose1, 16), p2 = max_abs(a.pose2, b.pose2, 16);
#define FIELD(f) if (a.f != b.f) { std::printf("DIVERGED [%s] at frame %zu: %s %lld vs %lld\n", what, k, #f, (long long)a.f, (long long)b.f); return (int)k; }
    FIELD(N) FIELD(h_feat) FIELD(n_frame) FIELD(h_assign1) FIELD(n_inl1) FIELD(h_outl1)
    if (p1 > kPoseTol) { std::printf("DIVERGED [%s] at frame %zu: pose after the first PoseOptimization differs by %g\n", what, k, p1); return (int)k; }
    FIELD(n_local_kfs) FIELD(n_local_pts) FIELD(h_local) FIELD(n_local) FIELD(h_assign2) FIELD(n_inl2) FIELD(h_outl2) FIELD(n_reloc) FIELD(h_reloc)
    if (p2 > kPoseTol) { std::printf("DIVERGED [%s] at frame %zu: pose after the second PoseOptimization differs by %g\n", what, k, p2); return (int)k; }
    if (std::getenv("CLOSED_LOOP_VERBOSE") && (p1 > 1e-7f || p2 > 1e-7f)) std::printf("   [%s] frame %zu: pose differences %.3g / %.3g\n", what, k, p1, p2);
    FIELD(n_tracked) FIELD(is_kf) FIELD(n_new) FIELD(n_culled) FIELD(n_fused) FIELD(lba_status) FIELD(it1) FIELD(it2) FIELD(n_fixed) FIELD(n_bad_total) FIELD(n_kf_matches) FIELD(h_obs)
#undef FIELD
    if (a.state.size() != b.state.size()) { std::printf("DIVERGED [%s] at frame %zu: map state of %zu vs %zu values\n", what, k, a.state.size(), b.state.size()); return (int)k; }
    const float s = a.state.empty() ? 0.f : max_abs(a.state.data(), b.state.data(), a.state.size());
    if (std::getenv("CLOSED_LOOP_VERBOSE") && s > 1e-6f) {
      int n4 = 0, n5 = 0; size_t worst = 0; float wv = 0;
      for (size_t q = 0; q < a.state.size(); q++) { const float e = std::fabs(a.state[q] - b.state[q]); n4 += e > 1e-4f; n5 += e > 1e-5f; if (e > wv) { wv = e; worst = q; } }
      float cat[4] = {0, 0, 0, 0};      // position, normal, distance range, keyframe pose
      for (size_t q = 0; q < a.state.size(); q++) { const float e = std::fabs(a.state[q] - b.state[q]); const int c = q >= a.n_point_values ? 3 : (q % 8) < 3 ? 0 : (q % 8) < 6 ? 1 : 2; cat[c] = std::max(cat[c], e); }
      std::printf("   [%s] frame %zu: map state: %zu values, %d differ by > 1e-5, %d by > 1e-4; worst: position %.3g normal %.3g range %.3g keyframe pose %.3g\n", what, k, a.state.size(), n5, n4, cat[0], cat[1], cat[2], cat[3]);
      (void)worst;
    }
    if (s > kStateTol) { std::printf("DIVERGED [%s] at frame %zu: map state after the local BA differs by %g\n", what, k, s); return (int)k; }
    if (a.state.size() > a.n_point_values) {
      const float kp = max_abs(a.state.data() + a.n_point_values, b.state.data() + a.n_point_values, a.state.size() - a.n_point_values);
      if (kp > kKfPoseTol) { std::printf("DIVERGED [%s] at frame %zu: keyframe poses after the local BA differ by %g\n", what, k, kp); return (int)k; }
    }
    *worst_pose = std::max(*worst_pose, std::max(p1, p2)); *worst_state = std::max(*worst_state, s);
  }
  return -1;
}

int main(int argc, char** argv) {
  int n_frames = 200, kf_every = 5; bool oracle_only = false;
  int pos = 0;
  for (int i = 1; i < argc; i++) {
    if (!std::strcmp(argv[i], "--oracle-only")) oracle_only = true;
    else if (!std::strncmp(argv[i], "--pose-tol=", 11)) kPoseTol = (float)std::atof(argv[i] + 11);
    else if (!std::strncmp(argv[i], "--state-tol=", 12)) kStateTol = (float)std::atof(argv[i] + 12);
    else if (pos == 0) { n_frames = std::atoi(argv[i]); pos++; }
    else if (pos == 1) { kf_every = std::atoi(argv[i]); pos++; }
  }
  if (n_frames < 3 || kf_every < 1) { std::printf("usage: closed_loop [n_frames >= 3] [kf_every >= 1] [--oracle-only]\n"); return 2; }
  try {
    const Sequence seq(n_frames);
    double tc = 0, tg = 0, tu = 0;
    const std::vector<Digest> c = run<OracleLoopOps, OracleFront>(seq, n_frames, kf_every, "oracle entry points", &tc);
    int n_kf = 0, n_lba = 0, max_it = 0; long matched = 0;
    for (auto& d : c) { n_kf += d.is_kf; n_lba += d.lba_status == LBA_APPLIED; max_it = std::max(max_it, d.it1 + d.it2); matched += d.n_tracked; }
    if ((int)c.size() != n_frames) { std::printf("the oracle run lost track: the scenario is broken\n"); return 1; }
    if (oracle_only) { std::printf("oracle-only run: %d keyframes, %d applied local BAs, %.1f tracked points a frame, %d frames through SearchByBoW, %d through the relocalisation search, %.1f s\n", n_kf, n_lba, (double)matched / n_frames, bow_frames(), reloc_frames(), tc); return 0; }
    const std::vector<Digest> g = run<GpuLoopOps, GpuFront>(seq, n_frames, kf_every, "product entry points (change-counter caches)", &tg);
    float wp = 0, ws = 0, wp2 = 0, ws2 = 0;
    const int d1 = compare(g, c, "product vs oracle", &wp, &ws);
    const std::vector<Digest> u = run<GpuUnmodifiedLoopOps, GpuFront>(seq, n_frames, kf_every, "product entry points (unmodified MapPoint: no caches)", &tu);
    const int d2 = compare(u, c, "product without caches vs oracle", &wp2, &ws2);
    int bit_equal = 0;                         // leading frames on which even the float digests are identical
    for (size_t k = 0; k < g.size() && k < c.size(); k++) { if (std::memcmp(g[k].pose1, c[k].pose1, 64) || std::memcmp(g[k].pose2, c[k].pose2, 64) || g[k].state != c[k].state) break; bit_equal++; }
    std::printf("closed loop: %d frames, %d keyframes, %d applied local BAs (up to %d LM iterations), %.1f tracked points a frame\n", n_frames, n_kf, n_lba, max_it, (double)matched / n_frames);
    std::printf("  shadow: %ld entry-point calls of the product runs repeated on the oracle with identical inputs, %ld mismatches; worst PoseOptimization pose difference %.3g, worst local-BA state difference %.3g\n",
                Shadow::calls(), Shadow::fails(), Shadow::worst_pose(), Shadow::worst_lba());
    std::printf("  independent runs: discrete digests equal on every frame: %s / %s; first %d frames bit-identical; max pose difference %.3g / %.3g, max map-state difference %.3g / %.3g\n",
                d1 < 0 ? "yes" : "NO", d2 < 0 ? "yes" : "NO", bit_equal, wp, wp2, ws, ws2);
    std::printf("  wall: %.1f s oracle, %.1f s product (+ shadow), %.1f s product without caches (+ shadow)\n", tc, tg, tu);
    const bool ok = d1 < 0 && d2 < 0 && Shadow::fails() == 0;
    std::printf("{\"closed_loop\": {\"frames\": %d, \"keyframes\": %d, \"local_bas_applied\": %d, \"shadow_calls\": %ld, \"shadow_mismatches\": %ld, \"shadow_max_pose_abs_diff\": %.3g, "
                "\"shadow_max_lba_abs_diff\": %.3g, \"first_divergent_frame\": %d, \"first_divergent_frame_no_caches\": %d, \"bit_identical_leading_frames\": %d, "
                "\"max_pose_abs_diff\": %.3g, \"max_map_state_abs_diff\": %.3g, \"frames_through_search_by_bow_per_run\": %d, \"frames_through_the_relocalisation_search_per_run\": %d, \"ok\": %s}}\n", n_frames, n_kf, n_lba, Shadow::calls(), Shadow::fails(), Shadow::worst_pose(), Shadow::worst_lba(),
                d1, d2, bit_equal, std::max(wp, wp2), std::max(ws, ws2), bow_frames() / 3, reloc_frames() / 3, ok ? "true" : "false");
    if (!ok) return 1;
    std::printf("closed loop ok\n");
    return 0;
  } catch (const std::runtime_error& e) {
    std::printf("runtime_error: %s\n", e.what());
    return 3;
  }
}
