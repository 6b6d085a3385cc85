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

struct GpuHeuristicOps : od::GpuOps {   // the product's entry points with the OPT-IN counter-less cache of the local map (-DORBGPU_DROPIN_HEURISTIC_LOCAL_MAP)
  static constexpr bool kNoChangeStamp = true;
  static constexpr bool kHeuristicLocalMap = true;
};
struct GpuUnmodifiedOps : od::GpuOps {  // the product's entry points as they compile against the reference's MapPoint as it is (+ edit E1): no counter, no cache
  static constexpr bool kNoChangeStamp = true;
  static constexpr bool kNoLbaCache = true;
};
#include "oracle_ops.hpp"

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

#include "scenario.hpp"

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

struct TrackOut { int n_visible, n_map, n_frame, n_bow, n_pose, n_fused, vis_sum; std::vector<int> a_map, a_frame, a_bow, a_fused; std::vector<bool> outl; std::vector<float> pose;
                  std::vector<std::vector<int>> a_varied; std::vector<int> vis_varied;
                  std::vector<int> a_cached, a_stale, a_moved, a_delta, a_delta_fresh;
                  int n_reloc1 = 0, n_reloc2 = 0; std::vector<int> a_reloc; };

template <class Ops>
static TrackOut run_tracking(orbgpu::ORBextractor& rig, const std::vector<uint8_t>& tex) {
  Agent A; TrackOut o;
  for (int k : {4, 5, 6}) make_frame(A, rig, tex, k, /*the frame being tracked stays on the device (product run)*/ k == 6);
  Frame &F0 = *A.frames[0], &F1 = *A.frames[1], &F2 = *A.frames[2];
  // ---- Tracking::SearchLocalPoints: isInFrustum + SearchByProjection(F, local map points)
  std::vector<MapPoint*> local; std::vector<int> src;
  make_points_from(A, F0, local, src);
  o.n_visible = od::isInFrustumAll<Ops>(F2, local, 0.5f);
  o.n_map = od::SearchByProjection<Ops>(F2, local, 3.0f, false, 50.0f, 0.8f);
  o.a_map = assignment_ids(F2, local);
  // ---- the same through the fused body of Tracking::SearchLocalPoints: some features already hold a point (those points are
  // skipped through mnLastFrameSeen), one of them is bad (dropped from the frame), a few local points are bad
  {
    std::fill(F2.mvpMapPoints.begin(), F2.mvpMapPoints.end(), nullptr);
    F2.mnId = 77;
    for (size_t j = 0; j < local.size(); j++) { local[j]->mbTrackInView = false; local[j]->mnVisible = 1; local[j]->mnLastFrameSeen = ~0ul; local[j]->mbBad = (j % 41) == 7; }
    int held = 0;
    for (int i = 0; i < F2.N && held < 60; i++) if (o.a_map[i] >= 0) { F2.mvpMapPoints[i] = local[o.a_map[i]]; held++; }
    o.n_fused = od::SearchLocalPoints<Ops>(F2, local, 3.0f, false, 50.0f, 0.8f);
    o.a_fused = assignment_ids(F2, local);
    o.vis_sum = 0;
    for (auto* p : local) o.vis_sum += p->mnVisible + 1000 * (int)p->mbTrackInView;
    // ---- the glue's cache of the flattened local map (round 4): the same call again (same pointers, same change index: statics and
    // device map reused), points moved WITHOUT the map's change index (the cache may not notice: the documented staleness until the
    // next write-back), the change index moved (everything re-read), another pointer sequence (known points copied, new ones read)
    auto again = [&](const std::vector<MapPoint*>& pts, unsigned long id) {
      std::fill(F2.mvpMapPoints.begin(), F2.mvpMapPoints.end(), nullptr);
      F2.mnId = id;
      for (auto* p : local) { p->mbTrackInView = false; p->mnVisible = 1; }
      int held2 = 0;
      for (int i = 0; i < F2.N && held2 < 60; i++) if (o.a_map[i] >= 0) { F2.mvpMapPoints[i] = local[o.a_map[i]]; held2++; }
      od::SearchLocalPoints<Ops>(F2, pts, 3.0f, false, 50.0f, 0.8f);
      return assignment_ids(F2, local);
    };
    o.a_cached = again(local, 78);
    // ---- per-frame state that CHANGES between the call that uploaded the resident map and the calls that reuse it (same pointers,
    // same change index): other features hold points (the previous frame's exclusions must not stick to the map), points turn bad,
    // Observations() move, and LocalMapping re-describes points without touching the change index (ProcessNewKeyFrame: observation
    // + descriptor + distance range, S/LocalMapping.cc:405-425) -- each call against the oracle's set, which reads everything fresh
    {
      auto varied = [&](unsigned long id, int hold_from, int hold_to) {
        std::fill(F2.mvpMapPoints.begin(), F2.mvpMapPoints.end(), nullptr);
        F2.mnId = id;
        for (auto* p : local) { p->mbTrackInView = false; p->mnVisible = 1; }
        int seen = 0;
        for (int i = 0; i < F2.N; i++) if (o.a_map[i] >= 0) { if (seen >= hold_from && seen < hold_to) F2.mvpMapPoints[i] = local[o.a_map[i]]; seen++; }
        od::SearchLocalPoints<Ops>(F2, local, 3.0f, false, 50.0f, 0.8f);
        o.a_varied.push_back(assignment_ids(F2, local));
        int vs = 0;
        for (auto* p : local) vs += p->mnVisible + 1000 * (int)p->mbTrackInView;
        o.vis_varied.push_back(vs);
      };
      struct Saved { int nObs; Mat desc; float dmin, dmax; };
      std::vector<Saved> keep;
      for (auto* p : local) keep.push_back(Saved{p->nObs, p->mDescriptor, p->mfMinDistance, p->mfMaxDistance});
      od::local_map_cache<Ops>().invalidate();
      varied(90, 0, 60);                                   // uploads the map: features 0..59 of the matched ones hold their points
      varied(91, 60, 120);                                 // cached map, OTHER held features: the first 60 points are candidates again
      varied(92, 0, 0);                                    // nothing held
      for (size_t j = 3; j < local.size(); j += 29) { local[j]->mbBad = true; local[j]->Touch(); }       // points culled by LocalMapping between two frames
      varied(93, 30, 90);
      for (size_t j = 1; j < local.size(); j += 7) { local[j]->nObs = (local[j]->nObs + 1) % 3; local[j]->Touch(); }     // Observations() moved (0 lets a feature be overwritten, S/ORBmatcher.cc:89-91)
      varied(94, 30, 90);
      for (size_t j = 2; j < local.size(); j += 5) {       // ProcessNewKeyFrame: one more observation, another medoid descriptor, a wider distance range
        MapPoint* p = local[j], *q = local[(j + 11) % local.size()];
        p->nObs += 1; p->mDescriptor = q->mDescriptor; p->mfMaxDistance *= 1.3f; p->mfMinDistance *= 0.8f; p->Touch();
      }
      varied(95, 10, 50);
      varied(96, 50, 10);                                  // (nothing held) the re-described points once more, from the refreshed cache
      for (size_t j = 0; j < local.size(); j++) { local[j]->nObs = keep[j].nObs; local[j]->mDescriptor = keep[j].desc; local[j]->mfMinDistance = keep[j].dmin; local[j]->mfMaxDistance = keep[j].dmax; local[j]->Touch(); }
    }
    od::local_map_cache<Ops>().invalidate();
    for (size_t j = 0; j < local.size(); j++) local[j]->mbBad = (j % 41) == 7;
    again(local, 78);                                      // (the cache is warm again, as after a_cached)
    std::vector<Mat> saved;
    for (size_t j = 0; j < local.size(); j += 3) {         // moved by somebody's SetWorldPos, the map's change index untouched
      saved.push_back(local[j]->mWorldPos);
      Mat X = local[j]->mWorldPos; X.ptr<float>(0)[2] += 40.0f; local[j]->SetWorldPos(X);
    }
    o.a_stale = again(local, 79);
    A.map.IncreaseChangeIndex();
    o.a_moved = again(local, 80);
    { size_t q = 0; for (size_t j = 0; j < local.size(); j += 3) local[j]->SetWorldPos(saved[q++]); }
    A.map.IncreaseChangeIndex();
    again(local, 81);
    std::vector<MapPoint*> other(local.begin() + 100, local.end());
    std::swap(other[3], other[200]); std::swap(other[10], other[11]);
    std::vector<MapPoint*> extra; std::vector<int> extra_src;
    make_points_from(A, F1, extra, extra_src);
    other.insert(other.begin() + 50, extra.begin(), extra.begin() + 40);
    auto ids_in = [&](const std::vector<MapPoint*>& pts) { std::vector<int> r(F2.N, -1); for (int i = 0; i < F2.N; i++) if (F2.mvpMapPoints[i]) r[i] = (int)(std::find(pts.begin(), pts.end(), F2.mvpMapPoints[i]) - pts.begin()); return r; };
    again(other, 82); o.a_delta = ids_in(other);
    od::local_map_cache<Ops>().invalidate();
    again(other, 83); o.a_delta_fresh = ids_in(other);
    for (auto* p : local) p->mbBad = false;
    std::fill(F2.mvpMapPoints.begin(), F2.mvpMapPoints.end(), nullptr);
  }
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
  // ---- Tracking::Relocalization (S/Tracking.cc:3372-3410): the BoW matches stay in the frame, SearchByProjection(F, pKF, sFound, 10, 100)
  // adds more, then a narrower pass (3, 64) with the enlarged set
  {
    F2.mvpMapPoints = bow;
    for (auto* p : lastpts) p->mbBad = false;
    for (size_t j = 5; j < lastpts.size(); j += 37) lastpts[j]->mbBad = true;
    std::set<MapPoint*> sFound;
    for (auto* p : F2.mvpMapPoints) if (p) sFound.insert(p);
    o.n_reloc1 = od::SearchByProjection<Ops>(F2, kf.get(), sFound, 10.0f, 100, true);
    for (auto* p : F2.mvpMapPoints) if (p) sFound.insert(p);
    o.n_reloc2 = od::SearchByProjection<Ops>(F2, kf.get(), sFound, 3.0f, 64, true);
    o.a_reloc = assignment_ids(F2, lastpts);
    for (auto* p : lastpts) p->mbBad = false;
  }
  return o;
}

// ------------------------------------------------------------------------------------------------ local BA scenario
struct BaOut { int status, num_fixed, erased, change_index, locked_poses, locked_points; std::vector<float> poses, points; std::vector<int> normal_updates; };

// raise_after_us >= 0: a second thread (Tracking calling LocalMapping::InterruptBA, S/LocalMapping.cc:381-386) sets the bool
// that many microseconds after the call starts
template <class Ops>
static BaOut run_lba(int n_local, int n_far, int n_pts, double outlier_frac, bool stop, unsigned seed, double raise_after_us = -1.0,
                     bool two_fisheye_rig = false) {
  od::LbaWindowCache<KeyFrame, MapPoint>::instance().clear();       // (the points of the last scene are gone: MapPoints never are, in the reference)
  Agent A; BaOut o;
  KeyFrame* cur = two_fisheye_rig ? build_lba_rig_scene(A, n_local, n_far, n_pts, outlier_frac, seed)
                                  : build_lba_scene(A, n_local, n_far, n_pts, outlier_frac, seed);
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
    EXPECT(g.n_fused == c.n_fused && g.a_fused == c.a_fused && g.vis_sum == c.vis_sum && g.n_fused > 100, "SearchLocalPoints (fused) %d vs %d, visible sums %d vs %d",
           g.n_fused, c.n_fused, g.vis_sum, c.vis_sum);
    auto ndiff = [](const std::vector<int>& a, const std::vector<int>& b) { int d = (int)(a.size() != b.size()); for (size_t i = 0; i < a.size() && i < b.size(); i++) d += a[i] != b[i]; return d; };
    // the local-map cache of the glue: with the mocks' change counter (exact) and with the counter-less rules
    auto check_local_map_cache = [&](const TrackOut& g, bool stamped, const char* what) {
      EXPECT(g.n_fused == c.n_fused && g.a_fused == c.a_fused && g.vis_sum == c.vis_sum, "[%s] SearchLocalPoints (fused) %d vs %d", what, g.n_fused, c.n_fused);
      EXPECT(g.a_cached == g.a_fused && c.a_cached == c.a_fused, "[%s] SearchLocalPoints on the cached local map differs from the first call (%d / %d features)", what,
             ndiff(g.a_cached, g.a_fused), ndiff(c.a_cached, c.a_fused));
      EXPECT(g.a_varied.size() == 7 && c.a_varied.size() == 7, "[%s] varied-state calls missing", what);
      for (size_t q = 0; q < g.a_varied.size() && q < c.a_varied.size(); q++) {
        int nm = 0; for (int v : g.a_varied[q]) nm += v >= 0;
        EXPECT(g.a_varied[q] == c.a_varied[q] && g.vis_varied[q] == c.vis_varied[q] && nm > 100,
               "[%s] SearchLocalPoints with changing per-frame state, call %zu: %d features differ from the oracle's fresh read (visible sums %d vs %d, %d matched)", what, q,
               ndiff(g.a_varied[q], c.a_varied[q]), g.vis_varied[q], c.vis_varied[q], nm);
      }
      EXPECT(g.vis_varied[1] != g.vis_varied[2] && ndiff(g.a_varied[2], g.a_varied[3]) > 0 && ndiff(g.a_varied[4], g.a_varied[5]) + ndiff(g.a_varied[3], g.a_varied[4]) > 0,
             "[%s] the varied-state calls do not exercise anything (visible sums %d %d; %d / %d / %d features differ between consecutive calls)", what, g.vis_varied[1],
             g.vis_varied[2], ndiff(g.a_varied[2], g.a_varied[3]), ndiff(g.a_varied[3], g.a_varied[4]), ndiff(g.a_varied[4], g.a_varied[5]));
      if (stamped)      // SetWorldPos moved the points' counters: seen at once, change index or not
        EXPECT(g.a_stale == c.a_stale && ndiff(g.a_stale, g.a_fused) > 20, "[%s] points moved by SetWorldPos: %d features differ from the oracle's fresh read, %d from the unmoved map", what,
               ndiff(g.a_stale, c.a_stale), ndiff(g.a_stale, g.a_fused));
      else              // without a counter only the map's change index tells: the documented staleness until the next write-back
        EXPECT(g.a_stale == g.a_fused, "[%s] moved points were noticed without a change of the map's change index (%d features)", what, ndiff(g.a_stale, g.a_fused));
      EXPECT(g.a_moved == c.a_moved && ndiff(g.a_moved, g.a_fused) > 20, "[%s] after IncreaseChangeIndex: %d features differ between the entry-point sets, %d from the unmoved map", what,
             ndiff(g.a_moved, c.a_moved), ndiff(g.a_moved, g.a_fused));
      EXPECT(g.a_delta == g.a_delta_fresh && c.a_delta == c.a_delta_fresh && g.a_delta == c.a_delta, "[%s] another pointer sequence: patched cache vs fresh read %d / %d, GPU vs oracle %d", what,
             ndiff(g.a_delta, g.a_delta_fresh), ndiff(c.a_delta, c.a_delta_fresh), ndiff(g.a_delta, c.a_delta));
      int nd = 0; for (int v : g.a_delta) nd += v >= 0;
      EXPECT(nd > 100, "[%s] the patched local map matched only %d features", what, nd);
    };
    check_local_map_cache(g, true, "change counter");
    g_seed = 7; const TrackOut gu = run_tracking<GpuUnmodifiedOps>(rig, tex);
    check_local_map_cache(gu, true, "unmodified MapPoint: every call reads every point (the default)");
    g_seed = 7; const TrackOut gh = run_tracking<GpuHeuristicOps>(rig, tex);
    check_local_map_cache(gh, false, "opt-in counter-less cache");
    EXPECT(g.n_bow == c.n_bow && g.a_bow == c.a_bow && g.n_bow > 20, "SearchByBoW %d vs %d", g.n_bow, c.n_bow);
    EXPECT(g.n_reloc1 == c.n_reloc1 && g.n_reloc2 == c.n_reloc2 && g.a_reloc == c.a_reloc && g.n_reloc1 > 20,
           "SearchByProjection(F, KF, sAlreadyFound): %d / %d vs %d / %d new matches", g.n_reloc1, g.n_reloc2, c.n_reloc1, c.n_reloc2);
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
    // ---- keyframes of the two-fisheye rig (mpCamera2 != NULL, NLeft != -1): a keyframe observes a point with the left camera, the right
    // one or both -- monocular edges through KannalaBrandt8, EdgeSE3ProjectXYZToBody for the right camera (S/Optimizer.cc:2021-2120) --
    // through the glue with the window cache (the mocks have the change counter) and without it
    {
      const BaOut bg = run_lba<od::GpuOps>(8, 4, 500, 0.03, false, 123, -1.0, true);
      const BaOut bu = run_lba<GpuUnmodifiedOps>(8, 4, 500, 0.03, false, 123, -1.0, true);
      const BaOut bc = run_lba<OracleOps>(8, 4, 500, 0.03, false, 123, -1.0, true);
      std::printf("LocalBundleAdjustment [two-fisheye rig]: status %d, %d fixed KFs, %d observations erased\n", bg.status, bg.num_fixed, bg.erased);
      EXPECT(bg.status == LBA_APPLIED && bc.status == LBA_APPLIED && bu.status == LBA_APPLIED, "rig window: status %d / %d / %d", bg.status, bu.status, bc.status);
      EXPECT(bg.num_fixed == bc.num_fixed && bg.erased > 0 && std::abs(bg.erased - bc.erased) <= 1, "rig window: %d fixed / %d erased vs %d / %d", bg.num_fixed,
             bg.erased, bc.num_fixed, bc.erased);
      EXPECT(max_abs_diff(bg.poses, bc.poses) <= 1e-4f && max_abs_diff(bg.points, bc.points) <= 1e-4f, "rig window: poses %g points %g",
             max_abs_diff(bg.poses, bc.poses), max_abs_diff(bg.points, bc.points));
      EXPECT(max_abs_diff(bg.poses, bu.poses) == 0.f && max_abs_diff(bg.points, bu.points) == 0.f && bg.erased == bu.erased,
             "rig window: the window cache changes the result (poses %g points %g)", max_abs_diff(bg.poses, bu.poses), max_abs_diff(bg.points, bu.points));
    }
    // ---- PoseOptimization of a Frame of that rig (S/Optimizer.cc:1085-1151)
    {
      auto run = [](auto ops_tag, std::vector<bool>& outl, std::vector<float>& pose) {
        using Ops = decltype(ops_tag);
        Agent A;
        Frame* F = build_rig_frame(A, 300, 220, 0.1, 321);
        const int n = od::PoseOptimization<Ops>(F);
        outl = F->mvbOutlier; pose.assign(F->mTcw.ptr<float>(0), F->mTcw.ptr<float>(0) + 16);
        return n;
      };
      std::vector<bool> og, oc; std::vector<float> pg, pc;
      const int ng = run(od::GpuOps{}, og, pg), nc = run(OracleOps{}, oc, pc);
      int nd = 0, nb = 0; for (size_t i = 0; i < og.size(); i++) { nd += og[i] != oc[i]; nb += og[i]; }
      std::printf("PoseOptimization [two-fisheye rig]: %d inliers, %d flagged\n", ng, nb);
      EXPECT(std::abs(ng - nc) <= 1 && nd <= 1 && ng > 350 && nb >= 20, "rig frame: inliers %d vs %d, %d flags differ, %d flagged", ng, nc, nd, nb);
      EXPECT(max_abs_diff(pg, pc) <= 1e-4f, "rig frame: pose differs by %g", max_abs_diff(pg, pc));
    }
    // ---- the matcher on a Frame of that rig (Nleft != -1): Tracking::SearchLocalPoints (isInFrustum through either camera, the
    // right camera's block of SearchByProjection, stereo partners) and SearchByProjection(CurrentFrame, LastFrame)
    {
      struct RigOut { std::vector<long> a_local, a_frame, a_bow; int n_local = 0, n_frame = 0, n_bow = 0, vis = 0, seen = 0; std::vector<float> fields; };
      auto run = [](auto ops_tag, double dz) {
        using Ops = decltype(ops_tag);
        RigOut o;
        Agent A;
        RigTrack S = build_rig_track_scene(A, 4242, 1200, 250, dz);
        auto ids = [](const Frame& F, std::vector<long>& out) { out.clear(); for (MapPoint* p : F.mvpMapPoints) out.push_back(p ? (long)p->mnId : -1); };
        Frame F1(*S.cur);                                          // SearchByProjection(Cur, Last) as TrackWithMotionModel calls it: on an empty frame
        std::fill(F1.mvpMapPoints.begin(), F1.mvpMapPoints.end(), nullptr);
        o.n_frame = od::SearchByProjection<Ops>(F1, *S.last, 7.0f, false, true);
        ids(F1, o.a_frame);
        { std::vector<MapPoint*> vm;                               // SearchByBoW(KF, F) as TrackReferenceKeyFrame calls it
          o.n_bow = od::SearchByBoW<Ops>(S.ref, *S.cur, vm, 0.7f, true);
          for (MapPoint* p : vm) o.a_bow.push_back(p ? (long)p->mnId : -1); }
        o.n_local = od::SearchLocalPoints<Ops>(*S.cur, S.local, 3.0f, true, 7.0f, 0.8f);
        ids(*S.cur, o.a_local);
        for (MapPoint* p : S.local) {
          o.vis += p->mnVisible; o.seen += p->mnLastFrameSeen == S.cur->mnId;
          if (p->mbTrackInView) { o.fields.push_back(p->mTrackProjX); o.fields.push_back(p->mTrackProjY); o.fields.push_back(p->mTrackDepth); o.fields.push_back(p->mTrackViewCos); o.fields.push_back((float)p->mnTrackScaleLevel); }
          else o.fields.push_back(-7.f);
          if (p->mbTrackInViewR) { o.fields.push_back(p->mTrackProjXR); o.fields.push_back(p->mTrackProjYR); o.fields.push_back(p->mTrackDepthR); o.fields.push_back(p->mTrackViewCosR); o.fields.push_back((float)p->mnTrackScaleLevelR); }
          else o.fields.push_back((float)p->mnTrackScaleLevelR - 9.f);
        }
        return o;
      };
      for (double dz : {0.02, 0.4, -0.4}) {
        const RigOut g = run(od::GpuOps{}, dz), c = run(OracleOps{}, dz);
        int nl = 0, nf = 0; for (size_t i = 0; i < g.a_local.size(); i++) nl += g.a_local[i] != c.a_local[i];
        for (size_t i = 0; i < g.a_frame.size(); i++) nf += g.a_frame[i] != c.a_frame[i];
        std::printf("two-camera Frame (last frame %+.2f m): SearchByProjection(Cur, Last) %d matches, SearchLocalPoints %d matches, SearchByBoW(KF, F) %d matches\n", dz, g.n_frame, g.n_local, g.n_bow);
        EXPECT(g.n_frame == c.n_frame && nf == 0 && g.n_frame > 150, "rig frame: SearchByProjection(Cur, Last) %d vs %d, %d features differ", g.n_frame, c.n_frame, nf);
        EXPECT(g.n_local == c.n_local && nl == 0 && g.n_local > 300, "rig frame: SearchLocalPoints %d vs %d, %d features differ", g.n_local, c.n_local, nl);
        EXPECT(g.n_bow == c.n_bow && g.a_bow == c.a_bow && g.n_bow > 100, "rig frame: SearchByBoW(KF, F) %d vs %d matches", g.n_bow, c.n_bow);
        EXPECT(g.vis == c.vis && g.seen == c.seen && g.seen > 20, "rig frame: visible sums %d vs %d, seen %d vs %d", g.vis, c.vis, g.seen, c.seen);
        EXPECT(g.fields.size() == c.fields.size() && max_abs_diff(g.fields, c.fields) <= 3e-4f, "rig frame: track fields differ by %g (%zu vs %zu values)",
               g.fields.size() == c.fields.size() ? max_abs_diff(g.fields, c.fields) : -1.f, g.fields.size(), c.fields.size());
      }
    }
    // ---- a MONOCULAR fisheye Frame (Nleft == -1, mpCamera a KannalaBrandt8, no mpCamera2): isInFrustum and SearchByProjection(Cur, Last)
    // project through the camera model (S/Frame.cc:489, S/ORBmatcher.cc:2012)
    {
      struct MonoOut { std::vector<long> a_local, a_frame, a_reloc; int n_local = 0, n_frame = 0, n_reloc = 0, vis = 0; };
      auto run = [](auto ops_tag) {
        using Ops = decltype(ops_tag);
        MonoOut o;
        Agent A;
        RigTrack S = build_rig_track_scene(A, 4343, 1200, 250, 0.02);
        auto mono_of = [](const Frame& F) {                          // the left camera alone
          Frame M(F);
          const int nl = F.Nleft;
          M.Nleft = -1; M.Nright = -1; M.N = nl; M.mpCamera2 = nullptr;
          M.mvKeys.resize(nl); M.mvKeysUn = M.mvKeys; M.mvKeysRight.clear(); M.mvuRight.assign(nl, -1.f); M.mvDepth.assign(nl, -1.f);
          M.mvpMapPoints.resize(nl); M.mvbOutlier.resize(nl);
          Mat d(nl, 32, 1); std::memcpy(d.ptr<uint8_t>(0), F.mDescriptors.ptr<uint8_t>(0), (size_t)nl * 32); M.mDescriptors = d;
          M.mvLeftToRightMatch.clear(); M.mvRightToLeftMatch.clear(); M.mFeatVec.clear();
          return M;
        };
        Frame cur = mono_of(*S.cur), last = mono_of(*S.last);
        auto ids = [](const Frame& F, std::vector<long>& out) { out.clear(); for (MapPoint* p : F.mvpMapPoints) out.push_back(p ? (long)p->mnId : -1); };
        Frame F1(cur);
        std::fill(F1.mvpMapPoints.begin(), F1.mvpMapPoints.end(), nullptr);
        o.n_frame = od::SearchByProjection<Ops>(F1, last, 15.0f, true, true);
        ids(F1, o.a_frame);
        { Frame F2(cur);                                           // Relocalization's guided search against the reference keyframe's points
          std::fill(F2.mvpMapPoints.begin(), F2.mvpMapPoints.end(), nullptr);
          std::set<MapPoint*> already; for (size_t i = 0; i < S.local.size(); i += 9) already.insert(S.local[i]);
          S.ref->mvKeysUn = S.ref->mvKeys; S.ref->mvKeysUn.insert(S.ref->mvKeysUn.end(), S.ref->mvKeysRight.begin(), S.ref->mvKeysRight.end());
          o.n_reloc = od::SearchByProjection<Ops>(F2, S.ref, already, 10.0f, 100, true);
          ids(F2, o.a_reloc); }
        o.n_local = od::SearchLocalPoints<Ops>(cur, S.local, 3.0f, true, 7.0f, 0.8f);
        ids(cur, o.a_local);
        for (MapPoint* p : S.local) o.vis += p->mnVisible;
        return o;
      };
      const MonoOut g = run(od::GpuOps{}), c = run(OracleOps{});
      {                                                            // ... and with the Frame's features already on the device (Frame::mpGpuFrame, INTEGRATION.md edit E3)
        Agent A;
        RigTrack S = build_rig_track_scene(A, 4343, 1200, 250, 0.02);
        Frame cur(*S.cur);
        const int nl = cur.Nleft;
        cur.Nleft = -1; cur.Nright = -1; cur.N = nl; cur.mpCamera2 = nullptr; cur.mvKeys.resize(nl); cur.mvKeysUn = cur.mvKeys; cur.mvKeysRight.clear();
        cur.mvuRight.assign(nl, -1.f); cur.mvDepth.assign(nl, -1.f); cur.mvpMapPoints.resize(nl); cur.mvbOutlier.resize(nl);
        { Mat d(nl, 32, 1); std::memcpy(d.ptr<uint8_t>(0), S.cur->mDescriptors.ptr<uint8_t>(0), (size_t)nl * 32); cur.mDescriptors = d; }
        od::FrameFlat host; od::flatten_frame<OracleOps>(cur, host);
        orbgpu::FrameOnDevice dev(4096);
        dev.Upload(host.v);
        cur.mpGpuFrame = &dev;
        const int n_res = od::SearchLocalPoints<od::GpuOps>(cur, S.local, 3.0f, true, 7.0f, 0.8f);
        std::vector<long> a_res; for (MapPoint* p : cur.mvpMapPoints) a_res.push_back(p ? (long)p->mnId : -1);
        EXPECT(n_res == g.n_local && a_res == g.a_local, "mono fisheye with a device-resident frame: SearchLocalPoints %d vs %d", n_res, g.n_local);
      }
      std::printf("monocular fisheye Frame: SearchByProjection(Cur, Last) %d matches, SearchLocalPoints %d matches, relocalisation search %d matches\n", g.n_frame, g.n_local, g.n_reloc);
      EXPECT(g.n_reloc == c.n_reloc && g.a_reloc == c.a_reloc && g.n_reloc > 100, "mono fisheye: SearchByProjection(F, KF, found) %d vs %d", g.n_reloc, c.n_reloc);
      EXPECT(g.n_frame == c.n_frame && g.a_frame == c.a_frame && g.n_frame > 100, "mono fisheye: SearchByProjection(Cur, Last) %d vs %d", g.n_frame, c.n_frame);
      EXPECT(g.n_local == c.n_local && g.a_local == c.a_local && g.vis == c.vis && g.n_local > 200, "mono fisheye: SearchLocalPoints %d vs %d, visible sums %d vs %d", g.n_local, c.n_local, g.vis, c.vis);
    }
    // ---- Frame::ComputeStereoFishEyeMatches of the two-fisheye Frame constructor (S/Frame.cc:1093-1150)
    {
      auto run = [](auto ops_tag, std::vector<int>& l2r, std::vector<int>& r2l, std::vector<float>& dep, std::vector<float>& pts) {
        using Ops = decltype(ops_tag);
        Agent A;
        Frame* F = build_fisheye_ctor_frame(A, 6161);
        const int n = od::ComputeStereoFishEyeMatches<Ops>(*F);
        l2r = F->mvLeftToRightMatch; r2l = F->mvRightToLeftMatch; dep = F->mvDepth; pts.clear();
        for (int i = 0; i < F->Nleft; i++) for (int a = 0; a < 3; a++) pts.push_back(l2r[i] >= 0 ? F->mvStereo3Dpoints[i].ptr<float>(0)[a] : 0.f);
        return n;
      };
      std::vector<int> lg, rg, lc, rc2; std::vector<float> dg, dc, pg, pc;
      const int ng = run(od::GpuOps{}, lg, rg, dg, pg), nc = run(OracleOps{}, lc, rc2, dc, pc);
      std::printf("ComputeStereoFishEyeMatches [two-fisheye rig]: %d matches of %zu + %zu features\n", ng, lg.size(), rg.size());
      EXPECT(ng == nc && lg == lc && rg == rc2 && ng > 150, "fisheye constructor: %d vs %d matches, partner arrays %s", ng, nc, (lg == lc && rg == rc2) ? "equal" : "differ");
      EXPECT(max_abs_diff(dg, dc) <= 1e-3f && max_abs_diff(pg, pc) <= 1e-3f, "fisheye constructor: depths differ by %g, points by %g", max_abs_diff(dg, dc), max_abs_diff(pg, pc));
    }
    // ---- five consecutive windows of one map (a new keyframe each, points moved / observations erased by the solves in between): the
    // product through the glue's window cache, the oracle reading every point of every window
    {
      auto windows = [](auto ops_tag) {
        using Ops = decltype(ops_tag);
        od::LbaWindowCache<KeyFrame, MapPoint>::instance().clear();
        Agent A; BaOut o{};
        KeyFrame* cur = build_lba_scene(A, 10, 5, 700, 0.03, 77);
        bool mbAbortBA = false;
        for (int w = 0; w < 5; w++) {
          o.status = od::LocalBundleAdjustment<Ops>(cur, &mbAbortBA, &A.map, o.num_fixed, 0);
          cur = next_keyframe(A, cur, w);
        }
        o.change_index = A.map.GetMapChangeIndex(); o.erased = 0;
        long copies = 0;
        for (auto& kf : A.kfs) { o.poses.insert(o.poses.end(), kf->Tcw.ptr<float>(0), kf->Tcw.ptr<float>(0) + 16); for (auto* p : kf->mvpMapPoints) o.erased += p == nullptr; }
        for (auto& p : A.points) { o.points.insert(o.points.end(), p->mWorldPos.ptr<float>(0), p->mWorldPos.ptr<float>(0) + 3); o.normal_updates.push_back(p->n_normal_updates); copies += p->n_obs_copies; }
        o.locked_points = (int)copies;
        od::LbaWindowCache<KeyFrame, MapPoint>::instance().clear();
        return o;
      };
      const BaOut wg = windows(od::GpuOps{}), wc = windows(OracleOps{});
      std::printf("LocalBundleAdjustment, five consecutive windows: %d observations erased, change index %d, GetObservations copies %d (window cache) vs %d\n",
                  wg.erased, wg.change_index, wg.locked_points, wc.locked_points);
      EXPECT(wg.status == wc.status && wg.erased == wc.erased && wg.change_index == wc.change_index && wg.change_index == 5, "windows: status %d vs %d, erased %d vs %d",
             wg.status, wc.status, wg.erased, wc.erased);
      EXPECT(max_abs_diff(wg.poses, wc.poses) <= 1e-4f && max_abs_diff(wg.points, wc.points) <= 1e-4f && wg.normal_updates == wc.normal_updates,
             "windows: poses %g points %g", max_abs_diff(wg.poses, wc.poses), max_abs_diff(wg.points, wc.points));
      EXPECT(2 * wg.locked_points < wc.locked_points, "the window cache re-read %d points where the uncached glue read %d", wg.locked_points, wc.locked_points);
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
