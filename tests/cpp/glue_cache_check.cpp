// The glue's local-BA window cache (include/orbgpu_dropin.hpp, LbaWindowCache) against the uncached glue, host only: two identical
// scenes go through six windows each with the same changes in between (observations added and erased, points moved, made bad, a new
// keyframe) and DURING each solve (observations added by "another thread" between the glue's read and its write-back), one through an entry-point set that allows the cache, one through a set that forbids it; every flattened problem, every
// status and the final map must be identical, and the cached run must have read far fewer points.
//   g++ -O2 -std=c++17 -I include -I tests/cpp tests/cpp/glue_cache_check.cpp -L multi_orbslam3_amd -lorbgpu -o glue_cache_check
#include <cstdio>
#include <cstring>
#include <vector>

#include "scenario.hpp"

namespace od = orbgpu::dropin;

struct Captured {
  std::vector<float> poses, points; std::vector<uint8_t> fixed; std::vector<lba_edge> edges;
};
static std::vector<Captured> g_cap[2];
static int g_which = 0;
static Agent* g_agent = nullptr;          // the scene of the run in progress: "another thread" changes points DURING the solve
static int g_round = 0;

template <bool NoCache>
struct CaptureOps {
  static constexpr bool kUsesResidentFrame = false;
  static constexpr bool kNoLbaCache = NoCache;
  static int lba(const lba_problem& p, const volatile bool*, lba_result& r) {
    Captured c;
    c.poses.assign(p.poses, p.poses + 16 * (size_t)p.n_poses); c.points.assign(p.points, p.points + 3 * (size_t)p.n_points);
    c.fixed.assign(p.pose_fixed, p.pose_fixed + p.n_poses); c.edges.assign(p.edges, p.edges + p.n_edges);
    g_cap[g_which].push_back(std::move(c));
    // while the solve runs (no map mutex is held, S/Optimizer.cc:2127-2263) another thread -- Tracking creating points, the Communicator
    // applying a server update -- gives a few of the window's points a new observation.  The glue read those points BEFORE the solve
    // and writes them back AFTER it: its record of such a point is stale the moment the write-back ends, and must not be kept.
    for (auto& up : g_agent->points) {
      MapPoint* mp = up.get();
      if (mp->isBad()) continue;
      const unsigned key = (unsigned)(mp->mnId * 2246822519u) ^ (unsigned)(g_round * 7919u);
      if (key % 53 != 2) continue;
      KeyFrame* kf = g_agent->kfs[(key >> 7) % g_agent->kfs.size()].get();
      if (mp->mObservations.count(kf)) continue;
      const int li = (int)kf->mvKeysUn.size();
      kf->mvKeysUn.push_back(KeyPoint{{120.f + (float)(key % 380), 90.f + (float)(key % 280)}, 31.f, 0.f, 20.f, (int)(key % 4)});
      kf->mvuRight.push_back((key & 2) ? -1.f : 70.f + (float)(key % 280));
      kf->mvpMapPoints.push_back(mp);
      mp->AddObservation(kf, li);
    }
    // a deterministic "solve": every free pose and every point moves a little, one edge in 41 is an outlier
    std::memcpy(r.poses, p.poses, sizeof(float) * 16 * (size_t)p.n_poses);
    for (int i = 0; i < p.n_poses; i++) if (!p.pose_fixed[i]) r.poses[16 * i + 3] += 0.001f * (float)(i % 7);
    for (int j = 0; j < p.n_points; j++) for (int a = 0; a < 3; a++) r.points[3 * j + a] = p.points[3 * j + a] + 0.0005f * (float)((j + a) % 5);
    for (int k = 0; k < p.n_edges; k++) { r.edge_outlier[k] = (k % 41) == 7; r.edge_depth_pos[k] = 1; r.edge_chi2[k] = 1.0; }
    r.status = LBA_APPLIED;
    return ORBG_OK;
  }
};

template <class Ops>
static void run(int which, long* obs_copies, long* pos_clones) {
  g_which = which;
  Agent B;
  KeyFrame* cur = build_lba_scene(B, 21, 10, 2000, 0.03, 99);
  bool stop = false; int nf = 0;
  g_agent = &B;
  for (int round = 0; round < 6; round++) {
    g_round = round;
    od::LocalBundleAdjustment<Ops>(cur, &stop, &B.map, nf, 0);
    // what LocalMapping / LoopClosing do between two windows, on points chosen by id (the scenes are built identically)
    for (size_t j = 0; j < B.points.size(); j++) {
      MapPoint* mp = B.points[j].get();
      if (mp->isBad()) continue;
      const unsigned key = (unsigned)(mp->mnId * 2654435761u) ^ (unsigned)(round * 40503u);
      if (key % 97 == 3 && !mp->mObservations.empty()) {                                                           // KeyFrameCulling
        KeyFrame* lowest = nullptr;                          // (by id: the map's own order is by address and differs between the two scenes)
        for (auto& ob : mp->mObservations) if (!lowest || ob.first->mnId < lowest->mnId) lowest = ob.first;
        mp->EraseObservation(lowest);
      }
      if (key % 89 == 5) { Mat X = mp->GetWorldPos(); X.ptr<float>(0)[1] += 0.01f; mp->SetWorldPos(X); }          // a correction
      if (key % 211 == 11) mp->SetBadFlag();                                                                       // MapPointCulling
      if (key % 83 == 9) {                                                                                         // SearchInNeighbors: a new observation
        KeyFrame* kf = B.kfs[(key >> 8) % B.kfs.size()].get();
        if (!mp->mObservations.count(kf)) {
          const int li = (int)kf->mvKeysUn.size();
          kf->mvKeysUn.push_back(KeyPoint{{100.f + (float)(key % 400), 80.f + (float)(key % 300)}, 31.f, 0.f, 20.f, (int)(key % 4)});
          kf->mvuRight.push_back((key & 1) ? -1.f : 60.f + (float)(key % 300));
          kf->mvpMapPoints.push_back(mp);
          mp->AddObservation(kf, li);
        }
      }
    }
    // the next window belongs to a NEW keyframe: covisible with the last one and most of its neighbours, seeing two thirds of its points
    {
      std::unique_ptr<KeyFrame> kf(new KeyFrame(*cur));
      kf->mnId = cur->mnId + 1; kf->mnBALocalForKF = ~0ul; kf->mnBAFixedForKF = ~0ul;
      kf->mvKeysUn.clear(); kf->mvuRight.clear(); kf->mvpMapPoints.clear();
      kf->Tcw.ptr<float>(0)[3] -= 0.1f;
      kf->mvpOrderedConnectedKeyFrames.clear();
      for (size_t q = 1; q < cur->mvpOrderedConnectedKeyFrames.size(); q++) kf->mvpOrderedConnectedKeyFrames.push_back(cur->mvpOrderedConnectedKeyFrames[q]);
      kf->mvpOrderedConnectedKeyFrames.push_back(cur);
      for (size_t li0 = 0; li0 < cur->mvpMapPoints.size(); li0++) {
        MapPoint* mp = cur->mvpMapPoints[li0];
        if (!mp || mp->isBad() || (mp->mnId + round) % 3 == 0) continue;
        const int li = (int)kf->mvKeysUn.size();
        KeyPoint kp = cur->mvKeysUn[li0]; kp.pt.x += 3.f;
        kf->mvKeysUn.push_back(kp); kf->mvuRight.push_back(cur->mvuRight[li0] < 0 ? -1.f : cur->mvuRight[li0] + 3.f); kf->mvpMapPoints.push_back(mp);
        mp->AddObservation(kf.get(), li);
      }
      cur = kf.get();
      B.kfs.push_back(std::move(kf));
    }
  }
  (void)cur;
  long a = 0, b = 0;
  for (auto& mp : B.points) { a += mp->n_obs_copies; b += mp->n_pos_clones; }
  *obs_copies = a; *pos_clones = b;
}

int main() {
  long oc[2], pc[2];
  run<CaptureOps<false>>(0, &oc[0], &pc[0]);
  od::LbaWindowCache<KeyFrame, MapPoint>::instance().clear();
  run<CaptureOps<true>>(1, &oc[1], &pc[1]);
  int bad = 0;
  if (g_cap[0].size() != g_cap[1].size()) { std::printf("different numbers of solves\n"); return 1; }
  for (size_t w = 0; w < g_cap[0].size(); w++) {
    const Captured &a = g_cap[0][w], &b = g_cap[1][w];
    const bool same = a.poses == b.poses && a.points == b.points && a.fixed == b.fixed && a.edges.size() == b.edges.size() &&
                      (a.edges.empty() || !std::memcmp(a.edges.data(), b.edges.data(), sizeof(lba_edge) * a.edges.size()));
    std::printf("window %zu: %zu poses, %zu points, %zu edges: %s\n", w, a.fixed.size(), a.points.size() / 3, a.edges.size(), same ? "identical" : "DIFFERENT");
    bad += !same;
    if (!same) {
      std::printf("   poses %d points %d fixed %d edges n %zu / %zu\n", a.poses == b.poses, a.points == b.points, a.fixed == b.fixed, a.edges.size(), b.edges.size());
      for (size_t k = 0; k < std::min(a.points.size(), b.points.size()); k++) if (a.points[k] != b.points[k]) { std::printf("   first point difference at %zu: %.7f vs %.7f\n", k, a.points[k], b.points[k]); break; }
      for (size_t k = 0; k < std::min(a.edges.size(), b.edges.size()); k++) if (std::memcmp(&a.edges[k], &b.edges[k], sizeof(lba_edge))) { std::printf("   first edge difference at %zu: (%d %d %.2f) vs (%d %d %.2f)\n", k, a.edges[k].pose, a.edges[k].point, a.edges[k].u, b.edges[k].pose, b.edges[k].point, b.edges[k].u);
        const int pt = a.edges[k].point;
        for (int z = 0; z < 2; z++) { const Captured& c = z ? b : a; std::printf("     %s point %d:", z ? "uncached" : "cached  ", pt); for (auto& e : c.edges) if (e.point == pt) std::printf(" (%d %.1f)", e.pose, e.u); std::printf("\n"); }
        break; }
    }
  }
  std::printf("GetObservations copies: cached %ld, uncached %ld; GetWorldPos clones: cached %ld, uncached %ld\n", oc[0], oc[1], pc[0], pc[1]);
  if (!(oc[0] * 3 < oc[1])) { std::printf("the cache did not spare the observation copies\n"); bad++; }
  // ---- the cache's bookkeeping alone: 12000 points seen in an old window, 1500 in recent ones -> the old records are dropped, the
  // table is rebuilt, every recent point is still found under its address and no address maps to a foreign record
  {
    auto& C = od::LbaWindowCache<KeyFrame, MapPoint>::instance();
    C.clear();
    std::vector<std::unique_ptr<MapPoint>> pts;
    for (int i = 0; i < 12000; i++) { pts.emplace_back(new MapPoint); pts.back()->mnId = 50000 + i; }
    C.call = 1;
    for (auto& p : pts) { auto& r = C.recs[C.find_or_add(p.get())]; r.id = p->mnId; r.valid = true; r.seen = 1; }
    C.call = 20;
    for (int i = 0; i < 1500; i++) C.recs[C.find_or_add(pts[8 * i].get())].seen = 20;
    const size_t before = C.recs.size();
    C.drop_unseen(1500);
    int wrong = 0;
    for (int i = 0; i < 1500; i++) { const auto& r = C.recs[C.find_or_add(pts[8 * i].get())]; wrong += !(r.mp == pts[8 * i].get() && r.valid && r.id == pts[8 * i]->mnId); }
    const size_t kept = C.recs.size();
    const int32_t again = C.find_or_add(pts[1].get());             // a dropped point comes back as a fresh (invalid) record
    wrong += C.recs[again].valid || C.recs[again].mp != pts[1].get();
    std::printf("cache bookkeeping: %zu records before, %zu after dropping the unseen, %d wrong look-ups\n", before, kept, wrong);
    if (before != 12000 || kept != 1500 || wrong) bad++;
    C.clear();
  }
  std::printf(bad ? "FAILED\n" : "ALL OK\n");
  return bad ? 1 : 0;
}
