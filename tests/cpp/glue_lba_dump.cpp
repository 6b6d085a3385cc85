// Host only: a local-BA window of mock KeyFrame / MapPoint objects goes through the glue (include/orbgpu_dropin.hpp,
// LocalBundleAdjustment) with an entry-point set that RECORDS the flattened problem and returns "aborted" (nothing is written back);
// the scene and the recorded problem are printed as JSON.  tests/test_reference_formulas.py rebuilds the scene as Python stand-ins, runs
// the reference's own graph construction (S/Optimizer.cc:1810-2124, transliterated) on them and compares what g2o would have been
// given with what the glue gave the C-ABI.
//   g++ -O1 -std=c++17 -I include -I tests/cpp tests/cpp/glue_lba_dump.cpp -L multi_orbslam3_amd -lorbgpu -o glue_lba_dump
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "scenario.hpp"

namespace od = orbgpu::dropin;

struct Captured { std::vector<float> poses, points; std::vector<uint8_t> fixed; std::vector<lba_edge> edges; double lambda_init; bool has_rig, has_right; };
static Captured g_cap;

struct RecordOps {
  static constexpr bool kUsesResidentFrame = false;
  static constexpr bool kNoLbaCache = true;
  static int lba(const lba_problem& p, const volatile bool*, lba_result& r) {
    g_cap.poses.assign(p.poses, p.poses + 16 * (size_t)p.n_poses); g_cap.points.assign(p.points, p.points + 3 * (size_t)p.n_points);
    g_cap.fixed.assign(p.pose_fixed, p.pose_fixed + p.n_poses); g_cap.edges.assign(p.edges, p.edges + p.n_edges);
    g_cap.lambda_init = p.lambda_init; g_cap.has_rig = p.rig != nullptr; g_cap.has_right = p.rig && p.rig->has_right;
    r.status = LBA_ABORTED_BEFORE_OPT;
    return ORBG_OK;
  }
};

static void dump_floats(const char* name, const float* v, size_t n) {
  std::printf("\"%s\": [", name);
  for (size_t i = 0; i < n; i++) std::printf("%s%.9g", i ? ", " : "", v[i]);
  std::printf("]");
}

static void dump(const char* scene_name, Agent& A, Map& other_map, KeyFrame* cur, int num_fixed, int status) {
  std::printf("{\"scene\": \"%s\", \"current_kf\": %lu, \"init_kf\": %lu, \"inertial\": %d, \"num_fixed\": %d, \"status\": %d,\n \"kfs\": [", scene_name, cur->mnId,
              A.map.GetInitKFid(), (int)A.map.IsInertial(), num_fixed, status);
  for (size_t k = 0; k < A.kfs.size(); k++) {
    KeyFrame* kf = A.kfs[k].get();
    std::printf("%s\n  {\"id\": %lu, \"client\": %d, \"bad\": %d, \"map\": %d, \"fx\": %.9g, \"fy\": %.9g, \"cx\": %.9g, \"cy\": %.9g, \"mbf\": %.9g, \"NLeft\": %d, \"camera2\": %d, ",
                k ? "," : "", kf->mnId, (int)kf->mnClientId, (int)kf->isBad(), kf->GetMap() == &A.map ? 0 : 1, kf->fx, kf->fy, kf->cx, kf->cy, kf->mbf, kf->NLeft, kf->mpCamera2 ? 1 : 0);
    dump_floats("pose", kf->Tcw.ptr<float>(0), 16);
    std::printf(", \"covisible\": [");
    for (size_t i = 0; i < kf->mvpOrderedConnectedKeyFrames.size(); i++) std::printf("%s%lu", i ? ", " : "", kf->mvpOrderedConnectedKeyFrames[i]->mnId);
    std::printf("], \"matches\": [");
    for (size_t i = 0; i < kf->mvpMapPoints.size(); i++) std::printf("%s%ld", i ? ", " : "", kf->mvpMapPoints[i] ? (long)kf->mvpMapPoints[i]->mnId : -1L);
    std::printf("], \"keysUn\": [");
    for (size_t i = 0; i < kf->mvKeysUn.size(); i++) std::printf("%s[%.9g, %.9g, %d]", i ? ", " : "", kf->mvKeysUn[i].pt.x, kf->mvKeysUn[i].pt.y, kf->mvKeysUn[i].octave);
    std::printf("], \"keysRight\": [");
    for (size_t i = 0; i < kf->mvKeysRight.size(); i++) std::printf("%s[%.9g, %.9g, %d]", i ? ", " : "", kf->mvKeysRight[i].pt.x, kf->mvKeysRight[i].pt.y, kf->mvKeysRight[i].octave);
    std::printf("], ");
    dump_floats("uRight", kf->mvuRight.data(), kf->mvuRight.size());
    std::printf(", ");
    dump_floats("invSigma2", kf->mvInvLevelSigma2.data(), kf->mvInvLevelSigma2.size());
    std::printf("}");
  }
  std::printf("],\n \"mps\": [");
  for (size_t j = 0; j < A.points.size(); j++) {
    MapPoint* mp = A.points[j].get();
    std::printf("%s\n  {\"id\": %lu, \"client\": %d, \"bad\": %d, \"map\": %d, ", j ? "," : "", mp->mnId, (int)mp->mnClientId, (int)mp->isBad(), mp->GetMap() == &A.map ? 0 : 1);
    dump_floats("pos", mp->mWorldPos.ptr<float>(0), 3);
    std::printf(", \"obs\": [");
    bool first = true;
    for (const auto& ob : mp->mObservations) {
      std::printf("%s[%lu, %d, %d]", first ? "" : ", ", ob.first->mnId, std::get<0>(ob.second), std::get<1>(ob.second));
      first = false;
    }
    std::printf("]}");
  }
  std::printf("],\n \"problem\": {");
  dump_floats("poses", g_cap.poses.data(), g_cap.poses.size());
  std::printf(", \"fixed\": [");
  for (size_t i = 0; i < g_cap.fixed.size(); i++) std::printf("%s%d", i ? ", " : "", (int)g_cap.fixed[i]);
  std::printf("], ");
  dump_floats("points", g_cap.points.data(), g_cap.points.size());
  std::printf(", \"lambda_init\": %.9g, \"has_rig\": %d, \"has_right\": %d, \"edges\": [", g_cap.lambda_init, (int)g_cap.has_rig, (int)g_cap.has_right);
  for (size_t k = 0; k < g_cap.edges.size(); k++)
    std::printf("%s[%d, %d, %.9g, %.9g, %.9g, %.9g]", k ? ", " : "", g_cap.edges[k].pose, g_cap.edges[k].point, g_cap.edges[k].u, g_cap.edges[k].v, g_cap.edges[k].ur, g_cap.edges[k].inv_sigma2);
  std::printf("]}}\n");
}

int main() {
  struct Scene { const char* name; int n_local, n_far, n_pts; bool rig; int variant; } scenes[] = {
      {"regular window", 8, 4, 300, false, 0}, {"no fixed keyframes: the two lowest ids get fixed", 6, 0, 200, false, 0},
      {"bad keyframe, bad point, keyframe and point of another map, inertial map", 8, 3, 300, false, 1}, {"two-fisheye rig", 7, 3, 250, true, 0}};
  for (const Scene& sc : scenes) {
    od::LbaWindowCache<KeyFrame, MapPoint>::instance().clear();
    Agent A; Map other;
    KeyFrame* cur = sc.rig ? build_lba_rig_scene(A, sc.n_local, sc.n_far, sc.n_pts, 0.03, 4242) : build_lba_scene(A, sc.n_local, sc.n_far, sc.n_pts, 0.03, 4242);
    if (sc.variant == 1) {
      A.kfs[sc.n_far + 2]->mbBad = true;                    // a covisible keyframe that is bad: not local, its observations make no edges
      A.kfs[1]->mpMap = &other;                             // an older keyframe of another map: never a fixed camera
      A.points[5]->mbBad = true; A.points[17]->mbBad = true;
      A.points[9]->mpMap = &other;
      A.map.mbIsInertial = true;                            // lambda init 100 (S/Optimizer.cc:1924-1925)
    }
    bool stop = false; int num_fixed = -1;
    const int status = od::LocalBundleAdjustment<RecordOps>(cur, &stop, &A.map, num_fixed, 0);
    dump(sc.name, A, other, cur, num_fixed, status);
  }
  return 0;
}
