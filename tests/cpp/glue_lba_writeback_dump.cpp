// Host only: a local-BA window of mock KeyFrame / MapPoint objects goes through the glue (include/orbgpu_dropin.hpp,
// LocalBundleAdjustment) with an entry-point set that records the flattened problem and ANSWERS with a synthetic solve -- moved poses
// and points, some observations flagged, one point turned bad while "solving" -- or with the 50 %-outlier refusal; the scene before,
// the synthetic answer and everything the glue's write-back left in the objects are printed as JSON.  tests/test_reference_formulas.py
// runs the reference's own text of that part (S/Optimizer.cc:2205-2400: vToErase, the map mutex, the erasures, SetPose / SetWorldPos /
// UpdateNormalAndDepth, IncreaseChangeIndex) on Python stand-ins of the same window and compares the final states.
#include <cstdio>
#include <cstring>
#include <vector>

#include "scenario.hpp"

namespace od = orbgpu::dropin;

struct Answer {
  std::vector<lba_edge> edges; std::vector<long> edge_kf, edge_mp; std::vector<uint8_t> eout; std::vector<float> poses_in, poses_out, points_in, points_out;
  std::vector<long> pose_kf, point_mp; std::vector<uint8_t> fixed; bool has_right = false; long turned_bad = -1;
};
static Answer g_ans;
static Agent* g_agent = nullptr;
static int g_status = LBA_APPLIED;

struct AnswerOps {
  static constexpr bool kUsesResidentFrame = false;
  static constexpr bool kNoLbaCache = true;
  static int lba(const lba_problem& p, const volatile bool*, lba_result& r) {
    Answer& a = g_ans; a = Answer();
    a.has_right = p.rig && p.rig->has_right;
    a.edges.assign(p.edges, p.edges + p.n_edges); a.fixed.assign(p.pose_fixed, p.pose_fixed + p.n_poses);
    a.poses_in.assign(p.poses, p.poses + 16 * (size_t)p.n_poses); a.points_in.assign(p.points, p.points + 3 * (size_t)p.n_points);
    for (int i = 0; i < p.n_poses; i++) {                        // which keyframe / point an index is: by value (ids are unique, so are the values)
      long id = -1;
      for (auto& kf : g_agent->kfs) if (!std::memcmp(kf->Tcw.ptr<float>(0), p.poses + 16 * (size_t)i, 64)) id = (long)kf->mnId;
      a.pose_kf.push_back(id);
    }
    for (int j = 0; j < p.n_points; j++) {
      long id = -1;
      for (auto& mp : g_agent->points) if (!std::memcmp(mp->mWorldPos.ptr<float>(0), p.points + 3 * (size_t)j, 12)) id = (long)mp->mnId;
      a.point_mp.push_back(id);
    }
    r.status = g_status;
    if (g_status != LBA_APPLIED) return ORBG_OK;
    for (int i = 0; i < p.n_poses; i++) {
      std::memcpy(r.poses + 16 * (size_t)i, p.poses + 16 * (size_t)i, 64);
      if (!p.pose_fixed[i]) { r.poses[16 * (size_t)i + 3] += 0.01f * (float)(i + 1); r.poses[16 * (size_t)i + 7] -= 0.005f * (float)(i + 1); }
    }
    for (int j = 0; j < p.n_points; j++) for (int c = 0; c < 3; c++) r.points[3 * (size_t)j + c] = p.points[3 * (size_t)j + c] + 0.001f * (float)((j % 5) - 2) * (float)(c + 1);
    a.poses_out.assign(r.poses, r.poses + 16 * (size_t)p.n_poses); a.points_out.assign(r.points, r.points + 3 * (size_t)p.n_points);
    for (int k = 0; k < p.n_edges; k++) {
      r.edge_outlier[k] = (k % 7) == 3; r.edge_depth_pos[k] = (k % 14) != 3; r.edge_chi2[k] = r.edge_outlier[k] && r.edge_depth_pos[k] ? 50.0 : 1.0;
      a.edge_kf.push_back(a.pose_kf[p.edges[k].pose]); a.edge_mp.push_back(a.point_mp[p.edges[k].point]);
    }
    a.eout.assign(r.edge_outlier, r.edge_outlier + p.n_edges);
    // another thread turns a point bad while the solve runs: its flagged observation must stay (S/Optimizer.cc:2214-2215)
    for (auto& mp : g_agent->points) if ((long)mp->mnId == a.edge_mp[3]) { mp->mbBad = true; a.turned_bad = (long)mp->mnId; }
    r.iters_round1 = 5; r.iters_round2 = 10;
    return ORBG_OK;
  }
};

static void dump_floats(const char* name, const float* v, size_t n) {
  std::printf("\"%s\": [", name);
  for (size_t i = 0; i < n; i++) std::printf("%s%.9g", i ? ", " : "", v[i]);
  std::printf("]");
}
template <class T> static void dump_ints(const char* name, const std::vector<T>& v) {
  std::printf("\"%s\": [", name);
  for (size_t i = 0; i < v.size(); i++) std::printf("%s%ld", i ? ", " : "", (long)v[i]);
  std::printf("]");
}
static void dump_state(const char* name, Agent& A) {
  std::printf("\"%s\": {\"change_index\": %d, \"kfs\": [", name, A.map.GetMapChangeIndex());
  for (size_t k = 0; k < A.kfs.size(); k++) {
    KeyFrame* kf = A.kfs[k].get();
    std::printf("%s{\"id\": %lu, \"client\": %d, \"locked_writes\": %d, ", k ? ", " : "", kf->mnId, (int)kf->mnClientId, kf->n_locked_pose_writes);
    dump_floats("pose", kf->Tcw.ptr<float>(0), 16);
    std::printf(", \"matches\": [");
    for (size_t i = 0; i < kf->mvpMapPoints.size(); i++) std::printf("%s%ld", i ? ", " : "", kf->mvpMapPoints[i] ? (long)kf->mvpMapPoints[i]->mnId : -1L);
    std::printf("]}");
  }
  std::printf("], \"mps\": [");
  for (size_t j = 0; j < A.points.size(); j++) {
    MapPoint* mp = A.points[j].get();
    std::printf("%s{\"id\": %lu, \"client\": %d, \"bad\": %d, \"locked_writes\": %d, \"normal_updates\": %d, ", j ? ", " : "", mp->mnId, (int)mp->mnClientId, (int)mp->isBad(), mp->n_locked_pos_writes, mp->n_normal_updates);
    dump_floats("pos", mp->mWorldPos.ptr<float>(0), 3);
    std::printf(", \"obs\": [");
    bool first = true;
    for (const auto& ob : mp->mObservations) { std::printf("%s%lu", first ? "" : ", ", ob.first->mnId); first = false; }
    std::printf("]}");
  }
  std::printf("]}");
}

int main() {
  struct Scene { const char* name; bool rig; int status; } scenes[] = {{"pinhole window", false, LBA_APPLIED}, {"two-fisheye rig", true, LBA_APPLIED},
                                                                     {"refused: most observations outliers", false, LBA_REJECTED_OUTLIERS}};
  for (const Scene& sc : scenes) {
    od::LbaWindowCache<KeyFrame, MapPoint>::instance().clear();
    Agent A; g_agent = &A; g_status = sc.status;
    KeyFrame* cur = sc.rig ? build_lba_rig_scene(A, 7, 3, 250, 0.03, 5151) : build_lba_scene(A, 8, 4, 300, 0.03, 5151);
    std::printf("{\"scene\": \"%s\", \"current_kf\": %lu, ", sc.name, cur->mnId);
    dump_state("before", A);
    bool stop = false; int num_fixed = -1;
    const int status = od::LocalBundleAdjustment<AnswerOps>(cur, &stop, &A.map, num_fixed, 0);
    std::printf(",\n \"status\": %d, \"has_right\": %d, \"turned_bad\": %ld, ", status, (int)g_ans.has_right, g_ans.turned_bad);
    dump_ints("pose_kf", g_ans.pose_kf); std::printf(", "); dump_ints("point_mp", g_ans.point_mp); std::printf(", "); dump_ints("fixed", g_ans.fixed); std::printf(", ");
    dump_floats("poses_out", g_ans.poses_out.data(), g_ans.poses_out.size()); std::printf(", "); dump_floats("points_out", g_ans.points_out.data(), g_ans.points_out.size());
    std::printf(", \"edges\": [");
    for (size_t k = 0; k < g_ans.edges.size() && k < g_ans.eout.size(); k++)
      std::printf("%s[%ld, %ld, %.9g, %d, %d]", k ? ", " : "", g_ans.edge_kf[k], g_ans.edge_mp[k], g_ans.edges[k].ur, (int)g_ans.eout[k], (int)((k % 14) != 3));
    std::printf("],\n ");
    dump_state("after", A);
    std::printf("}\n");
  }
  return 0;
}
