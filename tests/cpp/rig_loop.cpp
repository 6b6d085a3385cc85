// Closed loop of a two-fisheye agent through the reference-signature glue (include/orbgpu_dropin.hpp): K frames, each built from the SAME
// synthetic observations of a fixed map (features of both cameras: noisy projections, descriptors a few bits off, distractors), then
//   ComputeStereoFishEyeMatches -> motion model -> SearchByProjection(Cur, Last) -> PoseOptimization -> outliers dropped ->
//   SearchLocalPoints -> PoseOptimization -> outliers dropped -> mLastFrame; every eighth frame a keyframe + LocalBundleAdjustment
//   (write-back into this run's own keyframes and map points, flagged observations erased)
// as Tracking::Track does (S/Tracking.cc:2590-2811, TrackWithMotionModel :2928-3010, TrackLocalMap :3012-3081) -- once over liborbgpu,
// once over the CPU oracle, EACH RUN FEEDING ON ITS OWN poses, matches and outlier decisions.  Per-frame digests (stereo partners, both
// match arrays, outlier flags, inlier counts, local-BA status / fixed keyframes, observation counts) must be equal on every frame, poses and
// keyframe poses within 1e-4, point positions within 5e-2; the first divergent frame is reported.
//   rig_loop [frames]          exit code 0 = no divergence
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "scenario.hpp"
#include "../../oracle/orb_oracle.h"
#include "oracle_ops.hpp"

struct Digest { std::vector<int> l2r; std::vector<long> after_motion, after_local; std::vector<bool> outl1, outl2; int n_stereo = 0, n_motion = 0, in1 = 0, n_local = 0, in2 = 0; std::vector<float> pose;
                int lba_status = -99, lba_fixed = -1, obs_total = 0; std::vector<float> points, kf_poses; };

struct World { std::vector<std::unique_ptr<MapPoint>> pts; std::vector<double> base_angle; std::vector<int> lvl; Map map; };

static void make_world(World& W, int n_points) {
  g_seed = 20251u;
  for (int i = 0; i < n_points; i++) {
    std::unique_ptr<MapPoint> mp(new MapPoint);
    mp->mnId = 1000 + i; mp->mpMap = &W.map;
    // a shell of points in front of the start pose, 1.5 - 8 m away, spread over the fisheye field of view
    const double th = 0.95 * urand(), psi = 6.283185307179586 * urand(), depth = 1.5 + 6.5 * urand();
    const double X[3] = {depth * std::sin(th) * std::cos(psi), depth * std::sin(th) * std::sin(psi), depth * std::cos(th)};
    double nn = 0, nv[3];
    for (int a = 0; a < 3; a++) { nv[a] = X[a] / depth + 0.12 * nrand(); nn += nv[a] * nv[a]; }
    W.lvl.push_back(rnd() % 6 + 1); W.base_angle.push_back(360 * urand());
    for (int a = 0; a < 3; a++) { mp->mWorldPos.ptr<float>(0)[a] = (float)X[a]; mp->mNormalVector.ptr<float>(0)[a] = (float)(nv[a] / std::sqrt(nn)); }
    const double maxd = depth * std::pow(1.2, W.lvl.back() - 0.5);
    mp->mfMaxDistance = (float)maxd; mp->mfMinDistance = (float)(maxd / std::pow(1.2, 7));
    for (int b = 0; b < 32; b++) mp->mDescriptor.ptr<uint8_t>(0)[b] = (uint8_t)rnd();
    mp->nObs = 1 + (int)(rnd() % 6);
    W.pts.push_back(std::move(mp));
  }
}
static void true_pose(int k, double T[16]) {                      // a slow arc: forward and sideways, a little yaw and pitch
  double R[9]; rot(0.002 * k, 0.004 * k, -0.001 * k, R);
  const double t[3] = {-0.012 * k, 0.004 * k, -0.018 * k};
  for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0) ? 1 : 0;
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T[4 * i + j] = R[3 * i + j]; T[4 * i + 3] = t[i]; }
}

// the Frame of time k as its constructor leaves it before ComputeStereoFishEyeMatches: the same for every run
static std::unique_ptr<Frame> make_frame(Agent& A, const World& W, int k) {
  g_seed = 777u + 131u * (unsigned)k;
  const float size = 512.f;
  std::unique_ptr<Frame> F(new Frame);
  F->mnId = 100 + k;
  give_rig(A, F->mpCamera, F->mpCamera2, F->mTrl);
  double Trl[12]; rig_Trl(Trl);
  F->mTlr = Mat(3, 4, 4);
  for (int i = 0; i < 3; i++) { double tt = 0; for (int j = 0; j < 3; j++) { F->mTlr.ptr<float>(0)[4 * i + j] = (float)Trl[4 * j + i]; tt -= Trl[4 * j + i] * Trl[4 * j + 3]; }
                                F->mTlr.ptr<float>(0)[4 * i + 3] = (float)tt; }
  F->mnMinX = 0; F->mnMaxX = size; F->mnMinY = 0; F->mnMaxY = size; F->fx = KB8_L[0]; F->fy = KB8_L[1]; F->cx = KB8_L[2]; F->cy = KB8_L[3]; F->mbf = 0; F->mb = 0.1f;
  float sc = 1.f; for (int l = 0; l < 8; l++) { F->mvLevelSigma2.push_back(sc * sc); F->mvInvLevelSigma2.push_back(1.f / (sc * sc)); sc *= 1.2f; }
  double T[16]; true_pose(k, T);
  struct Feat { float x, y; int oct; float angle; uint8_t d[32]; };
  std::vector<Feat> fl, fr;
  auto noisy = [&](const uint8_t* d, int bits, uint8_t* o) { std::memcpy(o, d, 32); for (int b = 0; b < bits; b++) { const int q = rnd() % 256; o[q >> 3] ^= (uint8_t)(1u << (q & 7)); } };
  for (size_t i = 0; i < W.pts.size(); i++) {
    const float* Xw = W.pts[i]->mWorldPos.ptr<float>(0);
    double Xc[3], Xr[3];
    for (int a = 0; a < 3; a++) Xc[a] = T[4 * a] * Xw[0] + T[4 * a + 1] * Xw[1] + T[4 * a + 2] * Xw[2] + T[4 * a + 3];
    for (int a = 0; a < 3; a++) Xr[a] = Trl[4 * a] * Xc[0] + Trl[4 * a + 1] * Xc[1] + Trl[4 * a + 2] * Xc[2] + Trl[4 * a + 3];
    for (int side = 0; side < 2; side++) {
      const double* X = side ? Xr : Xc;
      if (X[2] <= 0.05 || urand() < 0.3) continue;
      double uv[2]; kb8_project(side ? KB8_R : KB8_L, X, uv);
      const double sig = 0.5 * std::pow(1.2, W.lvl[i] - 1);
      uv[0] += sig * nrand(); uv[1] += sig * nrand();
      if (urand() < 0.04) { uv[0] += 12 * nrand(); uv[1] += 12 * nrand(); }                   // a gross error now and then: PoseOptimization's business
      if (!(uv[0] > 2 && uv[0] < size - 2 && uv[1] > 2 && uv[1] < size - 2)) continue;
      Feat f; f.x = (float)uv[0]; f.y = (float)uv[1]; f.oct = std::max(0, W.lvl[i] - (urand() < 0.3 ? 1 : 0));
      f.angle = (float)std::fmod(W.base_angle[i] + 2 * nrand() + 720.0, 360.0);
      noisy(W.pts[i]->mDescriptor.ptr<uint8_t>(0), (int)(rnd() % 36), f.d);
      (side ? fr : fl).push_back(f);
    }
  }
  for (int side = 0; side < 2; side++) {
    std::vector<Feat>& v = side ? fr : fl;
    for (int q = 0; q < 150; q++) { Feat f; f.x = (float)(2 + (size - 4) * urand()); f.y = (float)(2 + (size - 4) * urand()); f.oct = rnd() % 8; f.angle = (float)(360 * urand());
                                    for (int b = 0; b < 32; b++) f.d[b] = (uint8_t)rnd(); v.push_back(f); }
    for (size_t q = v.size(); q > 1; q--) std::swap(v[q - 1], v[rnd() % q]);
  }
  const int nl = (int)fl.size(), nr = (int)fr.size();
  F->Nleft = nl; F->Nright = nr; F->N = nl + nr; F->monoLeft = 0; F->monoRight = 0;             // (every feature in the lapping area)
  F->mDescriptors = Mat(nl, 32, 1); F->mDescriptorsRight = Mat(nr, 32, 1);
  for (int i = 0; i < nl; i++) { F->mvKeys.push_back(KeyPoint{{fl[i].x, fl[i].y}, 31.f, fl[i].angle, 20.f, fl[i].oct}); std::memcpy(F->mDescriptors.ptr<uint8_t>(i), fl[i].d, 32); }
  for (int i = 0; i < nr; i++) { F->mvKeysRight.push_back(KeyPoint{{fr[i].x, fr[i].y}, 31.f, fr[i].angle, 20.f, fr[i].oct}); std::memcpy(F->mDescriptorsRight.ptr<uint8_t>(i), fr[i].d, 32); }
  F->mvKeysUn = F->mvKeys;
  return F;
}

static Mat mat_mul44(const Mat& A, const Mat& B) { Mat C(4, 4, 4); for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int q = 0; q < 4; q++) s += (double)A.at(i, q) * B.at(q, j); C.ptr<float>(0)[4 * i + j] = (float)s; } return C; }
static Mat inv_pose(const Mat& T) { Mat R(4, 4, 4); for (int i = 0; i < 3; i++) { double t = 0; for (int j = 0; j < 3; j++) { R.ptr<float>(0)[4 * i + j] = T.at(j, i); t -= (double)T.at(j, i) * T.at(j, 3); } R.ptr<float>(0)[4 * i + 3] = (float)t; }
                                 R.ptr<float>(0)[12] = R.ptr<float>(0)[13] = R.ptr<float>(0)[14] = 0; R.ptr<float>(0)[15] = 1; return R; }

template <class Ops>
static std::vector<Digest> run(int K) {
  Agent A; World W; make_world(W, 1400);
  std::vector<MapPoint*> local; for (auto& p : W.pts) local.push_back(p.get());
  std::vector<Digest> out;
  std::unique_ptr<Frame> last, last2;
  for (int k = 0; k < K; k++) {
    std::unique_ptr<Frame> F = make_frame(A, W, k);
    Digest d;
    d.n_stereo = od::ComputeStereoFishEyeMatches<Ops>(*F);                                      // the constructor's stereo stage ...
    d.l2r = F->mvLeftToRightMatch;
    { Mat all(F->N, 32, 1);                                                                     // ... vconcat(mDescriptors, mDescriptorsRight), :1083
      std::memcpy(all.ptr<uint8_t>(0), F->mDescriptors.ptr<uint8_t>(0), (size_t)F->Nleft * 32);
      std::memcpy(all.ptr<uint8_t>(F->Nleft), F->mDescriptorsRight.ptr<uint8_t>(0), (size_t)F->Nright * 32);
      F->mDescriptors = all; }
    F->mvpMapPoints.assign(F->N, nullptr); F->mvbOutlier.assign(F->N, false); F->mvuRight.assign(F->N, -1.f); F->mvDepth.resize(F->N, -1.f);
    auto ids = [](const Frame& Fr, std::vector<long>& o) { o.clear(); for (MapPoint* p : Fr.mvpMapPoints) o.push_back(p ? (long)p->mnId : -1); };
    auto drop_outliers = [](Frame& Fr) { for (int i = 0; i < Fr.N; i++) if (Fr.mvpMapPoints[i] && Fr.mvbOutlier[i]) { Fr.mvpMapPoints[i] = nullptr; Fr.mvbOutlier[i] = false; } };   // :2990-3003
    if (!last) {                                                                                // the first frame starts from the truth (initialisation is not on the path)
      double T[16]; true_pose(0, T); F->mTcw = mat44(T);
    } else {
      Mat V(4, 4, 4); for (int i = 0; i < 16; i++) V.ptr<float>(0)[i] = (i % 5 == 0) ? 1.f : 0.f;
      if (last2) V = mat_mul44(last->mTcw, inv_pose(last2->mTcw));                              // mVelocity = mLastFrame.mTcw * LastTwc (:2790-2797)
      F->mTcw = mat_mul44(V, last->mTcw);                                                       // :2949
      d.n_motion = od::SearchByProjection<Ops>(*F, *last, 15.0f, false, true);
      ids(*F, d.after_motion);
      d.in1 = od::PoseOptimization<Ops>(F.get());
      d.outl1 = F->mvbOutlier;
      drop_outliers(*F);
    }
    d.n_local = od::SearchLocalPoints<Ops>(*F, local, 3.0f, false, 0.f, 0.8f);                   // TrackLocalMap
    ids(*F, d.after_local);
    d.in2 = od::PoseOptimization<Ops>(F.get());
    d.outl2 = F->mvbOutlier;
    drop_outliers(*F);
    d.pose.assign(F->mTcw.ptr<float>(0), F->mTcw.ptr<float>(0) + 16);
    // every eighth frame becomes a keyframe and LocalMapping runs its local BA (S/LocalMapping.cc:245): observations of both cameras
    // (the tuple's second entry is NLeft + the right camera's index), the window = the last five keyframes, the write-back moves
    // keyframe poses and THIS run's map points, flagged observations are erased
    if (k % 8 == 0) {
      std::unique_ptr<KeyFrame> kf(new KeyFrame);
      kf->mnId = (long unsigned)A.kfs.size(); kf->mpMap = &W.map;
      kf->fx = F->fx; kf->fy = F->fy; kf->cx = F->cx; kf->cy = F->cy; kf->mbf = 0.f; kf->mvInvLevelSigma2 = F->mvInvLevelSigma2;
      kf->mpCamera = F->mpCamera; kf->mpCamera2 = F->mpCamera2; kf->mTrl = F->mTrl; kf->NLeft = F->Nleft; kf->NRight = F->Nright;
      kf->mvKeys = F->mvKeys; kf->mvKeysUn = F->mvKeys; kf->mvKeysRight = F->mvKeysRight; kf->mvuRight.assign(F->Nleft, -1.f);
      kf->Tcw = F->mTcw;
      kf->mvpMapPoints = F->mvpMapPoints;
      for (int i = 0; i < F->N; i++) {
        MapPoint* mp = F->mvpMapPoints[i];
        if (!mp) continue;
        auto it = mp->mObservations.find(kf.get());
        int li = -1, ri = -1;
        if (it != mp->mObservations.end()) { li = std::get<0>(it->second); ri = std::get<1>(it->second); } else mp->nObs++;
        if (i < F->Nleft) li = i; else ri = i;                                                    // (i >= Nleft IS NLeft + the right index)
        mp->mObservations[kf.get()] = std::make_tuple(li, ri);
        mp->Touch();                                                                                // (MapPoint::AddObservation moves the change counter, INTEGRATION.md edit E2)
      }
      for (int q = (int)A.kfs.size() - 1; q >= 0 && (int)kf->mvpOrderedConnectedKeyFrames.size() < 5; q--) kf->mvpOrderedConnectedKeyFrames.push_back(A.kfs[q].get());
      KeyFrame* cur_kf = kf.get();
      A.kfs.push_back(std::move(kf));
      if (A.kfs.size() >= 2) {
        bool stop = false; int num_fixed = -1;
        d.lba_status = od::LocalBundleAdjustment<Ops>(cur_kf, &stop, &W.map, num_fixed, 0);
        d.lba_fixed = num_fixed;
      }
      for (auto& p : W.pts) { d.obs_total += p->nObs; for (int a = 0; a < 3; a++) d.points.push_back(p->mWorldPos.ptr<float>(0)[a]); }
      for (auto& q : A.kfs) for (int a = 0; a < 16; a++) d.kf_poses.push_back(q->Tcw.ptr<float>(0)[a]);
    }
    out.push_back(d);
    last2 = std::move(last); last = std::move(F);
  }
  return out;
}

int main(int argc, char** argv) {
  const int K = argc > 1 ? std::atoi(argv[1]) : 40;
  if (orbg_device_count() <= 0) { std::printf("no usable HIP device\n"); return 3; }
  const std::vector<Digest> g = run<od::GpuOps>(K), c = run<OracleOps>(K);
  int first = -1, n_lba = 0, n_kf = 0; double worst = 0, worst_truth = 0, worst_pt = 0, worst_kf = 0; long matches = 0;
  for (int k = 0; k < K; k++) {
    const Digest &a = g[k], &b = c[k];
    const bool same = a.l2r == b.l2r && a.after_motion == b.after_motion && a.after_local == b.after_local && a.outl1 == b.outl1 && a.outl2 == b.outl2 &&
                      a.n_stereo == b.n_stereo && a.n_motion == b.n_motion && a.n_local == b.n_local && a.in1 == b.in1 && a.in2 == b.in2 &&
                      a.lba_status == b.lba_status && a.lba_fixed == b.lba_fixed && a.obs_total == b.obs_total && a.points.size() == b.points.size();
    double dp = 0; for (int i = 0; i < 16; i++) dp = std::max(dp, (double)std::fabs(a.pose[i] - b.pose[i]));
    worst = std::max(worst, dp);
    if (same) { for (size_t i = 0; i < a.points.size(); i++) worst_pt = std::max(worst_pt, (double)std::fabs(a.points[i] - b.points[i]));
                for (size_t i = 0; i < a.kf_poses.size() && i < b.kf_poses.size(); i++) worst_kf = std::max(worst_kf, (double)std::fabs(a.kf_poses[i] - b.kf_poses[i])); }
    if (a.lba_status == 0) n_lba++;
    if (a.lba_status != -99) n_kf++;
    double T[16]; true_pose(k, T); double dt = 0; for (int i = 0; i < 16; i++) dt = std::max(dt, std::fabs(a.pose[i] - T[i]));
    worst_truth = std::max(worst_truth, dt);
    matches += a.in2;
    // (gates as tests/cpp/closed_loop.cpp: a local BA that stops short of convergence returns weakly observed points millimetres apart for inputs
    //  micrometres apart -- poses and keyframe poses 1e-4, point positions 5e-2)
    if ((!same || dp > 1e-4 || worst_pt > 5e-2 || worst_kf > 1e-4) && first < 0) first = k;
    if (k < 3 || k == K - 1 || !same)
      std::printf("frame %2d: stereo %d, motion model %d matches -> %d inliers, local map %d matches -> %d inliers, pose diff %.2e, local BA %d (%d fixed), %d observations%s\n", k, a.n_stereo,
                  a.n_motion, a.in1, a.n_local, a.in2, dp, a.lba_status, a.lba_fixed, a.obs_total, same ? "" : "   DIGESTS DIFFER");
  }
  std::printf("{\"rig_loop\": {\"frames\": %d, \"first_divergent_frame\": %d, \"max_pose_diff\": %.3g, \"max_pose_error_vs_truth\": %.3g, \"mean_inliers\": %.1f, "
              "\"local_bas\": %d, \"local_bas_applied\": %d, \"max_point_diff\": %.3g, \"max_keyframe_pose_diff\": %.3g, \"ok\": %s}}\n", K, first, worst,
              worst_truth, (double)matches / K, n_kf, n_lba, worst_pt, worst_kf, first < 0 ? "true" : "false");
  return first < 0 ? 0 : 1;
}
