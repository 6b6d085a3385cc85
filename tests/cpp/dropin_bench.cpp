// What the drop-in costs END TO END: the reference-side glue of include/orbgpu_dropin.hpp (pointer graph -> flat views,
// GetWorldPos() / GetDescriptor() clones, write-back under Map::mMutexMapUpdate) timed next to the C-ABI calls it wraps, over the
// header-only mocks of Frame / KeyFrame / MapPoint / Map at the sizes of BASELINE.json configs[1] (C2: 1000 features per image,
// a local map of a few thousand points, a 20 + 10 keyframe local-BA window of 2000 points).
//   per call:  total = glue (flatten + write-back, host only)  +  frame upload (orbm_frame_upload of the flattened Frame)  +  C-ABI call
// Reference lines the glue stands for: I/ORBmatcher.h:39-88, S/ORBmatcher.cc:44-60, S/Tracking.cc:3083-3155,
// S/Optimizer.cc:1813-1908,2270-2408, S/Optimizer.cc:964-1060,1270-1278, S/Frame.cc:71-172.
// Prints one human-readable table (stderr) and ONE JSON line (stdout).  Exit code 3 = no GPU.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "scenario.hpp"

namespace od = orbgpu::dropin;

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The product's entry points with two stop-watches: time inside the C-ABI call proper, and time spent uploading the flattened
// Frame (a Frame that carries an orbgpu::FrameOnDevice member, as the Frame constructor adapter leaves it, skips that part).
struct TimedOps {
  static constexpr bool kUsesResidentFrame = true;
  static double t_call, t_upload;
  static orbgpu::FrameOnDevice& frame(const od::FrameKey& key, const orbm_frame_view& v) {
    const double t0 = now_us();
    orbgpu::FrameOnDevice& f = od::GpuOps::frame(key, v);
    t_upload += now_us() - t0;
    return f;
  }
  static int is_in_frustum(const od::FrameKey& key, const orbm_frame_view& v, const float* Tcw, const orbm_worldpoints_view& pts, float lim,
                           uint8_t* in_view, float* px, float* py, float* pxr, float* depth, int32_t* level, float* vcos) {
    orbm_frame* f = frame(key, v).handle();
    const double t0 = now_us();
    const int rc = orbm_is_in_frustum(f, Tcw, &pts, lim, in_view, px, py, pxr, depth, level, vcos);
    t_call += now_us() - t0;
    return rc;
  }
  static int search_mps(const od::FrameKey& key, const orbm_frame_view& v, const orbm_mappoints_view& mps, float th, int far_points, float th_far,
                        float nnratio, int32_t* amp, int32_t* aob, int* n) {
    orbm_frame* f = frame(key, v).handle();
    const double t0 = now_us();
    const int rc = orbm_search_by_projection_mps(f, &mps, th, far_points, th_far, nnratio, amp, aob, n);
    t_call += now_us() - t0;
    return rc;
  }
  static int search_frame(const od::FrameKey& key, const orbm_frame_view& v, const float* Tcw, const orbm_lastframe_view& last, float th, int mono,
                          int check_ori, int32_t* amp, int32_t* aob, int* n) {
    orbm_frame* f = frame(key, v).handle();
    const double t0 = now_us();
    const int rc = orbm_search_by_projection_frame(f, Tcw, &last, th, mono, check_ori, amp, aob, n);
    t_call += now_us() - t0;
    return rc;
  }
  static int search_local(const od::FrameKey& key, const orbm_frame_view& v, const float* Tcw, const orbm_worldpoints_view& pts, float th, int far_points,
                          float th_far, float nnratio, int32_t* amp, int32_t* aob, int* n, uint8_t* in_frustum) {
    orbm_frame* f = frame(key, v).handle();
    od::GpuOps::ThreadState& s = od::GpuOps::state();
    const double t0 = now_us();
    if (!s.local_map) s.local_map.reset(new orbgpu::MapPointsOnDevice(std::max(pts.m, 16384)));
    orbm_worldpoints_view up = pts;                    // (as GpuOps::search_local: the per-frame exclusions travel with the call)
    up.skip = nullptr;
    s.local_map->Upload(up);
    s.local_map_loaded = true;
    const int rc = orbm_search_local_points_vis(f, s.local_map->handle(), Tcw, pts.skip, th, far_points, th_far, nnratio, amp, aob, n, in_frustum);
    t_call += now_us() - t0;
    return rc;
  }
  static int search_local_resident(const od::FrameKey& key, const orbm_frame_view& v, const float* Tcw, const orbm_worldpoints_view& pts, const uint8_t* excluded,
                                   bool statics_same, float th, int far_points, float th_far, float nnratio, int32_t* amp, int32_t* aob, int* n,
                                   uint8_t* in_frustum) {
    od::GpuOps::ThreadState& s = od::GpuOps::state();
    if (!(statics_same && s.local_map && s.local_map_loaded))
      return search_local(key, v, Tcw, pts, th, far_points, th_far, nnratio, amp, aob, n, in_frustum);
    orbm_frame* f = frame(key, v).handle();
    const double t0 = now_us();
    s.local_map->SetObservations(pts.n_obs);
    const int rc = orbm_search_local_points_vis(f, s.local_map->handle(), Tcw, excluded, th, far_points, th_far, nnratio, amp, aob, n, in_frustum);
    t_call += now_us() - t0;
    return rc;
  }
  static int search_bow(const od::FrameKey& key, const orbm_frame_view& v, const orbm_featvec_view& fvF, const uint8_t* kf_desc, int nkf,
                        const uint8_t* kf_valid, const float* kf_angle, const orbm_featvec_view& fvKF, float nnratio, int check_ori,
                        int32_t* matches, int* n) {
    orbm_frame* f = frame(key, v).handle();
    const double t0 = now_us();
    const int rc = orbm_search_by_bow(f, &fvF, kf_desc, nkf, kf_valid, kf_angle, &fvKF, nnratio, check_ori, matches, n);
    t_call += now_us() - t0;
    return rc;
  }
  static int lba(const lba_problem& p, const volatile bool* stop, lba_result& r) {
    const double t0 = now_us();
    const int rc = od::GpuOps::lba(p, stop, r);
    t_call += now_us() - t0;
    return rc;
  }
  static int pose_opt(const pose_opt_problem& p, pose_opt_result& r) {
    const double t0 = now_us();
    const int rc = od::GpuOps::pose_opt(p, r);
    t_call += now_us() - t0;
    return rc;
  }
};
double TimedOps::t_call = 0, TimedOps::t_upload = 0;

struct Row { std::string name; double total = 0, call = 0, upload = 0; int reps = 0; int work = 0; };

template <class Fn, class Reset>
static Row measure(const char* name, int reps, int warm, Reset reset, Fn fn) {
  Row r; r.name = name;
  for (int i = 0; i < warm + reps; i++) {
    reset();
    TimedOps::t_call = TimedOps::t_upload = 0;
    const double t0 = now_us();
    const int w = fn();
    const double dt = now_us() - t0;
    if (i >= warm) { r.total += dt; r.call += TimedOps::t_call; r.upload += TimedOps::t_upload; r.reps++; r.work = w; }
  }
  r.total /= r.reps; r.call /= r.reps; r.upload /= r.reps;
  return r;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? std::atoi(argv[1]) : 40;
  try {
    g_seed = 20251003u;
    const std::vector<uint8_t> tex = make_texture();
    orbgpu::ORBextractor rig(1000, 1.2f, 8, 20, 7, W, H, /*n_cams*/ 2);
    Agent A;
    for (int k = 0; k <= 7; k++) make_frame(A, rig, tex, k, /*Tracking's current frame stays on the device*/ k == 7);
    orbgpu::ORBextractor rig2(1000, 1.2f, 8, 20, 7, W, H, /*n_cams*/ 2);      // (the constructor row must not disturb the frame `rig` holds)
    Frame& Last = *A.frames[6];
    Frame& Cur = *A.frames[7];
    // local map: the points the keyframes 0..5 created (a few thousand, as bench.py's local map of 20 keyframes)
    std::vector<MapPoint*> local; std::vector<int> src;
    for (int k = 0; k <= 5; k++) make_points_from(A, *A.frames[k], local, src);
    std::vector<MapPoint*> lastpts; std::vector<int> lastfeat;
    make_points_from(A, Last, lastpts, lastfeat);
    for (size_t j = 0; j < lastpts.size(); j++) Last.mvpMapPoints[lastfeat[j]] = lastpts[j];
    { float* T = Cur.mTcw.ptr<float>(0); T[3] += 0.012f; T[7] -= 0.008f; T[11] += 0.01f; }
    const Mat Tguess = Cur.mTcw;
    auto clear_cur = [&]() { std::fill(Cur.mvpMapPoints.begin(), Cur.mvpMapPoints.end(), nullptr); std::fill(Cur.mvbOutlier.begin(), Cur.mvbOutlier.end(), false); Cur.mTcw = Tguess; };
    std::vector<Row> rows;

    // ---- Frame::Frame(stereo) through the adapter: host images in, device-resident features + the host copies a Frame holds
    {
      double T[16], Tr[16]; pose_of(7, T);
      for (int i = 0; i < 16; i++) Tr[i] = T[i];
      Tr[3] -= BB;
      const std::vector<uint8_t> L = render(tex, T), R = render(tex, Tr);
      orbm_frame_view v{0, nullptr, nullptr, nullptr, nullptr, 0, (float)W, 0, (float)H, FX, FX, CX, CY, BF, BB, 8, 1.2f};
      orbgpu::FrameOnDevice dev(4096);
      std::vector<orbx_keypoint> keys; std::vector<uint8_t> desc; std::vector<float> ur, dp;
      Frame F;
      rows.push_back(measure("Frame::Frame(stereo, host images)", reps, 5, []() {}, [&]() {
        const double t0 = now_us();
        const int N = dev.StereoCtor(rig2, v, L.data(), R.data(), W, H, W, &keys, &desc, &ur, &dp);
        TimedOps::t_call += now_us() - t0;
        // what the reference's constructor leaves in the Frame (S/Frame.cc:97-160): mvKeys / mvKeysUn, mDescriptors, mvuRight, mvDepth
        F.N = N; F.mvKeys.resize(N); F.mDescriptors = Mat(N, 32, 1); F.mvuRight = ur; F.mvDepth = dp;
        for (int i = 0; i < N; i++) F.mvKeys[i] = KeyPoint{{keys[i].x, keys[i].y}, keys[i].size, keys[i].angle, keys[i].response, keys[i].octave};
        F.mvKeysUn = F.mvKeys;
        std::memcpy(F.mDescriptors.ptr<uint8_t>(0), desc.data(), desc.size());
        F.mvpMapPoints.assign(N, nullptr); F.mvbOutlier.assign(N, false);
        return N;
      }));
    }
    // ---- TrackWithMotionModel: SearchByProjection(Current, Last)
    rows.push_back(measure("SearchByProjection(Cur, Last)", reps, 5, clear_cur, [&]() { return od::SearchByProjection<TimedOps>(Cur, Last, 7.0f, false, true); }));
    // ---- Tracking::SearchLocalPoints: isInFrustum for every local point + SearchByProjection(F, local points)
    clear_cur();
    od::SearchByProjection<TimedOps>(Cur, Last, 7.0f, false, true);
    const std::vector<MapPoint*> after_frame = Cur.mvpMapPoints;
    auto reset_local = [&]() { Cur.mvpMapPoints = after_frame; Cur.mTcw = Tguess; };
    Cur.mnId = 7;
    auto reset_fused = [&]() { reset_local(); for (auto* p : local) p->mnLastFrameSeen = ~0ul; };
    // a frame whose local map is the previous frame's (4 of 5 frames at one keyframe per 5 frames): the flattened statics and the device
    // copy are reused; and a frame after a keyframe: the points the local BA moved are read again (all of them for a MapPoint type
    // without a change counter: the map's change index moved)
    rows.push_back(measure("SearchLocalPoints (fused; local map unchanged since the last frame)", reps, 5, reset_fused,
                           [&]() { return od::SearchLocalPoints<TimedOps>(Cur, local, 1.0f, false, 50.0f, 0.8f); }));
    // (what a keyframe does to the local map before the next frame: the local BA's write-back -- SetWorldPos + UpdateNormalAndDepth on
    // the window's points, four in five of the local map here, and the map's change index -- and ProcessNewKeyFrame's new observations)
    auto after_keyframe = [&]() {
      reset_fused();
      for (size_t j = 0; j < local.size(); j++) {
        if (j % 5 == 4) continue;
        local[j]->SetWorldPos(local[j]->GetWorldPos(), true);
        local[j]->UpdateNormalAndDepth();
      }
      if (!local.empty() && local[0]->GetMap()) local[0]->GetMap()->IncreaseChangeIndex();
    };
    rows.push_back(measure("SearchLocalPoints (fused; after a keyframe: the local BA moved 4 in 5 points, re-read + upload)", reps, 5, after_keyframe,
                           [&]() { return od::SearchLocalPoints<TimedOps>(Cur, local, 1.0f, false, 50.0f, 0.8f); }));
    rows.push_back(measure("SearchLocalPoints as two calls (isInFrustum loop + SearchByProjection(F, MPs))", reps, 5, reset_fused, [&]() {
      od::isInFrustumAll<TimedOps>(Cur, local, 0.5f);
      return od::SearchByProjection<TimedOps>(Cur, local, 1.0f, false, 50.0f, 0.8f);
    }));
    // ---- PoseOptimization after the two searches
    reset_local();
    od::isInFrustumAll<TimedOps>(Cur, local, 0.5f);
    od::SearchByProjection<TimedOps>(Cur, local, 1.0f, false, 50.0f, 0.8f);
    const std::vector<MapPoint*> after_map = Cur.mvpMapPoints;
    int n_corr = 0; for (auto* p : after_map) n_corr += p != nullptr;
    auto reset_po = [&]() { Cur.mvpMapPoints = after_map; Cur.mTcw = Tguess; std::fill(Cur.mvbOutlier.begin(), Cur.mvbOutlier.end(), false); };
    rows.push_back(measure("PoseOptimization", reps, 5, reset_po, [&]() { return od::PoseOptimization<TimedOps>(&Cur); }));
    rows.back().work = n_corr;
    // ... and the FIRST call of a frame (S/Tracking.cc:2649): on the matches of SearchByProjection(Cur, Last) alone
    int n_corr1 = 0; for (auto* p : after_frame) n_corr1 += p != nullptr;
    auto reset_po1 = [&]() { Cur.mvpMapPoints = after_frame; Cur.mTcw = Tguess; std::fill(Cur.mvbOutlier.begin(), Cur.mvbOutlier.end(), false); };
    rows.push_back(measure("PoseOptimization (first call of a frame: the frame-to-frame matches)", reps, 5, reset_po1, [&]() { return od::PoseOptimization<TimedOps>(&Cur); }));
    rows.back().work = n_corr1;
    const double po_second = rows[rows.size() - 2].total, po_first = rows.back().total;
    // ---- LocalMapping: LocalBundleAdjustment on 20 + 10 keyframe windows of 2000 points.  First row: consecutive windows of one map
    // (each for a new keyframe that sees a third of the last one's points; windows 3-8 of 8 are timed) -- what LocalMapping does, and
    // what the glue's window cache is for.  Second row: a fresh map per call (nothing to reuse: the cost of a first window).
    {
      Row r; r.name = "LocalBundleAdjustment (20 free + 10 fixed KFs, 2000 points; consecutive windows)";
      const int nscene = std::max(reps / 16, 2);
      for (int i = 0; i < nscene; i++) {
        od::LbaWindowCache<KeyFrame, MapPoint>::instance().clear();      // (the last scene's points are gone)
        Agent B;
        KeyFrame* cur = build_lba_scene(B, 21, 10, 2000, 0.03, 99);
        bool mbAbortBA = false; int num_fixed = 0;
        for (int w = 0; w < 8; w++) {
          TimedOps::t_call = TimedOps::t_upload = 0;
          const double t0 = now_us();
          od::LocalBundleAdjustment<TimedOps>(cur, &mbAbortBA, &B.map, num_fixed, 0);
          const double dt = now_us() - t0;
          if (w >= 2) { r.total += dt; r.call += TimedOps::t_call; r.reps++; int ne = 0; for (auto& kf : B.kfs) ne += (int)kf->mvKeysUn.size(); r.work = ne; }
          cur = next_keyframe(B, cur, w);
        }
      }
      r.total /= r.reps; r.call /= r.reps;
      rows.push_back(r);
    }
    {
      Row r; r.name = "LocalBundleAdjustment (first window of a map: every point read)";
      const int nrep = std::max(reps / 8, 3);
      for (int i = 0; i < nrep + 2; i++) {
        od::LbaWindowCache<KeyFrame, MapPoint>::instance().clear();
        Agent B;
        KeyFrame* cur = build_lba_scene(B, 21, 10, 2000, 0.03, 99);
        bool mbAbortBA = false; int num_fixed = 0;
        TimedOps::t_call = TimedOps::t_upload = 0;
        const double t0 = now_us();
        od::LocalBundleAdjustment<TimedOps>(cur, &mbAbortBA, &B.map, num_fixed, 0);
        const double dt = now_us() - t0;
        if (i >= 2) { r.total += dt; r.call += TimedOps::t_call; r.reps++; int ne = 0; for (auto& kf : B.kfs) ne += (int)kf->mvKeysUn.size(); r.work = ne; }
      }
      r.total /= r.reps; r.call /= r.reps;
      rows.push_back(r);
      od::LbaWindowCache<KeyFrame, MapPoint>::instance().clear();
    }
    // ---- report
    std::fprintf(stderr, "%-66s %10s %10s %10s %10s %7s\n", "call through the reference-side glue (C2 sizes)", "total us", "C-ABI us", "upload us", "glue us", "glue %");
    std::string js = "{\"dropin_bench\": {";
    double frame_path = 0;
    for (size_t i = 0; i < rows.size(); i++) {
      const Row& r = rows[i];
      const double glue = r.total - r.call - r.upload;
      std::fprintf(stderr, "%-66s %10.1f %10.1f %10.1f %10.1f %6.1f%%   (%d)\n", r.name.c_str(), r.total, r.call, r.upload, glue, 100.0 * glue / r.total, r.work);
      char buf[512];
      std::snprintf(buf, sizeof(buf), "%s\"%s\": {\"total_us\": %.1f, \"c_abi_us\": %.1f, \"frame_upload_us\": %.1f, \"glue_us\": %.1f, \"glue_frac\": %.3f, \"work\": %d}",
                    i ? ", " : "", r.name.c_str(), r.total, r.call, r.upload, glue, glue / r.total, r.work);
      js += buf;
      // constructor + SearchByProjection(Cur, Last) + fused SearchLocalPoints (4 frames on an unchanged local map, 1 after a keyframe)
      if (i < 2) frame_path += r.total;
      else if (i == 2) frame_path += 0.8 * r.total;
      else if (i == 3) frame_path += 0.2 * r.total;
    }
    char tail[768];
    // ... and what an unchanged Tracking thread gets per frame: the same + both PoseOptimization calls (S/Tracking.cc:2649, :2712)
    const double full_path = frame_path + po_first + po_second;
    std::snprintf(tail, sizeof(tail), "}, \"frame_path_us\": %.1f, \"frames_per_s_frame_path\": %.1f, \"frame_path_with_pose_opt_us\": %.1f, "
                  "\"frames_per_s_frame_path_with_pose_opt\": %.1f, \"local_map_points\": %zu, \"reps\": %d}", frame_path, 1e6 / frame_path, full_path, 1e6 / full_path,
                  local.size(), reps);
    js += tail;
    std::printf("%s\n", js.c_str());
    od::GpuOps::release();
    return 0;
  } catch (const std::runtime_error& e) {
    std::printf("runtime_error: %s\n", e.what());
    return 3;
  }
}
