// orbgpu_dropin.hpp -- the bodies INTEGRATION.md describes, as real code: function templates with the reference's
// signatures that flatten the reference's pointer graph (Frame / KeyFrame / MapPoint / Map), call the C-ABI and write the
// results back where the reference does.  They are templates over the object types (duck typing on the reference's own
// member names), so the same code compiles against yutongwangBIT/multi_orbslam3's classes
// (cv::Mat based; define HAVE_OPENCV) and against the header-only mocks of tests/cpp/mock_orbslam3.hpp, and over an
// `Ops` policy that names the entry points: orbgpu::dropin::GpuOps = liborbgpu (the product); tests/cpp adds an
// OracleOps over the CPU oracle so that both run through the same glue.
//
//   int  SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)   I/ORBmatcher.h:46,  S/ORBmatcher.cc:44-214
//   int  SearchByProjection(Frame& Current, const Frame& Last, th, bMono)                     I/ORBmatcher.h:50,  S/ORBmatcher.cc:1970-2186
//   int  SearchByProjection(Frame& Current, KeyFrame*, const set<MapPoint*>&, th, ORBdist)     I/ORBmatcher.h:54,  S/ORBmatcher.cc:2188-2310
//   int  SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)                                   I/ORBmatcher.h:64,  S/ORBmatcher.cc:269-471
//   void isInFrustum for a list of points (Tracking::SearchLocalPoints' loop)                 S/Tracking.cc:3111-3128, S/Frame.cc:466-543
//   void Tracking::SearchLocalPoints() body: the loop above + the matcher call as ONE device pass  S/Tracking.cc:3083-3155
//   void LocalBundleAdjustment(KeyFrame*, bool* pbStopFlag, Map*, int& num_fixedKF, int)      I/Optimizer.h:42,   S/Optimizer.cc:1810-2410
//   int  PoseOptimization(Frame*)                                                             I/Optimizer.h:47,   S/Optimizer.cc:964-1278
//
// Matrix access goes through mat_f32 / mat_u8 / make_mat (overloads for cv::Mat below, for the mock in the test header).
#ifndef ORBGPU_DROPIN_HPP_
#define ORBGPU_DROPIN_HPP_

#include <algorithm>
#include <climits>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "orbgpu_adapters.hpp"

#ifdef ORBGPU_GLUE_TRACE
#include <chrono>
#include <cstdio>
#define ORBGPU_GLUE_T(label) do { const auto _n = std::chrono::steady_clock::now(); std::fprintf(stderr, "  [glue] %-28s %8.1f us\n", label, std::chrono::duration<double, std::micro>(_n - _glue_t).count()); _glue_t = _n; } while (0)
#define ORBGPU_GLUE_T0() auto _glue_t = std::chrono::steady_clock::now()
#else
#define ORBGPU_GLUE_T(label) do { } while (0)
#define ORBGPU_GLUE_T0() do { } while (0)
#endif

namespace orbgpu {
namespace dropin {

#ifdef HAVE_OPENCV
inline const float* mat_f32(const cv::Mat& m) { return m.ptr<float>(0); }
inline const uint8_t* mat_u8(const cv::Mat& m, int row) { return m.ptr<uint8_t>(row); }
inline void make_mat(cv::Mat& out, int rows, int cols, const float* data) { out = cv::Mat(rows, cols, CV_32F, const_cast<float*>(data)).clone(); }
#endif

// What an entry-point set gets to identify the Frame a call works on: its address and, if the Frame carries one, the
// device-resident copy its constructor left behind (FrameOnDevice::StereoCtor / StereoCtorSubmitHost: features, stereo matches
// and grid are already on the device, nothing is flattened or uploaded).  A Frame type opts in by having a member
// `void* mpGpuFrame` (an orbgpu::FrameOnDevice*, nullptr = none).
struct FrameKey {
  const void* id = nullptr;
  FrameOnDevice* resident = nullptr;
};
template <class FrameT> auto resident_frame_of(const FrameT& F, int) -> decltype(static_cast<FrameOnDevice*>(F.mpGpuFrame)) { return static_cast<FrameOnDevice*>(F.mpGpuFrame); }
template <class FrameT> FrameOnDevice* resident_frame_of(const FrameT&, long) { return nullptr; }

// ------------------------------------------------------------------------------------------------ entry points
// The product: liborbgpu.  A Frame is uploaded on first use into one of a few device frames the calling thread keeps
// (least recently used first: Tracking works on mCurrentFrame / mLastFrame / a relocalisation candidate at a time; the
// reference's Frame would carry an orbgpu::FrameOnDevice member instead -- FrameOnDevice::StereoCtor leaves it on the
// device).  The local BA keeps ONE handle per calling thread (the LocalMapping thread), so device buffers, pinned staging
// and the stream survive from keyframe to keyframe.  GpuOps::release() drops everything the calling thread holds (call it
// before the thread ends if the HIP runtime may be torn down first).
struct GpuOps {
  static constexpr int kFrameSlots = 4;
  struct ThreadState {
    struct Slot { const void* key = nullptr; unsigned long long used = 0; std::unique_ptr<FrameOnDevice> dev; };
    Slot slots[kFrameSlots];
    unsigned long long tick = 0;
    std::unique_ptr<LocalBA> ba;
    std::unique_ptr<MapPointsOnDevice> local_map;      // Tracking's local map: resident from one SearchLocalPoints to the next while its
    bool local_map_loaded = false;                     // static fields do not change (search_local_resident)
    std::unique_ptr<MapPointsOnDevice> reloc_map;      // the candidate keyframe's points of SearchByProjection(F, pKF, sAlreadyFound, ..)
  };
  static ThreadState& state() { static thread_local ThreadState s; return s; }
  static void release() { ThreadState& s = state(); for (auto& sl : s.slots) { sl.dev.reset(); sl.key = nullptr; sl.used = 0; } s.ba.reset(); s.local_map.reset(); s.local_map_loaded = false; s.reloc_map.reset(); }
  static constexpr bool kUsesResidentFrame = true;
  static FrameOnDevice& frame(const FrameKey& fk, const orbm_frame_view& v) {
    if (fk.resident) return *fk.resident;
    const void* key = fk.id;
    ThreadState& s = state();
    ThreadState::Slot* hit = nullptr; ThreadState::Slot* lru = &s.slots[0];
    for (auto& sl : s.slots) {
      if (sl.key == key && sl.dev) hit = &sl;
      if (sl.used < lru->used) lru = &sl;
    }
    if (!hit) { hit = lru; hit->key = key; if (!hit->dev) hit->dev.reset(new FrameOnDevice(std::max(v.n, 4096))); }
    hit->used = ++s.tick;
    hit->dev->Upload(v);
    return *hit->dev;
  }
  static int is_in_frustum(const FrameKey& key, const orbm_frame_view& v, const float* Tcw, const orbm_worldpoints_view& pts, float lim,
                           uint8_t* in_view, float* px, float* py, float* pxr, float* depth, int32_t* level, float* vcos) {
    return orbm_is_in_frustum(frame(key, v).handle(), Tcw, &pts, lim, in_view, px, py, pxr, depth, level, vcos);
  }
  static int search_mps(const FrameKey& key, const orbm_frame_view& v, const orbm_mappoints_view& mps, float th, int far_points, float th_far,
                        float nnratio, int32_t* amp, int32_t* aob, int* n) {
    return orbm_search_by_projection_mps(frame(key, v).handle(), &mps, th, far_points, th_far, nnratio, amp, aob, n);
  }
  static int search_frame(const FrameKey& key, const orbm_frame_view& v, const float* Tcw, const orbm_lastframe_view& last, float th, int mono,
                          int check_ori, int32_t* amp, int32_t* aob, int* n) {
    return orbm_search_by_projection_frame(frame(key, v).handle(), Tcw, &last, th, mono, check_ori, amp, aob, n);
  }
  // ---- two-camera Frames (Nleft != -1): the cameras' features are two device frames
  static int is_in_frustum_rig(const FrameKey& key, const orbm_frame_view& v, const float* Tcw, const orbg_camera_rig& rig, const float* Tlr,
                               const orbm_worldpoints_view& pts, float lim, uint8_t* in_view, float* px, float* py, float* depth, int32_t* level,
                               float* vcos, uint8_t* in_view_r, float* px_r, float* py_r, float* depth_r, int32_t* level_r, float* vcos_r) {
    return orbm_is_in_frustum_rig(frame(key, v).handle(), Tcw, &rig, Tlr, &pts, lim, in_view, px, py, depth, level, vcos, in_view_r, px_r, py_r,
                                  depth_r, level_r, vcos_r);
  }
  static int search_mps_rig(const FrameKey& kl, const orbm_frame_view& vl, const FrameKey& kr, const orbm_frame_view& vr, const orbm_mappoints_view& mps,
                            const orbm_mappoints_view& mps_r, const int32_t* l2r, const int32_t* r2l, float th, int far_points, float th_far,
                            float nnratio, int32_t* amp, int32_t* aob, int* n) {
    FrameOnDevice& L = frame(kl, vl); FrameOnDevice& R = frame(kr, vr);
    return orbm_search_by_projection_mps_rig(L.handle(), R.handle(), &mps, &mps_r, l2r, r2l, th, far_points, th_far, nnratio, amp, aob, n);
  }
  static int search_frame_rig(const FrameKey& kl, const orbm_frame_view& vl, const FrameKey& kr, const orbm_frame_view& vr, const float* Tcw,
                              const orbg_camera_rig& rig, const orbm_lastframe_view& last, float th, int mono, int check_ori, int32_t* amp,
                              int32_t* aob, int* n) {
    FrameOnDevice& L = frame(kl, vl);
    if (!rig.has_right) return orbm_search_by_projection_frame_rig(L.handle(), nullptr, Tcw, &rig, &last, th, mono, check_ori, amp, aob, n);
    FrameOnDevice& R = frame(kr, vr);
    return orbm_search_by_projection_frame_rig(L.handle(), R.handle(), Tcw, &rig, &last, th, mono, check_ori, amp, aob, n);
  }
  // isInFrustum(., 0.5) for every non-skipped point + SearchByProjection(F, points) in one device pass; in_frustum[m] out
  static int search_local(const FrameKey& key, const orbm_frame_view& v, const float* Tcw, const orbm_worldpoints_view& pts, float th,
                          int far_points, float th_far, float nnratio, int32_t* amp, int32_t* aob, int* n, uint8_t* in_frustum) {
    ThreadState& s = state();
    if (!s.local_map) s.local_map.reset(new MapPointsOnDevice(std::max(pts.m, 16384)));
    // Only what does NOT depend on the frame is baked into the resident map: the static fields and isBad() (final once set).  The
    // per-frame exclusion -- mnLastFrameSeen == F.mnId, S/Tracking.cc:3105-3109 -- belongs to THIS frame: it travels as the call's skip
    // list, on the uploading call as on every later one (the kernels OR the map's flags with the call's: a skip flag uploaded with
    // the map would exclude the uploading frame's matches from every later frame that reuses the map).
    orbm_worldpoints_view up = pts;
    up.skip = nullptr;
    s.local_map->Upload(up);
    s.local_map_loaded = true;
    return orbm_search_local_points_vis(frame(key, v).handle(), s.local_map->handle(), Tcw, pts.skip, th, far_points, th_far, nnratio, amp, aob, n,
                                        in_frustum);
  }
  // The same when the STATIC fields of the points (position, normal, distance range, descriptor) are the ones of the previous call
  // (SearchLocalPoints' cache below): the map that call uploaded is still resident, nothing is uploaded; only Observations() is
  // refreshed (host side) and the per-frame exclusions -- already matched in this frame / bad -- travel as the call's skip list.
  // (statics_same: the caller vouches that pts' static fields are those of this thread's previous search_local* call; `excluded` =
  // pts.skip | pts.bad per point)
  static int search_local_resident(const FrameKey& key, const orbm_frame_view& v, const float* Tcw, const orbm_worldpoints_view& pts, const uint8_t* excluded,
                                   bool statics_same, float th, int far_points, float th_far, float nnratio, int32_t* amp, int32_t* aob, int* n,
                                   uint8_t* in_frustum) {
    ThreadState& s = state();
    if (!(statics_same && s.local_map && s.local_map_loaded))
      return search_local(key, v, Tcw, pts, th, far_points, th_far, nnratio, amp, aob, n, in_frustum);
    s.local_map->SetObservations(pts.n_obs);
    return orbm_search_local_points_vis(frame(key, v).handle(), s.local_map->handle(), Tcw, excluded, th, far_points, th_far, nnratio, amp, aob, n,
                                        in_frustum);
  }
  static int search_bow(const FrameKey& key, const orbm_frame_view& v, const orbm_featvec_view& fvF, const uint8_t* kf_desc, int nkf,
                        const uint8_t* kf_valid, const float* kf_angle, const orbm_featvec_view& fvKF, float nnratio, int check_ori,
                        int32_t* matches, int* n) {
    return orbm_search_by_bow(frame(key, v).handle(), &fvF, kf_desc, nkf, kf_valid, kf_angle, &fvKF, nnratio, check_ori, matches, n);
  }
  static int search_bow_rig(const FrameKey& key, const orbm_frame_view& v_all, int n_left, const orbm_featvec_view& fvF, const uint8_t* kf_desc, int nkf,
                            const uint8_t* kf_valid, const float* kf_angle, const orbm_featvec_view& fvKF, float nnratio, int check_ori,
                            int32_t* matches, int* n) {
    return orbm_search_by_bow_rig(frame(key, v_all).handle(), n_left, &fvF, kf_desc, nkf, kf_valid, kf_angle, &fvKF, nnratio, check_ori, matches, n);
  }
  static int fisheye_stereo(const orbx_fisheye_stereo_view& v, int32_t* l2r, int32_t* r2l, float* depth, float* p3d, int* n) {
    return orbx_fisheye_stereo_matches(0, &v, l2r, r2l, depth, p3d, n);
  }
  // the relocalisation overload: the keyframe's points go up as a map of their own (a candidate keyframe is searched once or twice)
  static int search_reloc(const FrameKey& key, const orbm_frame_view& v, const float* Tcw, const orbm_worldpoints_view& kf_pts, const uint8_t* found,
                          const float* kf_angle, float th, int orb_dist, int check_ori, int32_t* amp, int* n) {
    ThreadState& s = state();
    if (!s.reloc_map) s.reloc_map.reset(new MapPointsOnDevice(std::max(kf_pts.m, 4096)));
    s.reloc_map->Upload(kf_pts);
    return orbm_search_by_projection_reloc(frame(key, v).handle(), s.reloc_map->handle(), Tcw, found, kf_angle, th, orb_dist, check_ori, amp, n);
  }
  static int search_reloc_cam(const FrameKey& key, const orbm_frame_view& v, const float* Tcw, const orbg_camera& cam, const orbm_worldpoints_view& kf_pts,
                              const uint8_t* found, const float* kf_angle, float th, int orb_dist, int check_ori, int32_t* amp, int* n) {
    ThreadState& s = state();
    if (!s.reloc_map) s.reloc_map.reset(new MapPointsOnDevice(std::max(kf_pts.m, 4096)));
    s.reloc_map->Upload(kf_pts);
    return orbm_search_by_projection_reloc_cam(frame(key, v).handle(), s.reloc_map->handle(), Tcw, &cam, found, kf_angle, th, orb_dist, check_ori, amp, n);
  }
  // pbStopFlag goes through as it is: the library polls the caller's bool (lba_solve_hb)
  static int lba(const lba_problem& p, const volatile bool* stop, lba_result& r) {
    ThreadState& s = state();
    if (!s.ba) s.ba.reset(new LocalBA(p.device));
    s.ba->Run(p, stop, r);
    return ORBG_OK;
  }
  static int pose_opt(const pose_opt_problem& p, pose_opt_result& r) { return pose_optimize(&p, &r); }
};

// ------------------------------------------------------------------------------------------------ flattening helpers
struct FrameFlat {
  std::vector<orbx_keypoint> kps;
  std::vector<uint8_t> desc;
  std::vector<float> uright, depth;
  orbm_frame_view v;
  FrameKey key;
};

template <class Ops, class FrameT>
void flatten_frame(const FrameT& F, FrameFlat& o) {
  const int N = F.N;
  o.key.id = &F;
  o.key.resident = Ops::kUsesResidentFrame ? resident_frame_of(F, 0) : nullptr;
  if (o.key.resident) {
    // the features are on the device already: only the scalars of the view are needed
    o.v = orbm_frame_view{N, nullptr, nullptr, nullptr, nullptr, F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY,
                          F.fx, F.fy, F.cx, F.cy, F.mbf, F.mb, F.mnScaleLevels, F.mfScaleFactor};
    return;
  }
  o.kps.resize(N); o.desc.resize((size_t)N * 32); o.uright.resize(N); o.depth.resize(N);
  for (int i = 0; i < N; i++) {
    const auto& kp = F.mvKeysUn[i];
    o.kps[i] = orbx_keypoint{kp.pt.x, kp.pt.y, kp.size, kp.angle, kp.response, (int32_t)kp.octave};
    std::memcpy(&o.desc[(size_t)32 * i], mat_u8(F.mDescriptors, i), 32);
    o.uright[i] = F.mvuRight[i];
    o.depth[i] = F.mvDepth[i];
  }
  o.v = orbm_frame_view{N, o.kps.data(), o.desc.data(), o.uright.data(), o.depth.data(), F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY,
                        F.fx, F.fy, F.cx, F.cy, F.mbf, F.mb, F.mnScaleLevels, F.mfScaleFactor};
}

// F.mvpMapPoints <-> (assigned_mp, assigned_obs): entries that exist on entry are marked with a value no search writes
template <class FrameT>
void flatten_assignments(const FrameT& F, std::vector<int32_t>& amp, std::vector<int32_t>& aob) {
  amp.assign(F.N, -1); aob.assign(F.N, 0);
  for (int i = 0; i < F.N; i++)
    if (F.mvpMapPoints[i]) { amp[i] = INT32_MAX; aob[i] = F.mvpMapPoints[i]->Observations(); }
}

// A two-camera Frame (Nleft != -1, S/Frame.cc:1017-1091): the left camera's features are mvKeys[0, Nleft) with descriptor rows
// [0, Nleft) and mGrid, the right camera's mvKeysRight with rows Nleft + i and mGridRight (S/Frame.cc:360-391) -- two frames for the
// entry points (no uRight on either: mvuRight is -1 throughout such a frame).
template <class Ops, class FrameT>
void flatten_rig_frame(const FrameT& F, FrameFlat& L, FrameFlat& R) {
  const int nl = F.Nleft, nr = F.N - F.Nleft;
  auto fill = [&](FrameFlat& o, const auto& keys, int n, int row0, const void* id) {
    o.key.id = id; o.key.resident = nullptr;
    o.kps.resize(n); o.desc.resize((size_t)n * 32);
    for (int i = 0; i < n; i++) {
      const auto& kp = keys[i];
      o.kps[i] = orbx_keypoint{kp.pt.x, kp.pt.y, kp.size, kp.angle, kp.response, (int32_t)kp.octave};
      std::memcpy(&o.desc[(size_t)32 * i], mat_u8(F.mDescriptors, row0 + i), 32);
    }
    o.v = orbm_frame_view{n, o.kps.data(), o.desc.data(), nullptr, nullptr, F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY,
                          F.fx, F.fy, F.cx, F.cy, F.mbf, F.mb, F.mnScaleLevels, F.mfScaleFactor};
  };
  fill(L, F.mvKeys, nl, 0, &F);
  fill(R, F.mvKeysRight, nr, nl, reinterpret_cast<const char*>(&F) + 1);
}
// ... and ALL of its features as one frame (mvKeys then mvKeysRight: the index space of F.mFeatVec and of mvpMapPoints), for SearchByBoW
template <class Ops, class FrameT>
void flatten_rig_frame_all(const FrameT& F, FrameFlat& o) {
  const int N = F.N, nl = F.Nleft;
  o.key.id = reinterpret_cast<const char*>(&F) + 2; o.key.resident = nullptr;
  o.kps.resize(N); o.desc.resize((size_t)N * 32);
  for (int i = 0; i < N; i++) {
    const auto& kp = i < nl ? F.mvKeys[i] : F.mvKeysRight[i - nl];
    o.kps[i] = orbx_keypoint{kp.pt.x, kp.pt.y, kp.size, kp.angle, kp.response, (int32_t)kp.octave};
    std::memcpy(&o.desc[(size_t)32 * i], mat_u8(F.mDescriptors, i), 32);
  }
  o.v = orbm_frame_view{N, o.kps.data(), o.desc.data(), nullptr, nullptr, F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY,
                        F.fx, F.fy, F.cx, F.cy, F.mbf, F.mb, F.mnScaleLevels, F.mfScaleFactor};
}
template <class Ops, class = void> struct has_rig_matcher : std::false_type {};
template <class Ops> struct has_rig_matcher<Ops, decltype((void)&Ops::search_mps_rig)> : std::true_type {};

template <class FeatVecT>
struct FeatVecFlat {          // the std::map in key order
  std::vector<uint32_t> node, start, feat;
  orbm_featvec_view v;
  explicit FeatVecFlat(const FeatVecT& fv) {
    start.push_back(0);
    for (const auto& kv : fv) {
      node.push_back((uint32_t)kv.first);
      for (unsigned idx : kv.second) feat.push_back(idx);
      start.push_back((uint32_t)feat.size());
    }
    v = orbm_featvec_view{(int32_t)node.size(), node.data(), start.data(), feat.data()};
  }
};

// The cameras of a KeyFrame / Frame as the optimiser's edges use them (I/CameraModels/GeometricCamera.h:77-92): returns true -- and
// fills `rig` -- when the edges cannot be written with the five pinhole scalars alone: a second camera (mpCamera2, the two-fisheye
// rig, with mTrl) or a first one that is not a pinhole (monocular fisheye).
template <class CamT> inline void fill_camera(orbg_camera& c, CamT* cam) {
  c.model = (int32_t)cam->GetType();
  c.fx = cam->getParameter(0); c.fy = cam->getParameter(1); c.cx = cam->getParameter(2); c.cy = cam->getParameter(3);
  for (int i = 0; i < 4; i++) c.k[i] = cam->size() > (size_t)(4 + i) ? cam->getParameter(4 + i) : 0.f;
}
template <class ObjT> inline bool make_rig(ObjT* obj, orbg_camera_rig& rig) {
  std::memset(&rig, 0, sizeof(rig));
  if (!obj->mpCamera) return false;
  if (!obj->mpCamera2 && obj->mpCamera->GetType() == ORBG_CAM_PINHOLE) return false;
  fill_camera(rig.left, obj->mpCamera);
  if (obj->mpCamera2) {
    rig.has_right = 1;
    fill_camera(rig.right, obj->mpCamera2);
    const float* Trl = mat_f32(obj->mTrl);                     // 3 x 4 (or 4 x 4) CV_32F, row-major: Converter::toSE3Quat reads rows 0..2
    std::memcpy(rig.Trl, Trl, 12 * sizeof(float));
  }
  return true;
}

inline size_t vertex_id(long unsigned id, unsigned client, bool is_kf) {       // Optimizer::GetID, I/Optimizer.h:104-112
  const size_t IDRANGE = 1000000, MAXAGENTS = 4;                                // I/Optimizer.h:23-24
  return is_kf ? IDRANGE * client + id : IDRANGE * (MAXAGENTS + client) + id;
}

// ------------------------------------------------------------------------------------------------ MapPoint access
// The scale-invariance range as the members hold it.  mfMinDistance / mfMaxDistance are PROTECTED in the reference
// (I/MapPoint.h:244,281-282) and its public getters return 0.8f * / 1.2f * the members (S/MapPoint.cc:617-627), from which the
// members cannot be recovered bit for bit -- and PredictScale (S/MapPoint.cc:646-661), which the device evaluates, needs the raw
// mfMaxDistance.  INTEGRATION.md's list of reference-side edits therefore adds two public getters, GetMinDistance() /
// GetMaxDistance() (edit E1); a MapPoint type whose members are accessible (a friend declaration, an older fork) works unchanged.
template <class MapPointT> auto min_distance_raw(MapPointT* p, int) -> decltype((float)p->GetMinDistance()) { return p->GetMinDistance(); }
template <class MapPointT> auto min_distance_raw(MapPointT* p, long) -> decltype((float)p->mfMinDistance) { return p->mfMinDistance; }
template <class MapPointT> auto max_distance_raw(MapPointT* p, int) -> decltype((float)p->GetMaxDistance()) { return p->GetMaxDistance(); }
template <class MapPointT> auto max_distance_raw(MapPointT* p, long) -> decltype((float)p->mfMaxDistance) { return p->mfMaxDistance; }

// ------------------------------------------------------------------------------------------------ isInFrustum (batch)
// void Frame::ComputeStereoFishEyeMatches(), S/Frame.cc:1093-1150 -- the left-right matcher of the two-fisheye Frame constructor, called
// where the reference calls it (after the two ExtractORB threads have joined: mvKeys / mvKeysRight, mDescriptors / mDescriptorsRight,
// monoLeft / monoRight are set, mDescriptors still holds the left camera's rows only).  Leaves mvLeftToRightMatch, mvRightToLeftMatch,
// mvDepth, mvuRight (-1 throughout), mvStereo3Dpoints and mnCloseMPs = 0 as the reference does.
template <class Ops = GpuOps, class FrameT>
int ComputeStereoFishEyeMatches(FrameT& F) {
  const int nl = (int)F.mvKeys.size(), nr = (int)F.mvKeysRight.size();
  std::vector<orbx_keypoint> kl(nl), kr(nr); std::vector<uint8_t> dl((size_t)nl * 32), dr((size_t)nr * 32);
  for (int i = 0; i < nl; i++) { const auto& kp = F.mvKeys[i]; kl[i] = orbx_keypoint{kp.pt.x, kp.pt.y, kp.size, kp.angle, kp.response, (int32_t)kp.octave};
                                 std::memcpy(&dl[(size_t)32 * i], mat_u8(F.mDescriptors, i), 32); }
  for (int i = 0; i < nr; i++) { const auto& kp = F.mvKeysRight[i]; kr[i] = orbx_keypoint{kp.pt.x, kp.pt.y, kp.size, kp.angle, kp.response, (int32_t)kp.octave};
                                 std::memcpy(&dr[(size_t)32 * i], mat_u8(F.mDescriptorsRight, i), 32); }
  orbx_fisheye_stereo_view v{};
  v.n_left = nl; v.n_right = nr; v.mono_left = F.monoLeft; v.mono_right = F.monoRight;
  v.kps_left = kl.data(); v.kps_right = kr.data(); v.desc_left = dl.data(); v.desc_right = dr.data();
  v.level_sigma2 = F.mvLevelSigma2.data(); v.n_levels = (int32_t)F.mvLevelSigma2.size();
  fill_camera(v.left, F.mpCamera); fill_camera(v.right, F.mpCamera2);
  std::memcpy(v.Tlr, mat_f32(F.mTlr), 12 * sizeof(float));
  std::vector<int32_t> l2r(std::max(nl, 1)), r2l(std::max(nr, 1)); std::vector<float> depth(std::max(nl, 1)), p3d(3 * (size_t)std::max(nl, 1));
  int n = 0;
  check(Ops::fisheye_stereo(v, l2r.data(), r2l.data(), depth.data(), p3d.data(), &n), "ComputeStereoFishEyeMatches");
  F.mvLeftToRightMatch.assign(l2r.begin(), l2r.begin() + nl); F.mvRightToLeftMatch.assign(r2l.begin(), r2l.begin() + nr);
  F.mvDepth.assign(depth.begin(), depth.begin() + nl); F.mvuRight.assign(nl, -1.f);
  F.mvStereo3Dpoints.assign(nl, decltype(F.mTcw)());
  for (int i = 0; i < nl; i++)
    if (l2r[i] >= 0) make_mat(F.mvStereo3Dpoints[i], 3, 1, &p3d[3 * (size_t)i]);
  F.mnCloseMPs = 0;
  return n;
}

// The loop of Tracking::SearchLocalPoints (S/Tracking.cc:3111-3128): F.isInFrustum(pMP, 0.5) for every candidate, which
// stores the mTrack* fields in the map point (S/Frame.cc:529-538) and counts the visible ones.
template <class Ops = GpuOps, class FrameT, class MapPointT>
int isInFrustumAll(FrameT& F, const std::vector<MapPointT*>& vpMPs, float viewingCosLimit = 0.5f) {
  const int M = (int)vpMPs.size();
  FrameFlat ff; flatten_frame<Ops>(F, ff);
  std::vector<float> pos(3 * (size_t)M), nrm(3 * (size_t)M), dmin(M), dmax(M); std::vector<uint8_t> desc(32 * (size_t)M), bad(M);
  std::vector<int32_t> nobs(M);
  for (int i = 0; i < M; i++) {
    MapPointT* p = vpMPs[i];
    const auto X = p->GetWorldPos(); const auto nv = p->GetNormal();
    std::memcpy(&pos[3 * (size_t)i], mat_f32(X), 12); std::memcpy(&nrm[3 * (size_t)i], mat_f32(nv), 12);
    // the raw members (INTEGRATION.md edit E1); the 0.8 / 1.2 factors of GetMin/MaxDistanceInvariance (S/MapPoint.cc:617-627) are
    // applied inside the call
    dmin[i] = min_distance_raw(p, 0); dmax[i] = max_distance_raw(p, 0);
    const auto Dm = p->GetDescriptor();
    std::memcpy(&desc[32 * (size_t)i], mat_u8(Dm, 0), 32);
    bad[i] = p->isBad(); nobs[i] = p->Observations();
  }
  orbm_worldpoints_view wv{M, pos.data(), nrm.data(), dmin.data(), dmax.data(), desc.data(), nobs.data(), bad.data(), nullptr};
  std::vector<uint8_t> inv(M); std::vector<float> px(M), py(M), pxr(M), dep(M), vc(M); std::vector<int32_t> lvl(M);
  check(Ops::is_in_frustum(ff.key, ff.v, mat_f32(F.mTcw), wv, viewingCosLimit, inv.data(), px.data(), py.data(), pxr.data(), dep.data(),
                           lvl.data(), vc.data()), "isInFrustum");
  int n = 0;
  for (int i = 0; i < M; i++) {
    MapPointT* p = vpMPs[i];
    p->mbTrackInView = inv[i] != 0;
    if (inv[i]) { p->mTrackProjX = px[i]; p->mTrackProjY = py[i]; p->mTrackProjXR = pxr[i]; p->mTrackDepth = dep[i];
                  p->mnTrackScaleLevel = lvl[i]; p->mTrackViewCos = vc[i]; n++; }
  }
  return n;
}

// ------------------------------------------------------------------------------------------------ Tracking::SearchLocalPoints
// void Tracking::SearchLocalPoints(), S/Tracking.cc:3083-3155, with mCurrentFrame = F and mvpLocalMapPoints = vpLocalMapPoints:
// the first loop (points the frame already holds: drop bad ones, IncreaseVisible, mnLastFrameSeen) runs here as it does there;
// the second loop (isInFrustum per point) and matcher.SearchByProjection(F, points, th, bFarPoints, thFarPoints) are ONE device
// pass over the flattened points (world position, normal, distance range, descriptor, flags -- one GetWorldPos / GetNormal /
// GetDescriptor clone per point, as the reference's isInFrustum + SearchByProjection take).  Side effects kept: IncreaseVisible()
// and mbTrackInView for the points in the frustum (the other mTrack* fields are only read by the search itself and, for the
// viewer, through F.mmProjectPoints: call isInFrustumAll instead when they are needed).  `th` is the value the reference
// derives from the sensor / IMU / relocalisation state (:3131-3151).  Returns the number of matches (nToMatch == 0: 0).
//
// What is read when (round 6: the reference's semantics are the default).
//  * A MapPoint WITHOUT a change counter -- the reference's class as it is, plus edit E1 -- : every call reads every point's world
//    position, normal, distance range and descriptor, exactly as the reference's isInFrustum + SearchByProjection do (three clones
//    per point: 97 of the 148 us this body takes at C2 sizes).  No cache, no assumption.
//  * A MapPoint WITH the change counter of INTEGRATION.md edit E2 (std::atomic<unsigned long> mnChangeStamp, incremented once per
//    call by the six mutators of the fields read here, after their stores): the calling thread keeps the flattened static fields of
//    its last call and the device keeps the map that call uploaded; a point is re-read if and only if its counter moved -- by the
//    local BA's write-back, ProcessNewKeyFrame, a fusion, anybody.  Exact as well: nothing is assumed about which code paths
//    change a point.  The per-frame fields -- mnLastFrameSeen, isBad(), Observations() -- are read on every call in both forms.
//  * OPT-IN, counter-less cache (-DORBGPU_DROPIN_HEURISTIC_LOCAL_MAP, or an entry-point set with kHeuristicLocalMap = true), for
//    an integrator who wants the cached speed without touching MapPoint.cc: statics are reused while the map, its
//    GetMapChangeIndex() and the pointer sequence are the same and a point's Observations() and distance range are what they were
//    when it was read, for at most kLocalMapMaxAge calls.  It is NOT the reference's semantics: a point re-described without a
//    change of any of those (an erase + add pair on one point within one frame time; SetWorldPos by a thread that does not move the
//    change index) is seen only at the next refresh.  dropin_parity pins exactly that difference.
// -DORBGPU_DROPIN_EXACT_LOCAL_MAP forces the first form even with a counter (A/B measurements).
// A Map type without GetMapChangeIndex() compiles in the two exact forms; the heuristic one needs it (static_assert below).
// Frame::isInFrustum for every candidate of a two-camera Frame (S/Frame.cc:545-554): both cameras' checks in one device pass; the
// fields land in the map points as isInFrustumChecks leaves them (:1212-1227), the flags and the -1 levels of :546-551 included.
// Returns the number of points either camera sees.
template <class Ops, class FrameT, class MapPointT>
int isInFrustumRigAll(FrameT& F, const FrameFlat& L, const orbg_camera_rig& rig, const std::vector<MapPointT*>& vpMPs, float viewingCosLimit) {
  const int M = (int)vpMPs.size();
  std::vector<float> pos(3 * (size_t)M), nrm(3 * (size_t)M), dmin(M), dmax(M); std::vector<uint8_t> bad(M); std::vector<int32_t> nobs(M);
  for (int i = 0; i < M; i++) {
    MapPointT* p = vpMPs[i];
    const auto X = p->GetWorldPos(); const auto nv = p->GetNormal();
    std::memcpy(&pos[3 * (size_t)i], mat_f32(X), 12); std::memcpy(&nrm[3 * (size_t)i], mat_f32(nv), 12);
    dmin[i] = min_distance_raw(p, 0); dmax[i] = max_distance_raw(p, 0);
    bad[i] = p->isBad(); nobs[i] = p->Observations();
  }
  orbm_worldpoints_view wv{M, pos.data(), nrm.data(), dmin.data(), dmax.data(), nullptr, nobs.data(), bad.data(), nullptr};
  std::vector<uint8_t> inv[2]; std::vector<float> px[2], py[2], dep[2], vc[2]; std::vector<int32_t> lvl[2];
  for (int s = 0; s < 2; s++) { inv[s].resize(M); px[s].resize(M); py[s].resize(M); dep[s].resize(M); vc[s].resize(M); lvl[s].resize(M); }
  check(Ops::is_in_frustum_rig(L.key, L.v, mat_f32(F.mTcw), rig, mat_f32(F.mTlr), wv, viewingCosLimit, inv[0].data(), px[0].data(), py[0].data(),
                               dep[0].data(), lvl[0].data(), vc[0].data(), inv[1].data(), px[1].data(), py[1].data(), dep[1].data(), lvl[1].data(),
                               vc[1].data()), "isInFrustum (two cameras)");
  int n = 0;
  for (int i = 0; i < M; i++) {
    MapPointT* p = vpMPs[i];
    p->mbTrackInView = inv[0][i] != 0; p->mbTrackInViewR = inv[1][i] != 0;
    p->mnTrackScaleLevel = lvl[0][i]; p->mnTrackScaleLevelR = lvl[1][i];                     // (-1 where the checks failed)
    if (inv[0][i]) { p->mTrackProjX = px[0][i]; p->mTrackProjY = py[0][i]; p->mTrackDepth = dep[0][i]; p->mTrackViewCos = vc[0][i]; }
    if (inv[1][i]) { p->mTrackProjXR = px[1][i]; p->mTrackProjYR = py[1][i]; p->mTrackDepthR = dep[1][i]; p->mTrackViewCosR = vc[1][i]; }
    n += inv[0][i] || inv[1][i];
  }
  return n;
}

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints) on a two-camera Frame
// (S/ORBmatcher.cc:44-214 with the right camera's block): the track fields of both cameras as isInFrustum left them in the points.
template <class Ops, class FrameT, class MapPointT>
int SearchByProjectionRig(FrameT& F, const FrameFlat& L, const FrameFlat& R, const std::vector<MapPointT*>& vpMapPoints, const float th,
                          const bool bFarPoints, const float thFarPoints, float mfNNratio) {
  const int M = (int)vpMapPoints.size();
  std::vector<uint8_t> inv(M), invr(M), bad(M), desc(32 * (size_t)M); std::vector<float> px(M), py(M), dep(M), vc(M), pxr(M), pyr(M), vcr(M);
  std::vector<int32_t> lvl(M), lvlr(M), nobs(M);
  for (int i = 0; i < M; i++) {
    MapPointT* p = vpMapPoints[i];
    inv[i] = p->mbTrackInView; invr[i] = p->mbTrackInViewR; bad[i] = p->isBad(); nobs[i] = p->Observations(); dep[i] = p->mTrackDepth;
    px[i] = p->mTrackProjX; py[i] = p->mTrackProjY; lvl[i] = p->mnTrackScaleLevel; vc[i] = p->mTrackViewCos;
    pxr[i] = p->mTrackProjXR; pyr[i] = p->mTrackProjYR; lvlr[i] = p->mnTrackScaleLevelR; vcr[i] = p->mTrackViewCosR;
    const auto Dm = p->GetDescriptor();
    std::memcpy(&desc[32 * (size_t)i], mat_u8(Dm, 0), 32);
  }
  orbm_mappoints_view mv{M, inv.data(), bad.data(), px.data(), py.data(), px.data(), dep.data(), lvl.data(), vc.data(), desc.data(), nobs.data()};
  orbm_mappoints_view mvr{M, invr.data(), bad.data(), pxr.data(), pyr.data(), pxr.data(), dep.data(), lvlr.data(), vcr.data(), desc.data(), nobs.data()};
  std::vector<int32_t> l2r(F.mvLeftToRightMatch.begin(), F.mvLeftToRightMatch.end()), r2l(F.mvRightToLeftMatch.begin(), F.mvRightToLeftMatch.end());
  l2r.resize((size_t)F.Nleft, -1); r2l.resize((size_t)(F.N - F.Nleft), -1);
  std::vector<int32_t> amp, aob; flatten_assignments(F, amp, aob);
  int n = 0;
  check(Ops::search_mps_rig(L.key, L.v, R.key, R.v, mv, mvr, l2r.data(), r2l.data(), th, bFarPoints, thFarPoints, mfNNratio, amp.data(), aob.data(), &n),
        "SearchByProjection(F, MPs), two cameras");
  for (int i = 0; i < F.N; i++)
    if (amp[i] >= 0 && amp[i] != INT32_MAX) F.mvpMapPoints[i] = vpMapPoints[amp[i]];
  return n;
}

template <class Ops, class FrameT, class MapPointT>
int SearchByProjection(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, const float th, const bool bFarPoints, const float thFarPoints, float mfNNratio);
// Tracking::SearchLocalPoints on a monocular Frame whose camera is a model (Nleft == -1, mpCamera a KannalaBrandt8): isInFrustum's
// Nleft == -1 branch projects through mpCamera->project (S/Frame.cc:489); the search itself is the single-camera one.
template <class Ops, class FrameT, class MapPointT>
int SearchLocalPointsModelCamera(FrameT& F, const orbg_camera_rig& rig, const std::vector<MapPointT*>& vpLocalMapPoints, float th, bool bFarPoints,
                                 float thFarPoints, float mfNNratio) {
  std::vector<MapPointT*> cand;
  for (MapPointT* pMP : vpLocalMapPoints) {
    if (pMP->mnLastFrameSeen == F.mnId) continue;
    if (pMP->isBad()) continue;
    cand.push_back(pMP);
  }
  if (cand.empty()) return 0;
  const int M = (int)cand.size();
  FrameFlat ff; flatten_frame<Ops>(F, ff);                     // (a device-resident frame serves as it is: only its bounds, scale table and features are read)
  std::vector<float> pos(3 * (size_t)M), nrm(3 * (size_t)M), dmin(M), dmax(M); std::vector<uint8_t> bad(M, 0); std::vector<int32_t> nobs(M);
  for (int i = 0; i < M; i++) {
    MapPointT* p = cand[i];
    const auto X = p->GetWorldPos(); const auto nv = p->GetNormal();
    std::memcpy(&pos[3 * (size_t)i], mat_f32(X), 12); std::memcpy(&nrm[3 * (size_t)i], mat_f32(nv), 12);
    dmin[i] = min_distance_raw(p, 0); dmax[i] = max_distance_raw(p, 0); nobs[i] = p->Observations();
  }
  orbm_worldpoints_view wv{M, pos.data(), nrm.data(), dmin.data(), dmax.data(), nullptr, nobs.data(), bad.data(), nullptr};
  std::vector<uint8_t> inv(M); std::vector<float> px(M), py(M), dep(M), vc(M); std::vector<int32_t> lvl(M);
  check(Ops::is_in_frustum_rig(ff.key, ff.v, mat_f32(F.mTcw), rig, nullptr, wv, 0.5f, inv.data(), px.data(), py.data(), dep.data(), lvl.data(), vc.data(),
                               nullptr, nullptr, nullptr, nullptr, nullptr, nullptr), "isInFrustum (camera model)");
  int nToMatch = 0;
  for (int i = 0; i < M; i++) {
    MapPointT* p = cand[i];
    p->mbTrackInView = inv[i] != 0;
    if (!inv[i]) continue;
    p->mTrackProjX = px[i]; p->mTrackProjY = py[i]; p->mTrackDepth = dep[i]; p->mnTrackScaleLevel = lvl[i]; p->mTrackViewCos = vc[i];
    p->mTrackProjXR = px[i];                                  // (uv.x - mbf * invz with mbf = 0 on a monocular frame; read only for features with uRight)
    p->IncreaseVisible(); nToMatch++;
  }
  if (nToMatch == 0) return 0;
  return SearchByProjection<Ops>(F, cand, th, bFarPoints, thFarPoints, mfNNratio);
}

// Tracking::SearchLocalPoints on a two-camera Frame: the same three steps (S/Tracking.cc:3083-3155), every point read on every call.
template <class Ops, class FrameT, class MapPointT>
int SearchLocalPointsRig(FrameT& F, const std::vector<MapPointT*>& vpLocalMapPoints, float th, bool bFarPoints, float thFarPoints, float mfNNratio) {
  orbg_camera_rig rig;
  if (!make_rig(&F, rig) || !rig.has_right) throw std::runtime_error("orbgpu dropin: a Frame with Nleft != -1 needs mpCamera2");
  std::vector<MapPointT*> cand;
  for (MapPointT* pMP : vpLocalMapPoints) {                                                   // :3112-3115
    if (pMP->mnLastFrameSeen == F.mnId) continue;
    if (pMP->isBad()) continue;
    cand.push_back(pMP);
  }
  if (cand.empty()) return 0;
  FrameFlat L, R; flatten_rig_frame<Ops>(F, L, R);
  isInFrustumRigAll<Ops>(F, L, rig, cand, 0.5f);
  int nToMatch = 0;
  for (MapPointT* pMP : cand)
    if (pMP->mbTrackInView || pMP->mbTrackInViewR) { pMP->IncreaseVisible(); nToMatch++; }    // :3117-3121
  if (nToMatch == 0) return 0;
  // (the reference hands SearchByProjection ALL local points: the ones left out here were seen in this frame -- both flags cleared
  //  above -- or are bad, and SearchByProjection passes over either, :53-60)
  return SearchByProjectionRig<Ops>(F, L, R, cand, th, bFarPoints, thFarPoints, mfNNratio);
}

constexpr unsigned kLocalMapMaxAge = 30;
struct LocalMapCache {
  std::vector<const void*> ptrs;
  std::vector<float> pos, nrm, dmin, dmax;
  std::vector<int32_t> nobs_seen;                  // Observations() when the statics of point i were read
  std::vector<unsigned long> stamp;                // MapPoint::mnChangeStamp when they were read (points that carry one)
  std::vector<uint8_t> desc, have;                 // have[i]: the statics of point i were read (bad points are never read)
  const void* map = nullptr; long long change_index = -1;
  unsigned age = 0;
  std::unordered_map<const void*, int> index; bool index_valid = false;
  void invalidate() { ptrs.clear(); map = nullptr; change_index = -1; age = 0; index.clear(); index_valid = false; }
};
template <class Ops> inline LocalMapCache& local_map_cache() { static thread_local LocalMapCache c; return c; }    // (one per entry-point set)
// An entry-point set with `kNoChangeStamp = true` behaves as if MapPoint had no counter (tests: the mocks carry one).
template <class T, class = void> struct has_change_stamp : std::false_type {};
template <class T> struct has_change_stamp<T, std::void_t<decltype(std::declval<T&>().mnChangeStamp)>> : std::true_type {};
template <class Ops, class = void> struct stamp_off : std::false_type {};
template <class Ops> struct stamp_off<Ops, std::void_t<decltype(Ops::kNoChangeStamp)>> : std::integral_constant<bool, Ops::kNoChangeStamp> {};
template <class MapPointT> auto stamp_of(const MapPointT* p, int) -> decltype((unsigned long)p->mnChangeStamp) { return (unsigned long)p->mnChangeStamp; }
template <class MapPointT> unsigned long stamp_of(const MapPointT*, long) { return 0; }
template <class Ops, class = void> struct heuristic_local_map : std::false_type {};
template <class Ops> struct heuristic_local_map<Ops, std::void_t<decltype(Ops::kHeuristicLocalMap)>> : std::integral_constant<bool, Ops::kHeuristicLocalMap> {};
template <class Ops, class = void> struct has_search_local_resident : std::false_type {};
template <class Ops> struct has_search_local_resident<Ops, decltype((void)&Ops::search_local_resident)> : std::true_type {};
template <class MapPointT> auto change_index_of(MapPointT* p, int) -> decltype((long long)p->GetMap()->GetMapChangeIndex()) {
  auto* m = p->GetMap();
  return m ? (long long)m->GetMapChangeIndex() : 0;
}
// An entry-point set may ask for the reference's exact behaviour (every point's statics read on every call): `static constexpr bool
// kExactLocalMap = true` (tests/cpp: the oracle's set, so that the product's cache is checked against uncached semantics)
template <class Ops, class = void> struct exact_local_map : std::false_type {};
template <class Ops> struct exact_local_map<Ops, typename std::enable_if<Ops::kExactLocalMap>::type> : std::true_type {};
template <class MapPointT> long long change_index_of(MapPointT*, long) { return -1; }       // (no GetMapChangeIndex(): only the exact forms may be used)
template <class MapPointT, class = void> struct has_change_index : std::false_type {};
template <class MapPointT> struct has_change_index<MapPointT, std::void_t<decltype(std::declval<MapPointT&>().GetMap()->GetMapChangeIndex())>> : std::true_type {};

template <class Ops = GpuOps, class FrameT, class MapPointT>
int SearchLocalPoints(FrameT& F, const std::vector<MapPointT*>& vpLocalMapPoints, float th, bool bFarPoints, float thFarPoints, float mfNNratio = 0.8f) {
  for (auto& pMP : F.mvpMapPoints) {                                                          // :3086-3103
    if (!pMP) continue;
    if (pMP->isBad()) { pMP = nullptr; continue; }
    pMP->IncreaseVisible();
    pMP->mnLastFrameSeen = F.mnId;
    pMP->mbTrackInView = false;
    pMP->mbTrackInViewR = false;
  }
  const int M = (int)vpLocalMapPoints.size();
  if (M == 0) return 0;
  if (F.Nleft != -1) {                                                                        // a two-camera Frame
    if constexpr (has_rig_matcher<Ops>::value) return SearchLocalPointsRig<Ops>(F, vpLocalMapPoints, th, bFarPoints, thFarPoints, mfNNratio);
    else throw std::runtime_error("orbgpu dropin: this entry-point set has no two-camera matcher");
  }
  if constexpr (has_rig_matcher<Ops>::value) {                                                // a monocular Frame whose camera is a model (a fisheye)
    orbg_camera_rig rig1;
    if (make_rig(&F, rig1)) return SearchLocalPointsModelCamera<Ops>(F, rig1, vpLocalMapPoints, th, bFarPoints, thFarPoints, mfNNratio);
  }
  FrameFlat ff; flatten_frame<Ops>(F, ff);
  LocalMapCache& C = local_map_cache<Ops>();
  // ---- per-frame fields of every point (:3112-3115) + which map / change index the points belong to
  static thread_local std::vector<uint8_t> bad, skip, excl;
  static thread_local std::vector<int32_t> nobs;
  bad.resize(M); skip.resize(M); excl.resize(M); nobs.resize(M);
  const void* map = nullptr; long long ci = 0;
  for (int i = 0; i < M; i++) {
    MapPointT* p = vpLocalMapPoints[i];
    skip[i] = p->mnLastFrameSeen == F.mnId;
    bad[i] = p->isBad();
    excl[i] = skip[i] | bad[i];
    nobs[i] = bad[i] ? 0 : p->Observations();
    if (!map && !bad[i]) { map = (const void*)p->GetMap(); ci = change_index_of(p, 0); }
  }
  // ---- static fields: read (the default without a counter) / reuse, patch, re-read (counter: exact; heuristic: opt-in)
#ifdef ORBGPU_DROPIN_EXACT_LOCAL_MAP
  constexpr bool kStamped = false, kHeuristic = false;
#else
  constexpr bool kStamped = has_change_stamp<MapPointT>::value && !stamp_off<Ops>::value && !exact_local_map<Ops>::value;
#ifdef ORBGPU_DROPIN_HEURISTIC_LOCAL_MAP
  constexpr bool kHeuristic = !kStamped && !exact_local_map<Ops>::value;
#else
  constexpr bool kHeuristic = !kStamped && !exact_local_map<Ops>::value && heuristic_local_map<Ops>::value;
#endif
#endif
  static_assert(!kHeuristic || has_change_index<MapPointT>::value, "the counter-less local-map cache needs MapPoint::GetMap()->GetMapChangeIndex() (I/Map.h): "
                                                                    "without it nothing signals that the local BA moved the points");
  auto read_statics = [&](int i) {
    MapPointT* p = vpLocalMapPoints[i];
    C.stamp[i] = stamp_of(p, 0);                       // BEFORE the fields: a change during the read is seen on the next call
    const auto X = p->GetWorldPos(); const auto nv = p->GetNormal(); const auto Dm = p->GetDescriptor();
    std::memcpy(&C.pos[3 * (size_t)i], mat_f32(X), 12); std::memcpy(&C.nrm[3 * (size_t)i], mat_f32(nv), 12);
    C.dmin[i] = min_distance_raw(p, 0); C.dmax[i] = max_distance_raw(p, 0);
    std::memcpy(&C.desc[32 * (size_t)i], mat_u8(Dm, 0), 32);
    C.nobs_seen[i] = nobs[i];
    C.have[i] = 1;
  };
  // a cached point is DIRTY when its observation count or its distance range is not what it was when its statics were read
  auto dirty = [&](int i) {
    MapPointT* p = vpLocalMapPoints[i];
    if (kStamped) return C.stamp[i] != stamp_of(p, 0);
    return C.nobs_seen[i] != nobs[i] || C.dmin[i] != min_distance_raw(p, 0) || C.dmax[i] != max_distance_raw(p, 0);
  };
  const bool fresh = !(kStamped || kHeuristic) || C.ptrs.empty() || C.map != map || (kHeuristic && (C.change_index != ci || C.age >= kLocalMapMaxAge));
  bool statics_same = false;
  if (!fresh && (int)C.ptrs.size() == M && std::memcmp(C.ptrs.data(), vpLocalMapPoints.data(), sizeof(void*) * (size_t)M) == 0) {
    statics_same = true;
    for (int i = 0; i < M; i++) {                    // (a point that was bad when the cache was filled cannot come back: isBad is final)
      if (bad[i]) continue;
      if (!C.have[i] || dirty(i)) { read_statics(i); statics_same = false; }
    }
    C.age++;
  } else if (!fresh) {
    // another pointer sequence on the same map state: keep what is known, read the new points
    if (!C.index_valid) { C.index.clear(); C.index.reserve(C.ptrs.size() * 2); for (size_t j = 0; j < C.ptrs.size(); j++) if (C.have[j]) C.index.emplace(C.ptrs[j], (int)j); }
    std::vector<float> pos(3 * (size_t)M), nrm(3 * (size_t)M), dmin(M), dmax(M); std::vector<uint8_t> desc(32 * (size_t)M), have(M, 0);
    std::vector<int32_t> seen(M, 0);
    std::vector<unsigned long> stamps(M, 0);
    std::vector<int> todo;
    for (int i = 0; i < M; i++) {
      const auto it = C.index.find((const void*)vpLocalMapPoints[i]);
      MapPointT* p = vpLocalMapPoints[i];
      const size_t j = it != C.index.end() ? (size_t)it->second : 0;
      const bool known = it != C.index.end() && !bad[i] &&
                         (kStamped ? C.stamp[j] == stamp_of(p, 0) : C.nobs_seen[j] == nobs[i] && C.dmin[j] == min_distance_raw(p, 0) && C.dmax[j] == max_distance_raw(p, 0));
      if (known) {
        std::memcpy(&pos[3 * (size_t)i], &C.pos[3 * j], 12); std::memcpy(&nrm[3 * (size_t)i], &C.nrm[3 * j], 12);
        dmin[i] = C.dmin[j]; dmax[i] = C.dmax[j]; std::memcpy(&desc[32 * (size_t)i], &C.desc[32 * j], 32); have[i] = 1; seen[i] = C.nobs_seen[j];
        stamps[i] = C.stamp[j];
      } else if (!bad[i]) todo.push_back(i);
    }
    C.pos.swap(pos); C.nrm.swap(nrm); C.dmin.swap(dmin); C.dmax.swap(dmax); C.desc.swap(desc); C.have.swap(have); C.nobs_seen.swap(seen); C.stamp.swap(stamps);
    C.ptrs.assign((const void* const*)vpLocalMapPoints.data(), (const void* const*)vpLocalMapPoints.data() + M);
    for (int i : todo) read_statics(i);
    C.index_valid = false; C.age++;
  } else {
    C.pos.assign(3 * (size_t)M, 0.f); C.nrm.assign(3 * (size_t)M, 0.f); C.dmin.assign(M, 0.f); C.dmax.assign(M, 0.f); C.desc.assign(32 * (size_t)M, 0); C.have.assign(M, 0);
    C.nobs_seen.assign(M, 0); C.stamp.assign(M, 0);
    C.ptrs.assign((const void* const*)vpLocalMapPoints.data(), (const void* const*)vpLocalMapPoints.data() + M);
    for (int i = 0; i < M; i++) if (!bad[i]) read_statics(i);
    C.map = map; C.change_index = ci; C.age = 0; C.index_valid = false;
  }
  std::vector<int32_t> amp, aob; flatten_assignments(F, amp, aob);
  static thread_local std::vector<uint8_t> vis;
  vis.assign(M, 0);
  int n = 0;
  const orbm_worldpoints_view wv{M, C.pos.data(), C.nrm.data(), C.dmin.data(), C.dmax.data(), C.desc.data(), nobs.data(), bad.data(), skip.data()};
  if constexpr (has_search_local_resident<Ops>::value)
    check(Ops::search_local_resident(ff.key, ff.v, mat_f32(F.mTcw), wv, excl.data(), statics_same, th, bFarPoints, thFarPoints, mfNNratio, amp.data(),
                                     aob.data(), &n, vis.data()), "SearchLocalPoints");
  else
    check(Ops::search_local(ff.key, ff.v, mat_f32(F.mTcw), wv, th, bFarPoints, thFarPoints, mfNNratio, amp.data(), aob.data(), &n, vis.data()),
          "SearchLocalPoints");
  int nToMatch = 0;
  for (int i = 0; i < M; i++) {
    if (excl[i]) continue;
    MapPointT* p = vpLocalMapPoints[i];
    p->mbTrackInView = vis[i] != 0;                                                           // S/Frame.cc:468,529
    if (vis[i]) { p->IncreaseVisible(); nToMatch++; }                                         // :3118-3122
  }
  if (nToMatch == 0) return 0;                                                                // :3129 (nothing was written: no query was valid)
  for (int i = 0; i < F.N; i++)
    if (amp[i] >= 0 && amp[i] != INT32_MAX) F.mvpMapPoints[i] = vpLocalMapPoints[amp[i]];       // S/ORBmatcher.cc:139
  return n;
}

// ------------------------------------------------------------------------------------------------ ORBmatcher
// int ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th, const bool bFarPoints,
//                                    const float thFarPoints), S/ORBmatcher.cc:44-214
template <class Ops = GpuOps, class FrameT, class MapPointT>
int SearchByProjection(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, const float th, const bool bFarPoints,
                       const float thFarPoints, float mfNNratio) {
  const int M = (int)vpMapPoints.size();
  if (F.Nleft != -1) {
    if constexpr (has_rig_matcher<Ops>::value) { FrameFlat L, R; flatten_rig_frame<Ops>(F, L, R); return SearchByProjectionRig<Ops>(F, L, R, vpMapPoints, th, bFarPoints, thFarPoints, mfNNratio); }
    else throw std::runtime_error("orbgpu dropin: this entry-point set has no two-camera matcher");
  }
  FrameFlat ff; flatten_frame<Ops>(F, ff);
  std::vector<uint8_t> inv(M), bad(M), desc(32 * (size_t)M); std::vector<float> px(M), py(M), pxr(M), dep(M), vc(M);
  std::vector<int32_t> lvl(M), nobs(M);
  for (int i = 0; i < M; i++) {                                    // the fields isInFrustum() stored (S/Frame.cc:529-538)
    MapPointT* p = vpMapPoints[i];
    inv[i] = p->mbTrackInView; bad[i] = p->isBad(); px[i] = p->mTrackProjX; py[i] = p->mTrackProjY; pxr[i] = p->mTrackProjXR;
    dep[i] = p->mTrackDepth; lvl[i] = p->mnTrackScaleLevel; vc[i] = p->mTrackViewCos; nobs[i] = p->Observations();
    const auto Dm = p->GetDescriptor();
    std::memcpy(&desc[32 * (size_t)i], mat_u8(Dm, 0), 32);
  }
  orbm_mappoints_view mv{M, inv.data(), bad.data(), px.data(), py.data(), pxr.data(), dep.data(), lvl.data(), vc.data(), desc.data(), nobs.data()};
  std::vector<int32_t> amp, aob; flatten_assignments(F, amp, aob);
  int n = 0;
  check(Ops::search_mps(ff.key, ff.v, mv, th, bFarPoints, thFarPoints, mfNNratio, amp.data(), aob.data(), &n), "SearchByProjection(F, MPs)");
  for (int i = 0; i < F.N; i++)
    if (amp[i] >= 0 && amp[i] != INT32_MAX) F.mvpMapPoints[i] = vpMapPoints[amp[i]];         // :139
  return n;
}

// int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono),
// S/ORBmatcher.cc:1970-2186
template <class Ops = GpuOps, class FrameT>
int SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono, bool mbCheckOrientation) {
  const int NL = LastFrame.N;
  const bool kRig = CurrentFrame.Nleft != -1;
  FrameFlat ff, ffR;
  if (kRig) flatten_rig_frame<Ops>(CurrentFrame, ff, ffR); else flatten_frame<Ops>(CurrentFrame, ff);
  std::vector<uint8_t> valid(NL), outl(NL), desc(32 * (size_t)NL); std::vector<float> pos(3 * (size_t)NL), ang(NL);
  std::vector<int32_t> oct(NL), nobs(NL);
  for (int i = 0; i < NL; i++) {
    auto* p = LastFrame.mvpMapPoints[i];
    valid[i] = p != nullptr; outl[i] = LastFrame.mvbOutlier[i];
    const bool lastRight = LastFrame.Nleft != -1 && i >= LastFrame.Nleft;                     // :2018-2019, :2075-2077
    oct[i] = lastRight ? LastFrame.mvKeysRight[i - LastFrame.Nleft].octave : LastFrame.mvKeys[i].octave;
    ang[i] = LastFrame.Nleft == -1 ? LastFrame.mvKeysUn[i].angle : lastRight ? LastFrame.mvKeysRight[i - LastFrame.Nleft].angle : LastFrame.mvKeys[i].angle;
    if (p) { const auto Xm = p->GetWorldPos(); const auto Dm = p->GetDescriptor();
             std::memcpy(&pos[3 * (size_t)i], mat_f32(Xm), 12); std::memcpy(&desc[32 * (size_t)i], mat_u8(Dm, 0), 32);
             nobs[i] = p->Observations(); }
  }
  orbm_lastframe_view lv{NL, valid.data(), outl.data(), pos.data(), desc.data(), oct.data(), ang.data(), nobs.data(), {0}};
  std::memcpy(lv.Tcw, mat_f32(LastFrame.mTcw), 64);
  std::vector<int32_t> amp, aob; flatten_assignments(CurrentFrame, amp, aob);
  int n = 0;
  orbg_camera_rig rig1;
  const bool kModel = !kRig && has_rig_matcher<Ops>::value && make_rig(&CurrentFrame, rig1);          // a monocular Frame whose camera is a model
  if (kModel) {
    if constexpr (has_rig_matcher<Ops>::value) {
      check(Ops::search_frame_rig(ff.key, ff.v, FrameKey{}, orbm_frame_view{}, mat_f32(CurrentFrame.mTcw), rig1, lv, th, bMono, mbCheckOrientation, amp.data(),
                                  aob.data(), &n), "SearchByProjection(Cur, Last), camera model");
    }
  } else
  if (kRig) {
    if constexpr (has_rig_matcher<Ops>::value) {
      orbg_camera_rig rig;
      if (!make_rig(&CurrentFrame, rig) || !rig.has_right) throw std::runtime_error("orbgpu dropin: a Frame with Nleft != -1 needs mpCamera2");
      check(Ops::search_frame_rig(ff.key, ff.v, ffR.key, ffR.v, mat_f32(CurrentFrame.mTcw), rig, lv, th, bMono, mbCheckOrientation, amp.data(),
                                  aob.data(), &n), "SearchByProjection(Cur, Last), two cameras");
    } else throw std::runtime_error("orbgpu dropin: this entry-point set has no two-camera matcher");
  } else
  check(Ops::search_frame(ff.key, ff.v, mat_f32(CurrentFrame.mTcw), lv, th, bMono, mbCheckOrientation, amp.data(), aob.data(), &n),
        "SearchByProjection(Cur, Last)");
  for (int i = 0; i < CurrentFrame.N; i++) {
    if (amp[i] >= 0 && amp[i] != INT32_MAX) CurrentFrame.mvpMapPoints[i] = LastFrame.mvpMapPoints[amp[i]];   // :2077
    else if (amp[i] == -1) CurrentFrame.mvpMapPoints[i] = nullptr;      // rotation-histogram rejects (:2170) -- also of a feature that held a point without observations
  }
  return n;
}

// int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th,
//                                    const int ORBdist), S/ORBmatcher.cc:2188-2310 -- Tracking::Relocalization's guided search
template <class Ops = GpuOps, class FrameT, class KeyFrameT, class MapPointT>
int SearchByProjection(FrameT& CurrentFrame, KeyFrameT* pKF, const std::set<MapPointT*>& sAlreadyFound, const float th, const int ORBdist,
                       bool mbCheckOrientation) {
  const std::vector<MapPointT*> vpMPs = pKF->GetMapPointMatches();                            // :2203
  const int NK = (int)vpMPs.size();
  FrameFlat ff; flatten_frame<Ops>(CurrentFrame, ff);
  std::vector<float> pos(3 * (size_t)NK), nrm(3 * (size_t)NK, 0.f), dmin(NK, 0.f), dmax(NK, 1.f), ang(NK);
  std::vector<uint8_t> desc(32 * (size_t)NK), bad(NK, 1), found(NK, 0);
  std::vector<int32_t> nobs(NK, 0);
  for (int i = 0; i < NK; i++) {
    ang[i] = pKF->mvKeysUn[i].angle;                                                          // :2261
    MapPointT* p = vpMPs[i];
    if (!p || p->isBad()) continue;                                                           // :2209-2211
    bad[i] = 0;
    found[i] = sAlreadyFound.count(p) ? 1 : 0;                                                // :2211
    if (found[i]) continue;
    const auto X = p->GetWorldPos(); const auto Dm = p->GetDescriptor();
    std::memcpy(&pos[3 * (size_t)i], mat_f32(X), 12); std::memcpy(&desc[32 * (size_t)i], mat_u8(Dm, 0), 32);
    dmin[i] = min_distance_raw(p, 0); dmax[i] = max_distance_raw(p, 0);                       // raw members (edit E1): the 0.8 / 1.2 factors are applied inside
  }
  orbm_worldpoints_view wv{NK, pos.data(), nrm.data(), dmin.data(), dmax.data(), desc.data(), nobs.data(), bad.data(), nullptr};
  // CurrentFrame.mvpMapPoints[i2] != NULL blocks feature i2, whatever the point's observations (:2246-2247)
  std::vector<int32_t> amp(CurrentFrame.N);
  for (int i = 0; i < CurrentFrame.N; i++) amp[i] = CurrentFrame.mvpMapPoints[i] ? INT32_MAX : -1;
  int n = 0;
  orbg_camera_rig rig1;
  const bool kModel = has_rig_matcher<Ops>::value && CurrentFrame.Nleft == -1 && make_rig(&CurrentFrame, rig1);   // a monocular Frame whose camera is a model (:2217)
  if (kModel) {
    if constexpr (has_rig_matcher<Ops>::value) {
      check(Ops::search_reloc_cam(ff.key, ff.v, mat_f32(CurrentFrame.mTcw), rig1.left, wv, found.data(), ang.data(), th, ORBdist, mbCheckOrientation, amp.data(), &n),
            "SearchByProjection(Cur, KF, sAlreadyFound), camera model");
    }
  } else
  check(Ops::search_reloc(ff.key, ff.v, mat_f32(CurrentFrame.mTcw), wv, found.data(), ang.data(), th, ORBdist, mbCheckOrientation, amp.data(), &n),
        "SearchByProjection(Cur, KF, sAlreadyFound)");
  for (int i = 0; i < CurrentFrame.N; i++)
    if (amp[i] >= 0 && amp[i] != INT32_MAX) CurrentFrame.mvpMapPoints[i] = vpMPs[amp[i]];      // :2267 (rotation-vote rejects come back as -1, :2300)
  return n;
}

// int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches), S/ORBmatcher.cc:269-471
template <class Ops = GpuOps, class KeyFrameT, class FrameT, class MapPointT>
int SearchByBoW(KeyFrameT* pKF, FrameT& F, std::vector<MapPointT*>& vpMapPointMatches, float mfNNratio, bool mbCheckOrientation) {
  const std::vector<MapPointT*> vpMapPointsKF = pKF->GetMapPointMatches();
  const int NK = (int)vpMapPointsKF.size();
  const bool kRig = F.Nleft != -1;                                                            // a two-camera Frame: :342-430
  FrameFlat ff;
  if (kRig) flatten_rig_frame_all<Ops>(F, ff); else flatten_frame<Ops>(F, ff);
  std::vector<uint8_t> kdesc(32 * (size_t)NK), kvalid(NK); std::vector<float> kang(NK);
  for (int i = 0; i < NK; i++) {
    std::memcpy(&kdesc[32 * (size_t)i], mat_u8(pKF->mDescriptors, i), 32);
    kvalid[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();                               // :322-325
    kang[i] = !pKF->mpCamera2 ? pKF->mvKeysUn[i].angle                                          // :379-382
              : i >= pKF->NLeft ? pKF->mvKeysRight[i - pKF->NLeft].angle : pKF->mvKeys[i].angle;
  }
  FeatVecFlat<decltype(F.mFeatVec)> fF(F.mFeatVec);
  FeatVecFlat<decltype(pKF->mFeatVec)> fK(pKF->mFeatVec);
  std::vector<int32_t> matches(F.N, -1);
  int n = 0;
  if (kRig) {
    if constexpr (has_rig_matcher<Ops>::value)
      check(Ops::search_bow_rig(ff.key, ff.v, F.Nleft, fF.v, kdesc.data(), NK, kvalid.data(), kang.data(), fK.v, mfNNratio, mbCheckOrientation,
                                matches.data(), &n), "SearchByBoW(KF, F), two cameras");
    else throw std::runtime_error("orbgpu dropin: this entry-point set has no two-camera matcher");
  } else
  check(Ops::search_bow(ff.key, ff.v, fF.v, kdesc.data(), NK, kvalid.data(), kang.data(), fK.v, mfNNratio, mbCheckOrientation, matches.data(), &n),
        "SearchByBoW(KF, F)");
  vpMapPointMatches.assign(F.N, static_cast<MapPointT*>(nullptr));                           // :273
  for (int i = 0; i < F.N; i++)
    if (matches[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[matches[i]];
  return n;
}

// ------------------------------------------------------------------------------------------------ local-BA window cache
// Consecutive local-BA windows share ~95 % of their map points.  Reading a point costs the reference a std::map copy
// (GetObservations) and a matrix clone (GetWorldPos), both under the point's mutexes, plus a sort of its observations: two thirds of
// the glue's time.  A MapPoint type that carries a change counter
//     std::atomic<unsigned long> mnChangeStamp{0};     // +1 per call of AddObservation, EraseObservation, SetWorldPos, SetBadFlag,
//                                                      // UpdateNormalAndDepth, ComputeDistinctiveDescriptors (INTEGRATION.md edit E2)
// lets the calling thread (LocalMapping) keep each point's flattened observation list and position from window to window and
// re-read only the points whose counter moved -- by this function's own write-back, which knows what it wrote, or by anybody else.
// Without the member the points are read as the reference reads them; an entry-point set with `kNoLbaCache = true` also does (tests).
template <class Ops, class = void> struct lba_cache_off : std::false_type {};
template <class Ops> struct lba_cache_off<Ops, std::void_t<decltype(Ops::kNoLbaCache)>> : std::integral_constant<bool, Ops::kNoLbaCache> {};

template <class KeyFrameT, class MapPointT>
struct LbaWindowCache {
  // li / u v ur w: the left camera's observation; ri / u2 v2 w2: the right camera's (rigs only: get<1>(indexes) - NLeft, -1 otherwise)
  struct Obs { KeyFrameT* kf; size_t kf_vid; int li; float u, v, ur, w; int ri; float u2, v2, w2; };
  struct Rec {
    MapPointT* mp = nullptr; long unsigned id = 0, stamp = 0, seen = 0; bool valid = false;
    size_t vid = 0; float pos[3] = {0, 0, 0};
    std::vector<Obs> obs;                    // every observation (li < 0 ones too: their keyframes still count as fixed cameras), by keyframe vertex id
  };
  std::vector<Rec> recs;
  std::vector<int32_t> tab;                  // open addressing on the point's address -> index into recs
  long unsigned call = 0;
  static LbaWindowCache& instance() { static thread_local LbaWindowCache c; return c; }
  size_t slot_of(const void* p) const { return ((reinterpret_cast<uintptr_t>(p) >> 4) * 0x9E3779B97F4A7C15ull >> 16) & (tab.size() - 1); }
  void rebuild_table(size_t want) {
    size_t n = 1024;
    while (n < 2 * want) n *= 2;
    tab.assign(n, -1);
    for (size_t i = 0; i < recs.size(); i++) {
      size_t h = slot_of(recs[i].mp);
      while (tab[h] >= 0) h = (h + 1) & (tab.size() - 1);
      tab[h] = (int32_t)i;
    }
  }
  int32_t find_or_add(MapPointT* mp) {
    if (tab.empty() || 2 * (recs.size() + 1) > tab.size()) rebuild_table(2 * (recs.size() + 1));
    size_t h = slot_of(mp);
    while (tab[h] >= 0) {
      if (recs[tab[h]].mp == mp) return tab[h];
      h = (h + 1) & (tab.size() - 1);
    }
    recs.emplace_back();
    recs.back().mp = mp;
    tab[h] = (int32_t)recs.size() - 1;
    return tab[h];
  }
  // points that left the window stay until they outnumber the window four to one
  void drop_unseen(size_t window) {
    if (recs.size() <= 4 * window + 1024) return;
    size_t k = 0;
    for (size_t i = 0; i < recs.size(); i++)
      if (recs[i].seen + 8 >= call) { if (k != i) recs[k] = std::move(recs[i]); k++; }
    recs.resize(k);
    rebuild_table(recs.size());
  }
  void clear() { recs.clear(); tab.clear(); }
};

// ------------------------------------------------------------------------------------------------ Optimizer
// void Optimizer::LocalBundleAdjustment(KeyFrame *pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int LocalBASize),
// S/Optimizer.cc:1810-2410.  Graph collection (:1813-1908) and write-back (:2263-2408) act on the caller's objects exactly
// as the reference does; the g2o block in between (:1917-2261) is one lba_solve.  Returns the LBA_* status for callers
// that want it (the reference returns void).
template <class Ops = GpuOps, class KeyFrameT, class MapT>
int LocalBundleAdjustment(KeyFrameT* pKF, bool* pbStopFlag, MapT* pMap, int& num_fixedKF, int /*LocalBASize: unused in the reference too*/) {
  using MapPointT = typename std::remove_pointer<typename std::decay<decltype(pKF->GetMapPointMatches())>::type::value_type>::type;
  ORBGPU_GLUE_T0();
  // ---- local keyframes: pKF and its covisible neighbours in this map (:1813-1826)
  std::list<KeyFrameT*> lLocalKeyFrames{pKF};
  pKF->mnBALocalForKF = pKF->mnId;
  auto* pCurrentMap = pKF->GetMap();
  for (KeyFrameT* n : pKF->GetVectorCovisibleKeyFrames()) {
    n->mnBALocalForKF = pKF->mnId;
    if (!n->isBad() && n->GetMap() == pCurrentMap) lLocalKeyFrames.push_back(n);
  }
  // ---- local map points: seen by the local keyframes (:1828-1854)
  num_fixedKF = 0;
  static thread_local std::vector<MapPointT*> lLocalMapPoints;       // (a std::list in the reference: 2000 nodes allocated and freed per window)
  lLocalMapPoints.clear();
  for (KeyFrameT* kf : lLocalKeyFrames) {
    if (kf->mnId == pMap->GetInitKFid()) num_fixedKF = 1;
    for (MapPointT* mp : kf->GetMapPointMatches())
      if (mp && !mp->isBad() && mp->GetMap() == pCurrentMap && mp->mnBALocalForKF != pKF->mnId) {
        lLocalMapPoints.push_back(mp);
        mp->mnBALocalForKF = pKF->mnId;
      }
  }
  ORBGPU_GLUE_T("local keyframes + points");
  // ---- fixed keyframes: observe local points without being local (:1856-1873).  GetObservations() hands out a COPY of the
  // point's std::map (under the point's mutex): it is taken once per point and kept for the edge pass below -- or, with the window
  // cache, taken only for the points that changed since this thread's last window
  constexpr bool kCached = has_change_stamp<MapPointT>::value && !lba_cache_off<Ops>::value;
  using Cache = LbaWindowCache<KeyFrameT, MapPointT>;
  using ObsMap = typename std::decay<decltype(std::declval<MapPointT>().GetObservations())>::type;
  struct LocalPoint { MapPointT* mp; ObsMap obs; size_t vid; };
  std::vector<LocalPoint> vLP;
  std::vector<int32_t> vRec;                 // (cached path) the cache record of each local point
  std::list<KeyFrameT*> lFixedCameras;
  auto note_camera = [&](KeyFrameT* kf) {
    if (kf->mnBALocalForKF != pKF->mnId && kf->mnBAFixedForKF != pKF->mnId) {
      kf->mnBAFixedForKF = pKF->mnId;
      if (!kf->isBad() && kf->GetMap() == pCurrentMap) lFixedCameras.push_back(kf);
    }
  };
  // (cached path) keyframes by vertex id relative to the window's keyframe: the marks of a keyframe are looked at once per window,
  // not once per observation, and the edge pass finds a keyframe's column without hashing its address
  constexpr size_t kVidSpan = 4096;
  const size_t vid_hi = vertex_id(pKF->mnId, pKF->mnClientId, true);
  static thread_local std::vector<uint8_t> kfSeen; static thread_local std::vector<int32_t> kfCol;
  if constexpr (kCached) {
    Cache& C = Cache::instance();
    C.call++;
    kfSeen.assign(kVidSpan, 0);
    vRec.reserve(lLocalMapPoints.size());
    for (MapPointT* mp : lLocalMapPoints) {
      const int32_t ri = C.find_or_add(mp);
      typename Cache::Rec& rc = C.recs[ri];
      const long unsigned stamp = stamp_of(mp, 0);          // read BEFORE the point: a change during the read is seen next time
      if (!rc.valid || rc.id != mp->mnId || rc.stamp != stamp) {
        rc.id = mp->mnId; rc.stamp = stamp; rc.valid = true; rc.vid = vertex_id(mp->mnId, mp->mnClientId, false);
        rc.obs.clear();
        for (const auto& ob : mp->GetObservations()) {
          KeyFrameT* kf = ob.first;
          typename Cache::Obs o{kf, vertex_id(kf->mnId, kf->mnClientId, true), std::get<0>(ob.second), 0.f, 0.f, 0.f, 0.f, -1, 0.f, 0.f, 0.f};
          if (o.li >= 0) {
            const auto& kp = kf->mvKeysUn[o.li];
            o.u = kp.pt.x; o.v = kp.pt.y; o.ur = kf->mvuRight[o.li]; o.w = kf->mvInvLevelSigma2[kp.octave];
          }
          if (kf->mpCamera2 && std::get<1>(ob.second) != -1) {                               // :2086-2093
            o.ri = std::get<1>(ob.second) - kf->NLeft;
            const auto& kp = kf->mvKeysRight[o.ri];
            o.u2 = kp.pt.x; o.v2 = kp.pt.y; o.w2 = kf->mvInvLevelSigma2[kp.octave];
          }
          rc.obs.push_back(o);
        }
        std::sort(rc.obs.begin(), rc.obs.end(), [](const typename Cache::Obs& a, const typename Cache::Obs& b) { return a.kf_vid < b.kf_vid; });
        const auto Xm = mp->GetWorldPos();
        std::memcpy(rc.pos, mat_f32(Xm), 12);
      }
      rc.seen = C.call;
      vRec.push_back(ri);
      for (const auto& o : rc.obs) {
        const size_t d = vid_hi - o.kf_vid;                  // (unsigned: a keyframe with a higher id, or of another client, wraps past the span)
        if (d < kVidSpan) { if (kfSeen[d]) continue; kfSeen[d] = 1; }
        note_camera(o.kf);
      }
    }
  } else {
    vLP.reserve(lLocalMapPoints.size());
    for (MapPointT* mp : lLocalMapPoints) {
      vLP.push_back(LocalPoint{mp, mp->GetObservations(), vertex_id(mp->mnId, mp->mnClientId, false)});
      for (const auto& ob : vLP.back().obs) note_camera(ob.first);
    }
  }
  ORBGPU_GLUE_T("observations + fixed cameras");
  num_fixedKF += (int)lFixedCameras.size();
  if (num_fixedKF < 2) {
    // fewer than two fixed keyframes leave the scale free: the one / two local keyframes with the lowest ids are fixed (:1875-1910)
    KeyFrameT *low = nullptr, *second = nullptr;
    long unsigned lowId = pKF->mnId, secondId = pKF->mnId;
    for (KeyFrameT* kf : lLocalKeyFrames) {
      if (kf == pKF || kf->mnId == pMap->GetInitKFid()) continue;
      if (kf->mnId < lowId) { lowId = kf->mnId; low = kf; }
      else if (kf->mnId < secondId) { secondId = kf->mnId; second = kf; }
    }
    if (low) { lFixedCameras.push_back(low); lLocalKeyFrames.remove(low); num_fixedKF++; }
    if (num_fixedKF < 2 && second) { lFixedCameras.push_back(second); lLocalKeyFrames.remove(second); num_fixedKF++; }
  }
  // ---- the flattened problem: vertices in ascending g2o id, edges per point in observation order
  std::vector<KeyFrameT*> vKF(lLocalKeyFrames.begin(), lLocalKeyFrames.end());
  const size_t n_local = vKF.size();
  vKF.insert(vKF.end(), lFixedCameras.begin(), lFixedCameras.end());
  std::vector<std::pair<KeyFrameT*, bool>> kf_fixed(vKF.size());
  for (size_t i = 0; i < vKF.size(); i++) kf_fixed[i] = {vKF[i], i >= n_local || vKF[i]->mnId == pMap->GetInitKFid()};   // :1940, :1951
  std::sort(kf_fixed.begin(), kf_fixed.end(), [](const auto& a, const auto& b) {
    return vertex_id(a.first->mnId, a.first->mnClientId, true) < vertex_id(b.first->mnId, b.first->mnClientId, true); });
  // keyframe -> index in the flattened problem: a few dozen keyframes, looked up once per observation: a small open-addressing
  // table keyed by the keyframe's address
  size_t tab_size = 64;
  while (tab_size < 4 * vKF.size()) tab_size *= 2;
  // (per keyframe: its column, its vertex id and whether its observations make edges -- !isBad() && GetMap() == pCurrentMap, :2003 --
  // evaluated once per call instead of once per observation: both are stable while this thread, LocalMapping, is in here)
  struct KfRec { KeyFrameT* kf; int col; size_t vid; bool usable; };
  std::vector<KfRec> kfTab(tab_size, KfRec{nullptr, -1, 0, false});
  auto kf_slot = [&](KeyFrameT* kf) { return ((reinterpret_cast<uintptr_t>(kf) >> 4) * 0x9E3779B97F4A7C15ull >> 20) & (tab_size - 1); };
  std::vector<float> poses(16 * vKF.size()); std::vector<uint8_t> fixed(vKF.size());
  for (size_t i = 0; i < vKF.size(); i++) {
    vKF[i] = kf_fixed[i].first;
    size_t h = kf_slot(vKF[i]);
    while (kfTab[h].kf) h = (h + 1) & (tab_size - 1);
    kfTab[h] = KfRec{vKF[i], (int)i, vertex_id(vKF[i]->mnId, vKF[i]->mnClientId, true), !vKF[i]->isBad() && vKF[i]->GetMap() == pCurrentMap};
    const auto Tm = vKF[i]->GetPose();
    std::memcpy(&poses[16 * i], mat_f32(Tm), 64);
    fixed[i] = kf_fixed[i].second;
  }
  auto kf_rec = [&](KeyFrameT* kf) -> const KfRec* {
    for (size_t h = kf_slot(kf);; h = (h + 1) & (tab_size - 1)) {
      if (kfTab[h].kf == kf) return &kfTab[h];
      if (!kfTab[h].kf) return nullptr;
    }
  };
  auto kf_index = [&](KeyFrameT* kf) -> int { const KfRec* r = kf_rec(kf); return r ? r->col : -1; };
  ORBGPU_GLUE_T("keyframe table");
  // the points in ascending vertex id: a permutation is sorted, not the records (each holds a std::map)
  const size_t n_pts = kCached ? vRec.size() : vLP.size();
  std::vector<std::pair<size_t, uint32_t>> order(n_pts);
  if constexpr (kCached) { for (size_t j = 0; j < n_pts; j++) order[j] = {Cache::instance().recs[vRec[j]].vid, (uint32_t)j}; }
  else { for (size_t j = 0; j < n_pts; j++) order[j] = {vLP[j].vid, (uint32_t)j}; }
  // (a window's points mostly come in ascending id already: nothing to do then)
  if (!std::is_sorted(order.begin(), order.end())) std::sort(order.begin(), order.end());
  ORBGPU_GLUE_T("sort points");
  std::vector<MapPointT*> vMP(n_pts);
  std::vector<int32_t> recOf(kCached ? n_pts : 0);           // (cached path) record of problem point j
  // (the flat arrays of a call live in the calling thread -- LocalMapping -- across calls: no allocation, no zero fill per keyframe)
  static thread_local std::vector<float> pts, oposes, opts; static thread_local std::vector<lba_edge> edges;
  static thread_local std::vector<uint8_t> eout, edep; static thread_local std::vector<double> echi;
  static thread_local std::vector<std::pair<KeyFrameT*, MapPointT*>> edgeOwner;
  pts.clear(); edges.clear(); edgeOwner.clear();
  pts.reserve(3 * n_pts); edges.reserve(8 * n_pts); edgeOwner.reserve(8 * n_pts);
  if constexpr (kCached) {
    Cache& C = Cache::instance();
    kfCol.assign(kVidSpan, -2);                              // -2: not looked up yet, -1: not in the problem / not usable (:2003)
    for (size_t j = 0; j < n_pts; j++) {
      const int32_t ri = vRec[order[j].second];
      const typename Cache::Rec& rc = C.recs[ri];
      vMP[j] = rc.mp; recOf[j] = ri;
      pts.insert(pts.end(), rc.pos, rc.pos + 3);
      for (const auto& o : rc.obs) {                         // already by keyframe vertex id
        if (o.li < 0 && o.ri < 0) continue;                                                  // :2007, :2086
        const size_t d = vid_hi - o.kf_vid;
        int32_t col;
        if (d < kVidSpan && kfCol[d] != -2) col = kfCol[d];
        else {
          const KfRec* kr = kf_rec(o.kf);
          col = kr && kr->usable ? kr->col : -1;                                             // :2003
          if (d < kVidSpan) kfCol[d] = col;
        }
        if (col < 0) continue;
        if (o.li >= 0) {
          edges.push_back(lba_edge{col, (int32_t)j, o.u, o.v, o.ur /* < 0: monocular (:2007) */, o.w});
          edgeOwner.push_back({o.kf, rc.mp});
        }
        if (o.ri >= 0) {                                                                     // the right camera's edge (:2086-2120)
          edges.push_back(lba_edge{col, (int32_t)j, o.u2, o.v2, LBA_UR_RIGHT_CAMERA, o.w2});
          edgeOwner.push_back({o.kf, rc.mp});
        }
      }
    }
  } else {
    struct ObsRef { size_t vid; KeyFrameT* kf; int li; int col; int ri; };
    std::vector<ObsRef> obs;
    for (size_t j = 0; j < n_pts; j++) {
      const LocalPoint& lp = vLP[order[j].second];
      MapPointT* mp = lp.mp;
      vMP[j] = mp;
      const auto Xm = mp->GetWorldPos();                     // a clone (S/MapPoint.cc:GetWorldPos): keep it alive while it is read
      const float* X = mat_f32(Xm);
      pts.insert(pts.end(), X, X + 3);
      // the reference walks a std::map<KeyFrame*, ...> (address order); here: by vertex id, which only permutes sums (E-6)
      obs.clear();
      for (const auto& ob : lp.obs) {
        KeyFrameT* kf = ob.first;
        const KfRec* kr = kf_rec(kf);
        if (!kr || !kr->usable) continue;                                                    // :2003
        const int li = std::get<0>(ob.second);
        const int ri = kf->mpCamera2 && std::get<1>(ob.second) != -1 ? std::get<1>(ob.second) - kf->NLeft : -1;   // :2086-2093
        if (li < 0 && ri < 0) continue;                                                      // :2007, :2086
        obs.push_back(ObsRef{kr->vid, kf, li, kr->col, ri});
      }
      std::sort(obs.begin(), obs.end(), [](const ObsRef& a, const ObsRef& b) { return a.vid < b.vid; });
      for (const ObsRef& o : obs) {
        if (o.li >= 0) {
          const auto& kp = o.kf->mvKeysUn[o.li];
          edges.push_back(lba_edge{o.col, (int32_t)j, kp.pt.x, kp.pt.y, o.kf->mvuRight[o.li] /* < 0: monocular (:2007) */, o.kf->mvInvLevelSigma2[kp.octave]});
          edgeOwner.push_back({o.kf, mp});
        }
        if (o.ri >= 0) {                                                                     // the right camera's edge (:2086-2120)
          const auto& kp = o.kf->mvKeysRight[o.ri];
          edges.push_back(lba_edge{o.col, (int32_t)j, kp.pt.x, kp.pt.y, LBA_UR_RIGHT_CAMERA, o.kf->mvInvLevelSigma2[kp.octave]});
          edgeOwner.push_back({o.kf, mp});
        }
      }
    }
  }
  ORBGPU_GLUE_T("points + edges");
  lba_problem P{(int32_t)vKF.size(), (int32_t)vMP.size(), (int32_t)edges.size(), poses.data(), fixed.data(), pts.data(), edges.data(),
                pKF->fx, pKF->fy, pKF->cx, pKF->cy, pKF->mbf, pMap->IsInertial() ? 100.0 : 0.0 /* :1924-1925 */, 5, 10, 0, nullptr};
  // (the keyframes of a map hold the same camera objects: the window's cameras are the current keyframe's)
  orbg_camera_rig rig;
  if (make_rig(pKF, rig)) P.rig = &rig;
  oposes.resize(poses.size()); opts.resize(pts.size()); eout.resize(edges.size()); edep.resize(edges.size()); echi.resize(edges.size());
  lba_result R{}; R.poses = oposes.data(); R.points = opts.data(); R.edge_outlier = eout.data(); R.edge_depth_pos = edep.data(); R.edge_chi2 = echi.data();
  // *pbStopFlag is LocalMapping::mbAbortBA, a bool Tracking raises through InterruptBA() while this runs (S/LocalMapping.cc:381-386):
  // the pointer goes to the library unchanged and is polled there between LM iterations / trials, where g2o polls it
  // (G/core/sparse_optimizer.cpp:376, G/core/optimization_algorithm_levenberg.cpp:149) and for bDoMore (:2135-2139)
  ORBGPU_GLUE_T("result buffers");
  check(Ops::lba(P, pbStopFlag, R), "LocalBundleAdjustment");
  ORBGPU_GLUE_T("solve (Ops::lba)");
  if (R.status != LBA_APPLIED) return R.status;                                              // :2127-2129 and :2257-2261: nothing is written
  std::vector<std::pair<KeyFrameT*, MapPointT*>> vToErase;                                   // :2207-2253
  std::vector<uint8_t> obsErased(kCached ? n_pts : 0, 0);    // (cached path) points that lose an observation below: re-read next time
  for (size_t k = 0; k < edges.size(); k++)
    if (!edgeOwner[k].second->isBad() && eout[k]) { vToErase.push_back(edgeOwner[k]); if (kCached) obsErased[edges[k].point] = 1; }
  std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);                                   // :2263
  for (auto& e : vToErase) { e.first->EraseMapPointMatch(e.second); e.second->EraseObservation(e.first); }   // :2279-2286
  for (KeyFrameT* kf : lLocalKeyFrames) {                                                    // :2318-2372 (SetPose)
    decltype(kf->GetPose()) T; make_mat(T, 4, 4, &oposes[16 * (size_t)kf_index(kf)]);
    kf->SetPose(T, true);                                                                    // :2327 (bLock: mbPoseLock on the server)
  }
  if (!vMP.empty()) {                                                                        // :2375-2383
    decltype(vMP[0]->GetWorldPos()) X; make_mat(X, 3, 1, &opts[0]);                          // ONE 3 x 1 matrix, refilled per point:
    float* Xd = const_cast<float*>(mat_f32(X));                                              // SetWorldPos copies it (Pos.copyTo, S/MapPoint.cc:125)
    for (size_t j = 0; j < vMP.size(); j++) {
      Xd[0] = opts[3 * j]; Xd[1] = opts[3 * j + 1]; Xd[2] = opts[3 * j + 2];
      vMP[j]->SetWorldPos(X, true);                                                          // :2386
      vMP[j]->UpdateNormalAndDepth();
      if constexpr (kCached) {
        // What this thread knows of the point is current again IF nobody else touched it since it was read at the top of this call
        // (the solve ran without the map mutex, and the point's own mutexes are not held here): the counter must have moved by
        // exactly this thread's own two mutator calls (edit E2: once per call of SetWorldPos / UpdateNormalAndDepth).  Anything
        // else -- AddObservation / EraseObservation / SetWorldPos / SetBadFlag by Tracking or the Communicator meanwhile -- and
        // the record is dropped: the point is read again in the next window.
        typename Cache::Rec& rc = Cache::instance().recs[recOf[j]];
        const long unsigned now = stamp_of(vMP[j], 0);
        if (obsErased[j] || now - rc.stamp != 2) rc.valid = false;
        else { rc.pos[0] = Xd[0]; rc.pos[1] = Xd[1]; rc.pos[2] = Xd[2]; rc.stamp = now; }
      }
    }
  }
  if constexpr (kCached) Cache::instance().drop_unseen(n_pts);
  pMap->IncreaseChangeIndex();                                                               // :2397 (Tracking reads it: mbMapUpdated)
  ORBGPU_GLUE_T("write-back");
  return R.status;
}

// int Optimizer::PoseOptimization(Frame *pFrame), S/Optimizer.cc:964-1278: the rectified-stereo / mono frames of every BASELINE
// configuration (mpCamera2 == NULL, :1012-1083) and the two-fisheye rig (:1085-1151: features i < Nleft are mvKeys[i] seen by mpCamera,
// the others mvKeysRight[i - Nleft] seen by mpCamera2 after mTrl)
template <class Ops = GpuOps, class FrameT>
int PoseOptimization(FrameT* pFrame) {
  const int N = pFrame->N;
  std::vector<float> Xw, u, v, ur, w; std::vector<int> featIdx;
  const bool two_cameras = pFrame->mpCamera2 != nullptr;
  for (int i = 0; i < N; i++) {
    auto* pMP = pFrame->mvpMapPoints[i];
    if (!pMP) continue;
    pFrame->mvbOutlier[i] = false;                                                           // :1030, :1061, :1094, :1128
    const bool right = two_cameras && i >= pFrame->Nleft;
    const auto& kp = !two_cameras ? pFrame->mvKeysUn[i] : right ? pFrame->mvKeysRight[i - pFrame->Nleft] : pFrame->mvKeys[i];
    const auto Xm = pMP->GetWorldPos();
    const float* X = mat_f32(Xm);
    Xw.insert(Xw.end(), X, X + 3); u.push_back(kp.pt.x); v.push_back(kp.pt.y);
    ur.push_back(!two_cameras ? pFrame->mvuRight[i] : right ? LBA_UR_RIGHT_CAMERA : -1.f);
    w.push_back(pFrame->mvInvLevelSigma2[kp.octave]); featIdx.push_back(i);
  }
  if (featIdx.size() < 3) return 0;                                                          // :1160-1161
  pose_opt_problem P{(int32_t)featIdx.size(), Xw.data(), u.data(), v.data(), ur.data(), w.data(), pFrame->fx, pFrame->fy, pFrame->cx, pFrame->cy,
                     pFrame->mbf, {0}, 0, nullptr};
  orbg_camera_rig rig;
  if (make_rig(pFrame, rig)) P.rig = &rig;
  std::memcpy(P.Tcw, mat_f32(pFrame->mTcw), 64);
  std::vector<uint8_t> outlier(featIdx.size());
  pose_opt_result R{}; R.outlier = outlier.data();
  check(Ops::pose_opt(P, R), "PoseOptimization");
  for (size_t k = 0; k < featIdx.size(); k++) pFrame->mvbOutlier[featIdx[k]] = outlier[k] != 0;
  decltype(pFrame->mTcw) T; make_mat(T, 4, 4, R.Tcw);
  pFrame->SetPose(T);                                                                        // :1273-1275
  return R.n_inliers;                                                                        // nInitialCorrespondences - nBad
}

}  // namespace dropin
}  // namespace orbgpu

#endif  // ORBGPU_DROPIN_HPP_
