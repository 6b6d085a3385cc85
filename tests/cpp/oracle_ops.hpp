// The glue's entry points (include/orbgpu_dropin.hpp: `Ops`) over the CPU oracle: views instead of device handles.  Test infrastructure,
// shared by tests/cpp/dropin_parity.cpp and tests/cpp/closed_loop.cpp; the product's set is orbgpu::dropin::GpuOps.
#pragma once
#include <cstdint>
#include <vector>

#include "orbgpu_dropin.hpp"
#include "../../oracle/orb_oracle.h"

namespace od = orbgpu::dropin;

struct OracleOps {       // the same entry points over the CPU oracle: views instead of device handles
  static constexpr bool kUsesResidentFrame = false;
  static constexpr bool kExactLocalMap = true;      // the reference's semantics: every point's fields are read on every call (no cache)
  static constexpr bool kNoLbaCache = true;         // ... and every local-BA window reads every point (no window cache)
  static int is_in_frustum(const od::FrameKey&, const orbm_frame_view& v, const float* Tcw, const orbm_worldpoints_view& pts, float lim, uint8_t* in_view,
                           float* px, float* py, float* pxr, float* depth, int32_t* level, float* vcos) {
    return oracle_is_in_frustum(&v, Tcw, &pts, lim, in_view, px, py, pxr, depth, level, vcos);
  }
  static int search_mps(const od::FrameKey&, const orbm_frame_view& v, const orbm_mappoints_view& mps, float th, int far_points, float th_far, float nnratio,
                        int32_t* amp, int32_t* aob, int* n) {
    return oracle_search_by_projection_mps(&v, &mps, th, far_points, th_far, nnratio, amp, aob, n);
  }
  static int search_frame(const od::FrameKey&, const orbm_frame_view& v, const float* Tcw, const orbm_lastframe_view& last, float th, int mono, int check_ori,
                          int32_t* amp, int32_t* aob, int* n) {
    return oracle_search_by_projection_frame(&v, Tcw, &last, th, mono, check_ori, amp, aob, n);
  }
  static int is_in_frustum_rig(const od::FrameKey&, const orbm_frame_view& v, const float* Tcw, const orbg_camera_rig& rig, const float* Tlr,
                               const orbm_worldpoints_view& pts, float lim, uint8_t* in_view, float* px, float* py, float* depth, int32_t* level,
                               float* vcos, uint8_t* in_view_r, float* px_r, float* py_r, float* depth_r, int32_t* level_r, float* vcos_r) {
    return oracle_is_in_frustum_rig(&v, Tcw, &rig, Tlr, &pts, lim, in_view, px, py, depth, level, vcos, in_view_r, px_r, py_r, depth_r, level_r, vcos_r);
  }
  static int search_mps_rig(const od::FrameKey&, const orbm_frame_view& vl, const od::FrameKey&, const orbm_frame_view& vr, const orbm_mappoints_view& mps,
                            const orbm_mappoints_view& mps_r, const int32_t* l2r, const int32_t* r2l, float th, int far_points, float th_far,
                            float nnratio, int32_t* amp, int32_t* aob, int* n) {
    return oracle_search_by_projection_mps_rig(&vl, &vr, &mps, &mps_r, l2r, r2l, th, far_points, th_far, nnratio, amp, aob, n);
  }
  static int search_frame_rig(const od::FrameKey&, const orbm_frame_view& vl, const od::FrameKey&, const orbm_frame_view& vr, const float* Tcw,
                              const orbg_camera_rig& rig, const orbm_lastframe_view& last, float th, int mono, int check_ori, int32_t* amp,
                              int32_t* aob, int* n) {
    return oracle_search_by_projection_frame_rig(&vl, rig.has_right ? &vr : nullptr, Tcw, &rig, &last, th, mono, check_ori, amp, aob, n);
  }
  static int search_bow_rig(const od::FrameKey&, const orbm_frame_view& v_all, int n_left, const orbm_featvec_view& fvF, const uint8_t* kf_desc, int nkf,
                            const uint8_t* kf_valid, const float* kf_angle, const orbm_featvec_view& fvKF, float nnratio, int check_ori, int32_t* matches, int* n) {
    return oracle_search_by_bow_rig(&v_all, n_left, &fvF, kf_desc, nkf, kf_valid, kf_angle, &fvKF, nnratio, check_ori, matches, n);
  }
  static int fisheye_stereo(const orbx_fisheye_stereo_view& v, int32_t* l2r, int32_t* r2l, float* depth, float* p3d, int* n) {
    return oracle_fisheye_stereo_matches(&v, l2r, r2l, depth, p3d, n);
  }
  static int search_reloc_cam(const od::FrameKey&, const orbm_frame_view& v, const float* Tcw, const orbg_camera& cam, const orbm_worldpoints_view& kf_pts,
                              const uint8_t* found, const float* kf_angle, float th, int orb_dist, int check_ori, int32_t* amp, int* n) {
    return oracle_search_by_projection_reloc_cam(&v, Tcw, &cam, &kf_pts, found, kf_angle, th, orb_dist, check_ori, amp, n);
  }
  static int search_reloc(const od::FrameKey&, const orbm_frame_view& v, const float* Tcw, const orbm_worldpoints_view& kf_pts, const uint8_t* found,
                          const float* kf_angle, float th, int orb_dist, int check_ori, int32_t* amp, int* n) {
    return oracle_search_by_projection_reloc(&v, Tcw, &kf_pts, found, kf_angle, th, orb_dist, check_ori, amp, n);
  }
  static int search_bow(const od::FrameKey&, const orbm_frame_view& v, const orbm_featvec_view& fvF, const uint8_t* kf_desc, int nkf, const uint8_t* kf_valid,
                        const float* kf_angle, const orbm_featvec_view& fvKF, float nnratio, int check_ori, int32_t* matches, int* n) {
    return oracle_search_by_bow(&v, &fvF, kf_desc, nkf, kf_valid, kf_angle, &fvKF, nnratio, check_ori, matches, n);
  }
  static int search_local(const od::FrameKey&, const orbm_frame_view& v, const float* Tcw, const orbm_worldpoints_view& pts, float th, int far_points,
                          float th_far, float nnratio, int32_t* amp, int32_t* aob, int* n, uint8_t* in_frustum) {
    const int m = pts.m;
    std::vector<float> px(m), py(m), pxr(m), dep(m), vc(m); std::vector<int32_t> lvl(m);
    oracle_is_in_frustum(&v, Tcw, &pts, 0.5f, in_frustum, px.data(), py.data(), pxr.data(), dep.data(), lvl.data(), vc.data());
    for (int i = 0; i < m; i++)
      if ((pts.skip && pts.skip[i]) || pts.bad[i]) in_frustum[i] = 0;
    return oracle_search_local_points(&v, &pts, Tcw, th, far_points, th_far, nnratio, amp, aob, n);
  }
  // (the oracle polls an int32: the bool is sampled -- the cases that raise it DURING the solve go through OracleAtTrialOps)
  static int lba(const lba_problem& p, const volatile bool* stop, lba_result& r) {
    volatile int32_t s = (stop && *stop) ? 1 : 0;
    return oracle_lba_solve(&p, &s, &r);
  }
  static int pose_opt(const pose_opt_problem& p, pose_opt_result& r) { return oracle_pose_optimize(&p, &r); }
};

