// orbgpu_adapters.hpp -- header-only C++ host side above the C-ABI (include/orbgpu.h).
//
// Thin adapter classes that keep the reference's class / method names and argument meaning so that the Tracking and
// LocalMapping threads of yutongwangBIT/multi_orbslam3 call the MI355X path through the symbols they already use:
//   ORB_SLAM3::ORBextractor            I/ORBextractor.h:47-113, S/ORBextractor.cc:408-468,1068-1150
//   ORB_SLAM3::ORBmatcher              I/ORBmatcher.h:35-108,   S/ORBmatcher.cc:44-214,269-471,1970-2186
//   ORB_SLAM3::Optimizer::LocalBundleAdjustment   I/Optimizer.h:42, S/Optimizer.cc:1810-2410
// The reference's pointer graph (Frame, KeyFrame, MapPoint) stays on the host; these adapters take the flattened
// views of SURVEY.md Appendix E (see INTEGRATION.md for the 20-line glue that fills them from the reference objects).
// Error behaviour mirrors the reference: the extractor returns -1 on an empty image, the matchers return the match
// count, LocalBundleAdjustment returns silently when aborted / rejected; any other non-zero status (in particular
// ORBG_NO_DEVICE -- there is no CPU fallback) throws std::runtime_error.
//
// Define ORBGPU_DROPIN_NAMESPACE to also export the classes as ORB_SLAM3::*; define HAVE_OPENCV for the cv::Mat /
// cv::KeyPoint overloads with the reference's exact signatures.
#ifndef ORBGPU_ADAPTERS_HPP_
#define ORBGPU_ADAPTERS_HPP_

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "orbgpu.h"

#ifdef HAVE_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#endif

namespace orbgpu {

inline void check(int rc, const char* where) {
  if (rc != ORBG_OK) throw std::runtime_error(std::string(where) + ": " + orbg_strerror(rc));
}

// ------------------------------------------------------------------------------------------------ extractor
class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };   // I/ORBextractor.h:51

  // ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST), I/ORBextractor.h:53-54.
  // max_width / max_height only pre-size device buffers (they grow on demand); one instance per camera as in
  // S/Tracking.cc:145-151, or n_cams = 2 for the batched stereo rig.
  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int max_width = 752,
               int max_height = 480, int n_cams = 1, int device = 0)
      : nfeatures_(nfeatures), nlevels_(nlevels), scaleFactor_(scaleFactor), iniThFAST_(iniThFAST), minThFAST_(minThFAST) {
    orbx_config cfg{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_width, max_height, n_cams, device};
    check(orbx_create(&cfg, &h_), "orbx_create");
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels); mnFeaturesPerLevel.resize(nlevels);
    check(orbx_get_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                          mnFeaturesPerLevel.data()), "orbx_get_tables");
  }
  ~ORBextractor() { if (h_) orbx_destroy(h_); }
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  // Host destinations of the two-halves stereo constructor (FrameOnDevice::StereoCtorSubmit* / StereoCtorWait below): the Frame's
  // mvKeys / mDescriptors / mvuRight / mvDepth storage, cap entries each; StereoCtorWait fills them (orbx_set_frame_outputs).
  void SetFrameOutputs(orbx_keypoint* mvKeys, uint8_t* mDescriptors, float* mvuRight, float* mvDepth, int cap) {
    check(orbx_set_frame_outputs(h_, mvKeys, mDescriptors, mvuRight, mvDepth, cap), "orbx_set_frame_outputs");
  }

  // int operator()(InputArray image, InputArray mask, vector<KeyPoint>& keypoints, OutputArray descriptors,
  //                vector<int>& vLappingArea), S/ORBextractor.cc:1068-1150 -- flattened: 8-bit gray image, row stride.
  // Returns monoIndex, or -1 for an empty image (:1072-1073).
  int operator()(const uint8_t* image, int width, int height, int stride, std::vector<orbx_keypoint>& keypoints,
                 std::vector<uint8_t>& descriptors, const std::vector<int>& vLappingArea) {
    const int cap = 2 * nfeatures_ + 256;
    keypoints.resize(cap);
    descriptors.resize((size_t)cap * 32);
    int n = 0, n_mono = 0;
    const int rc = orbx_extract(h_, 0, image, width, height, stride, vLappingArea.at(0), vLappingArea.at(1), keypoints.data(),
                                descriptors.data(), cap, &n, &n_mono);
    if (rc == ORBG_EMPTY) { keypoints.clear(); descriptors.clear(); return -1; }
    check(rc, "orbx_extract");
    keypoints.resize(n);
    descriptors.resize((size_t)n * 32);
    return n_mono;
  }

  // Both ExtractORB calls of the stereo Frame ctor (S/Frame.cc:92-95) in one batched submission (n_cams == 2).
  void ExtractStereo(const uint8_t* left, const uint8_t* right, int width, int height, int stride,
                     std::vector<orbx_keypoint>& kpsL, std::vector<uint8_t>& descL, std::vector<orbx_keypoint>& kpsR,
                     std::vector<uint8_t>& descR) {
    const int cap = 2 * nfeatures_ + 256;
    kpsL.resize(cap); kpsR.resize(cap); descL.resize((size_t)cap * 32); descR.resize((size_t)cap * 32);
    int nl = 0, nr = 0;
    check(orbx_extract_stereo(h_, left, right, width, height, stride, kpsL.data(), descL.data(), cap, &nl, kpsR.data(),
                              descR.data(), cap, &nr), "orbx_extract_stereo");
    kpsL.resize(nl); kpsR.resize(nr); descL.resize((size_t)nl * 32); descR.resize((size_t)nr * 32);
  }

  // Frame::ComputeStereoMatches (S/Frame.cc:785-963) on the features of the last ExtractStereo.
  void ComputeStereoMatches(float bf, float b, std::vector<float>& mvuRight, std::vector<float>& mvDepth, int n_left) {
    mvuRight.assign(n_left, -1.0f);
    mvDepth.assign(n_left, -1.0f);
    check(orbx_stereo_match(h_, bf, b, mvuRight.data(), mvDepth.data()), "orbx_stereo_match");
  }

  // mvImagePyramid[level] (public member, I/ORBextractor.h:87): copy of one level without its border.
  void GetPyramidLevel(int cam, int level, std::vector<uint8_t>& out, int& width, int& height) {
    check(orbx_get_level(h_, cam, level, nullptr, &width, &height), "orbx_get_level");
    out.resize((size_t)width * height);
    check(orbx_get_level(h_, cam, level, out.data(), &width, &height), "orbx_get_level");
  }

#ifdef HAVE_OPENCV
  // The reference's exact signature (I/ORBextractor.h:61-63); the mask is ignored there too.
  int operator()(cv::InputArray _image, cv::InputArray, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors,
                 std::vector<int>& vLappingArea) {
    if (_image.empty()) return -1;
    cv::Mat image = _image.getMat();
    CV_Assert(image.type() == CV_8UC1);
    std::vector<orbx_keypoint> k;
    std::vector<uint8_t> d;
    const int mono = (*this)(image.data, image.cols, image.rows, (int)image.step, k, d, vLappingArea);
    _keypoints.resize(k.size());
    for (size_t i = 0; i < k.size(); i++)
      _keypoints[i] = cv::KeyPoint(k[i].x, k[i].y, k[i].size, k[i].angle, k[i].response, k[i].octave);
    if (k.empty()) { _descriptors.release(); return mono; }
    _descriptors.create((int)k.size(), 32, CV_8U);
    std::memcpy(_descriptors.getMat().data, d.data(), d.size());
    return mono;
  }
#endif

  // getters, I/ORBextractor.h:65-85
  int GetLevels() const { return nlevels_; }
  float GetScaleFactor() const { return scaleFactor_; }
  std::vector<float> GetScaleFactors() const { return mvScaleFactor; }
  std::vector<float> GetInverseScaleFactors() const { return mvInvScaleFactor; }
  std::vector<float> GetScaleSigmaSquares() const { return mvLevelSigma2; }
  std::vector<float> GetInverseScaleSigmaSquares() const { return mvInvLevelSigma2; }
  orbx_handle* handle() const { return h_; }

 protected:
  int nfeatures_, nlevels_;
  float scaleFactor_;
  int iniThFAST_, minThFAST_;
  std::vector<int> mnFeaturesPerLevel;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  orbx_handle* h_ = nullptr;
};

// ------------------------------------------------------------------------------------------------ frame + matcher
// Device-resident view of a Frame (features + grid).  Fill an orbm_frame_view from the Frame's members
//  or take the features straight from the extractor.
class FrameOnDevice {
 public:
  explicit FrameOnDevice(int cap_features = 4096, int device = 0) { check(orbm_frame_create(device, cap_features, &f_), "orbm_frame_create"); }
  ~FrameOnDevice() { if (f_) orbm_frame_destroy(f_); }
  FrameOnDevice(const FrameOnDevice&) = delete;
  FrameOnDevice& operator=(const FrameOnDevice&) = delete;
  void Upload(const orbm_frame_view& v) { check(orbm_frame_upload(f_, &v), "orbm_frame_upload"); n_ = v.n; }
  void FromExtractor(const ORBextractor& ex, orbm_frame_view v, int n_left) {
    v.n = n_left;
    check(orbm_frame_from_extractor(f_, ex.handle(), &v), "orbm_frame_from_extractor");
    n_ = n_left;
  }
  // Frame::Frame(imLeft, imRight, ...) numerical part (S/Frame.cc:71-172): ExtractORB(L) || ExtractORB(R) ->
  // ComputeStereoMatches -> AssignFeaturesToGrid with ONE submission and ONE host sync; the features stay on the device.
  // Host copies are optional (pass empty vectors' data() == nullptr to skip them).
  int StereoCtor(ORBextractor& ex, orbm_frame_view v, const uint8_t* left, const uint8_t* right, int width, int height, int stride,
                 std::vector<orbx_keypoint>* mvKeys = nullptr, std::vector<uint8_t>* mDescriptors = nullptr,
                 std::vector<float>* mvuRight = nullptr, std::vector<float>* mvDepth = nullptr, int cap = 4096) {
    if (mvKeys) mvKeys->resize(cap);
    if (mDescriptors) mDescriptors->resize((size_t)cap * 32);
    if (mvuRight) mvuRight->resize(cap);
    if (mvDepth) mvDepth->resize(cap);
    int nl = 0, nr = 0;
    check(orbx_frame_stereo(ex.handle(), f_, &v, left, right, width, height, stride, v.bf, v.b, mvKeys ? mvKeys->data() : nullptr,
                            mDescriptors ? mDescriptors->data() : nullptr, mvuRight ? mvuRight->data() : nullptr,
                            mvDepth ? mvDepth->data() : nullptr, cap, &nl, &nr), "orbx_frame_stereo");
    if (mvKeys) mvKeys->resize(nl);
    if (mDescriptors) mDescriptors->resize((size_t)nl * 32);
    if (mvuRight) mvuRight->resize(nl);
    if (mvDepth) mvDepth->resize(nl);
    n_ = nl;
    return nl;
  }
  // The same constructor in two halves for images that are already in device memory: Submit enqueues the chain and
  // returns, Wait completes it.  Between the two calls `ex` and this frame are busy; other extractors / frames (the
  // tracking of the previous frame) may be used and overlap on the GPU.
  void StereoCtorSubmit(ORBextractor& ex, orbm_frame_view v, const uint8_t* d_left, const uint8_t* d_right, int width, int height,
                        int stride) {
    check(orbx_frame_stereo_dev_submit(ex.handle(), f_, &v, d_left, d_right, width, height, stride, v.bf, v.b),
          "orbx_frame_stereo_dev_submit");
  }
  // ... and for HOST images (the cv::Mat data Tracking::GrabImageStereo holds): Submit packs the rows into the extractor's pinned
  // staging slot and enqueues copy + chain; with async_ingest the library's ingest thread does that and the images must stay
  // valid until StereoCtorWait.
  void StereoCtorSubmitHost(ORBextractor& ex, orbm_frame_view v, const uint8_t* left, const uint8_t* right, int width, int height,
                            int stride, bool async_ingest = false) {
    check(orbx_frame_stereo_submit(ex.handle(), f_, &v, left, right, width, height, stride, v.bf, v.b, async_ingest ? ORBX_SUBMIT_ASYNC : 0),
          "orbx_frame_stereo_submit");
  }
  int StereoCtorWait(ORBextractor& ex, int* n_right = nullptr) {
    int nl = 0, nr = 0;
    check(orbx_frame_stereo_dev_wait(ex.handle(), &nl, &nr), "orbx_frame_stereo_dev_wait");
    if (n_right) *n_right = nr;
    n_ = nl;
    return nl;
  }
  int N() const { return n_; }
  orbm_frame* handle() const { return f_; }

 private:
  orbm_frame* f_ = nullptr;
  int n_ = 0;
};

// Map points kept resident on the device (the local map of Tracking, or the candidate points of a server search).
class MapPointsOnDevice {
 public:
  explicit MapPointsOnDevice(int cap_points = 16384, int device = 0) { check(orbm_map_create(device, cap_points, &m_), "orbm_map_create"); }
  ~MapPointsOnDevice() { if (m_) orbm_map_destroy(m_); }
  MapPointsOnDevice(const MapPointsOnDevice&) = delete;
  MapPointsOnDevice& operator=(const MapPointsOnDevice&) = delete;
  void Upload(const orbm_worldpoints_view& v) { check(orbm_map_upload(m_, &v), "orbm_map_upload"); }
  // the resident map stays as it is; only MapPoint::Observations() of its points is refreshed (host side, no device traffic)
  void SetObservations(const int32_t* n_obs) { check(orbm_map_set_observations(m_, n_obs), "orbm_map_set_observations"); }
  orbm_map* handle() const { return m_; }

 private:
  orbm_map* m_ = nullptr;
};

class ORBmatcher {
 public:
  static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;   // S/ORBmatcher.cc:36-38
  ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

  // int SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th, bFarPoints, thFarPoints), :44-214.
  // mvpMapPoints is flattened as (assigned_mp, assigned_obs): SURVEY.md Appendix E-2.
  int SearchByProjection(FrameOnDevice& F, const orbm_mappoints_view& vpMapPoints, std::vector<int32_t>& assigned_mp,
                         std::vector<int32_t>& assigned_obs, const float th = 3, const bool bFarPoints = false,
                         const float thFarPoints = 50.0f) {
    int n = 0;
    check(orbm_search_by_projection_mps(F.handle(), &vpMapPoints, th, bFarPoints, thFarPoints, mfNNratio, assigned_mp.data(),
                                        assigned_obs.data(), &n), "orbm_search_by_projection_mps");
    return n;
  }
  // int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono), :1970-2186.
  int SearchByProjection(FrameOnDevice& CurrentFrame, const float* Tcw_current, const orbm_lastframe_view& LastFrame,
                         std::vector<int32_t>& assigned_mp, std::vector<int32_t>& assigned_obs, const float th, const bool bMono) {
    int n = 0;
    check(orbm_search_by_projection_frame(CurrentFrame.handle(), Tcw_current, &LastFrame, th, bMono, mbCheckOrientation,
                                          assigned_mp.data(), assigned_obs.data(), &n), "orbm_search_by_projection_frame");
    return n;
  }
  // Tracking::SearchLocalPoints body (S/Tracking.cc:3111-3153): isInFrustum(pMP, 0.5) for every candidate + the search
  // above in one launch, map points resident on the device.  skip[i] = 1 for points already matched in this frame.
  int SearchLocalPoints(FrameOnDevice& F, MapPointsOnDevice& vpLocalMapPoints, const float* Tcw, const uint8_t* skip,
                        std::vector<int32_t>& assigned_mp, std::vector<int32_t>& assigned_obs, const float th = 1,
                        const bool bFarPoints = false, const float thFarPoints = 50.0f) {
    int n = 0;
    check(orbm_search_local_points(F.handle(), vpLocalMapPoints.handle(), Tcw, skip, th, bFarPoints, thFarPoints, mfNNratio,
                                   assigned_mp.data(), assigned_obs.data(), &n), "orbm_search_local_points");
    return n;
  }
  // int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched, int th,
  //                        float ratioHamming), :473-587 (bWithKFs = false) and the overload with vpPointsKFs / vpMatchedKF,
  // :589-700 (bWithKFs = true; the caller sets vpMatchedKF[idx] = vpPointsKFs[vpMatched[idx]] for the entries written).
  int SearchByProjection(FrameOnDevice& pKF, const float* Scw, MapPointsOnDevice& vpPoints, const uint8_t* already_found,
                         std::vector<int32_t>& vpMatched, int th, float ratioHamming = 1.0f, bool bWithKFs = false) {
    int n = 0;
    check(orbm_search_by_projection_sim3(pKF.handle(), vpPoints.handle(), Scw, already_found, th, ratioHamming, bWithKFs ? 0 : 1,
                                         vpMatched.data(), &n), "orbm_search_by_projection_sim3");
    return n;
  }
  // int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12), :819-959.  vpMatches12[idx1] = idx2.
  int SearchByBoW(FrameOnDevice& pKF2, const orbm_featvec_view& fv2, const uint8_t* mp_valid2, const uint8_t* desc1, int n1,
                  const uint8_t* mp_valid1, const float* angle1, const orbm_featvec_view& fv1, std::vector<int32_t>& vpMatches12) {
    int n = 0;
    vpMatches12.assign(n1, -1);
    check(orbm_search_by_bow_kf(pKF2.handle(), &fv2, mp_valid2, desc1, n1, mp_valid1, angle1, &fv1, mfNNratio, mbCheckOrientation,
                                vpMatches12.data(), &n), "orbm_search_by_bow_kf");
    return n;
  }
  // int SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches), :269-471.
  int SearchByBoW(FrameOnDevice& F, const orbm_featvec_view& fvF, const uint8_t* kf_desc, int nkf, const uint8_t* kf_mp_valid,
                  const float* kf_angle, const orbm_featvec_view& fvKF, std::vector<int32_t>& vpMapPointMatches) {
    int n = 0;
    vpMapPointMatches.assign(F.N(), -1);
    check(orbm_search_by_bow(F.handle(), &fvF, kf_desc, nkf, kf_mp_valid, kf_angle, &fvKF, mfNNratio, mbCheckOrientation,
                             vpMapPointMatches.data(), &n), "orbm_search_by_bow");
    return n;
  }

 protected:
  float mfNNratio;
  bool mbCheckOrientation;
};

// ------------------------------------------------------------------------------------------------ optimizer
class Optimizer {
 public:
  // void static LocalBundleAdjustment(KeyFrame* pKF, bool *pbStopFlag, Map *pMap, int& num_fixedKF, int LocalBASize),
  // I/Optimizer.h:42 -- numerical core over the flattened problem of SURVEY.md Appendix E-6.  Returns the status
  // (LBA_APPLIED / LBA_ABORTED_BEFORE_OPT / LBA_REJECTED_OUTLIERS); the caller writes the result back under
  // Map::mMutexMapUpdate only for LBA_APPLIED, exactly where the reference does (S/Optimizer.cc:2257-2270).
  static int LocalBundleAdjustment(const lba_problem& problem, const volatile int32_t* pbStopFlag, lba_result& result) {
    check(lba_solve(&problem, pbStopFlag, &result), "lba_solve");
    return result.status;
  }
  // ... with the reference's own flag: LocalMapping hands &mbAbortBA, a one-byte bool that InterruptBA() sets while the solve
  // runs (S/LocalMapping.cc:245,381-386); the library polls that byte where g2o polls it, nothing on the reference side changes
  static int LocalBundleAdjustment(const lba_problem& problem, const volatile bool* pbStopFlag, lba_result& result) {
    check(lba_solve_b(&problem, stop_byte(pbStopFlag), &result), "lba_solve_b");
    return result.status;
  }
  static const volatile uint8_t* stop_byte(const volatile bool* pbStopFlag) {
    static_assert(sizeof(bool) == 1, "bool* pbStopFlag is polled as one byte");
    return reinterpret_cast<const volatile uint8_t*>(pbStopFlag);
  }
  // int static PoseOptimization(Frame* pFrame), I/Optimizer.h:47, S/Optimizer.cc:964-1278: returns nInitialCorrespondences -
  // nBad; result.Tcw is what the reference writes with pFrame->SetPose, result.outlier[i] is pFrame->mvbOutlier.
  static int PoseOptimization(const pose_opt_problem& problem, pose_opt_result& result) {
    check(pose_optimize(&problem, &result), "pose_optimize");
    return result.n_inliers;
  }
};

// LocalMapping-side handle: keeps the device buffers across keyframes and can run the solve on its own worker thread
// next to Tracking (S/ClientSystem.cc:105-106), as the reference's LocalMapping thread does.
class LocalBA {
 public:
  explicit LocalBA(int device = 0) { check(lba_create(device, 0, 0, 0, &h_), "lba_create"); }
  ~LocalBA() { if (h_) lba_destroy(h_); }
  LocalBA(const LocalBA&) = delete;
  LocalBA& operator=(const LocalBA&) = delete;
  int Run(const lba_problem& problem, const volatile int32_t* pbStopFlag, lba_result& result) {
    check(lba_solve_h(h_, &problem, pbStopFlag, &result), "lba_solve_h");
    return result.status;
  }
  void Submit(const lba_problem& problem, const volatile int32_t* pbStopFlag, lba_result& result) {
    check(lba_solve_async(h_, &problem, pbStopFlag, &result), "lba_solve_async");
  }
  // the reference's bool flag (&LocalMapping::mbAbortBA), polled in place
  int Run(const lba_problem& problem, const volatile bool* pbStopFlag, lba_result& result) {
    check(lba_solve_hb(h_, &problem, Optimizer::stop_byte(pbStopFlag), &result), "lba_solve_hb");
    return result.status;
  }
  void Submit(const lba_problem& problem, const volatile bool* pbStopFlag, lba_result& result) {
    check(lba_solve_async_b(h_, &problem, Optimizer::stop_byte(pbStopFlag), &result), "lba_solve_async_b");
  }
  void Wait(double* solve_ms = nullptr) { check(lba_wait(h_, solve_ms), "lba_wait"); }

 private:
  lba_handle* h_ = nullptr;
};

}  // namespace orbgpu

#ifdef ORBGPU_DROPIN_NAMESPACE
namespace ORB_SLAM3 {
using orbgpu::ORBextractor;
using orbgpu::ORBmatcher;
using orbgpu::Optimizer;
}  // namespace ORB_SLAM3
#endif

#endif  // ORBGPU_ADAPTERS_HPP_
