// Header-only stand-ins for the reference's Frame / KeyFrame / MapPoint / Map with exactly the members the hot path reads
// or writes, under the reference's own names, so that include/orbgpu_dropin.hpp -- the glue that
// would be pasted into S/ORBmatcher.cc / S/Optimizer.cc -- compiles and runs without OpenCV / g2o / ROS.
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace mock {

struct Mat {                       // the slice of cv::Mat the glue uses: continuous rows of float or bytes
  int rows = 0, cols = 0;
  std::vector<uint8_t> bytes;
  int elem = 4;
  Mat() {}
  Mat(int r, int c, int elem_size) : rows(r), cols(c), bytes((size_t)r * c * elem_size), elem(elem_size) {}
  bool empty() const { return bytes.empty(); }
  template <class T> const T* ptr(int row = 0) const { return reinterpret_cast<const T*>(bytes.data() + (size_t)row * cols * elem); }
  template <class T> T* ptr(int row = 0) { return reinterpret_cast<T*>(bytes.data() + (size_t)row * cols * elem); }
  float at(int r, int c) const { return ptr<float>(r)[c]; }
};
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave; };

class Map; class KeyFrame;

// Access sections mirror the reference's headers: what is `protected:` there is MOCK_PROTECTED here -- public for the tests (which set
// scenes up by writing members directly), protected under -DMOCK_STRICT_ACCESS, the build in which tests/cpp/glue_access_check.cpp
// instantiates every template of include/orbgpu_dropin.hpp: the glue then compiles only if it stays within what the reference's
// classes let an outsider touch, plus the edits INTEGRATION.md lists.  tests/test_reference_access.py holds the partition below
// against the reference's own headers (I/MapPoint.h, I/KeyFrame.h, I/Frame.h, I/Map.h) where they are present.
#ifdef MOCK_STRICT_ACCESS
#define MOCK_PROTECTED protected
#else
#define MOCK_PROTECTED public
#endif

class MapPoint {
 public:      // ---- public in the reference (I/MapPoint.h:118-242)
  long unsigned mnId = 0; uint8_t mnClientId = 0;
  long unsigned mnBALocalForKF = ~0ul;
  // fields Frame::isInFrustum stores (S/Frame.cc:529-538)
  bool mbTrackInView = false; float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0, mTrackDepth = 0, mTrackViewCos = 0; int mnTrackScaleLevel = 0;
  // ... and Frame::isInFrustumChecks for the right camera of a rig (S/Frame.cc:1212-1218; I/MapPoint.h:208-213)
  bool mbTrackInViewR = false; float mTrackProjYR = 0, mTrackDepthR = 0, mTrackViewCosR = 0; int mnTrackScaleLevelR = 0;
  int nObs = 0;
  long unsigned mnLastFrameSeen = ~0ul;
  void IncreaseVisible(int n = 1) { mnVisible += n; }                   // S/MapPoint.cc:427-431
  bool isBad() const { return mbBad; }
  Map* GetMap() const { return mpMap; }
  int Observations() const { return nObs; }
  Mat GetWorldPos() const { n_pos_clones++; return mWorldPos; }
  Mat GetNormal() const { return mNormalVector; }
  Mat GetDescriptor() const { return mDescriptor; }
  void SetWorldPos(const Mat& X, bool bLock = false, bool /*bLockSend*/ = false) { mWorldPos = X; n_locked_pos_writes += bLock; mnChangeStamp++; }   // I/MapPoint.h:126
  std::map<KeyFrame*, std::tuple<int, int>> GetObservations() const { n_obs_copies++; return mObservations; }
  // (virtual: tests/cpp/closed_loop.cpp derives points whose mutators follow S/MapPoint.cc statement by statement -- stereo
  //  observations count twice, a point that falls to two observations turns bad, the normal and the distance range are recomputed)
  virtual ~MapPoint() {}
  virtual void AddObservation(KeyFrame* kf, int idx) { if (!mObservations.count(kf)) nObs++; mObservations[kf] = std::make_tuple(idx, -1); mnChangeStamp++; }
  virtual void EraseObservation(KeyFrame* kf) { if (mObservations.erase(kf)) nObs--; mnChangeStamp++; }
  virtual void SetBadFlag() { mbBad = true; mObservations.clear(); nObs = 0; mnChangeStamp++; }
  virtual void UpdateNormalAndDepth() { n_normal_updates++; mnChangeStamp++; }
 public:      // ---- reference-side edits ("Reference-side edits": E1 two getters, E2 the change counter)
  float GetMinDistance() const { return mfMinDistance; }
  float GetMaxDistance() const { return mfMaxDistance; }
  std::atomic<unsigned long> mnChangeStamp{0};       // +1 per call of the six mutators of what the glue reads, after their stores
 MOCK_PROTECTED:   // ---- protected in the reference (I/MapPoint.h:244-300)
  float mfMinDistance = 0, mfMaxDistance = 0;
  Mat mWorldPos{3, 1, 4}, mNormalVector{3, 1, 4}, mDescriptor{1, 32, 1};
  std::map<KeyFrame*, std::tuple<int, int>> mObservations;
  bool mbBad = false; Map* mpMap = nullptr;
  int mnVisible = 1;
 public:      // ---- test instrumentation (no counterpart in the reference; the glue never names these)
  int n_normal_updates = 0;
  int n_locked_pos_writes = 0;
  mutable int n_obs_copies = 0, n_pos_clones = 0;
  void Touch() { mnChangeStamp++; }      // what ComputeDistinctiveDescriptors / UpdateNormalAndDepth do to the counter when a test writes their fields directly
};

class Map {
 public:      // ---- public in the reference (I/Map.h:56-162)
  std::mutex mMutexMapUpdate;
  void IncreaseChangeIndex() { mnMapChange++; }                       // I/Map.h:100-101
  int GetMapChangeIndex() const { return mnMapChange; }
  long unsigned GetInitKFid() const { return mnInitKFid; }
  bool IsInertial() const { return mbIsInertial; }
 MOCK_PROTECTED:   // ---- protected in the reference (I/Map.h:164-210)
  long unsigned mnInitKFid = 0; bool mbIsInertial = false;
  int mnMapChange = 0;
};

typedef std::map<unsigned, std::vector<unsigned>> FeatureVector;     // DBoW2::FeatureVector (D/FeatureVector.h:24-25)

// GeometricCamera (I/CameraModels/GeometricCamera.h:50-105): the three members the glue reads, all public there
class GeometricCamera {
 public:
  GeometricCamera(unsigned type, std::vector<float> params) : mnType(type), mvParameters(std::move(params)) {}
  float getParameter(const int i) { return mvParameters[i]; }
  size_t size() { return mvParameters.size(); }
  unsigned int GetType() { return mnType; }
 protected:
  unsigned int mnType;
  std::vector<float> mvParameters;
};

class KeyFrame {
 public:      // ---- public in the reference (I/KeyFrame.h:268-533)
  long unsigned mnId = 0; uint8_t mnClientId = 0;
  long unsigned mnBALocalForKF = ~0ul, mnBAFixedForKF = ~0ul;
  float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
  std::vector<KeyPoint> mvKeys, mvKeysUn; std::vector<float> mvuRight, mvInvLevelSigma2;
  // (I/KeyFrame.h:634-648) the cameras and, for the two-fisheye rig, the right camera's keypoints
  GeometricCamera* mpCamera = nullptr; GeometricCamera* mpCamera2 = nullptr;
  Mat mTrl{3, 4, 4};
  std::vector<KeyPoint> mvKeysRight;
  int NLeft = -1, NRight = -1;
  Mat mDescriptors; FeatureVector mFeatVec;
  bool isBad() const { return mbBad; }
  Map* GetMap() const { return mpMap; }
  std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() const { return mvpOrderedConnectedKeyFrames; }
  std::vector<MapPoint*> GetMapPointMatches() const { return mvpMapPoints; }
  Mat GetPose() const { return Tcw; }
  void SetPose(const Mat& T, bool bLock = false, bool /*bLockSend*/ = false) { Tcw = T; n_locked_pose_writes += bLock; }          // I/KeyFrame.h:276
  void EraseMapPointMatch(MapPoint* mp) { for (auto& p : mvpMapPoints) if (p == mp) p = nullptr; }
 MOCK_PROTECTED:   // ---- protected in the reference (I/KeyFrame.h:535-632)
  std::vector<MapPoint*> mvpMapPoints; std::vector<KeyFrame*> mvpOrderedConnectedKeyFrames;
  Mat Tcw{4, 4, 4};
  bool mbBad = false; Map* mpMap = nullptr;
 public:      // ---- test instrumentation
  int n_locked_pose_writes = 0;
};

class Frame {
 public:      // ---- reference-side edits (optional)
  void* mpGpuFrame = nullptr;           // the device-resident copy the constructor adapter left (an orbgpu::FrameOnDevice*), or none
 public:      // ---- public in the reference (I/Frame.h:49-252)
  long unsigned mnId = 0;
  int N = 0;
  std::vector<KeyPoint> mvKeys, mvKeysUn; Mat mDescriptors;
  std::vector<float> mvuRight, mvDepth, mvInvLevelSigma2;
  std::vector<MapPoint*> mvpMapPoints; std::vector<bool> mvbOutlier;
  FeatureVector mFeatVec;
  Mat mTcw{4, 4, 4};
  float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0, fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0, mb = 0;   // statics in the reference
  int mnScaleLevels = 8; float mfScaleFactor = 1.2f;
  void SetPose(const Mat& T) { mTcw = T; }
  // (I/Frame.h:169, 276-297) the cameras and, for the two-fisheye rig, the right camera's keypoints (features i >= Nleft)
  GeometricCamera* mpCamera = nullptr; GeometricCamera* mpCamera2 = nullptr;
  Mat mTrl{3, 4, 4};
  std::vector<KeyPoint> mvKeysRight;
  int Nleft = -1, Nright = -1;
  Mat mTlr{3, 4, 4};
  std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch;      // (I/Frame.h:285) stereo partners of the two cameras' features, -1 = none
  // (I/Frame.h:183,188,232,282,292) what Frame::ComputeStereoFishEyeMatches reads and leaves
  Mat mDescriptorsRight; int monoLeft = 0, monoRight = 0; std::vector<float> mvLevelSigma2; std::vector<Mat> mvStereo3Dpoints; int mnCloseMPs = 0;
};

// the matrix access points include/orbgpu_dropin.hpp asks for
inline const float* mat_f32(const Mat& m) { return m.ptr<float>(0); }
inline const uint8_t* mat_u8(const Mat& m, int row) { return m.ptr<uint8_t>(row); }
inline void make_mat(Mat& out, int rows, int cols, const float* data) { out = Mat(rows, cols, 4); std::memcpy(out.ptr<float>(0), data, sizeof(float) * rows * cols); }

}  // namespace mock

namespace orbgpu { namespace dropin { using mock::mat_f32; using mock::mat_u8; using mock::make_mat; } }
