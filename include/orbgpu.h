/*
 * orbgpu.h -- C-ABI of the MI355X-native ORB-SLAM3 hot path (liborbgpu.so).
 *
 * This is the drop-in boundary for the one hot path of yutongwangBIT/multi_orbslam3 that this
 * repository re-implements for gfx950: ORBextractor, stereo matching, the Hamming matchers and
 * Optimizer::LocalBundleAdjustment.  The reference has no FFI of its own (its interface is three
 * C++ headers full of cv::Mat / STL / pointer graphs), so every entry point below cites the C++
 * member it replaces; INTEGRATION.md shows the adapter a maintainer adds on the reference side.
 *
 * Conventions: plain pointers + sizes, no C++/torch types, every function returns an int status
 * (ORBG_OK == 0, negative == error, never throws).  Unless a parameter is named d_*, pointers are
 * HOST memory owned by the caller; handles own all device memory, pinned staging and HIP streams.
 * Handles are thread-compatible (one thread at a time per handle), the library is re-entrant
 * across handles.  File:line citations use the path shorthand of SURVEY.md
 * (S/ = src/orb_slam3_ros/orb_slam3/src/, I/ = .../include/, G/ = .../Thirdparty/g2o/g2o/).
 */
#ifndef ORBGPU_H_
#define ORBGPU_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- status codes */
enum {
  ORBG_OK = 0,
  ORBG_EMPTY = -1,         /* empty image: ORBextractor::operator() returns -1, S/ORBextractor.cc:1072-1073 */
  ORBG_BAD_ARG = -2,
  ORBG_CAP_EXCEEDED = -3,  /* caller-provided capacity (or a handle capacity) too small */
  ORBG_HIP_ERROR = -4,
  ORBG_NO_DEVICE = -5,     /* no HIP device / HIP runtime unusable: the product path never falls back to CPU */
  ORBG_INTERNAL = -6
};

/* LBA outcome, lba_result.status (not errors: they mirror the reference's early returns) */
enum {
  LBA_APPLIED = 0,
  LBA_ABORTED_BEFORE_OPT = 1,   /* *pbStopFlag set before optimize(): S/Optimizer.cc:2127-2129 */
  LBA_REJECTED_OUTLIERS = 2     /* >= 50 % of the edges are outliers: S/Optimizer.cc:2257-2261 */
};

#define ORBG_MAX_LEVELS 16
#define ORBG_DESC_BYTES 32
#define ORBG_GRID_COLS 64      /* FRAME_GRID_COLS, I/Frame.h:39 */
#define ORBG_GRID_ROWS 48      /* FRAME_GRID_ROWS, I/Frame.h:38 */
/* Feature indices travel in 16 bits inside the matchers and the stereo matcher (0xFFFF = none): frames / keypoint sets of
 * this many features or more are refused with ORBG_CAP_EXCEEDED (the reference runs with 1000-2000 features per image). */
#define ORBG_MAX_FRAME_FEATURES 65535

/* ---------------------------------------------------------------- ORB extractor */

/* ORBextractor ctor arguments (S/ORBextractor.cc:408-411, values from ORBParameters I/Datatypes.h:43-55)
 * plus the sizes the handle pre-allocates for. */
typedef struct orbx_config {
  int32_t n_features;     /* nfeatures, 1 .. 3500 (the reference's settings use 1000 - 2000); more: ORBG_BAD_ARG */
  float   scale_factor;   /* scaleFactor */
  int32_t n_levels;       /* nlevels (<= ORBG_MAX_LEVELS) */
  int32_t ini_th_fast;    /* iniThFAST   */
  int32_t min_th_fast;    /* minThFAST   */
  int32_t max_width;      /* largest image the handle will be given (<= 4000 x 4000); every pyramid level of an image
                           * must stay larger than the 19-pixel border (else ORBG_BAD_ARG from the extract call) */
  int32_t max_height;
  int32_t n_cams;         /* 1 = mono, 2 = stereo rig (left = cam 0, right = cam 1) */
  int32_t device;         /* HIP device ordinal (1 agent <-> 1 GPU) */
  /* Deployment variants for the two places where the reference's output depends on its build, not on its source
   * (all zero = the defaults the parity tests pin):
   * gauss_taps: outer-to-centre half of the 7-tap Q8 kernel cv::GaussianBlur(7x7, sigma 2) uses on CV_8U
   *   {18,34,49,55} (sum 257, default): OpenCV <= 3.4.1, and the 3.4.2 - 4.4 fixed-point path (each tap rounded on its own);
   *   {18,34,48,56} (sum 256): OpenCV >= 4.5 (getGaussianKernelFixedPoint_ED: rounding error diffused, centre = remainder).
   *   Any taps with 2 (t0+t1+t2) + t3 <= 257 are accepted.
   * octree_oldest_first: DistributeOctTree sorts (size, ExtractorNode*) and splits from the back (S/ORBextractor.cc:679-682):
   *   among equally populated nodes the heap address decides.  0 (default): the most recently created node first
   *   (addresses grow with allocation order); 1: the oldest first (allocators that hand out descending addresses). */
  int32_t gauss_taps[4];
  int32_t octree_oldest_first;
} orbx_config;

/* The cv::KeyPoint fields the reference sets/uses. 24 bytes. */
typedef struct orbx_keypoint {
  float   x, y;       /* pt, level-0 pixels (S/ORBextractor.cc:1131-1133) */
  float   size;       /* (int)(31*mvScaleFactor[l]) (S/ORBextractor.cc:862,871) */
  float   angle;      /* degrees [0,360) (S/ORBextractor.cc:475) */
  float   response;   /* FAST score */
  int32_t octave;     /* pyramid level (S/ORBextractor.cc:870) */
} orbx_keypoint;

typedef struct orbx_handle orbx_handle;

/* ORBextractor::ORBextractor, S/ORBextractor.cc:408-468. */
int orbx_create(const orbx_config* cfg, orbx_handle** out);
int orbx_destroy(orbx_handle* h);

/* Scale tables the reference exposes through getters (I/ORBextractor.h:65-85):
 * mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2 (each n_levels floats) and
 * mnFeaturesPerLevel.  Any output pointer may be NULL. */
int orbx_get_tables(const orbx_handle* h, float* scale, float* inv_scale, float* sigma2,
                    float* inv_sigma2, int32_t* features_per_level);

/* int ORBextractor::operator()(image, mask, keypoints, descriptors, vLappingArea),
 * S/ORBextractor.cc:1068-1150.  img: u8 gray, row stride `stride` bytes.  Keypoints with
 * lap0 <= x <= lap1 are written from the back (S/ORBextractor.cc:1135-1144); *n_mono is the
 * reference's return value (monoIndex).  kps/desc hold `cap` entries; *n is the total count.
 * Returns ORBG_EMPTY for a NULL / zero-sized image. */
int orbx_extract(orbx_handle* h, int cam, const uint8_t* img, int width, int height, int stride,
                 int lap0, int lap1, orbx_keypoint* kps, uint8_t* desc, int cap, int* n, int* n_mono);

/* The stereo Frame ctor's two concurrent ExtractORB calls (S/Frame.cc:92-95,393-400) as ONE
 * batched submission: left+right share every kernel launch.  vLappingArea = {0,0} as there.
 * Output pointers may be NULL (results then stay device-resident for orbx_stereo_match /
 * orbm_frame_from_extractor). */
int orbx_extract_stereo(orbx_handle* h, const uint8_t* img_left, const uint8_t* img_right,
                        int width, int height, int stride,
                        orbx_keypoint* kps_left, uint8_t* desc_left, int cap_left, int* n_left,
                        orbx_keypoint* kps_right, uint8_t* desc_right, int cap_right, int* n_right);

/* Same, images already resident in device memory (d_img_*: device pointers, stride in bytes). */
int orbx_extract_stereo_dev(orbx_handle* h, const uint8_t* d_img_left, const uint8_t* d_img_right,
                            int width, int height, int stride,
                            orbx_keypoint* kps_left, uint8_t* desc_left, int cap_left, int* n_left,
                            orbx_keypoint* kps_right, uint8_t* desc_right, int cap_right, int* n_right);

/* mvImagePyramid[level] (public member read by Frame::ComputeStereoMatches, I/ORBextractor.h:87):
 * dimensions, and a copy of the level (without its 19-px border) into host memory (may be NULL). */
int orbx_get_level(orbx_handle* h, int cam, int level, uint8_t* host_out, int* width, int* height);
/* Same level together with the 19-px REFLECT_101 border copyMakeBorder puts around it (S/ORBextractor.cc:1167-1173):
 * (width+38) x (height+38) bytes, tightly packed.  *width / *height still report the level size without border. */
int orbx_get_level_bordered(orbx_handle* h, int cam, int level, uint8_t* host_out, int* width, int* height);

/* Test/diagnostic view of ComputeKeyPointsOctTree's vToDistributeKeys (S/ORBextractor.cc:776-853)
 * for the last extraction: per level, FAST candidates in the reference's cell-major order,
 * coordinates relative to (minBorderX,minBorderY).  xys = cap x {x,y,score} int32. */
int orbx_get_candidates(orbx_handle* h, int cam, int level, int32_t* xys, int cap, int* n);

/* Frame::ComputeStereoMatches, S/Frame.cc:785-963, on the device-resident result of the last
 * orbx_extract_stereo*: fills uright[n_left] / depth[n_left] (mvuRight / mvDepth, -1 = none).
 * bf = mbf, b = mb (S/Frame.cc:815-817).  Host outputs may be NULL. */
int orbx_stereo_match(orbx_handle* h, float bf, float b, float* uright, float* depth);

/* ---------------------------------------------------------------- frame view + matchers */

/* What the matchers read from a Frame. */
typedef struct orbm_frame_view {
  int32_t n;                      /* Frame::N */
  const orbx_keypoint* kps;       /* mvKeysUn (== mvKeys when k1 == 0, S/Frame.cc:723-727) */
  const uint8_t* desc;            /* mDescriptors, n x 32 */
  const float* uright;            /* mvuRight, NULL = all -1 */
  const float* depth;             /* mvDepth,  NULL = all -1 */
  float min_x, max_x, min_y, max_y;   /* mnMinX.. (S/Frame.cc:127-144) */
  float fx, fy, cx, cy, bf, b;    /* S/Frame.cc:136-146 */
  int32_t n_levels;
  float   scale_factor;           /* mfScaleFactor; mvScaleFactors rebuilt as S/ORBextractor.cc:413-421 */
} orbm_frame_view;

typedef struct orbm_frame orbm_frame;

int orbm_frame_create(int device, int cap_features, orbm_frame** out);
int orbm_frame_destroy(orbm_frame* f);
/* Upload a host frame view and build the 64x48 feature grid on the device
 * (Frame::AssignFeaturesToGrid / PosInGrid, S/Frame.cc:360-391,699-709). */
int orbm_frame_upload(orbm_frame* f, const orbm_frame_view* view);
/* Same, but features/uright/depth are taken device-to-device from the extractor handle's left
 * camera (no host round trip); view->kps/desc/uright/depth are ignored, view->n must be the
 * left feature count or -1. */
int orbm_frame_from_extractor(orbm_frame* f, orbx_handle* h, const orbm_frame_view* view);
/* Frame::Frame(imLeft, imRight, ...) (S/Frame.cc:71-172) as ONE submission with ONE final synchronisation: both
 * ExtractORB calls (:92-95), ComputeStereoMatches (:117) and AssignFeaturesToGrid (:160).  Images are host u8 gray;
 * `frame` (may be NULL) afterwards views the left features on the device exactly as after
 * orbm_frame_from_extractor.  Host outputs (kps_left, desc_left, uright, depth: cap_left entries) may be NULL. */
int orbx_frame_stereo(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const uint8_t* img_left,
                      const uint8_t* img_right, int width, int height, int stride, float bf, float b,
                      orbx_keypoint* kps_left, uint8_t* desc_left, float* uright, float* depth, int cap_left,
                      int* n_left, int* n_right);
/* Same with the two images already in device memory. */
int orbx_frame_stereo_dev(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const uint8_t* d_img_left,
                          const uint8_t* d_img_right, int width, int height, int stride, float bf, float b,
                          orbx_keypoint* kps_left, uint8_t* desc_left, float* uright, float* depth, int cap_left,
                          int* n_left, int* n_right);
/* The same constructor in two halves: _submit enqueues the whole chain (pyramids .. grid) on the handle's stream and
 * returns, _wait completes it (redoing the frame with the host quad-trees if a device list overflowed) and returns the
 * feature counts.  Between the two calls the handle, `frame` and the images must be left alone; work on OTHER handles /
 * frames (the searches of the previous frame, S/Tracking.cc:2629-2735) may run meanwhile and overlaps on the GPU.
 * One submission per handle at a time; no host copies of the features in this form. */
int orbx_frame_stereo_dev_submit(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const uint8_t* d_img_left,
                                 const uint8_t* d_img_right, int width, int height, int stride, float bf, float b);
int orbx_frame_stereo_dev_wait(orbx_handle* h, int* n_left, int* n_right);
/* The two-halves constructor with HOST images (cv::Mat data, any host memory): Frame::Frame(imLeft, imRight, ...)
 * (S/Frame.cc:71-172) as Tracking::GrabImageStereo (S/Tracking.cc:1014-1083) receives them.  _submit packs the rows into the
 * handle's pinned staging slot, enqueues one copy kernel (host -> HBM over PCIe, ordered before the pyramid on the handle's
 * stream, overlapping the kernels of other handles) and the constructor chain; _wait is orbx_frame_stereo_dev_wait.  The
 * images may be reused as soon as _submit returns with flags == 0.  With ORBX_SUBMIT_ASYNC the packing and the launches are
 * done by the library's ingest thread (one per process, created by the first such call; it inherits that caller's CPU
 * affinity): the call returns at once and the images must stay valid until _wait.  Typical use: the image grabber / camera
 * thread submits frame t+1 (flags 0) while Tracking works on frame t, or Tracking itself submits with ORBX_SUBMIT_ASYNC.
 * One submission per handle at a time (ORBG_BAD_ARG otherwise). */
#define ORBX_SUBMIT_ASYNC 1
/* Host arrays for the features of the two-halves constructor: from now on every _submit of this handle also delivers the LEFT image's
 * mvKeys / mDescriptors / mvuRight / mvDepth -- what the output arguments of orbx_frame_stereo deliver for the synchronous
 * constructor, S/Frame.cc:100-118 -- into these arrays (cap_left entries each; a NULL array is not delivered) by the time _wait
 * returns; ORBG_CAP_EXCEEDED from _wait when the frame has more features.  The arrays belong to the caller and must stay valid while
 * a submission is in flight; all NULL / cap 0 switches the delivery off again.  The handle must be idle. */
int orbx_set_frame_outputs(orbx_handle* h, orbx_keypoint* kps_left, uint8_t* desc_left, float* uright, float* depth, int cap_left);
int orbx_frame_stereo_submit(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const uint8_t* img_left,
                             const uint8_t* img_right, int width, int height, int stride, float bf, float b, int flags);
int orbx_frame_stereo_wait(orbx_handle* h, int* n_left, int* n_right);
/* ---- the monocular Frame constructor (mono agents of BASELINE configs 1 and 5)
 * Frame::Frame(imGray, ..., pCamera, distCoef, ...), S/Frame.cc:260-358: ExtractORB(0, imGray, 0, 1000) (:289 -- the lapping area
 * {0, 1000} covers every image up to 1000 px wide, so ALL keypoints are written from the back: reversed order, monoIndex = 0,
 * S/ORBextractor.cc:1135-1144), UndistortKeyPoints (:301, :721-754), mvuRight = mvDepth = -1 (:303-304), AssignFeaturesToGrid (:344)
 * as ONE submission with ONE final synchronisation, exactly like the stereo constructor above; `frame` views mvKeysUn + mDescriptors
 * with its grid on the device afterwards.
 * dist = mDistCoef {k1, k2, p1, p2, k3} (S/Tracking.cc:71-81); NULL or k1 == 0: mvKeysUn = mvKeys (S/Frame.cc:723-727) and the view's
 * bounds must be the image rectangle.  Otherwise every keypoint goes through cv::undistortPoints(mat, mat, K, mDistCoef, Mat(), mK)
 * (S/Frame.cc:740; K = mK = the view's fx, fy, cx, cy; OpenCV 3.2 cvUndistortPoints: 5 fixed-point iterations in double) on the
 * device, and the view's bounds are the caller's ComputeImageBounds (S/Frame.cc:756-783: orbx_undistort_points of the four corners).
 * Host outputs (cap entries each, any may be NULL): kps = mvKeys, kps_un = mvKeysUn, desc = mDescriptors. */
typedef struct orbx_distortion { float k1, k2, p1, p2, k3; } orbx_distortion;
int orbx_frame_mono(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const orbx_distortion* dist, const uint8_t* img,
                    int width, int height, int stride, orbx_keypoint* kps, orbx_keypoint* kps_un, uint8_t* desc, int cap, int* n);
int orbx_frame_mono_dev(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const orbx_distortion* dist, const uint8_t* d_img,
                        int width, int height, int stride, orbx_keypoint* kps, orbx_keypoint* kps_un, uint8_t* desc, int cap, int* n);
/* The two-halves form (see orbx_frame_stereo_submit): host image / image resident in HBM; _wait collects it (orbx_frame_stereo_dev_wait
 * works too and reports n_right = 0).  orbx_set_frame_outputs' kps_left / desc_left arrays receive mvKeys / mDescriptors,
 * orbx_set_frame_outputs_un's array receives mvKeysUn. */
int orbx_frame_mono_submit(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const orbx_distortion* dist, const uint8_t* img,
                           int width, int height, int stride, int flags);
int orbx_frame_mono_dev_submit(orbx_handle* h, orbm_frame* frame, const orbm_frame_view* view, const orbx_distortion* dist,
                               const uint8_t* d_img, int width, int height, int stride);
int orbx_frame_mono_wait(orbx_handle* h, int* n);
int orbx_set_frame_outputs_un(orbx_handle* h, orbx_keypoint* kps_un /* cap_left of orbx_set_frame_outputs entries, or NULL */);
/* cv::undistortPoints(pts, pts, K, mDistCoef, Mat(), K) for n points (xy_in / xy_out: n x {x, y} float32, host memory; in place
 * allowed), on the device: what Frame::ComputeImageBounds (S/Frame.cc:756-783) runs on the four image corners. */
int orbx_undistort_points(int device, const float* xy_in, int n, float fx, float fy, float cx, float cy, const orbx_distortion* dist,
                          float* xy_out);
/* Grid as CSR for tests: cell id = ix*48+iy, items in keypoint-index order (Appendix E-2). */
int orbm_frame_get_grid(orbm_frame* f, int32_t* cell_start /*64*48+1*/, int32_t* cell_items /*n*/);

/* int ORBmatcher::DescriptorDistance(a,b), S/ORBmatcher.cc:2358-2374, as a dense nq x nt matrix
 * (row-major int32) -- the raw Hamming kernel. */
int orbm_hamming_matrix(int device, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* dist);
/* Best / second-best over all t for every q: out4 = nq x {best_dist,best_idx,second_dist,second_idx}. */
int orbm_hamming_best2(int device, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* out4);

/* MapPoint fields read by SearchByProjection(Frame&, vector<MapPoint*>&...) after isInFrustum
 * filled them (SURVEY.md Appendix E-3, S/Frame.cc:529-538).  SoA, m entries. */
typedef struct orbm_mappoints_view {
  int32_t m;
  const uint8_t* track_in_view;   /* mbTrackInView */
  const uint8_t* bad;             /* isBad() */
  const float* proj_x;            /* mTrackProjX */
  const float* proj_y;            /* mTrackProjY */
  const float* proj_xr;           /* mTrackProjXR */
  const float* track_depth;       /* mTrackDepth */
  const int32_t* scale_level;     /* mnTrackScaleLevel */
  const float* view_cos;          /* mTrackViewCos */
  const uint8_t* desc;            /* GetDescriptor(), m x 32 */
  const int32_t* n_obs;           /* Observations() */
} orbm_mappoints_view;

/* World-space map points for the fused isInFrustum + search path (Appendix E-3, second half). */
typedef struct orbm_worldpoints_view {
  int32_t m;
  const float* pos;        /* m x 3  GetWorldPos() */
  const float* normal;     /* m x 3  GetNormal() */
  const float* min_dist;   /* m      mfMinDistance (raw; x0.8 applied as S/MapPoint.cc:617-621) */
  const float* max_dist;   /* m      mfMaxDistance (raw; x1.2 applied as S/MapPoint.cc:623-627) */
  const uint8_t* desc;     /* m x 32 */
  const int32_t* n_obs;    /* m */
  const uint8_t* bad;      /* m */
  const uint8_t* skip;     /* m, 1 = not a candidate (already matched / mnLastFrameSeen, S/Tracking.cc:3088-3109); may be NULL */
} orbm_worldpoints_view;

/* bool Frame::isInFrustum(MapPoint*, 0.5) for m points, S/Frame.cc:466-543 (Nleft == -1 branch):
 * fills the track fields.  Outputs are host arrays of m entries. */
int orbm_is_in_frustum(orbm_frame* f, const float* Tcw /*16, row-major*/, const orbm_worldpoints_view* pts,
                       float viewing_cos_limit, uint8_t* track_in_view, float* proj_x, float* proj_y,
                       float* proj_xr, float* track_depth, int32_t* scale_level, float* view_cos);

/* int ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*>&, th, bFarPoints, thFarPoints),
 * S/ORBmatcher.cc:44-214 (Nleft == -1).  assigned_mp[n] (in/out) is F.mvpMapPoints flattened:
 * -1 = NULL, otherwise an index; indices written by this call are positions in `mps`.
 * assigned_obs[n] (in/out) is Observations() of that map point.  *nmatches = return value. */
int orbm_search_by_projection_mps(orbm_frame* f, const orbm_mappoints_view* mps, float th,
                                  int far_points, float th_far_points, float nnratio,
                                  int32_t* assigned_mp, int32_t* assigned_obs, int* nmatches);

/* Fused Tracking::SearchLocalPoints body (S/Tracking.cc:3111-3153): isInFrustum(.,0.5) for every
 * non-skipped point, then the search above, map points staying resident on the device. */
typedef struct orbm_map orbm_map;
int orbm_map_create(int device, int cap_points, orbm_map** out);
int orbm_map_destroy(orbm_map* m);
int orbm_map_upload(orbm_map* m, const orbm_worldpoints_view* pts);
/* MapPoint::Observations() of the points of the last upload, refreshed without re-uploading the map (host side only: the serial commit
 * reads it, S/ORBmatcher.cc:89-91).  With it a local map stays resident across frames: its static fields (position, normal, distance
 * range, descriptor) are uploaded when Tracking::UpdateLocalPoints / the local BA's write-back changed them, the per-frame exclusions
 * (points the frame already holds, points that became bad) go through `skip`. */
int orbm_map_set_observations(orbm_map* m, const int32_t* n_obs /* m */);
int orbm_search_local_points(orbm_frame* f, orbm_map* m, const float* Tcw, const uint8_t* skip /*m or NULL*/,
                             float th, int far_points, float th_far_points, float nnratio,
                             int32_t* assigned_mp, int32_t* assigned_obs, int* nmatches);
/* Same, and in_frustum[m] (may be NULL) reports for which points isInFrustum() returned true -- the points the loop of
 * Tracking::SearchLocalPoints calls IncreaseVisible() for and counts in nToMatch (S/Tracking.cc:3118-3122). */
int orbm_search_local_points_vis(orbm_frame* f, orbm_map* m, const float* Tcw, const uint8_t* skip /*m or NULL*/,
                                 float th, int far_points, float th_far_points, float nnratio,
                                 int32_t* assigned_mp, int32_t* assigned_obs, int* nmatches, uint8_t* in_frustum /*m or NULL*/);

/* LastFrame fields read by SearchByProjection(Frame &Cur, const Frame &Last, th, bMono)
 * (SURVEY.md Appendix E-4). SoA, n entries (= LastFrame.N). */
typedef struct orbm_lastframe_view {
  int32_t n;
  const uint8_t* mp_valid;    /* LastFrame.mvpMapPoints[i] != NULL */
  const uint8_t* outlier;     /* LastFrame.mvbOutlier[i] */
  const float* world_pos;     /* n x 3, pMP->GetWorldPos() */
  const uint8_t* desc;        /* n x 32, pMP->GetDescriptor() */
  const int32_t* octave;      /* LastFrame.mvKeys[i].octave */
  const float* angle;         /* LastFrame.mvKeysUn[i].angle */
  const int32_t* n_obs;       /* pMP->Observations() */
  float Tcw[16];              /* LastFrame.mTcw, row-major */
} orbm_lastframe_view;

/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono),
 * S/ORBmatcher.cc:1970-2186 (Nleft == -1).  Tcw_cur = CurrentFrame.mTcw.  assigned_* as above
 * (indices are positions i in the last frame). */
int orbm_search_by_projection_frame(orbm_frame* cur, const float* Tcw_cur, const orbm_lastframe_view* last,
                                    float th, int mono, int check_orientation,
                                    int32_t* assigned_mp, int32_t* assigned_obs, int* nmatches);

/* The same search with the last frame's view RESIDENT on the device.  Tracking knows mLastFrame's map points when it has finished
 * tracking that frame (mLastFrame = Frame(mCurrentFrame), S/Tracking.cc:2086-2090) -- a frame time before the next
 * SearchByProjection(Current, Last) reads them: orbm_lastview_upload takes the view then (one packed copy on the library's M stream,
 * asynchronous; the view's arrays may be reused as soon as it returns), orbm_search_by_projection_frame_resident reads it from HBM
 * instead of from pinned host memory over PCIe at the moment the next constructor's images cross it.  Same results as
 * orbm_search_by_projection_frame on the same view. */
typedef struct orbm_lastview orbm_lastview;
int orbm_lastview_create(int device, int cap_features, orbm_lastview** out);
int orbm_lastview_destroy(orbm_lastview* v);
int orbm_lastview_upload(orbm_lastview* v, const orbm_lastframe_view* last);
int orbm_search_by_projection_frame_resident(orbm_frame* cur, const float* Tcw_cur, orbm_lastview* last, float th, int mono,
                                             int check_orientation, int32_t* assigned_mp, int32_t* assigned_obs, int* nmatches);

/* DBoW2::FeatureVector flattened: sorted node ids, CSR feature lists. */
typedef struct orbm_featvec_view {
  int32_t n_nodes;
  const uint32_t* node_id;    /* ascending */
  const uint32_t* start;      /* n_nodes+1 */
  const uint32_t* feat_idx;   /* start[n_nodes] entries */
} orbm_featvec_view;

/* int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame &F, vector<MapPoint*>& vpMapPointMatches),
 * S/ORBmatcher.cc:269-471 (Nleft == -1).  kf_desc: nkf x 32; kf_mp_valid[i] = pMP && !isBad();
 * kf_angle = pKF->mvKeysUn[i].angle.  matches[f->n] out: KF feature index or -1. */
int orbm_search_by_bow(orbm_frame* f, const orbm_featvec_view* fv_frame,
                       const uint8_t* kf_desc, int nkf, const uint8_t* kf_mp_valid, const float* kf_angle,
                       const orbm_featvec_view* fv_kf, float nnratio, int check_orientation,
                       int32_t* matches, int* nmatches);

/* Server-side KeyFrame matchers.  A KeyFrame carries the same undistorted keypoints, descriptors
 * and 64x48 grid as the Frame it was made from (S/KeyFrame.cc:889-940 copies mGrid), so it is an orbm_frame here.
 *
 * int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints,
 *                                    vector<MapPoint*>& vpMatched, int th, float ratioHamming), S/ORBmatcher.cc:473-587
 *   -> camera_project = 1 (projects with pKF->mpCamera->project)
 * and the overload with vpPointsKFs / vpMatchedKF, :589-700 -> camera_project = 0 (float invz projection); the adapter
 * fills vpMatchedKF[idx] = vpPointsKFs[matched[idx]].
 * pts: candidate points resident on the device (pts.skip/bad exclude); already_found[m] (may be NULL) = point is already
 * in vpMatched on entry.  matched[n] in/out: -1 = NULL; entries written are indices into pts. */
int orbm_search_by_projection_sim3(orbm_frame* kf, orbm_map* pts, const float* Scw /*16, row-major Sim3*/,
                                   const uint8_t* already_found, int th, float ratio_hamming, int camera_project,
                                   int32_t* matched, int* nmatches);

/* int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12), S/ORBmatcher.cc:819-959.
 * kf2 = pKF2 with its feature vector fv2 and mp_valid2[i] = (MapPoint present and not bad); the pKF1 side is flattened
 * as for orbm_search_by_bow.  matches12[n1] out: feature index in pKF2 (-> vpMapPoints2[idx]) or -1.
 * Two-camera keyframes (NLeft != -1): the reference leaves out every feature with index >= mvKeysUn.size() on either side
 * (:854-856, :874-876) -- pass mp_valid1 / mp_valid2 = 0 for those. */
/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th,
 * const int ORBdist) -- the relocalisation overload, I/ORBmatcher.h:54, S/ORBmatcher.cc:2188-2310 (Tracking::Relocalization,
 * S/Tracking.cc:3372-3410: th = 10 / ORBdist = 100, then th = 3 / ORBdist = 64).  kf_points: pKF->GetMapPointMatches() uploaded
 * feature by feature with orbm_map_upload (m = pKF->N; bad[i] = 1 where the keyframe has no point or the point isBad());
 * already_found[i] = sAlreadyFound.count(pMP) (NULL: none); kf_angle[i] = pKF->mvKeysUn[i].angle (read when check_orientation).
 * assigned_mp (f's N entries) in: >= 0 where CurrentFrame.mvpMapPoints[idx] != NULL -- here ANY map point blocks a feature
 * (:2246-2247); out: newly matched features hold the index i of the keyframe feature whose point they took. */
int orbm_search_by_projection_reloc(orbm_frame* cur, orbm_map* kf_points, const float* Tcw_cur /*16*/, const uint8_t* already_found /*m or NULL*/,
                                    const float* kf_angle /*m*/, float th, int orb_dist, int check_orientation,
                                    int32_t* assigned_mp, int* nmatches);
struct orbg_camera;
/* orbm_search_by_projection_sim3 with camera_project = 1 and pKF->mpCamera a camera model (a fisheye keyframe): :515 projects through it. */
int orbm_search_by_projection_sim3_cam(orbm_frame* kf, orbm_map* pts, const float* Scw /*16, row-major Sim3*/, const struct orbg_camera* cam,
                                       const uint8_t* already_found, int th, float ratio_hamming, int32_t* matched, int* nmatches);
/* ... the same with CurrentFrame.mpCamera a camera model (a monocular fisheye frame): :2217 projects through it. */
int orbm_search_by_projection_reloc_cam(orbm_frame* cur, orbm_map* kf_points, const float* Tcw_cur /*16*/, const struct orbg_camera* cam,
                                        const uint8_t* already_found /*m or NULL*/, const float* kf_angle /*m*/, float th, int orb_dist,
                                        int check_orientation, int32_t* assigned_mp, int* nmatches);
int orbm_search_by_bow_kf(orbm_frame* kf2, const orbm_featvec_view* fv2, const uint8_t* mp_valid2,
                          const uint8_t* desc1, int n1, const uint8_t* mp_valid1, const float* angle1,
                          const orbm_featvec_view* fv1, float nnratio, int check_orientation,
                          int32_t* matches12, int* nmatches);

/* ---------------------------------------------------------------- bag of words */

/* DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> flattened (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:60-130,
 * m_nodes): node 0 is the root; the children of node i are child_ids[child_start[i] .. child_start[i+1]) in the order of
 * m_nodes[i].children; a node without children is a leaf (= word). */
enum { ORBV_TF_IDF = 0, ORBV_TF = 1, ORBV_IDF = 2, ORBV_BINARY = 3 };           /* WeightingType, BowVector.h:24-30 */
enum { ORBV_NORM_NONE = 0, ORBV_NORM_L1 = 1, ORBV_NORM_L2 = 2 };                /* what mustNormalize() reports (ORBvoc: L1) */
typedef struct orbv_vocab_view {
  int32_t n_nodes;
  int32_t L;                    /* m_L */
  int32_t weighting;            /* ORBV_TF_IDF for ORBvoc.txt */
  int32_t scoring_norm;         /* ORBV_NORM_L1 for ORBvoc.txt (L1_NORM scoring) */
  const int32_t* child_start;   /* n_nodes + 1 */
  const int32_t* child_ids;     /* child_start[n_nodes] node ids */
  const uint8_t* desc;          /* n_nodes x 32, m_nodes[i].descriptor (the root's is unused) */
  const double* weight;         /* n_nodes, m_nodes[i].weight */
  const int32_t* word_id;       /* n_nodes, m_nodes[i].word_id (meaningful for leaves) */
} orbv_vocab_view;
typedef struct orbv_vocab orbv_vocab;
int orbv_vocab_create(int device, const orbv_vocab_view* view, orbv_vocab** out);
int orbv_vocab_destroy(orbv_vocab* v);
/* The vocabulary in the reference's own file format: bool TemplatedVocabulary::loadFromTextFile(const std::string&),
 * Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1427 -- the ORBvoc.txt that System / ClientSystem hand to it -- parsed on the
 * host into an orbv_vocab_view (line 1: `k L scoring weighting`; line i: `parent isLeaf d0..d31 weight`, node ids in file order from 1,
 * children in file order, word ids to the isLeaf > 0 nodes in file order).  m_nodes is protected in the reference
 * (TemplatedVocabulary.h:405-423): this is the way in that needs no edit there.
 * flags: ORBV_TEXT_KEEP_TRAILING_NODE reproduces the extra node the reference's `while(!f.eof()) getline` loop makes of the empty
 * line after the last newline (a child of the root, no children, weight 0; its descriptor is indeterminate in the reference and zero
 * here); default: the tree the file describes.  A malformed file (header out of range, a node line with fewer than 35 fields, a
 * parent that is not an earlier node, node lines after a blank line): ORBG_BAD_ARG.
 * orbv_text_* is the host-only half (no GPU needed); orbv_vocab_from_text = load + orbv_vocab_create + free.
 * scoring (ScoringType, BowVector.h:48-56; ORBvoc.txt: 0 = L1_NORM) selects the BowVector normalisation as mustNormalize() reports it;
 * orbv_score_l1 / orbd_detect_n_best_candidates implement the L1 score only. */
enum { ORBV_TEXT_KEEP_TRAILING_NODE = 1 };
typedef struct orbv_text orbv_text;
int orbv_text_load(const char* path, int flags, orbv_text** out);
/* the parsed tree as a view (pointers valid until orbv_text_free); k / scoring / n_words may be NULL */
int orbv_text_view(const orbv_text* t, orbv_vocab_view* view, int32_t* k, int32_t* scoring, int32_t* n_words);
int orbv_text_free(orbv_text* t);
int orbv_vocab_from_text(int device, const char* path, int flags, orbv_vocab** out);
/* void transform(const TDescriptor& feature, WordId&, WordValue&, NodeId* nid, int levelsup), TemplatedVocabulary.h:1214-1260,
 * for n descriptors (host memory, n x 32): word_id[n], node_id[n] (node at level L - levelsup; 0 = root when that level is
 * <= 0 or -- pinned, the reference leaves it uninitialised -- when a leaf is reached above it), weight[n]. */
int orbv_transform(orbv_vocab* v, const uint8_t* desc, int n, int levelsup, int32_t* word_id, int32_t* node_id, double* weight);
/* the same for the features resident in a device frame (orbm_frame_upload / orbm_frame_from_extractor / orbx_frame_stereo) */
int orbv_transform_frame(orbv_vocab* v, orbm_frame* f, int levelsup, int32_t* word_id, int32_t* node_id, double* weight);
/* void transform(const vector<TDescriptor>&, BowVector&, FeatureVector&, int levelsup), :1127-1199, second half: BowVector
 * (ascending word ids, values weighted / normalised as the vocabulary prescribes) and FeatureVector (orbm_featvec_view layout)
 * from the per-feature outputs above.  Output arrays hold up to n entries (fv_start: n + 1). */
int orbv_bow_assemble(const orbv_vocab* v, const int32_t* word_id, const int32_t* node_id, const double* weight, int n,
                      int32_t* bow_word, double* bow_value, int32_t* n_words,
                      uint32_t* fv_node, uint32_t* fv_start, uint32_t* fv_feat, int32_t* n_fv_nodes);

/* void MapPoint::ComputeDistinctiveDescriptors(), S/MapPoint.cc:448-522, for m map points at once: the descriptors of point p's
 * observations are desc[start[p] .. start[p+1]) (x 32 bytes, in the order the reference pushes them into vDescriptors);
 * best[p] = index inside that list of the descriptor with the least median Hamming distance to the others (first one on
 * ties), or -1 for a point without observations (the reference returns early and keeps the old descriptor). */
int orbm_distinctive_descriptors(int device, const uint8_t* desc, const int32_t* start, int m, int32_t* best);

/* double L1Scoring::score(v1, v2) (Thirdparty/DBoW2/DBoW2/ScoringObject.cpp:23-68) of one query BowVector against m candidates
 * (the inner loop of KeyFrameDatabase::DetectNBestCandidates, S/KeyFrameDatabase.cc:594-761): candidates as CSR over
 * ascending word ids.  score[m] in [0, 1]. */
int orbv_score_l1(int device, const int32_t* q_word, const double* q_value, int nq, const int32_t* cand_start,
                  const int32_t* cand_word, const double* cand_value, int m, double* score);

/* ---------------------------------------------------------------- place recognition database */

/* What KeyFrameDatabase::DetectNBestCandidates (S/KeyFrameDatabase.cc:594-761) reads, flattened; keyframes are indices
 * 0 .. n_kfs-1 (the caller's KeyFrame* table). */
typedef struct orbd_database_view {
  int32_t n_kfs, n_words;
  const int32_t* inv_start;    /* n_words + 1: CSR over words */
  const int32_t* inv_kf;       /* mvInvertedFile[word] (I/KeyFrameDatabase.h:82): keyframes in insertion order */
  const int32_t* bow_start;    /* n_kfs + 1 */
  const int32_t* bow_word;     /* pKFi->mBowVec, ascending word ids */
  const double*  bow_value;
  const int32_t* covis_start;  /* n_kfs + 1 */
  const int32_t* covis_kf;     /* pKFi->GetBestCovisibilityKeyFrames(10), best first (:682) */
  const int32_t* map_id;       /* n_kfs: pKFi->GetMap() as an id */
  const uint8_t* bad;          /* n_kfs: pKFi->isBad() */
  const uint8_t* map_bad;      /* n_kfs: pKFi->GetMap()->IsBad() */
} orbd_database_view;
typedef struct orbd_database orbd_database;
int orbd_database_create(int device, const orbd_database_view* view, orbd_database** out);   /* uploads the view */
int orbd_database_destroy(orbd_database* d);
/* void KeyFrameDatabase::DetectNBestCandidates(KeyFrame* pKF, vector<KeyFrame*>& vpLoopCand, vector<KeyFrame*>& vpMergeCand,
 *                                              int nNumCandidates), S/KeyFrameDatabase.cc:594-761, for the query keyframe's
 * BowVector (q_word ascending / q_value), its connected keyframes (connected[n_kfs] = spConnectedKF.count, :603,618) and its
 * map.  On the device: the inverted-file walk with the common-word counts (:605-633), the 0.8 * max filter (:638-646), the
 * L1 scores (:652-663) and the covisibility accumulation (:673-701); the ranking (stable sort by accumulated score, :705)
 * and the selection (:713-735) run on the host over the few scored keyframes.  place_score[n_kfs] is
 * pKFi->mPlaceRecognitionScore, in/out: keyframes that share a word without reaching the word threshold keep the score of
 * an earlier query, and the accumulation reads it (the reference does).  Pinned: a bad keyframe is skipped in the selection
 * (the reference's `continue` at :718-719 does not advance its iterator and never terminates).
 * loop_cand / merge_cand: up to n_candidates keyframe indices each. */
int orbd_detect_n_best_candidates(orbd_database* d, const int32_t* q_word, const double* q_value, int nq, const uint8_t* connected,
                                  int32_t query_map_id, int n_candidates, float* place_score, int32_t* loop_cand, int32_t* n_loop,
                                  int32_t* merge_cand, int32_t* n_merge);

/* ---------------------------------------------------------------- KeyFrame wire blocks */

/* What a client sends per keyframe feature in orb_slam3_ros/KF (R/msg/KF.msg:29-31): CvKeyPoint {f32 x, f32 y, u8 size, f32 angle,
 * u8 response, i8 octave} (R/msg/CvKeyPoint.msg:1-9, 15 bytes packed) and Descriptor {u8[32]}, as one block [N x 15 | N x 32]
 * with the conversions of Converter::toCvKeyPointMsg / fromCvKeyPointMsg (S/Converter.cc:217-245).  The blocks are what agents
 * exchange (RCCL all-gather on device memory, see multi_orbslam3_amd/harness.py) so that a server GPU can run the KeyFrame
 * matchers on other agents' keyframes. */
int orbk_wire_bytes(int n);                                              /* 47 * n */
int orbk_pack_frame(orbm_frame* f, uint8_t* wire, int wire_on_device);
int orbk_frame_from_wire(orbm_frame* f, const orbm_frame_view* view, const uint8_t* wire, int n, int wire_on_device);
/* host copies of a device-resident frame's features */
int orbm_frame_download(orbm_frame* f, orbx_keypoint* kps, uint8_t* desc);

/* ---------------------------------------------------------------- local bundle adjustment */

/* A GeometricCamera as the optimiser's edges use it (I/CameraModels/GeometricCamera.h:77-92): GetType() and the parameter vector
 * mvParameters = {fx, fy, cx, cy} (Pinhole) / {fx, fy, cx, cy, k1, k2, k3, k4} (KannalaBrandt8), float32 as the reference keeps them. */
#define ORBG_CAM_PINHOLE 0           /* GeometricCamera::CAM_PINHOLE */
#define ORBG_CAM_KANNALA_BRANDT8 1   /* GeometricCamera::CAM_FISHEYE */
typedef struct orbg_camera {
  int32_t model;
  float fx, fy, cx, cy;
  float k[4];                        /* k1..k4 of KannalaBrandt8 (mvParameters[4..7]); ignored for a pinhole */
} orbg_camera;
/* The cameras of a KeyFrame / Frame whose edges go through GeometricCamera::project / projectJac: mpCamera, and -- for the
 * two-fisheye rig (NLeft != -1) -- mpCamera2 with mTrl, the right camera's pose in the left camera's frame
 * (I/KeyFrame.h:635-642, I/Frame.h:277-297).  One rig per problem: every keyframe of a map holds the same camera objects. */
typedef struct orbg_camera_rig {
  orbg_camera left;                  /* mpCamera: EdgeSE3ProjectXYZ / EdgeSE3ProjectXYZOnlyPose (I/OptimizableTypes.h:31-57,89-115) */
  int32_t has_right;                 /* mpCamera2 != NULL */
  orbg_camera right;                 /* mpCamera2: EdgeSE3ProjectXYZToBody / EdgeSE3ProjectXYZOnlyPoseToBody (:59-87,117-144) */
  float Trl[12];                     /* mTrl, 3 x 4 row-major float32 (Converter::toSE3Quat reads exactly these, S/Converter.cc:34-44) */
} orbg_camera_rig;
/* ---- two-camera rig frames in the matcher (Frame::Nleft != -1, S/Frame.cc:1017-1091): the two cameras' features are two orbm_frame
 * objects -- `left` holds mvKeys[0, Nleft) with mGrid, `right` holds mvKeysRight with mGridRight (S/Frame.cc:360-391), neither has
 * uRight; entry Nleft + i of Frame::mvpMapPoints / mDescriptors is feature i of `right`.
 *
 * orbm_is_in_frustum_rig: Frame::isInFrustum for such a frame (S/Frame.cc:545-554): isInFrustumChecks (:1154-1231) through
 * rig->left for the left camera and through rig->right after mTrl for the right one; Tlr = Frame::mTlr (3 x 4 row-major, its
 * translation enters the right camera's centre, :1164).  Outputs, m entries each, per camera: mbTrackInView(R), mTrackProjX(R) /
 * mTrackProjY(R), mTrackDepth(R), mnTrackScaleLevel(R) (-1 where the checks fail, :546-547), mTrackViewCos(R); fields of a point that
 * fails a camera's checks are 0 (the reference leaves the previous frame's values there).
 * rig->has_right == 0: ONE camera behind a model -- a monocular fisheye Frame (Nleft == -1, mpCamera a KannalaBrandt8): the Nleft == -1
 * branch of Frame::isInFrustum (S/Frame.cc:466-543) makes the same checks through mpCamera->project; the left outputs are written, the
 * right ones (and Tlr) may be NULL.  (orbm_is_in_frustum is that branch for a pinhole.) */
int orbm_is_in_frustum_rig(orbm_frame* left, const float* Tcw /*16*/, const orbg_camera_rig* rig, const float* Tlr /*12*/,
                           const orbm_worldpoints_view* pts, float viewing_cos_limit, uint8_t* in_view, float* proj_x, float* proj_y,
                           float* track_depth, int32_t* scale_level, float* view_cos, uint8_t* in_view_r, float* proj_x_r, float* proj_y_r,
                           float* track_depth_r, int32_t* scale_level_r, float* view_cos_r);
/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints) on such a frame (S/ORBmatcher.cc:44-214
 * with the right camera's block :145-211).  mps: the left camera's track fields and the per-point fields (bad, track_depth, desc,
 * n_obs; proj_xr is not read); mps_r: track_in_view = mbTrackInViewR, proj_x / proj_y = mTrackProjXR / mTrackProjYR, scale_level =
 * mnTrackScaleLevelR, view_cos = mTrackViewCosR (its other fields are not read).  left_to_right[Nleft] = Frame::mvLeftToRightMatch,
 * right_to_left[Nright] = mvRightToLeftMatch (-1 = none): a match on one side is also written to the stereo partner on the other
 * (:132-136,199-203).  assigned_mp / assigned_obs have Nleft + Nright entries (in/out, as orbm_search_by_projection_mps). */
int orbm_search_by_projection_mps_rig(orbm_frame* left, orbm_frame* right, const orbm_mappoints_view* mps, const orbm_mappoints_view* mps_r,
                                      const int32_t* left_to_right, const int32_t* right_to_left, float th, int far_points,
                                      float th_far_points, float nnratio, int32_t* assigned_mp, int32_t* assigned_obs, int* nmatches);

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) with CurrentFrame.Nleft != -1
 * (S/ORBmatcher.cc:1970-2186): per map point of the last frame the left camera's search through rig->left (mpCamera->project), then
 * (:2092-2160) the point in the right camera's frame (rig->Trl) projected -- through mpCamera again, as the reference does -- and
 * searched in the right camera's grid; a point whose left window holds no feature is not searched on the right either (:2033).
 * `last`: the last frame's Nleft + Nright entries (octave / angle of mvKeys resp. mvKeysRight).  assigned_*: Nleft + Nright entries.
 * rig->has_right == 0 and right == NULL: a monocular Frame whose camera is a model (Nleft == -1, mpCamera a KannalaBrandt8): the left
 * camera's search alone, through mpCamera->project (orbm_search_by_projection_frame is the same for a pinhole). */
int orbm_search_by_projection_frame_rig(orbm_frame* left, orbm_frame* right, const float* Tcw_cur, const orbg_camera_rig* rig,
                                        const orbm_lastframe_view* last, float th, int mono, int check_orientation, int32_t* assigned_mp,
                                        int32_t* assigned_obs, int* nmatches);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame &F, vector<MapPoint*>&) with F.Nleft != -1 (S/ORBmatcher.cc:269-471, the branch
 * :342-430): `f` holds ALL of the Frame's features -- mvKeys then mvKeysRight, descriptor rows as in mDescriptors, indices as in
 * F.mFeatVec -- and n_left = F.Nleft.  Per keyframe feature the best two of a vocabulary bucket are kept per camera; the left
 * camera's best passes TH_LOW and the ratio test, the right camera's best only TH_LOW, and only when the left one passed TH_LOW
 * (:373-428).  kf_angle[i]: the keyframe keypoint's angle (mvKeysUn / mvKeys / mvKeysRight as :379-382 picks).  Other arguments and
 * `matches` as orbm_search_by_bow. */
int orbm_search_by_bow_rig(orbm_frame* f, int n_left, const orbm_featvec_view* fv_frame, const uint8_t* kf_desc, int nkf,
                           const uint8_t* kf_mp_valid, const float* kf_angle, const orbm_featvec_view* fv_kf, float nnratio,
                           int check_orientation, int32_t* matches, int* nmatches);

/* Frame::ComputeStereoFishEyeMatches (S/Frame.cc:1093-1150): the left-right matcher of the two-fisheye Frame constructor.  The
 * features of the two cameras' lapping areas -- [mono_left, n_left) and [mono_right, n_right): ORBextractor::operator() puts them
 * behind the monocular ones (S/ORBextractor.cc:1136-1160, orbx_extract's lapping arguments) -- are matched brute force (the two
 * nearest right descriptors per left one: cv::BFMatcher::knnMatch, k = 2, NORM_HAMMING), kept under Lowe's ratio 0.7, and
 * triangulated by KannalaBrandt8::TriangulateMatches (parallax, positive depth in both cameras, reprojection error against
 * 5.991 sigma^2 in both; S/CameraModels/KannalaBrandt8.cpp:335-420).  Both cameras must be ORBG_CAM_KANNALA_BRANDT8.
 * Outputs: left_to_right[n_left] = mvLeftToRightMatch, right_to_left[n_right] = mvRightToLeftMatch (-1 = none; of several left features
 * that pick one right feature the last one stays there, as in the reference's loop), depth[n_left] = mvDepth (-1), points3d[3 n_left] =
 * mvStereo3Dpoints (written where a match was accepted), *n_matches = nMatches.
 * The homogeneous solve of Triangulate is cv::SVD::compute in the reference (float32, one-sided Jacobi); here: the smallest
 * eigenvector of A^T A in float64 -- depths and points agree with an OpenCV build to float32 rounding, not to the bit. */
typedef struct orbx_fisheye_stereo_view {
  int32_t n_left, n_right, mono_left, mono_right;    /* Nleft, Nright, monoLeft, monoRight */
  const orbx_keypoint* kps_left;                     /* mvKeys */
  const orbx_keypoint* kps_right;                    /* mvKeysRight */
  const uint8_t* desc_left;                          /* mDescriptors, n_left x 32 */
  const uint8_t* desc_right;                         /* mDescriptorsRight, n_right x 32 */
  const float* level_sigma2;                         /* mvLevelSigma2 */
  int32_t n_levels;
  orbg_camera left, right;                           /* mpCamera, mpCamera2 */
  float Tlr[12];                                     /* mTlr, 3 x 4 row-major: mRlr, mtlr (S/Frame.cc:1073-1074) */
} orbx_fisheye_stereo_view;
int orbx_fisheye_stereo_matches(int device, const orbx_fisheye_stereo_view* view, int32_t* left_to_right, int32_t* right_to_left, float* depth,
                                float* points3d, int* n_matches);

/* `ur` of an observation made by the RIGHT camera of the rig (get<1>(indexes) != -1, S/Optimizer.cc:2086-2120; i >= Nleft,
 * :1121-1150): u, v are then mvKeysRight[rightIndex].pt and the edge is the *ToBody kind.  In a problem whose rig has a right camera
 * any ur <= -1.5 reads as this (mvuRight is -1 throughout on such frames); in every other problem a negative ur is a monocular
 * observation, as it always was. */
#define LBA_UR_RIGHT_CAMERA (-2.0f)

/* One reprojection edge (S/Optimizer.cc:2021-2120): stereo if ur >= 0, the right camera's if ur == LBA_UR_RIGHT_CAMERA, mono
 * otherwise (the reference's mvuRight is -1 there). */
typedef struct lba_edge {
  int32_t pose;        /* index into poses[] */
  int32_t point;       /* index into points[] */
  float   u, v, ur;    /* kpUn.pt.x, kpUn.pt.y, mvuRight (<0 => monocular edge) */
  float   inv_sigma2;  /* mvInvLevelSigma2[kpUn.octave] */
} lba_edge;

/* Everything S/Optimizer.cc:1934-2124 reads.  Poses MUST be ordered
 * free/fixed arbitrarily but listed in ascending vertex id (Optimizer::GetID, I/Optimizer.h:104-112),
 * points likewise; edges in creation order (per point, its observations). */
typedef struct lba_problem {
  int32_t n_poses, n_points, n_edges;
  const float*   poses;        /* n_poses x 16, Tcw row-major float32 (KeyFrame::GetPose) */
  const uint8_t* pose_fixed;   /* n_poses */
  const float*   points;       /* n_points x 3 */
  const lba_edge* edges;
  float fx, fy, cx, cy, bf;
  double lambda_init;          /* 0 = auto (tau*max diag); 100 for inertial maps, S/Optimizer.cc:1924-1925 */
  int32_t its_round1, its_round2;  /* 5 and 10, S/Optimizer.cc:2132,2203 */
  int32_t device;
  /* NULL: mpCamera is the pinhole {fx, fy, cx, cy} above and there is no second camera (every BASELINE configuration).  Otherwise
   * the monocular edges project through rig->left and the right camera's edges through rig->right after mTrl; stereo edges
   * (g2o::EdgeStereoSE3ProjectXYZ carries its own fx, fy, cx, cy, bf, S/Optimizer.cc:2071-2075) keep using the five scalars. */
  const orbg_camera_rig* rig;
} lba_problem;

typedef struct lba_result {
  float*   poses;          /* n_poses x 16 (Converter::toCvMat(SE3Quat)), caller-allocated */
  float*   points;         /* n_points x 3 */
  double*  edge_chi2;      /* n_edges, e->chi2() at S/Optimizer.cc:2219,2249 */
  uint8_t* edge_depth_pos; /* n_edges, e->isDepthPositive() */
  uint8_t* edge_outlier;   /* n_edges, the vToErase predicate (S/Optimizer.cc:2219-2253) */
  int32_t  status;         /* LBA_* */
  int32_t  iters_round1, iters_round2;  /* LM iterations actually run */
  int32_t  n_outliers;
  double   chi2_initial, chi2_final;    /* activeRobustChi2 at first linearisation / after last accepted step */
  double*  trace;          /* optional (may be NULL): per LM iteration {lambda, chi2, trials}; cap trace_cap rows */
  int32_t  trace_cap, trace_len;
} lba_result;

/* void Optimizer::LocalBundleAdjustment(KeyFrame*, bool* pbStopFlag, Map*, int&, int) numerical core,
 * S/Optimizer.cc:1917-2267 + 2321-2396 (state write-back into result).  stop_flag may be NULL;
 * it is polled between LM iterations/trials exactly where g2o polls forceStopFlag
 * (G/core/sparse_optimizer.cpp:376, G/core/optimization_algorithm_levenberg.cpp:149).  A positive value is the
 * reference's `true` (raised by Tracking, S/LocalMapping.cc:381-386).  A NEGATIVE value -k is a deterministic form for
 * tests: the flag reads as raised once k Levenberg-Marquardt trials have been evaluated, whatever the timing (INT32_MIN:
 * k = 0, i.e. raised right after the check that precedes optimize(), S/Optimizer.cc:2127-2129). */
int lba_solve(const lba_problem* p, const volatile int32_t* stop_flag, lba_result* r);
/* The same solve polling the reference's OWN flag: `bool* pbStopFlag` (I/Optimizer.h:42) points at LocalMapping::mbAbortBA,
 * a one-byte bool that LocalMapping::InterruptBA sets from the Tracking thread while the solve runs (S/LocalMapping.cc:381-386,
 * :118 hands &mbAbortBA to the optimiser).  stop_bool is that address, passed through unchanged (any non-zero byte = raised);
 * it is read -- never written -- at exactly the poll points above, so the reference's InterruptBA() needs no change. */
int lba_solve_b(const lba_problem* p, const volatile uint8_t* stop_bool, lba_result* r);

/* Persistent LBA workspace variant: avoids per-call device allocation. */
typedef struct lba_handle lba_handle;
int lba_create(int device, int cap_poses, int cap_points, int cap_edges, lba_handle** out);
int lba_destroy(lba_handle* h);
int lba_solve_h(lba_handle* h, const lba_problem* p, const volatile int32_t* stop_flag, lba_result* r);
int lba_solve_hb(lba_handle* h, const lba_problem* p, const volatile uint8_t* stop_bool, lba_result* r);   /* bool flag, see lba_solve_b */
/* The same solve on a worker thread owned by the handle, as the reference runs it on its LocalMapping thread next to
 * Tracking (S/ClientSystem.cc:105-106).  lba_solve_async returns immediately; lba_wait blocks until the solve is done
* and returns its status (ORBG_OK when nothing was submitted).  problem / stop_flag / result must stay valid until then;
 * one solve in flight per handle (a second lba_solve_async before lba_wait returns ORBG_BAD_ARG). */
int lba_solve_async(lba_handle* h, const lba_problem* problem, const volatile int32_t* stop_flag, lba_result* result);
int lba_solve_async_b(lba_handle* h, const lba_problem* problem, const volatile uint8_t* stop_bool, lba_result* result);
int lba_wait(lba_handle* h, double* solve_ms /* wall time of that solve on the worker, may be NULL */);
/* Measurement hooks (bench.py roofline): with profiling on, one launch per solve of the reduced-camera-system LDL^T
 * (G/solvers/linear_solver_eigen.h:94-124 behind G/core/block_solver.hpp:447) is bracketed by a HIP event pair on the
 * handle's stream; the stats are the summed bracket time, the number of brackets, the unknowns of the last system
 * (6 x free poses) and whether the FP64 matrix-core kernel solved it.  lba_event_overhead: cost of an empty pair. */
int lba_set_profiling(lba_handle* h, int on, int reset);
int lba_get_solver_stats(lba_handle* h, double* sum_ms, int64_t* n_brackets, int32_t* n_unknowns, int32_t* matrix_core);
int lba_event_overhead(lba_handle* h, int reps, float* ms);
/* Health of the handle's solver: how many launches of the eight-workgroup LDL^T (windows of 21 .. 50 free poses) gave up waiting
 * for a participant the dispatcher did not place within 2 s.  Such a launch is NOT taken for a non-positive-definite system (which
 * g2o answers with a rejected LM step, G/core/optimization_algorithm_levenberg.cpp:118-127): the window is solved again on the
 * one-workgroup kernels, which the handle then keeps using, and the event is counted here.  0 on a healthy box. */
int lba_get_watchdog_count(lba_handle* h, int64_t* n_timeouts);

/* ---------------------------------------------------------------- pose-only optimisation */

/* Everything Optimizer::PoseOptimization(Frame*) reads (S/Optimizer.cc:964-1278): one entry per
 * feature that holds a map point. */
typedef struct pose_opt_problem {
  int32_t n;               /* nInitialCorrespondences */
  const float* Xw;         /* n x 3, pMP->GetWorldPos() */
  const float* u;          /* mvKeysUn[i].pt.x */
  const float* v;          /* mvKeysUn[i].pt.y */
  const float* ur;         /* mvuRight[i]; < 0 => monocular edge */
  const float* inv_sigma2; /* mvInvLevelSigma2[octave] */
  float fx, fy, cx, cy, bf;
  float Tcw[16];           /* pFrame->mTcw (row-major), the estimate every round restarts from */
  int32_t device;
  /* NULL: Frame::mpCamera is the pinhole above, mpCamera2 == NULL.  Otherwise as in lba_problem: entries with
   * ur == LBA_UR_RIGHT_CAMERA are features i >= Nleft (u, v = mvKeysRight[i - Nleft].pt, S/Optimizer.cc:1121-1150), the other
   * monocular entries project through rig->left (u, v = mvKeys[i].pt, :1091-1119). */
  const orbg_camera_rig* rig;
} pose_opt_problem;

typedef struct pose_opt_result {
  float    Tcw[16];        /* optimised pose (pFrame->SetPose) */
  uint8_t* outlier;        /* n, pFrame->mvbOutlier for the correspondences (caller-allocated) */
  int32_t  n_inliers;      /* return value: nInitialCorrespondences - nBad (0 if n < 3) */
  int32_t  n_bad;
  int32_t  iters[4];       /* LM iterations run in each of the 4 rounds */
  double   chi2[4];        /* robustified chi2 at the end of each round (active edges) */
} pose_opt_result;

/* int Optimizer::PoseOptimization(Frame *pFrame), S/Optimizer.cc:964-1278: 4 rounds x 10 LM iterations on one SE3
 * vertex with unary reprojection edges (I/OptimizableTypes.h:31-57, G/types/types_six_dof_expmap.h:208-236), outliers
 * re-classified after every round (chi2 > 5.991 / 7.815), Huber kernels dropped for the last round.  The whole solve
 * runs in ONE kernel launch (LM control flow on the device). */
int pose_optimize(const pose_opt_problem* p, pose_opt_result* r);

/* ---------------------------------------------------------------- streams, hardware queues, host threads
 * Streams.  Every handle enqueues its work on ONE HIP stream.  By default that stream comes from a per-device pool the library
 * creates with its first handle: four hipStreamNonBlocking streams -- L (local BA handles), E0 / E1 (extractor handles,
 * alternating; a frame built by a constructor is on its extractor's stream until the constructor has been COLLECTED -- from then on its
 * searches run on the frame's own stream, M, and never queue behind constructors of later frames), M (map uploads, PoseOptimization, vocabulary and
 * database handles, host-built frames, the stand-alone utilities).  None of them is the legacy null stream: nothing the library
 * launches joins, or is joined by, the blocking streams of the application (SURVEY.md 8(b) "no hidden global state": the pool is the
 * one piece of per-device state the handles share, and it can be replaced).  *_set_stream hands a handle the CALLER's stream instead
 * (a hipStream_t passed as void*; never the null stream -- NULL means "back to the pool"); the handle must be idle, its old stream
 * is drained first, and the caller's stream is never destroyed by the library.  ORBG_STREAM_POOL=0 in the environment gives every
 * handle a private stream.
 * Hardware queues.  The ROCm runtime multiplexes the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (4 by default); the
 * streams that share a queue are serialised.  The pool's four streams are created together so that they occupy the four default
 * queues.  An application that keeps further streams of its own busy next to the library should start with GPU_MAX_HW_QUEUES =
 * 4 + its own busy streams in the environment (it is read when the runtime initialises; measured here: 4, 5 and 6 are equally good
 * for an agent, 3 costs a third of the frame rate, 8 half of it).
 * Host threads.  A synchronous call blocks its caller by SPINNING on a completion word in pinned memory for up to a bounded time
 * (then it falls back to the runtime's blocking wait); the asynchronous forms own one thread each -- lba_solve_async: one worker per
 * handle, ORBX_SUBMIT_ASYNC: one ingest thread per process -- which spin for a few hundred microseconds between jobs before they
 * sleep.  An agent that uses all three therefore keeps up to three cores busy (tracking thread, local-BA worker, ingest thread);
 * The policy is per ROLE of the waiting thread (orbg_set_wait_policy below): ORBG_ROLE_CALLER (any thread of the application that
 * calls the library: Tracking), ORBG_ROLE_LBA_WORKER, ORBG_ROLE_INGEST.  A role that does not spin waits in the runtime's blocking
 * form / on a condition variable (6-10 us more latency per wait, no busy core).  A host whose CPU quota cannot feed three spinning
 * threads per agent keeps the tracking thread spinning (it is on the frame's critical path) and lets the other two block.
 * ORBG_NO_POLL in the environment sets the start-up policy: "1" / "all" = no role spins; a comma list of caller / lba / ingest =
 * those roles block.
 * Cameras.  The fused Frame constructors (orbx_frame_stereo*) build the grid from the extracted keypoints as they are, i.e. they
 * implement Frame::UndistortKeyPoints for mDistCoef[0] == 0 (rectified stereo: mvKeysUn = mvKeys, S/Frame.cc:723-727).  The
 * reference's image bounds are exactly the image rectangle in that case (S/Frame.cc:775-783) and the undistorted corners otherwise
 * (:753-773): a view whose bounds are not (0, width, 0, height) is refused with ORBG_BAD_ARG.  Distorted cameras (the mono agents
 * with EuRoC intrinsics) use orbx_frame_mono*, which undistorts on the device. */
enum { ORBG_ROLE_CALLER = 0, ORBG_ROLE_LBA_WORKER = 1, ORBG_ROLE_INGEST = 2 };
int orbg_set_wait_policy(int role, int spin);   /* spin != 0: waits of that role spin on completion words; 0: they block.  Process-wide, any time */
int orbg_get_wait_policy(int role);             /* 1 = spins, 0 = blocks, ORBG_BAD_ARG for an unknown role */
int orbx_set_stream(orbx_handle* h, void* hip_stream);
int orbm_frame_set_stream(orbm_frame* f, void* hip_stream);
int orbm_map_set_stream(orbm_map* m, void* hip_stream);
int lba_set_stream(lba_handle* h, void* hip_stream);
int orbv_vocab_set_stream(orbv_vocab* v, void* hip_stream);
int orbd_database_set_stream(orbd_database* d, void* hip_stream);
int pose_opt_set_stream(int device, void* hip_stream);      /* the calling thread's pose_optimize calls on `device` */

/* ---------------------------------------------------------------- misc */
const char* orbg_version(void);
const char* orbg_strerror(int code);
int orbg_device_count(void);
/* Spins until everything enqueued so far on the library's pooled streams of `device` has completed (handles on caller-supplied
 * streams are the caller's to wait for).  After it, the runtime's own device synchronisation finds nothing outstanding. */
int orbg_quiesce(int device);
/* Device timings (ms, hipEvent on the handle's stream) of the most recent call, for bench.py.
 * ms[0] pyramid, [1] FAST+gather, [2] host quad-trees (wall), [3] orientation+descriptors, [4] stereo match,
 * [5] fast_cells_kernel alone.  orbx_set_profiling: 0 = record nothing, 1 = only [2] and [5] (default),
 * 2 = every stage (more hipEventRecord calls per frame). */
int orbx_set_profiling(orbx_handle* h, int level);
int orbx_get_timings(orbx_handle* h, float* ms /*8*/);
/* Average elapsed time of an EMPTY event pair on the handle's stream: the constant the profiling brackets add to a
 * bracketed kernel's duration (bench.py reports the bracketed time both raw and net of this). */
int orbx_event_overhead(orbx_handle* h, int reps, float* ms);
/* Profiling level 1 brackets fast_cells_kernel with an event pair; with interval k only every k-th extraction is bracketed.
 * The bracket times accumulate in the handle (sum in ms, number of samples) until reset. */
int orbx_set_profile_interval(orbx_handle* h, int interval, int reset);
/* Which kernel of the Frame-constructor chain the level-1 event pair brackets (default: fast_cells_kernel); the
   accumulated times are read with orbx_get_fast_kernel_stats and reset by this call.  For bench.py's roofline line,
   which has to describe whichever kernel dominates the step. */
#define ORBX_PROF_FAST 0
#define ORBX_PROF_OCTREE 1
#define ORBX_PROF_ORIENT_DESC 2
#define ORBX_PROF_PYRAMID 3
int orbx_set_profile_kernel(orbx_handle* h, int which);
/* Host-side timeline of the handle's last Frame constructors (<= 512 kept), oldest first: out[i * 5 + f] in microseconds,
 * f = 0 queue (hand-over to the ingest thread -> it starts; 0 for synchronous submissions), 1 pack (rows copied into the pinned
 * staging slot), 2 enqueue (host time of the launches), 3 wait (time the collecting thread was blocked), 4 latency (hand-over ->
 * constructor complete, as the collecting thread sees it; the synchronous orbx_frame_stereo records pack and latency only).
 * *n = entries written; reset != 0 forgets them.  For bench.py's per-step diagnosis of a shared host. */
int orbx_get_ctor_timeline(orbx_handle* h, float* out, int cap, int* n, int reset);
int orbx_get_fast_kernel_stats(orbx_handle* h, double* sum_ms, int64_t* n);

#ifdef __cplusplus
}
#endif
#endif /* ORBGPU_H_ */
