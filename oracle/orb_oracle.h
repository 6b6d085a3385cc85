/*
 * orb_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A dependency-free C++17 restatement of the reference's hot path (ORBextractor, stereo matching,
 * Hamming matchers, LocalBundleAdjustment of yutongwangBIT/multi_orbslam3) used ONLY by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker / CPU baseline.  The
 * product library (multi_orbslam3_amd/liborbgpu.so) never links, loads or calls anything in here.
 *
 * PARITY STATUS: "parity unpinned" -- the reference ships no tests, golden vectors or fixtures for
 * this path and cannot be compiled in the authoring container (needs
 * OpenCV, Eigen, ROS), so this restatement is pinned only against first-principles known-answer
 * tests (tests/test_oracle_*.py) and -- where the reference's text is plain scalar / loop code -- against that
 * text executed statement by statement (tests/test_reference_formulas.py: tables, cell tiling, IC_Angle, rBRIEF, grid, stereo
 * sub-pixel step, camera models, g2o's stereo edge, both tracking SearchByProjection overloads whole).  Arithmetic that lives in un-vendored
 * third parties (OpenCV 3.2-era cv::resize / cv::FAST / cv::GaussianBlur / cv::fastAtan2 /
 * cvRound; Eigen >= 3.1 fixed-size algebra + SimplicialLDLT) is restated from their published
 * algorithms.
 *
 * Shares only the plain-C struct layouts of include/orbgpu.h with the product.
 */
#ifndef ORB_ORACLE_H_
#define ORB_ORACLE_H_

#include "../include/orbgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_extractor oracle_extractor;

/* ---- image primitives (Appendix A), exported for the known-answer tests */
void oracle_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);
void oracle_border_reflect101(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int border, int dstride);
/* FAST-9/16 score (largest threshold at which the pixel is a corner, -1 if it is not one at t=0);
 * needs a 3-px margin around (x,y). */
int  oracle_fast_score(const uint8_t* img, int stride, int x, int y);
/* cv::FAST(img, kps, threshold, nonmax=true) on a w x h sub-image: out = cap x {x,y,score}. */
int  oracle_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold, int nonmax,
                        int32_t* xys, int cap);
void oracle_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
void oracle_gaussian_blur7_taps(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, const int* taps4);
float oracle_fast_atan2(float y, float x);
int  oracle_hamming(const uint8_t* a, const uint8_t* b);
/* the cells cv::FAST runs on at a level of that size (S/ORBextractor.cc:771-804): n x {x0, x1, y0, y1, row, column}, returns n */
int  oracle_fast_cell_grid(int level_w, int level_h, int32_t* rects6, int cap, int32_t* w_cell, int32_t* h_cell);
/* IC_Angle (S/ORBextractor.cc:75-102) of one keypoint on a level image; the radius-15 disc must lie inside */
float oracle_ic_angle(const oracle_extractor* e, const uint8_t* img, int stride, int x, int y);
/* computeOrbDescriptor (S/ORBextractor.cc:105-145) of one keypoint on an already blurred image; the 31 x 31 patch must lie inside */
void oracle_orb_descriptor(float kp_angle_deg, const uint8_t* img, int stride, int x, int y, uint8_t* desc32);

/* ---- extractor (S/ORBextractor.cc) */
int oracle_extractor_create(const orbx_config* cfg, oracle_extractor** out);
int oracle_extractor_destroy(oracle_extractor* e);
int oracle_get_tables(const oracle_extractor* e, float* scale, float* inv_scale, float* sigma2,
                      float* inv_sigma2, int32_t* features_per_level);
int oracle_get_umax(const oracle_extractor* e, int32_t* umax16);
int oracle_extract(oracle_extractor* e, const uint8_t* img, int width, int height, int stride,
                   int lap0, int lap1, orbx_keypoint* kps, uint8_t* desc, int cap, int* n, int* n_mono);
int oracle_get_level(oracle_extractor* e, int level, uint8_t* host_out, int* width, int* height);
int oracle_get_level_bordered(oracle_extractor* e, int level, uint8_t* host_out, int* width, int* height);
int oracle_get_candidates(oracle_extractor* e, int level, int32_t* xys, int cap, int* n);
/* DistributeOctTree alone (S/ORBextractor.cc:537-761): in = n x {x,y,score} (relative coords),
 * out = kept candidates in list order. */
int oracle_distribute_octree(const int32_t* xys, int n, int min_x, int max_x, int min_y, int max_y,
                             int n_target, int32_t* out_xys, int cap);

/* ---- Frame pieces (S/Frame.cc) */
int oracle_stereo_match(oracle_extractor* left, oracle_extractor* right,
                        const orbx_keypoint* kps_l, const uint8_t* desc_l, int n_l,
                        const orbx_keypoint* kps_r, const uint8_t* desc_r, int n_r,
                        float bf, float b, float* uright, float* depth);
/* cv::undistortPoints(src, dst, K, distCoeffs, Mat(), P = K) as Frame::UndistortKeyPoints / ComputeImageBounds call it
 * (S/Frame.cc:740,767): xy = n x {x, y} float32, in place allowed.  dist NULL or k1 == 0: copy (S/Frame.cc:723-727). */
int oracle_undistort_points(const float* xy_in, int n, float fx, float fy, float cx, float cy, const orbx_distortion* dist, float* xy_out);
/* the parabola fit and disparity test at the end of a left keypoint's stereo match (S/Frame.cc:918-946); 1: uright / depth written */
int oracle_stereo_subpixel(const float* dists, int L, int bestincR, float scaleduR0, float scale_of_level, float uL, float minD, float maxD,
                           float bf, float* uright, float* depth);
int oracle_build_grid(const orbm_frame_view* view, int32_t* cell_start, int32_t* cell_items);
int oracle_features_in_area(const orbm_frame_view* view, float x, float y, float r, int min_level,
                            int max_level, int32_t* out_idx, int cap);
int oracle_is_in_frustum(const orbm_frame_view* view, const float* Tcw, const orbm_worldpoints_view* pts,
                         float viewing_cos_limit, uint8_t* track_in_view, float* proj_x, float* proj_y,
                         float* proj_xr, float* track_depth, int32_t* scale_level, float* view_cos);

/* ---- matchers (S/ORBmatcher.cc) */
/* ORBmatcher::ComputeThreeMaxima (:2312-2353) on the bin sizes of a rotation histogram (ind1..3 start at -1), and the bin of a pair of
 * keypoint angles (:2082-2087 with factor = 1.0f / HISTO_LENGTH, :1978) */
void oracle_three_maxima(const int32_t* bin_sizes, int L, int32_t* ind3);
int  oracle_rot_bin(float angle_a, float angle_b);
int oracle_hamming_matrix(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* dist);
int oracle_hamming_best2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* out4);
int oracle_search_by_projection_mps(const orbm_frame_view* view, const orbm_mappoints_view* mps, float th,
                                    int far_points, float th_far_points, float nnratio,
                                    int32_t* assigned_mp, int32_t* assigned_obs, int* nmatches);
/* Two-camera rig frames (Frame::Nleft != -1): Frame::isInFrustum with isInFrustumChecks for both cameras (S/Frame.cc:545-554,1154-1231;
 * Tlr = Frame::mTlr, 3 x 4) and SearchByProjection(Frame, MapPoints) with the right camera's block (S/ORBmatcher.cc:44-214) */
int oracle_is_in_frustum_rig(const orbm_frame_view* view, const float* Tcw, const orbg_camera_rig* rig, const float* Tlr,
                             const orbm_worldpoints_view* pts, float viewing_cos_limit, uint8_t* in_view, float* px, float* py, float* depth,
                             int32_t* level, float* view_cos, uint8_t* in_view_r, float* px_r, float* py_r, float* depth_r,
                             int32_t* level_r, float* view_cos_r);
int oracle_search_by_projection_mps_rig(const orbm_frame_view* left, const orbm_frame_view* right, const orbm_mappoints_view* mps,
                                        const orbm_mappoints_view* mps_r, const int32_t* left_to_right, const int32_t* right_to_left,
                                        float th, int far_points, float th_far_points, float nnratio, int32_t* assigned_mp,
                                        int32_t* assigned_obs, int* nmatches);
/* SearchByProjection(CurrentFrame, LastFrame, th, bMono) with CurrentFrame.Nleft != -1 (S/ORBmatcher.cc:1970-2186 incl. :2092-2160) */
int oracle_search_by_projection_frame_rig(const orbm_frame_view* left, const orbm_frame_view* right, const float* Tcw_cur,
                                          const orbg_camera_rig* rig, const orbm_lastframe_view* last, float th, int mono,
                                          int check_orientation, int32_t* assigned_mp, int32_t* assigned_obs, int* nmatches);
/* SearchByBoW(KeyFrame*, Frame&, ...) with F.Nleft != -1 (S/ORBmatcher.cc:342-430): view = all Nleft + Nright features */
int oracle_search_by_bow_rig(const orbm_frame_view* view, int n_left, const orbm_featvec_view* fv_frame, const uint8_t* kf_desc, int nkf,
                             const uint8_t* kf_mp_valid, const float* kf_angle, const orbm_featvec_view* fv_kf, float nnratio,
                             int check_orientation, int32_t* matches, int* nmatches);
/* Frame::ComputeStereoFishEyeMatches (S/Frame.cc:1093-1150) with KannalaBrandt8::TriangulateMatches -- oracle/fisheye.cc */
int oracle_fisheye_stereo_matches(const orbx_fisheye_stereo_view* view, int32_t* left_to_right, int32_t* right_to_left, float* depth,
                                  float* points3d, int* n_matches);
void oracle_kb8_unproject(const orbg_camera* cam, float u, float v, float* ray3);
float oracle_kb8_triangulate_matches(const orbg_camera* cam1, const orbg_camera* cam2, const float* uv1, const float* uv2, const float* Tlr,
                                     float sigma1, float sigma2, float* p3D);
/* SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th, ratioHamming) with pKF->mpCamera a camera model (S/ORBmatcher.cc:515) */
int oracle_search_by_projection_sim3_cam(const orbm_frame_view* kf, const orbm_worldpoints_view* pts, const float* Scw, const orbg_camera* cam,
                                         const uint8_t* already_found, int th, float ratio_hamming, int32_t* matched, int* nmatches);
/* the relocalisation overload with CurrentFrame.mpCamera a camera model (S/ORBmatcher.cc:2217) */
int oracle_search_by_projection_reloc_cam(const orbm_frame_view* cur, const float* Tcw_cur, const orbg_camera* cam, const orbm_worldpoints_view* pts,
                                          const uint8_t* already_found, const float* kf_angle, float th, int orb_dist, int check_orientation,
                                          int32_t* assigned_mp, int* nmatches);
int oracle_search_local_points(const orbm_frame_view* view, const orbm_worldpoints_view* pts, const float* Tcw,
                               float th, int far_points, float th_far_points, float nnratio,
                               int32_t* assigned_mp, int32_t* assigned_obs, int* nmatches);
int oracle_search_by_projection_frame(const orbm_frame_view* cur, const float* Tcw_cur,
                                      const orbm_lastframe_view* last, float th, int mono, int check_orientation,
                                      int32_t* assigned_mp, int32_t* assigned_obs, int* nmatches);
int oracle_search_by_bow(const orbm_frame_view* view, const orbm_featvec_view* fv_frame,
                         const uint8_t* kf_desc, int nkf, const uint8_t* kf_mp_valid, const float* kf_angle,
                         const orbm_featvec_view* fv_kf, float nnratio, int check_orientation,
                         int32_t* matches, int* nmatches);

/* ---- server-side KeyFrame matchers */
int oracle_search_by_projection_reloc(const orbm_frame_view* cur, const float* Tcw_cur, const orbm_worldpoints_view* pts,
                                      const uint8_t* already_found, const float* kf_angle, float th, int orb_dist,
                                      int check_ori, int32_t* assigned_mp, int* nmatches_out);
int oracle_search_by_projection_sim3(const orbm_frame_view* kf, const orbm_worldpoints_view* pts, const float* Scw,
                                     const uint8_t* already_found, int th, float ratio_hamming, int camera_project,
                                     int32_t* matched, int* nmatches);
int oracle_search_by_bow_kf(const orbm_frame_view* kf2, const orbm_featvec_view* fv2, const uint8_t* mp_valid2,
                            const uint8_t* desc1, int n1, const uint8_t* mp_valid1, const float* angle1,
                            const orbm_featvec_view* fv1, float nnratio, int check_orientation,
                            int32_t* matches12, int* nmatches);

/* ---- bag of words (Thirdparty/DBoW2, S/MapPoint.cc:448-522) */
int oracle_vocab_transform(const orbv_vocab_view* v, const uint8_t* desc, int n, int levelsup, int32_t* word_id, int32_t* node_id,
                           double* weight);
int oracle_vocab_bow(const orbv_vocab_view* v, const uint8_t* desc, int n, int levelsup, int32_t* bow_word, double* bow_value,
                     int32_t* n_words, uint32_t* fv_node, uint32_t* fv_start, uint32_t* fv_feat, int32_t* n_fv_nodes);
/* the vocabulary's text format (TemplatedVocabulary.h:1338-1450): reader with the reference's stream operations, writer of
 * saveToTextFile's bytes (six significant digits per weight) */
typedef struct oracle_vocab oracle_vocab;
int oracle_vocab_load_text(const char* path, int keep_trailing_node, oracle_vocab** out);
int oracle_vocab_text_view(const oracle_vocab* t, orbv_vocab_view* view, int32_t* k, int32_t* scoring, int32_t* n_words);
int oracle_vocab_text_free(oracle_vocab* t);
int oracle_vocab_save_text(const orbv_vocab_view* v, int k, int scoring, const char* path);
int oracle_distinctive_descriptors(const uint8_t* desc, const int32_t* start, int m, int32_t* best);
int oracle_score_l1(const int32_t* q_word, const double* q_value, int nq, const int32_t* cand_start, const int32_t* cand_word,
                    const double* cand_value, int m, double* score);
int oracle_detect_n_best_candidates(const orbd_database_view* v, const int32_t* q_word, const double* q_value, int nq,
                                    const uint8_t* connected, int32_t query_map_id, int n_candidates, float* place_score,
                                    int32_t* loop_cand, int32_t* n_loop, int32_t* merge_cand, int32_t* n_merge);
int oracle_wire_pack(const orbx_keypoint* kps, const uint8_t* desc, int n, uint8_t* wire);
int oracle_wire_unpack(const uint8_t* wire, int n, orbx_keypoint* kps, uint8_t* desc);

/* ---- LBA (S/Optimizer.cc:1810-2410 + vendored g2o) */
int oracle_lba_solve(const lba_problem* p, const volatile int32_t* stop_flag, lba_result* r);
int oracle_pose_optimize(const pose_opt_problem* p, pose_opt_result* r);
/* pieces for the known-answer tests */
/* RobustKernelHuber::robustify (G/core/robust_kernel_impl.cpp:78-91): rho2 = {rho(e), rho'(e)} */
void oracle_robust_huber(double e, double delta, double* rho2);
/* GeometricCamera::project / projectJac (Eigen forms) of a pinhole / KannalaBrandt8 camera: uv2 and the 2 x 3 row-major Jacobian */
void oracle_camera_project(const orbg_camera* cam, const double* X3, double* uv2, double* J6);
/* one edge of a problem with a camera rig (see oracle_lba_edge_eval); Xcam3 (may be NULL): the point in the observing camera's frame */
void oracle_lba_edge_eval_rig(const double* q4_xyzw, const double* t3, const double* X3, const float* cam5, const orbg_camera_rig* rig,
                              const lba_edge* e, double* err3, double* Jpoint9, double* Jpose18, double* Xcam3);
void oracle_se3_exp(const double* upd6 /*omega,upsilon*/, double* q4_xyzw, double* t3);
void oracle_lba_edge_eval(const double* q4_xyzw, const double* t3, const double* X3, const float* cam5 /*fx fy cx cy bf*/,
                          const lba_edge* e, double* err3, double* Jpoint9, double* Jpose18);

#ifdef __cplusplus
}
#endif
#endif
