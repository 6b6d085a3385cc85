// libagentloop -- the Tracking thread of one client, in C++ above the C-ABI of liborbgpu (include/orbgpu.h).
//
// The reference's per-frame path is C++: ClientNode::ImageCallbackStereo -> ClientSystem::TrackStereo -> Tracking::GrabImageStereo
// (Frame constructor, S/Tracking.cc:1014-1083) -> Tracking::Track: TrackWithMotionModel (SearchByProjection(Current, Last) +
// PoseOptimization, :2592-2660) and TrackLocalMap (UpdateLocalMap, SearchLocalPoints + PoseOptimization, :2700-2735), with
// LocalMapping running Optimizer::LocalBundleAdjustment on its own thread for every keyframe (S/LocalMapping.cc:114-133,245).
// This file is that loop over flat inputs, calling ONLY the C-ABI (no kernels, no HIP calls of its own): bench.py prepares the
// handles and the per-frame views once and times agent_run(); a deployment has Tracking.cc in this place.
//
//   step(i):  frame k = seq[i % n_seq]
//     pipelined:   [submit Frame(t+1 .. t+ahead) on the other extractor handles]  wait Frame(t)   (orbx_frame_stereo_submit / _dev_submit / _wait;
//                  a monocular client: orbx_frame_mono_submit / _dev_submit)
//     synchronous: Frame(t) = orbx_frame_stereo / orbx_frame_stereo_dev
//     SearchByProjection(Current, Last)  [PoseOptimization]  SearchLocalPoints  [PoseOptimization]
//     keyframe step (i % frames_per_kf == 0): local map refresh (orbm_map_upload), wait for the previous local BA, submit the next
//
// Build: g++ -O2 -shared -fPIC agent_loop.cpp -I include -L. -lorbgpu  (multi_orbslam3_amd/csrc/build.sh)
#include <time.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/orbgpu.h"

extern "C" {

typedef struct agent_frame_in {            // what Tracking holds for one frame of the sequence
  const uint8_t* host_left; const uint8_t* host_right;       // cv::Mat data of the stereo pair
  const uint8_t* dev_left; const uint8_t* dev_right;         // the same images resident in HBM (device-image accounting)
  const float* Tcw_guess;                                    // 16 floats: motion-model prediction (mCurrentFrame.mTcw)
  const orbm_lastframe_view* last_view;                      // this frame AS the last frame of its successor (mLastFrame)
} agent_frame_in;

typedef struct agent_cfg {
  orbx_handle* ex[4]; orbm_frame* fr[4];                     // ring of (extractor handle, frame object) pairs: Frame(t+1 .. t+ahead) are built on the others
  orbm_map* local_map; lba_handle* lba;
  const orbm_frame_view* frame_view;                         // intrinsics / bounds (features come from the constructor)
  int width, height, stride; float bf, b;
  const agent_frame_in* frames; int n_frames;
  const int32_t* seq; int n_seq;                             // frame index per step (ping-pong over the distinct frames)
  const orbm_worldpoints_view* const* kf_maps; int n_kf_maps; // local map to upload at keyframe step i: kf_maps[(i / frames_per_kf) % n_kf_maps]
  const lba_problem* lba_prob; lba_result* lba_out;
  const pose_opt_problem* po[2]; pose_opt_result* po_out[2]; // the two PoseOptimization calls of a frame (NULL: skipped)
  int frames_per_kf;
  int pipelined, host_images, ingest_async, submit_first, lba_async, pose_opt;
  float th_frame; int mono; float nn_frame, nn_map;
  int32_t* amp; int32_t* aob; int cap;                       // F.mvpMapPoints as (assigned_mp, assigned_obs), cap entries each
  int32_t in_flight[4];                                      // pipelined constructor submitted on ex[c] and not yet collected (state across calls)
  int32_t ahead, ring;                                       // pipelined: frames handed over ahead of the one being tracked (1 .. ring - 1), pairs in the ring (2 .. 4;
                                                             // an even ring keeps consecutive frames on alternating extractor streams)
  int32_t lba_in_flight;                                     // a local BA submitted and not yet collected
  orbm_lastview* last_view_dev;                              // NULL: SearchByProjection(Current, Last) reads the view in place (pinned memory)
  int32_t last_view_frame;                                   // frame whose view is resident in last_view_dev (-1: none)
  int32_t* amp_after_frame;                                  // NULL, or cap entries: F.mvpMapPoints as SearchByProjection(Current, Last) left it
                                                             // (bench.py's in-job parity gate compares both searches with the oracle)
  int32_t mono_agent;                                        // 1: a monocular client -- Frame::Frame(mono) (S/Frame.cc:260-358) from host_left / dev_left
  const orbx_distortion* dist;                               // mono: mDistCoef (NULL: none), undistorted on the device by the constructor
  // host arrays the SYNCHRONOUS constructor delivers the left features into (the output arguments of orbx_frame_stereo / orbx_frame_mono;
  // the two-halves constructor delivers through orbx_set_frame_outputs); all NULL / 0: counts only
  orbx_keypoint* sync_kps; orbx_keypoint* sync_kps_un; uint8_t* sync_desc; float* sync_uright; float* sync_depth; int32_t sync_cap;
} agent_cfg;

typedef struct agent_stats {
  double stage_s[8];       // extract(wait / ctor), match_frame, match_map, pose_opt, map_upload, lba, last-frame view upload, (spare)
  double lba_s; int64_t lba_calls, lba_iters;
  int64_t kp, m_frame, m_map;
  int32_t error, error_step;
  double worst_step_s, worst_stage_s[8];   // the slowest timed step of the call and its stages (same order as stage_s)
  int64_t worst_step_index;                // ... and which step of the sequence it was (first_step + s)
  int32_t last_nl, last_nr, last_n1, last_n2;   // the LAST step of the call: features left / right, matches of the two searches
} agent_stats;

// struct sizes for the binding's layout check (multi_orbslam3_amd/agent.py)
int agent_sizeof(int which) { return which == 0 ? (int)sizeof(agent_cfg) : which == 1 ? (int)sizeof(agent_stats) : (int)sizeof(agent_frame_in); }

static inline double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static int submit_ctor(agent_cfg* c, int slot, int k) {
  const agent_frame_in& f = c->frames[k];
  if (c->mono_agent) {
    if (c->host_images)
      return orbx_frame_mono_submit(c->ex[slot], c->fr[slot], c->frame_view, c->dist, f.host_left, c->width, c->height, c->stride,
                                    c->ingest_async ? ORBX_SUBMIT_ASYNC : 0);
    return orbx_frame_mono_dev_submit(c->ex[slot], c->fr[slot], c->frame_view, c->dist, f.dev_left, c->width, c->height, c->stride);
  }
  if (c->host_images)
    return orbx_frame_stereo_submit(c->ex[slot], c->fr[slot], c->frame_view, f.host_left, f.host_right, c->width, c->height, c->stride, c->bf, c->b,
                                    c->ingest_async ? ORBX_SUBMIT_ASYNC : 0);
  return orbx_frame_stereo_dev_submit(c->ex[slot], c->fr[slot], c->frame_view, f.dev_left, f.dev_right, c->width, c->height, c->stride, c->bf, c->b);
}

// Collects the local BA in flight (if any); its wall time and iteration counts go to the statistics when `timed`.
static int collect_lba(agent_cfg* c, agent_stats* st, int timed) {
  if (!c->lba_in_flight) return ORBG_OK;
  double ms = 0;
  const int rc = lba_wait(c->lba, &ms);
  c->lba_in_flight = 0;
  if (rc) return rc;
  if (timed && st) { st->lba_s += 1e-3 * ms; st->lba_calls++; st->lba_iters += c->lba_out->iters_round1 + c->lba_out->iters_round2; }
  return ORBG_OK;
}

// Runs steps first_step .. first_step + n_steps - 1.  last_is_final: the last step hands no further frame over (a timed region
// holds exactly n_steps constructors).  step_s (n_steps doubles, may be NULL) receives the wall time of every step.  State that
// lives across calls (constructors / local BA in flight) is kept in cfg.
int agent_run(agent_cfg* c, int64_t first_step, int n_steps, int last_is_final, int timed, double* step_s, agent_stats* st) {
  if (!c || !st || n_steps < 0) return ORBG_BAD_ARG;
  int rc = ORBG_OK, s_done = 0;
  for (int s = 0; s < n_steps && rc == ORBG_OK; s++) {
    s_done = s;
    const int64_t i = first_step + s;
    const int k = c->seq[i % c->n_seq], k_last = c->seq[(i + c->n_seq - 1) % c->n_seq];
    const agent_frame_in& fin = c->frames[k];
    const double t0 = now_s();
    int nl = 0, nr = 0, cur = 0;
    if (c->pipelined) {
      // ring of (handle, frame) pairs: frame t lives in pair t % ring.  Frame(t+1 .. t+ahead) are handed over (their
      // staging copies and launches overlap frame t's constructor tail and tracking) either before or after frame t is collected;
      // a final region hands nothing over beyond its last step, so it holds exactly n_steps constructors
      const int ring = c->ring < 2 ? 2 : c->ring > 4 ? 4 : c->ring, ahead = c->ahead < 1 ? 1 : c->ahead > ring - 1 ? ring - 1 : c->ahead;
      cur = (int)(i % ring);
      auto submit_ahead = [&](int d0, int d1) -> int {
        for (int d = d0; d <= d1; d++) {
          const int slot = (int)((i + d) % ring);
          if (c->in_flight[slot] || (last_is_final && s + d >= n_steps)) continue;
          const int r = submit_ctor(c, slot, c->seq[(i + d) % c->n_seq]);
          if (r) return r;
          c->in_flight[slot] = 1;
        }
        return ORBG_OK;
      };
      if ((rc = submit_ahead(0, 0))) break;                    // the first step of a sequence only (nothing was handed over ahead)
      if (!c->submit_first || (rc = submit_ahead(1, ahead)) == ORBG_OK) rc = orbx_frame_stereo_dev_wait(c->ex[cur], &nl, &nr);
      if (rc) break;
      c->in_flight[cur] = 0;
      if (!c->submit_first && (rc = submit_ahead(1, ahead))) break;
    } else if (c->mono_agent) {
      rc = c->host_images ? orbx_frame_mono(c->ex[0], c->fr[0], c->frame_view, c->dist, fin.host_left, c->width, c->height, c->stride, c->sync_kps,
                                            c->sync_kps_un, c->sync_desc, c->sync_cap, &nl)
                          : orbx_frame_mono_dev(c->ex[0], c->fr[0], c->frame_view, c->dist, fin.dev_left, c->width, c->height, c->stride, c->sync_kps,
                                                c->sync_kps_un, c->sync_desc, c->sync_cap, &nl);
    } else if (c->host_images) {
      rc = orbx_frame_stereo(c->ex[0], c->fr[0], c->frame_view, fin.host_left, fin.host_right, c->width, c->height, c->stride, c->bf, c->b,
                             c->sync_kps, c->sync_desc, c->sync_uright, c->sync_depth, c->sync_cap, &nl, &nr);
    } else {
      rc = orbx_frame_stereo_dev(c->ex[0], c->fr[0], c->frame_view, fin.dev_left, fin.dev_right, c->width, c->height, c->stride, c->bf, c->b,
                                 c->sync_kps, c->sync_desc, c->sync_uright, c->sync_depth, c->sync_cap, &nl, &nr);
    }
    if (rc) break;
    if (nl > c->cap) { rc = ORBG_CAP_EXCEEDED; break; }
    orbm_frame* F = c->fr[cur];
    const double t1 = now_s();
    // F.mvpMapPoints starts empty (S/Frame.cc:113)
    for (int j = 0; j < nl; j++) c->amp[j] = -1;
    memset(c->aob, 0, sizeof(int32_t) * (size_t)nl);
    int n1 = 0, n2 = 0;
    if (c->last_view_dev) {
      // mLastFrame's view is resident: it went up when the tracking of that frame ended (below); only the first step after a change
      // of the sequence position uploads here
      if (c->last_view_frame != k_last) {
        if ((rc = orbm_lastview_upload(c->last_view_dev, c->frames[k_last].last_view))) break;
        c->last_view_frame = k_last;
      }
      if ((rc = orbm_search_by_projection_frame_resident(F, fin.Tcw_guess, c->last_view_dev, c->th_frame, c->mono, 1, c->amp, c->aob, &n1))) break;
    } else if ((rc = orbm_search_by_projection_frame(F, fin.Tcw_guess, c->frames[k_last].last_view, c->th_frame, c->mono, 1, c->amp, c->aob, &n1))) break;
    if (c->amp_after_frame) memcpy(c->amp_after_frame, c->amp, sizeof(int32_t) * (size_t)nl);
    const double t2 = now_s();
    double t_po = 0;
    if (c->pose_opt && c->po[0]) {                                 // TrackWithMotionModel: Optimizer::PoseOptimization(&mCurrentFrame), :2649
      const double a = now_s();
      if ((rc = pose_optimize(c->po[0], c->po_out[0]))) break;
      t_po += now_s() - a;
    }
    const double t2b = now_s();
    if ((rc = orbm_search_local_points(F, c->local_map, fin.Tcw_guess, nullptr, 1.0f, 0, 0.0f, c->nn_map, c->amp, c->aob, &n2))) break;
    const double t3 = now_s();
    if (c->pose_opt && c->po[1]) {                                 // TrackLocalMap: Optimizer::PoseOptimization(&mCurrentFrame), :2712
      const double a = now_s();
      if ((rc = pose_optimize(c->po[1], c->po_out[1]))) break;
      t_po += now_s() - a;
    }
    double t_lv = 0;
    if (c->last_view_dev) {
      // end of Track(): mLastFrame = Frame(mCurrentFrame) (S/Tracking.cc:2086-2090) -- this frame's map points are known now, the next
      // frame's SearchByProjection(Current, Last) reads them a frame time later: the view goes to the device while nothing else crosses PCIe
      const double a = now_s();
      if ((rc = orbm_lastview_upload(c->last_view_dev, fin.last_view))) break;
      c->last_view_frame = k;
      t_lv = now_s() - a;
    }
    const double t4 = now_s();
    double t5 = t4, t6 = t4;
    if (i % c->frames_per_kf == 0) {
      // keyframe: Tracking::UpdateLocalMap (the points of the last 20 / 50 keyframes), LocalMapping gets a new keyframe
      if ((rc = orbm_map_upload(c->local_map, c->kf_maps[(i / c->frames_per_kf) % c->n_kf_maps]))) break;
      t5 = now_s();
      if (c->lba_async) {
        if ((rc = collect_lba(c, st, timed))) break;               // the previous keyframe's local BA (long finished in steady state)
        if ((rc = lba_solve_async(c->lba, c->lba_prob, nullptr, c->lba_out))) break;
        c->lba_in_flight = 1;
      } else {
        const double a = now_s();
        if ((rc = lba_solve_h(c->lba, c->lba_prob, nullptr, c->lba_out))) break;
        if (timed) { st->lba_s += now_s() - a; st->lba_calls++; st->lba_iters += c->lba_out->iters_round1 + c->lba_out->iters_round2; }
      }
      t6 = now_s();
    }
    st->last_nl = nl; st->last_nr = nr; st->last_n1 = n1; st->last_n2 = n2;
    if (timed) {
      st->stage_s[0] += t1 - t0; st->stage_s[1] += t2 - t1; st->stage_s[2] += t3 - t2b; st->stage_s[3] += t_po;
      st->stage_s[4] += t5 - t4; st->stage_s[5] += t6 - t5; st->stage_s[6] += t_lv;
      st->kp += nl + nr; st->m_frame += n1; st->m_map += n2;
      const double t_step = now_s() - t0;
      if (step_s) step_s[s] = t_step;
      if (t_step > st->worst_step_s) {
        st->worst_step_s = t_step;
        st->worst_step_index = i;
        const double w[8] = {t1 - t0, t2 - t1, t3 - t2b, t_po, t5 - t4, t6 - t5, t_lv, 0.0};
        for (int q = 0; q < 8; q++) st->worst_stage_s[q] = w[q];
      }
    }
  }
  st->error = rc;
  if (rc) st->error_step = (int32_t)s_done;
  return rc;
}

// End of a region: every local BA triggered in it has finished and every constructor handed over has been collected.
int agent_drain(agent_cfg* c, agent_stats* st, int timed) {
  if (!c) return ORBG_BAD_ARG;
  int rc = collect_lba(c, st, timed);
  for (int s = 0; s < 4; s++)
    if (c->in_flight[s]) {
      int nl, nr;
      const int r2 = orbx_frame_stereo_dev_wait(c->ex[s], &nl, &nr);
      c->in_flight[s] = 0;
      if (!rc) rc = r2;
    }
  return rc;
}

}  // extern "C"
