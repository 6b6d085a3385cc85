"""Wall time of the matcher's two-camera forms (Frame::Nleft != -1: isInFrustum through either camera, SearchByProjection(Frame, MapPoints)
with the right camera's block, SearchByProjection(CurrentFrame, LastFrame)) next to the CPU oracle on the same scene:
python tools/rig_match_time.py  (on a GPU box)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from multi_orbslam3_amd import api, synth, views  # noqa: E402
from oracle import binding as ob  # noqa: E402
import helpers  # noqa: E402


def timed(fn, reps):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts))


def main():
    for n_points, n_distract in ((1500, 300), (4000, 600)):
        sc = synth.make_rig_track_scene(n_points=n_points, n_distract=n_distract)
        fl, fr, wv, rig, keep = helpers.rig_track_views(sc)
        FL, FR = api.Frame().upload(fl, keep[0]), api.Frame().upload(fr, keep[1])
        a, b = ob.is_in_frustum_rig(fl, sc["Tcw"], rig, sc["Tlr"], wv)
        mv, mvr, keep2 = helpers.rig_mappoint_views(sc, a, b)
        m = api.ORBmatcher(0.8, True)
        last = synth.rig_last_frame(sc, n_last=min(1000, n_points))
        lv, keep3 = views.lastframe_view(last["mp_valid"], last["outlier"], last["world_pos"], last["desc"], last["octave"], last["angle"], last["n_obs"], last["Tcw"])
        args = (sc["left_to_right"], sc["right_to_left"], 3.0, True, 6.0, sc["assigned_mp"], sc["assigned_obs"])
        g1 = timed(lambda: FL.isInFrustumRig(sc["Tcw"], rig, sc["Tlr"], wv), 50)
        o1 = timed(lambda: ob.is_in_frustum_rig(fl, sc["Tcw"], rig, sc["Tlr"], wv), 10)
        g2 = timed(lambda: m.SearchByProjectionRig(FL, FR, mv, mvr, *args), 50)
        o2 = timed(lambda: ob.search_by_projection_mps_rig(fl, fr, mv, mvr, sc["left_to_right"], sc["right_to_left"], 3.0, True, 6.0, 0.8, sc["assigned_mp"], sc["assigned_obs"]), 10)
        g3 = timed(lambda: m.SearchByProjectionFrameRig(FL, FR, sc["Tcw"], rig, lv, 7.0, False, sc["assigned_mp"], sc["assigned_obs"]), 50)
        o3 = timed(lambda: ob.search_by_projection_frame_rig(fl, fr, sc["Tcw"], rig, lv, 7.0, 0, 1, sc["assigned_mp"], sc["assigned_obs"]), 10)
        # the usual local map: every point observed -- no stereo-partner write can free a feature, one pass
        sc1 = synth.make_rig_track_scene(n_points=n_points, n_distract=n_distract, zero_obs_frac=0.0)
        sc1["assigned_obs"][:] = np.maximum(sc1["assigned_obs"], 1)
        fl1, fr1, wv1, rig1, keep1 = helpers.rig_track_views(sc1)
        FL1, FR1 = api.Frame().upload(fl1, keep1[0]), api.Frame().upload(fr1, keep1[1])
        a1, b1 = ob.is_in_frustum_rig(fl1, sc1["Tcw"], rig1, sc1["Tlr"], wv1)
        mv1, mvr1, keep4 = helpers.rig_mappoint_views(sc1, a1, b1)
        g2b = timed(lambda: m.SearchByProjectionRig(FL1, FR1, mv1, mvr1, sc1["left_to_right"], sc1["right_to_left"], 3.0, True, 6.0, sc1["assigned_mp"], sc1["assigned_obs"]), 50)
        o2b = timed(lambda: ob.search_by_projection_mps_rig(fl1, fr1, mv1, mvr1, sc1["left_to_right"], sc1["right_to_left"], 3.0, True, 6.0, 0.8, sc1["assigned_mp"], sc1["assigned_obs"]), 10)
        # SearchByBoW(KeyFrame, Frame): all features in one frame object, 900 keyframe features, 32 words
        import test_gpu_parity as tg
        fva, keepa, fvF, fvK, kf_desc, kf_angle, valid, keepb = tg._rig_bow_scene(sc, shift=3)
        FA = api.Frame().upload(fva, keepa)
        mb = api.ORBmatcher(0.7, True)
        g4 = timed(lambda: mb.SearchByBoWRig(FA, len(sc["kps_left"]), fvF, kf_desc, valid, kf_angle, fvK), 50)
        o4 = timed(lambda: ob.search_by_bow_rig(fva, len(sc["kps_left"]), fvF, kf_desc, valid, kf_angle, fvK, 0.7, True), 10)
        print(f"two-camera frame, {len(sc['kps_left'])} + {len(sc['kps_right'])} features, {n_points} map points:")
        print(f"  isInFrustum (both cameras)                 {g1:7.1f} us   oracle {o1:8.1f} us")
        print(f"  SearchByProjection(Frame, MapPoints)       {g2:7.1f} us   oracle {o2:8.1f} us")
        print(f"  SearchByProjection(CurrentFrame, LastFrame) {g3:6.1f} us   oracle {o3:8.1f} us   ({len(last['mp_valid'])} last-frame entries)")
        print(f"  SearchByProjection(Frame, MapPoints), every point observed {g2b:6.1f} us   oracle {o2b:8.1f} us")
        print(f"  SearchByBoW(KeyFrame, Frame), 900 keyframe features {g4:6.1f} us   oracle {o4:8.1f} us")


def fisheye_ctor():
    """Frame::ComputeStereoFishEyeMatches (the left-right matcher of the two-fisheye Frame constructor)."""
    for kw in (dict(), dict(n_stereo=1500, n_mono_left=600, n_mono_right=550, n_distract=300)):
        sc = synth.make_fisheye_stereo_scene(**kw)
        v, keep = views.fisheye_stereo_view(sc["kps_left"], sc["desc_left"], sc["mono_left"], sc["kps_right"], sc["desc_right"], sc["mono_right"], sc["left"],
                                            sc["right"], sc["Tlr"], sc["level_sigma2"])
        g = timed(lambda: api.ComputeStereoFishEyeMatches(v), 50)
        o = timed(lambda: ob.fisheye_stereo_matches(v), 5)
        print(f"ComputeStereoFishEyeMatches, {v.n_left} + {v.n_right} features ({v.n_left - v.mono_left} x {v.n_right - v.mono_right} in the lapping areas): "
              f"{g:.1f} us   oracle {o:.1f} us   ({api.ComputeStereoFishEyeMatches(v)[4]} matches)")


if __name__ == "__main__":
    main()
    fisheye_ctor()
