#!/usr/bin/env python3
"""Generate the committed golden fixtures (small, seeded) with the CPU oracle.

The reference ships no golden vectors and cannot be built here, so these fixtures are produced
by this repo's oracle; they freeze its outputs so that (a) an accidental change of the oracle is caught on CPU and
(b) the HIP path is checked against committed data on the GPU box.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from multi_orbslam3_amd import synth, views  # noqa: E402
from oracle import binding as ob  # noqa: E402
import helpers  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    sc = synth.Scene(160, 120, tex_size=(400, 300), px_per_m=50.0)
    L, R, Tcw = sc.stereo_pair(0)
    exL = ob.Extractor(n_features=300, n_levels=4, max_width=160, max_height=120)
    exR = ob.Extractor(n_features=300, n_levels=4, max_width=160, max_height=120)
    rc, kl, dl, _ = exL.extract(L)
    rc, kr, dr, _ = exR.extract(R)
    ur, dp = ob.stereo_match(exL, exR, kl, dl, kr, dr, float(sc.cam["bf"]), float(sc.cam["b"]))
    cands = [exL.candidates(l) for l in range(4)]
    np.savez_compressed(os.path.join(OUT, "extract_160x120.npz"), L=L, R=R, kps_l=kl, desc_l=dl, kps_r=kr, desc_r=dr,
                        uright=ur, depth=dp, level1=exL.level(1), level3=exL.level(3),
                        cand0=cands[0], cand1=cands[1], cand2=cands[2], cand3=cands[3])
    rng = np.random.RandomState(42)
    q = rng.randint(0, 256, (24, 32)).astype(np.uint8)
    t = rng.randint(0, 256, (40, 32)).astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "hamming_24x40.npz"), q=q, t=t, dist=ob.hamming_matrix(q, t), best2=ob.hamming_best2(q, t))
    prob = synth.make_lba_problem(n_free=5, n_fixed=2, n_points=60, width=320, height=240, mono_frac=0.15, seed=99)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    o = ob.lba_solve(p)
    np.savez_compressed(os.path.join(OUT, "lba_5p2_60.npz"), poses=prob["poses"], pose_fixed=prob["pose_fixed"],
                        points=prob["points"], edges=prob["edges"], cam=np.array(prob["cam"], np.float32),
                        out_poses=o.poses, out_points=o.points, out_outlier=o.edge_outlier, out_chi2=o.edge_chi2,
                        trace=o.trace_rows(), iters=np.array(o.iters), status=np.array([o.status]))
    print("kps", len(kl), len(kr), "stereo", int((ur > 0).sum()), "lba iters", o.iters, "edges", p.n_edges)
    make_matching_pose_bow(sc)


def make_matching_pose_bow(sc):
    """Matchers (frame / KeyFrame variants), PoseOptimization, bag-of-words pieces, KeyFrame wire block: inputs + oracle outputs."""
    rng = np.random.RandomState(7)
    f0, f1, kf = [helpers.oracle_stereo_frame(sc, k, 300) for k in (0, 2, 1)]
    mp = helpers.local_map_from(sc, [f0], rng)
    n = len(kf["kps"])
    T = synth.perturb_pose(kf["Tcw"], rng).astype(np.float32)
    fvp = sc.frame_view_params()
    fv, keep = helpers.frame_view_of(sc, kf)
    wv, keep2 = helpers.world_view_of(mp)
    amp0 = np.full(n, -1, np.int32); aob0 = np.zeros(n, np.int32)
    loc = ob.search_local_points(fv, wv, T, 3.0, False, 0.0, 0.8, amp0, aob0)
    S = T.copy(); S[:3, :] *= np.float32(1.3)
    sim = ob.search_by_projection_sim3(fv, wv, S, amp0, 5, 1.5)
    node = lambda d: (d[:, 0].astype(np.int64) >> 4)
    n1, s1, i1 = views.featvec_from_nodes(node(f1["desc"])); n2, s2, i2 = views.featvec_from_nodes(node(kf["desc"]))
    fv1, k1 = views.featvec_view(n1, s1, i1); fv2, k2 = views.featvec_view(n2, s2, i2)
    v1 = (f1["depth"] > 0).astype(np.uint8); v2 = (kf["depth"] > 0).astype(np.uint8)
    bow = ob.search_by_bow_kf(fv, fv2, v2, f1["desc"], v1, f1["kps"]["angle"], fv1, 0.8, True)
    pr = synth.make_pose_opt_problem(n=200, outlier_frac=0.1, mono_frac=0.25, seed=21)
    pp, keep3 = views.pose_opt_problem(pr["Xw"], pr["u"], pr["v"], pr["ur"], pr["inv_sigma2"], pr["cam"], pr["Tcw"])
    po = ob.pose_optimize(pp)
    voc = synth.make_vocabulary(k=6, L=3, seed=3, descriptors=kf["desc"])
    vv, keep4 = views.vocab_view(voc["child_start"], voc["child_ids"], voc["desc"], voc["weight"], voc["word_id"], 3)
    (bw, bv), (fn, fs, ff) = ob.vocab_bow(vv, kf["desc"], 1)
    dd_start = np.array([0, 1, 4, 4, 11, 30], np.int32)
    dd = ob.distinctive_descriptors(kf["desc"][:30], dd_start)
    wire = ob.wire_pack(kf["kps"], kf["desc"])
    np.savez_compressed(os.path.join(OUT, "matching_pose_bow.npz"),
                        kf_kps=kf["kps"], kf_desc=kf["desc"], kf_uright=kf["uright"], kf_depth=kf["depth"],
                        bounds=np.array(fvp["bounds"], np.float32), cam=np.array(fvp["cam"], np.float32), T=T, S=S,
                        mp_pos=mp["pos"], mp_normal=mp["normal"], mp_min=mp["min_dist"], mp_max=mp["max_dist"], mp_desc=mp["desc"],
                        mp_nobs=mp["n_obs"], mp_bad=mp["bad"],
                        loc_amp=loc[0], loc_aob=loc[1], loc_n=np.array([loc[2]]), sim_matched=sim[0], sim_n=np.array([sim[1]]),
                        f1_desc=f1["desc"], f1_angle=f1["kps"]["angle"], v1=v1, v2=v2, n1=n1, s1=s1, i1=i1, n2=n2, s2=s2, i2=i2,
                        bow_m12=bow[0], bow_n=np.array([bow[1]]),
                        po_Xw=pr["Xw"], po_u=pr["u"], po_v=pr["v"], po_ur=pr["ur"], po_w=pr["inv_sigma2"], po_cam=np.array(pr["cam"], np.float32),
                        po_T0=pr["Tcw"], po_T=po.Tcw, po_out=po.outliers, po_iters=np.array(po.iters), po_inl=np.array([po.n_inliers]),
                        voc_cs=voc["child_start"], voc_ci=voc["child_ids"], voc_desc=voc["desc"], voc_w=voc["weight"], voc_word=voc["word_id"],
                        bow_word=bw, bow_value=bv, fv_node=fn, fv_start=fs, fv_feat=ff, dd_start=dd_start, dd_best=dd, wire=wire)
    print("local", loc[2], "sim3", sim[1], "bow_kf", bow[1], "pose_opt", po.iters, po.n_inliers, "words", len(bw))


def make_rig_and_fisheye():
    """Two-camera frames (Frame::Nleft != -1) and the fisheye constructor's stereo matcher: the scene's INPUTS and the oracle's outputs."""
    sc = synth.make_rig_track_scene(n_points=350, n_distract=60, seed=0x601D)
    fl, fr, wv, rig, keep = helpers.rig_track_views(sc)
    a, b = ob.is_in_frustum_rig(fl, sc["Tcw"], rig, sc["Tlr"], wv)
    mv, mvr, keep2 = helpers.rig_mappoint_views(sc, a, b)
    srch = ob.search_by_projection_mps_rig(fl, fr, mv, mvr, sc["left_to_right"], sc["right_to_left"], 3.0, True, 6.0, 0.8, sc["assigned_mp"], sc["assigned_obs"])
    last = synth.rig_last_frame(sc, n_last=300, seed=3, motion=(0.03, 0.01, 0.02))
    lv, keep3 = views.lastframe_view(last["mp_valid"], last["outlier"], last["world_pos"], last["desc"], last["octave"], last["angle"], last["n_obs"], last["Tcw"])
    frm = ob.search_by_projection_frame_rig(fl, fr, sc["Tcw"], rig, lv, 7.0, 0, 1, sc["assigned_mp"], sc["assigned_obs"])
    fs = synth.make_fisheye_stereo_scene(n_stereo=220, n_mono_left=60, n_mono_right=50, n_distract=40, seed=0x601E)
    v, keep4 = views.fisheye_stereo_view(fs["kps_left"], fs["desc_left"], fs["mono_left"], fs["kps_right"], fs["desc_right"], fs["mono_right"], fs["left"], fs["right"],
                                         fs["Tlr"], fs["level_sigma2"])
    st = ob.fisheye_stereo_matches(v)
    # the two optimisers on a rig: KannalaBrandt8 models, the right camera's ToBody edges (S/Optimizer.cc:1085-1151, 2021-2120)
    pr = synth.make_lba_rig_problem(n_free=4, n_fixed=2, n_points=90, seed=0x601F)
    p, keepp = views.lba_problem(pr["poses"], pr["pose_fixed"], pr["points"], pr["edges"], pr["cam"], rig=views.camera_rig(*pr["rig"]))
    lo = ob.lba_solve(p)
    po = synth.make_pose_opt_rig_problem(n_left=120, n_right=80, seed=0x6020)
    q, keepq = views.pose_opt_problem(po["Xw"], po["u"], po["v"], po["ur"], po["inv_sigma2"], po["cam"], po["Tcw"], rig=views.camera_rig(*po["rig"]))
    oo = ob.pose_optimize(q)
    opt = dict(lba_poses=pr["poses"], lba_pose_fixed=pr["pose_fixed"], lba_points=pr["points"], lba_edges=pr["edges"], lba_cam=np.float32(pr["cam"]), lba_Trl=np.float32(pr["rig"][2]),
               lba_out_poses=lo.poses, lba_out_points=lo.points, lba_out_outlier=lo.edge_outlier, lba_iters=np.array(lo.iters), lba_status=np.array([lo.status]),
               po_Xw=po["Xw"], po_u=po["u"], po_v=po["v"], po_ur=po["ur"], po_w=po["inv_sigma2"], po_cam=np.float32(po["cam"]), po_Tcw=po["Tcw"],
               po_out_T=oo.Tcw, po_out_outliers=oo.outliers, po_out_inliers=np.array([oo.n_inliers]), po_out_iters=np.array(oo.iters))
    scene = {"sc_" + k: np.asarray(val) for k, val in sc.items() if k not in ("left", "right", "size")}
    scene.update({"last_" + k: np.asarray(val) for k, val in last.items()})
    scene.update({"fs_" + k: np.asarray(val) for k, val in fs.items() if k not in ("left", "right")})
    np.savez_compressed(os.path.join(OUT, "rig_and_fisheye.npz"), cam_left=np.float64(sc["left"]), cam_right=np.float64(sc["right"]), size=np.float64([sc["size"]]),
                        **scene, **{"fl_" + k: a[k] for k in ob.RIG_TRACK_KEYS}, **{"fr_" + k: b[k] for k in ob.RIG_TRACK_KEYS},
                        srch_amp=srch[0], srch_aob=srch[1], srch_n=np.int32([srch[2]]), frm_amp=frm[0], frm_aob=frm[1], frm_n=np.int32([frm[2]]),
                        st_l2r=st[0], st_r2l=st[1], st_depth=st[2], st_p3d=st[3], st_n=np.int32([st[4]]), **opt)


def make_appendix_f():
    """SURVEY.md Appendix F's larger fixtures: the 96x72 and 640x480 extractions (per-level candidates, kept keypoints, angles,
    descriptors, the constructor's tables) and the C2-sized local BA (20 free + 10 fixed keyframes, 2000 points) with its
    per-iteration (lambda, chi2, trials) trace."""
    # (Appendix F names 64 x 48; that size has NO FAST cell -- the detection window [16, h - 16) is 16 rows, nRows = 16 / 30 = 0, and
    # the reference divides by it, S/ORBextractor.cc:779-782 -- so the smallest fixture is a 96 x 72 crop of the 640 x 480 frame)
    sc = synth.Scene(640, 480)
    L, R, Tcw = sc.stereo_pair(3)
    crop = np.ascontiguousarray(L[200:272, 300:396])
    ex = ob.Extractor(n_features=100, n_levels=3, max_width=96, max_height=72)
    rc, kl, dl, nm = ex.extract(crop)
    np.savez_compressed(os.path.join(OUT, "extract_96x72.npz"), L=crop, kps=kl, desc=dl, cand0=ex.candidates(0), cand1=ex.candidates(1),
                        cand2=ex.candidates(2), level2=ex.level(2, border=True))
    print("96x72: kps", len(kl))
    sc = synth.Scene(640, 480)
    L, R, Tcw = sc.stereo_pair(3)
    ex = ob.Extractor(n_features=1000, max_width=640, max_height=480)
    rc, kl, dl, nm = ex.extract(L)
    cands = {"cand%d" % l: ex.candidates(l) for l in range(8)}
    tabs = ex.tables()
    rc, km, dm, nmono = ex.extract(L, (0, 1000))                # the monocular Frame constructor's lapping area (S/Frame.cc:289)
    np.savez_compressed(os.path.join(OUT, "extract_640x480.npz"), L=L, kps=kl, desc=dl, kps_mono_order=km, desc_mono_order=dm,
                        n_mono=np.array([nm, nmono]), level7=ex.level(7, border=True), scale=tabs[0], inv_scale=tabs[1], sigma2=tabs[2],
                        inv_sigma2=tabs[3], features_per_level=tabs[4], **cands)
    print("640x480: kps", len(kl), "candidates", [len(cands["cand%d" % l]) for l in range(8)], "monoIndex", nm, nmono)
    prob = synth.make_lba_problem(n_free=20, n_fixed=10, n_points=2000, width=640, height=480, seed=synth.SEED_LBA)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    o = ob.lba_solve(p)
    np.savez_compressed(os.path.join(OUT, "lba_20p10_2000.npz"), poses=prob["poses"], pose_fixed=prob["pose_fixed"], points=prob["points"],
                        edges=prob["edges"], cam=np.array(prob["cam"], np.float32), out_poses=o.poses, out_points=o.points,
                        out_outlier=o.edge_outlier, out_depth_pos=o.edge_depth_pos, out_chi2=o.edge_chi2, trace=o.trace_rows(),
                        iters=np.array(o.iters), status=np.array([o.status]), chi2=np.array(o.chi2), n_outliers=np.array([o.n_outliers]))
    print("lba 20+10/2000: edges", p.n_edges, "iters", o.iters, "trace rows", len(o.trace_rows()), "chi2", o.chi2, "outliers", o.n_outliers)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "appendix_f":
        make_appendix_f()
    elif len(sys.argv) > 1 and sys.argv[1] == "rig":
        make_rig_and_fisheye()
    else:
        main()
        make_appendix_f()
        make_rig_and_fisheye()
