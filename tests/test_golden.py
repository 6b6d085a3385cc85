"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py with the oracle):
the oracle must keep reproducing them on CPU; the HIP path must reproduce them on the GPU box."""
import os

import numpy as np
import pytest

from multi_orbslam3_amd import synth, views
from oracle import binding as ob
import helpers

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def test_oracle_reproduces_golden_extraction():
    g = _load("extract_160x120.npz")
    sc = synth.Scene(160, 120, tex_size=(400, 300), px_per_m=50.0)
    exL = ob.Extractor(n_features=300, n_levels=4, max_width=160, max_height=120)
    exR = ob.Extractor(n_features=300, n_levels=4, max_width=160, max_height=120)
    rc, kl, dl, _ = exL.extract(g["L"])
    rc, kr, dr, _ = exR.extract(g["R"])
    assert np.array_equal(kl, g["kps_l"]) and np.array_equal(dl, g["desc_l"])
    assert np.array_equal(kr, g["kps_r"]) and np.array_equal(dr, g["desc_r"])
    assert np.array_equal(exL.level(1), g["level1"]) and np.array_equal(exL.level(3), g["level3"])
    for l in range(4):
        assert np.array_equal(exL.candidates(l), g["cand%d" % l])
    ur, dp = ob.stereo_match(exL, exR, kl, dl, kr, dr, float(sc.cam["bf"]), float(sc.cam["b"]))
    assert np.array_equal(ur, g["uright"]) and np.array_equal(dp, g["depth"])
    assert len(kl) > 100 and (ur > 0).sum() > 20


def test_oracle_reproduces_golden_hamming_and_lba():
    g = _load("hamming_24x40.npz")
    assert np.array_equal(ob.hamming_matrix(g["q"], g["t"]), g["dist"])
    assert np.array_equal(ob.hamming_best2(g["q"], g["t"]), g["best2"])
    assert np.array_equal(g["dist"], np.unpackbits(g["q"][:, None, :] ^ g["t"][None, :, :], axis=2).sum(axis=2))
    b = _load("lba_5p2_60.npz")
    p, keep = views.lba_problem(b["poses"], b["pose_fixed"], b["points"], b["edges"], tuple(b["cam"]))
    o = ob.lba_solve(p)
    assert o.status == int(b["status"][0]) and tuple(o.iters) == tuple(b["iters"])
    assert np.allclose(o.poses, b["out_poses"], atol=1e-6) and np.allclose(o.points, b["out_points"], atol=1e-6)
    assert np.array_equal(o.edge_outlier, b["out_outlier"])


@pytest.mark.gpu
def test_gpu_reproduces_golden():
    from multi_orbslam3_amd import api
    g = _load("extract_160x120.npz")
    ex = api.ORBextractor(300, 1.2, 4, 20, 7, 160, 120, n_cams=2)
    (kl, dl), (kr, dr) = ex.extract_stereo(g["L"], g["R"])
    assert np.array_equal(kl, g["kps_l"]) and np.array_equal(dl, g["desc_l"])
    assert np.array_equal(kr, g["kps_r"]) and np.array_equal(dr, g["desc_r"])
    assert np.array_equal(ex.level(0, 1), g["level1"]) and np.array_equal(ex.level(0, 3), g["level3"])
    for l in range(4):
        assert np.array_equal(ex.candidates(0, l), g["cand%d" % l])
    sc = synth.Scene(160, 120, tex_size=(400, 300), px_per_m=50.0)
    ur, dp = ex.ComputeStereoMatches(float(sc.cam["bf"]), float(sc.cam["b"]), n_left=len(kl))
    assert np.array_equal(ur, g["uright"]) and np.array_equal(dp, g["depth"])
    h = _load("hamming_24x40.npz")
    m = api.ORBmatcher()
    assert np.array_equal(m.DescriptorDistance(h["q"], h["t"]), h["dist"]) and np.array_equal(m.best2(h["q"], h["t"]), h["best2"])
    b = _load("lba_5p2_60.npz")
    p, keep = views.lba_problem(b["poses"], b["pose_fixed"], b["points"], b["edges"], tuple(b["cam"]))
    o = api.Optimizer().LocalBundleAdjustment(p)
    assert o.status == int(b["status"][0]) and tuple(o.iters) == tuple(b["iters"])
    assert np.abs(o.poses - b["out_poses"]).max() <= 1e-4 and np.abs(o.points - b["out_points"]).max() <= 1e-4
    assert np.array_equal(o.edge_outlier, b["out_outlier"])


def _fixture_views(g):
    from multi_orbslam3_amd import views as V
    fv, k1 = V.frame_view(g["kf_kps"], g["kf_desc"], g["kf_uright"], g["kf_depth"], tuple(g["bounds"]), tuple(g["cam"]), 8, 1.2)
    wv, k2 = V.worldpoints_view(g["mp_pos"], g["mp_normal"], g["mp_min"], g["mp_max"], g["mp_desc"], g["mp_nobs"], g["mp_bad"])
    fv1, k3 = V.featvec_view(g["n1"], g["s1"], g["i1"]); fv2, k4 = V.featvec_view(g["n2"], g["s2"], g["i2"])
    pp, k5 = V.pose_opt_problem(g["po_Xw"], g["po_u"], g["po_v"], g["po_ur"], g["po_w"], tuple(g["po_cam"]), g["po_T0"])
    vv, k6 = V.vocab_view(g["voc_cs"], g["voc_ci"], g["voc_desc"], g["voc_w"], g["voc_word"], 3)
    return fv, wv, fv1, fv2, pp, vv, (k1, k2, k3, k4, k5, k6)


def test_oracle_reproduces_golden_matching_pose_bow():
    g = _load("matching_pose_bow.npz")
    fv, wv, fv1, fv2, pp, vv, keep = _fixture_views(g)
    n = len(g["kf_kps"])
    amp0 = np.full(n, -1, np.int32); aob0 = np.zeros(n, np.int32)
    loc = ob.search_local_points(fv, wv, g["T"], 3.0, False, 0.0, 0.8, amp0, aob0)
    assert loc[2] == int(g["loc_n"][0]) and np.array_equal(loc[0], g["loc_amp"]) and np.array_equal(loc[1], g["loc_aob"])
    sim = ob.search_by_projection_sim3(fv, wv, g["S"], amp0, 5, 1.5)
    assert sim[1] == int(g["sim_n"][0]) and np.array_equal(sim[0], g["sim_matched"])
    bow = ob.search_by_bow_kf(fv, fv2, g["v2"], g["f1_desc"], g["v1"], g["f1_angle"], fv1, 0.8, True)
    assert bow[1] == int(g["bow_n"][0]) and np.array_equal(bow[0], g["bow_m12"])
    po = ob.pose_optimize(pp)
    assert tuple(po.iters) == tuple(g["po_iters"]) and po.n_inliers == int(g["po_inl"][0]) and np.array_equal(po.outliers, g["po_out"])
    assert np.allclose(po.Tcw, g["po_T"], atol=1e-6)
    (bw, bv), (fn, fs, ff) = ob.vocab_bow(vv, g["kf_desc"], 1)
    assert np.array_equal(bw, g["bow_word"]) and np.array_equal(bv, g["bow_value"])
    assert np.array_equal(fn, g["fv_node"]) and np.array_equal(fs, g["fv_start"]) and np.array_equal(ff, g["fv_feat"])
    assert np.array_equal(ob.distinctive_descriptors(g["kf_desc"][:30], g["dd_start"]), g["dd_best"])
    assert np.array_equal(ob.wire_pack(g["kf_kps"], g["kf_desc"]), g["wire"])


@pytest.mark.gpu
def test_gpu_reproduces_golden_matching_pose_bow():
    from multi_orbslam3_amd import api
    g = _load("matching_pose_bow.npz")
    fv, wv, fv1, fv2, pp, vv, keep = _fixture_views(g)
    n = len(g["kf_kps"])
    amp0 = np.full(n, -1, np.int32); aob0 = np.zeros(n, np.int32)
    F = api.Frame().upload(fv, keep)
    LM = api.LocalMap().upload(wv)
    loc = api.ORBmatcher(0.8).SearchLocalPoints(F, LM, g["T"], 3.0, False, 0.0, amp0, aob0)
    assert loc[2] == int(g["loc_n"][0]) and np.array_equal(loc[0], g["loc_amp"]) and np.array_equal(loc[1], g["loc_aob"])
    sim = api.ORBmatcher(0.8).SearchByProjectionSim3(F, g["S"], LM, amp0, 5, 1.5)
    assert sim[1] == int(g["sim_n"][0]) and np.array_equal(sim[0], g["sim_matched"])
    bow = api.ORBmatcher(0.8, True).SearchByBoWKF(F, fv2, g["v2"], g["f1_desc"], g["v1"], g["f1_angle"], fv1)
    assert bow[1] == int(g["bow_n"][0]) and np.array_equal(bow[0], g["bow_m12"])
    po = api.Optimizer().PoseOptimization(pp)
    assert tuple(po.iters) == tuple(g["po_iters"]) and po.n_inliers == int(g["po_inl"][0]) and np.array_equal(po.outliers, g["po_out"])
    assert np.abs(po.Tcw - g["po_T"]).max() <= 1e-4
    voc = api.ORBVocabulary(vv, keep)
    (bw, bv), (fn, fs, ff) = voc.transform(g["kf_desc"], 1)
    assert np.array_equal(bw, g["bow_word"]) and np.array_equal(bv, g["bow_value"])
    assert np.array_equal(fn, g["fv_node"]) and np.array_equal(fs, g["fv_start"]) and np.array_equal(ff, g["fv_feat"])
    assert np.array_equal(api.ComputeDistinctiveDescriptors(g["kf_desc"][:30], g["dd_start"]), g["dd_best"])
    assert np.array_equal(F.pack_wire(), g["wire"])


# ---- the larger fixtures (96 x 72 and 640 x 480 extractions, the C2-sized local BA with its LM trace)
def _check_extraction(g, n_features, n_levels, w, h, extract, level_of, cands_of):
    rc, kl, dl, nm = extract(g["L"], (0, 0))
    assert np.array_equal(kl, g["kps"]) and np.array_equal(dl, g["desc"])
    for l in range(n_levels):
        assert np.array_equal(cands_of(l), g["cand%d" % l]), l
    return kl


def test_oracle_reproduces_appendix_f_extractions():
    g = _load("extract_96x72.npz")
    ex = ob.Extractor(n_features=100, n_levels=3, max_width=96, max_height=72)
    kl = _check_extraction(g, 100, 3, 96, 72, ex.extract, ex.level, ex.candidates)
    assert np.array_equal(ex.level(2, border=True), g["level2"]) and len(kl) == 15
    g = _load("extract_640x480.npz")
    ex = ob.Extractor(n_features=1000, max_width=640, max_height=480)
    kl = _check_extraction(g, 1000, 8, 640, 480, ex.extract, ex.level, ex.candidates)
    assert np.array_equal(ex.level(7, border=True), g["level7"]) and len(kl) == 1008
    for got, key in zip(ex.tables(), ("scale", "inv_scale", "sigma2", "inv_sigma2", "features_per_level")):
        assert np.array_equal(got, g[key]), key
    # the monocular Frame constructor's order: every keypoint inside the lapping area {0, 1000} => written from the back
    rc, km, dm, nmono = ex.extract(g["L"], (0, 1000))
    assert nmono == int(g["n_mono"][1]) == 0 and int(g["n_mono"][0]) == len(kl)
    assert np.array_equal(km, g["kps_mono_order"]) and np.array_equal(dm, g["desc_mono_order"])
    assert np.array_equal(km, kl[::-1]) and np.array_equal(dm, g["desc"][::-1])


def _check_lba_20p10(solve, tol):
    b = _load("lba_20p10_2000.npz")
    p, keep = views.lba_problem(b["poses"], b["pose_fixed"], b["points"], b["edges"], tuple(b["cam"]))
    assert (p.n_poses, p.n_points, p.n_edges) == (30, 2000, 11039)
    o = solve(p)
    assert o.status == int(b["status"][0]) and tuple(o.iters) == tuple(b["iters"]) and o.n_outliers == int(b["n_outliers"][0])
    assert np.abs(o.poses - b["out_poses"]).max() <= tol and np.abs(o.points - b["out_points"]).max() <= tol
    assert np.array_equal(o.edge_outlier, b["out_outlier"]) and np.array_equal(o.edge_depth_pos, b["out_depth_pos"])
    tr = o.trace_rows()
    assert tr.shape == b["trace"].shape and np.array_equal(tr[:, 2], b["trace"][:, 2])                       # LM trials per iteration
    assert np.allclose(tr[:, :2], b["trace"][:, :2], rtol=1e-9, atol=0)                                       # (lambda, chi2) per iteration
    assert np.allclose(o.chi2, b["chi2"], rtol=1e-9)


def test_oracle_reproduces_appendix_f_lba_with_its_trace():
    _check_lba_20p10(ob.lba_solve, 1e-6)


@pytest.mark.gpu
def test_gpu_reproduces_appendix_f_fixtures():
    from multi_orbslam3_amd import api
    g = _load("extract_96x72.npz")
    ex = api.ORBextractor(100, 1.2, 3, 20, 7, 96, 72)
    nm, kl, dl = ex(g["L"])
    assert np.array_equal(kl, g["kps"]) and np.array_equal(dl, g["desc"]) and nm == len(kl) == 15
    for l in range(3):
        assert np.array_equal(ex.candidates(0, l), g["cand%d" % l])
    assert np.array_equal(ex.level(0, 2, border=True), g["level2"])
    g = _load("extract_640x480.npz")
    ex = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480)
    nm, kl, dl = ex(g["L"])
    assert np.array_equal(kl, g["kps"]) and np.array_equal(dl, g["desc"]) and nm == int(g["n_mono"][0])
    for l in range(8):
        assert np.array_equal(ex.candidates(0, l), g["cand%d" % l]), l
    assert np.array_equal(ex.level(0, 7, border=True), g["level7"])
    nm, km, dm = ex(g["L"], (0, 1000))
    assert nm == 0 and np.array_equal(km, g["kps_mono_order"]) and np.array_equal(dm, g["desc_mono_order"])
    _check_lba_20p10(api.Optimizer().LocalBundleAdjustment, 1e-4)


@pytest.mark.gpu
def test_gpu_extractor_tables_equal_the_oracles_and_the_fixture():
    """Row a1 (ORBextractor::ORBextractor, S/ORBextractor.cc:408-468): the getters of I/ORBextractor.h:65-85 through the C-ABI on the GPU box
    against the oracle's and the committed 640 x 480 fixture, for the settings of BASELINE's configurations and two odd ones."""
    from multi_orbslam3_amd import api
    g = _load("extract_640x480.npz")
    for nf, sf, nl, w, h in ((1000, 1.2, 8, 640, 480), (2000, 1.2, 8, 1280, 720), (1500, 1.1, 12, 752, 480), (500, 1.5, 4, 320, 240)):
        ex = api.ORBextractor(nf, sf, nl, 20, 7, w, h)
        o = ob.Extractor(n_features=nf, scale_factor=sf, n_levels=nl, max_width=w, max_height=h)
        got, want = ex.tables(), o.tables()
        for a, b_, key in zip(got, want, ("scale", "inv_scale", "sigma2", "inv_sigma2", "features_per_level")):
            assert a.dtype == b_.dtype and a.tobytes() == b_.tobytes(), (nf, sf, nl, key)
            if (nf, sf, nl) == (1000, 1.2, 8):
                assert np.array_equal(a, g[key]), key
        assert int(got[4].sum()) == nf
        assert np.array_equal(ex.GetScaleFactors(), got[0]) and np.array_equal(ex.GetInverseScaleSigmaSquares(), got[3])


def _rig_fixture():
    g = _load("rig_and_fisheye.npz")
    sc = {k[3:]: g[k] for k in g.files if k.startswith("sc_")}
    sc["left"], sc["right"], sc["size"] = tuple([int(g["cam_left"][0])] + [float(x) for x in g["cam_left"][1:]]), tuple([int(g["cam_right"][0])] + [float(x) for x in g["cam_right"][1:]]), float(g["size"][0])
    last = {k[5:]: g[k] for k in g.files if k.startswith("last_")}
    fs = {k[3:]: g[k] for k in g.files if k.startswith("fs_")}
    fs["left"], fs["right"], fs["mono_left"], fs["mono_right"] = sc["left"], sc["right"], int(fs["mono_left"]), int(fs["mono_right"])
    return g, sc, last, fs


def _rig_fixture_checks(g, frustum, search, frame_search, stereo, exact_projections):
    a, b = frustum
    for side, d in (("fl_", a), ("fr_", b)):
        for k in ("track_in_view", "scale_level"):
            assert np.array_equal(d[k], g[side + k]), (side, k)
        for k in ("proj_x", "proj_y", "track_depth", "view_cos"):
            if exact_projections or k in ("track_depth", "view_cos"):
                assert np.array_equal(d[k].view(np.uint32), g[side + k].view(np.uint32)), (side, k)
            else:
                assert np.abs(d[k].astype(np.float64) - g[side + k]).max() <= 3e-4, (side, k)
    assert search[2] == int(g["srch_n"][0]) and np.array_equal(search[0], g["srch_amp"]) and np.array_equal(search[1], g["srch_aob"])
    assert frame_search[2] == int(g["frm_n"][0]) and np.array_equal(frame_search[0], g["frm_amp"]) and np.array_equal(frame_search[1], g["frm_aob"])
    assert stereo[4] == int(g["st_n"][0]) and np.array_equal(stereo[0], g["st_l2r"]) and np.array_equal(stereo[1], g["st_r2l"])
    hit = g["st_l2r"] >= 0
    assert np.allclose(stereo[2][hit], g["st_depth"][hit], rtol=1e-5) and np.allclose(stereo[3][hit], g["st_p3d"][hit], rtol=1e-5, atol=1e-6)


def _rig_optimiser_problems(g):
    rig = views.camera_rig(tuple([int(g["cam_left"][0])] + [float(x) for x in g["cam_left"][1:]]), tuple([int(g["cam_right"][0])] + [float(x) for x in g["cam_right"][1:]]), g["lba_Trl"])
    p, keep = views.lba_problem(g["lba_poses"], g["lba_pose_fixed"], g["lba_points"], g["lba_edges"], tuple(g["lba_cam"]), rig=rig)
    q, keep2 = views.pose_opt_problem(g["po_Xw"], g["po_u"], g["po_v"], g["po_ur"], g["po_w"], tuple(g["po_cam"]), g["po_Tcw"], rig=rig)
    return p, q, [keep, keep2, rig]


def _rig_optimiser_checks(g, lo, po, tol):
    assert lo.status == int(g["lba_status"][0]) and tuple(lo.iters) == tuple(g["lba_iters"]) and int((g["lba_edges"]["ur"] <= -1.5).sum()) > 100
    assert np.allclose(lo.poses, g["lba_out_poses"], atol=tol) and np.allclose(lo.points, g["lba_out_points"], atol=tol)
    assert int((lo.edge_outlier != g["lba_out_outlier"]).sum()) <= (0 if tol < 1e-5 else 1)
    assert po.n_inliers == int(g["po_out_inliers"][0]) and np.array_equal(po.outliers, g["po_out_outliers"]) and np.allclose(po.Tcw, g["po_out_T"], atol=max(tol / 10, 1e-6))


def test_oracle_reproduces_golden_rig_and_fisheye():
    """tests/golden/rig_and_fisheye.npz: a two-camera frame with its local map and last frame, and the two feature sets of a fisheye
    constructor -- the oracle's isInFrustum (both cameras), SearchByProjection(F, MPs), SearchByProjection(Cur, Last) and
    ComputeStereoFishEyeMatches outputs as they were when the fixture was made."""
    g, sc, last, fs = _rig_fixture()
    assert int(g["srch_n"][0]) > 100 and int(g["frm_n"][0]) > 60 and int(g["st_n"][0]) > 40
    fl, fr, wv, rig, keep = helpers.rig_track_views(sc)
    a, b = ob.is_in_frustum_rig(fl, sc["Tcw"], rig, sc["Tlr"], wv)
    mv, mvr, keep2 = helpers.rig_mappoint_views(sc, a, b)
    srch = ob.search_by_projection_mps_rig(fl, fr, mv, mvr, sc["left_to_right"], sc["right_to_left"], 3.0, True, 6.0, 0.8, sc["assigned_mp"], sc["assigned_obs"])
    lv, keep3 = views.lastframe_view(last["mp_valid"], last["outlier"], last["world_pos"], last["desc"], last["octave"], last["angle"], last["n_obs"], last["Tcw"])
    frm = ob.search_by_projection_frame_rig(fl, fr, sc["Tcw"], rig, lv, 7.0, 0, 1, sc["assigned_mp"], sc["assigned_obs"])
    v, keep4 = views.fisheye_stereo_view(fs["kps_left"], fs["desc_left"], fs["mono_left"], fs["kps_right"], fs["desc_right"], fs["mono_right"], fs["left"], fs["right"],
                                         fs["Tlr"], fs["level_sigma2"])
    _rig_fixture_checks(g, (a, b), srch, frm, ob.fisheye_stereo_matches(v), exact_projections=True)
    p, q, keep5 = _rig_optimiser_problems(g)
    _rig_optimiser_checks(g, ob.lba_solve(p), ob.pose_optimize(q), 1e-6)


@pytest.mark.gpu
def test_gpu_reproduces_golden_rig_and_fisheye():
    """The HIP path against the committed fixture (not against a fresh oracle run): the track fields of the fixture go into the searches."""
    from multi_orbslam3_amd import api
    g, sc, last, fs = _rig_fixture()
    fl, fr, wv, rig, keep = helpers.rig_track_views(sc)
    FL, FR = api.Frame().upload(fl, keep[0]), api.Frame().upload(fr, keep[1])
    frustum = FL.isInFrustumRig(sc["Tcw"], rig, sc["Tlr"], wv)
    fix = [{k: g[side + k] for k in ob.RIG_TRACK_KEYS} for side in ("fl_", "fr_")]
    mv, mvr, keep2 = helpers.rig_mappoint_views(sc, fix[0], fix[1])
    m = api.ORBmatcher(0.8, True)
    srch = m.SearchByProjectionRig(FL, FR, mv, mvr, sc["left_to_right"], sc["right_to_left"], 3.0, True, 6.0, sc["assigned_mp"], sc["assigned_obs"])
    lv, keep3 = views.lastframe_view(last["mp_valid"], last["outlier"], last["world_pos"], last["desc"], last["octave"], last["angle"], last["n_obs"], last["Tcw"])
    frm = m.SearchByProjectionFrameRig(FL, FR, sc["Tcw"], rig, lv, 7.0, False, sc["assigned_mp"], sc["assigned_obs"])
    v, keep4 = views.fisheye_stereo_view(fs["kps_left"], fs["desc_left"], fs["mono_left"], fs["kps_right"], fs["desc_right"], fs["mono_right"], fs["left"], fs["right"],
                                         fs["Tlr"], fs["level_sigma2"])
    _rig_fixture_checks(g, frustum, srch, frm, api.ComputeStereoFishEyeMatches(v), exact_projections=False)
    p, q, keep5 = _rig_optimiser_problems(g)
    opt = api.Optimizer()
    _rig_optimiser_checks(g, opt.LocalBundleAdjustment(p), opt.PoseOptimization(q), 1e-4)
