"""Builders for the flattened views of include/orbgpu.h from numpy arrays.

Each builder returns (ctypes struct, keepalive list): the struct only holds raw pointers, so the
caller must keep the keepalive list referenced while the struct is in use.
"""
import ctypes as C

import numpy as np

from . import _capi as capi


def _c(a, dtype):
    if a is None:
        return None
    return np.ascontiguousarray(a, dtype=dtype)


def frame_view(kps, desc, uright=None, depth=None, bounds=None, cam=None, n_levels=8, scale_factor=1.2):
    """bounds = (min_x, max_x, min_y, max_y); cam = (fx, fy, cx, cy, bf, b)."""
    kps = np.ascontiguousarray(kps, dtype=capi.KEYPOINT_DTYPE)
    desc = _c(desc, np.uint8)
    uright = _c(uright, np.float32)
    depth = _c(depth, np.float32)
    v = capi.FrameView()
    v.n = len(kps)
    v.kps, v.desc, v.uright, v.depth = capi.ptr(kps), capi.ptr(desc), capi.ptr(uright), capi.ptr(depth)
    v.min_x, v.max_x, v.min_y, v.max_y = [float(b) for b in bounds]
    v.fx, v.fy, v.cx, v.cy, v.bf, v.b = [float(c) for c in cam]
    v.n_levels = int(n_levels)
    v.scale_factor = float(scale_factor)
    return v, [kps, desc, uright, depth]


def mappoints_view(track_in_view, bad, proj_x, proj_y, proj_xr, track_depth, scale_level, view_cos, desc, n_obs):
    arrs = [_c(track_in_view, np.uint8), _c(bad, np.uint8), _c(proj_x, np.float32), _c(proj_y, np.float32),
            _c(proj_xr, np.float32), _c(track_depth, np.float32), _c(scale_level, np.int32),
            _c(view_cos, np.float32), _c(desc, np.uint8), _c(n_obs, np.int32)]
    v = capi.MapPointsView()
    v.m = len(arrs[0])
    (v.track_in_view, v.bad, v.proj_x, v.proj_y, v.proj_xr, v.track_depth, v.scale_level, v.view_cos, v.desc,
     v.n_obs) = [capi.ptr(a) for a in arrs]
    return v, arrs


def worldpoints_view(pos, normal, min_dist, max_dist, desc, n_obs, bad, skip=None):
    arrs = [_c(pos, np.float32), _c(normal, np.float32), _c(min_dist, np.float32), _c(max_dist, np.float32),
            _c(desc, np.uint8), _c(n_obs, np.int32), _c(bad, np.uint8), _c(skip, np.uint8)]
    v = capi.WorldPointsView()
    v.m = len(arrs[2])
    v.pos, v.normal, v.min_dist, v.max_dist, v.desc, v.n_obs, v.bad, v.skip = [capi.ptr(a) for a in arrs]
    return v, arrs


def lastframe_view(mp_valid, outlier, world_pos, desc, octave, angle, n_obs, Tcw):
    arrs = [_c(mp_valid, np.uint8), _c(outlier, np.uint8), _c(world_pos, np.float32), _c(desc, np.uint8),
            _c(octave, np.int32), _c(angle, np.float32), _c(n_obs, np.int32)]
    v = capi.LastFrameView()
    v.n = len(arrs[0])
    v.mp_valid, v.outlier, v.world_pos, v.desc, v.octave, v.angle, v.n_obs = [capi.ptr(a) for a in arrs]
    T = np.asarray(Tcw, dtype=np.float32).reshape(16)
    for i in range(16):
        v.Tcw[i] = float(T[i])
    return v, arrs


def featvec_view(node_id, start, feat_idx):
    arrs = [_c(node_id, np.uint32), _c(start, np.uint32), _c(feat_idx, np.uint32)]
    v = capi.FeatVecView()
    v.n_nodes = len(arrs[0])
    v.node_id, v.start, v.feat_idx = [capi.ptr(a) for a in arrs]
    return v, arrs


def featvec_from_nodes(node_of_feature):
    """Group feature indices by node id (ascending), as DBoW2::FeatureVector does (std::map + push_back)."""
    node_of_feature = np.asarray(node_of_feature, dtype=np.int64)
    order = np.argsort(node_of_feature, kind="stable")
    nodes, counts = np.unique(node_of_feature, return_counts=True)
    start = np.zeros(len(nodes) + 1, dtype=np.uint32)
    start[1:] = np.cumsum(counts)
    return nodes.astype(np.uint32), start, order.astype(np.uint32)


def vocab_view(child_start, child_ids, desc, weight, word_id, L, weighting=capi.ORBV_TF_IDF, scoring_norm=capi.ORBV_NORM_L1):
    """DBoW2 vocabulary tree flattened (include/orbgpu.h: orbv_vocab_view)."""
    arrs = [_c(child_start, np.int32), _c(child_ids, np.int32), _c(desc, np.uint8), _c(weight, np.float64), _c(word_id, np.int32)]
    v = capi.VocabView()
    v.n_nodes = len(arrs[0]) - 1
    v.L, v.weighting, v.scoring_norm = int(L), int(weighting), int(scoring_norm)
    v.child_start, v.child_ids, v.desc, v.weight, v.word_id = [capi.ptr(a) for a in arrs]
    return v, arrs


def camera_rig(left, right=None, Trl=None):
    """orbg_camera_rig from (model, fx, fy, cx, cy[, k1, k2, k3, k4]) tuples: mpCamera, mpCamera2 and mTrl (3x4 or 4x4)."""
    rig = capi.CameraRig()

    def fill(dst, cam):
        dst.model = int(cam[0])
        dst.fx, dst.fy, dst.cx, dst.cy = [float(np.float32(c)) for c in cam[1:5]]
        ks = list(cam[5:9]) + [0.0] * (4 - len(cam[5:9]))
        for i in range(4):
            dst.k[i] = float(np.float32(ks[i]))

    fill(rig.left, left)
    rig.has_right = 0 if right is None else 1
    if right is not None:
        fill(rig.right, right)
        T = np.asarray(Trl, np.float32).reshape(-1)[:12]
        for i in range(12):
            rig.Trl[i] = float(T[i])
    return rig


def fisheye_stereo_view(kps_left, desc_left, mono_left, kps_right, desc_right, mono_right, left, right, Tlr, level_sigma2):
    """orbx_fisheye_stereo_view: mvKeys / mvKeysRight with their descriptors, monoLeft / monoRight (the lapping-area features sit
    behind them), the two KannalaBrandt8 cameras as (model, fx, fy, cx, cy, k1..k4) tuples, mTlr (3x4 or 4x4), mvLevelSigma2."""
    kl = np.ascontiguousarray(kps_left, dtype=capi.KEYPOINT_DTYPE); kr = np.ascontiguousarray(kps_right, dtype=capi.KEYPOINT_DTYPE)
    dl = _c(desc_left, np.uint8); dr = _c(desc_right, np.uint8); sg = _c(level_sigma2, np.float32)
    v = capi.FisheyeStereoView()
    v.n_left, v.n_right, v.mono_left, v.mono_right = len(kl), len(kr), int(mono_left), int(mono_right)
    v.kps_left, v.kps_right, v.desc_left, v.desc_right, v.level_sigma2 = capi.ptr(kl), capi.ptr(kr), capi.ptr(dl), capi.ptr(dr), capi.ptr(sg)
    v.n_levels = len(sg)
    rig = camera_rig(left, right, np.eye(4)[:3])
    v.left, v.right = rig.left, rig.right
    T = np.asarray(Tlr, np.float32).reshape(-1)[:12]
    for i in range(12):
        v.Tlr[i] = float(T[i])
    return v, [kl, kr, dl, dr, sg]


def lba_problem(poses, pose_fixed, points, edges, cam, lambda_init=0.0, its=(5, 10), device=0, rig=None):
    """poses: (P,16) or (P,4,4) float32; edges: structured EDGE_DTYPE; cam = (fx, fy, cx, cy, bf); rig: camera_rig(...) or None."""
    poses = np.ascontiguousarray(np.asarray(poses, dtype=np.float32).reshape(-1, 16))
    pose_fixed = _c(pose_fixed, np.uint8)
    points = np.ascontiguousarray(np.asarray(points, dtype=np.float32).reshape(-1, 3))
    edges = np.ascontiguousarray(edges, dtype=capi.EDGE_DTYPE)
    p = capi.LbaProblem()
    p.n_poses, p.n_points, p.n_edges = len(poses), len(points), len(edges)
    p.poses, p.pose_fixed, p.points, p.edges = capi.ptr(poses), capi.ptr(pose_fixed), capi.ptr(points), capi.ptr(edges)
    p.fx, p.fy, p.cx, p.cy, p.bf = [float(c) for c in cam]
    p.lambda_init = float(lambda_init)
    p.its_round1, p.its_round2 = int(its[0]), int(its[1])
    p.device = int(device)
    if rig is not None:
        p.rig = C.pointer(rig)
    return p, [poses, pose_fixed, points, edges, rig]


class LbaOutput:
    """Owns the result arrays of one LBA call."""

    def __init__(self, n_poses, n_points, n_edges, trace_cap=64):
        self.poses = np.zeros((n_poses, 16), dtype=np.float32)
        self.points = np.zeros((n_points, 3), dtype=np.float32)
        self.edge_chi2 = np.zeros(n_edges, dtype=np.float64)
        self.edge_depth_pos = np.zeros(n_edges, dtype=np.uint8)
        self.edge_outlier = np.zeros(n_edges, dtype=np.uint8)
        self.trace = np.zeros((trace_cap, 3), dtype=np.float64)
        r = capi.LbaResult()
        r.poses, r.points = capi.ptr(self.poses), capi.ptr(self.points)
        r.edge_chi2, r.edge_depth_pos, r.edge_outlier = capi.ptr(self.edge_chi2), capi.ptr(self.edge_depth_pos), capi.ptr(self.edge_outlier)
        r.trace, r.trace_cap, r.trace_len = capi.ptr(self.trace), trace_cap, 0
        self.c = r

    @property
    def status(self):
        return self.c.status

    @property
    def iters(self):
        return (self.c.iters_round1, self.c.iters_round2)

    @property
    def n_outliers(self):
        return self.c.n_outliers

    @property
    def chi2(self):
        return (self.c.chi2_initial, self.c.chi2_final)

    def trace_rows(self):
        return self.trace[: self.c.trace_len].copy()


def pose_opt_problem(Xw, u, v, ur, inv_sigma2, cam, Tcw, device=0, rig=None):
    """cam = (fx, fy, cx, cy, bf); Tcw = pFrame->mTcw (4x4); rig: camera_rig(...) or None."""
    arrs = [_c(np.asarray(Xw, np.float32).reshape(-1, 3), np.float32), _c(u, np.float32), _c(v, np.float32), _c(ur, np.float32),
            _c(inv_sigma2, np.float32)]
    p = capi.PoseOptProblem()
    p.n = len(arrs[1])
    p.Xw, p.u, p.v, p.ur, p.inv_sigma2 = [capi.ptr(a) for a in arrs]
    p.fx, p.fy, p.cx, p.cy, p.bf = [float(c) for c in cam]
    T = np.asarray(Tcw, np.float32).reshape(16)
    for i in range(16):
        p.Tcw[i] = float(T[i])
    p.device = int(device)
    if rig is not None:
        p.rig = C.pointer(rig)
        arrs.append(rig)
    return p, arrs


class PoseOptOutput:
    def __init__(self, n):
        self.outlier = np.zeros(max(n, 1), np.uint8)
        self.c = capi.PoseOptResult()
        self.c.outlier = capi.ptr(self.outlier)
        self.n = n

    @property
    def Tcw(self):
        return np.array(list(self.c.Tcw), np.float32).reshape(4, 4)

    @property
    def n_inliers(self):
        return self.c.n_inliers

    @property
    def iters(self):
        return tuple(self.c.iters)

    @property
    def chi2(self):
        return tuple(self.c.chi2)

    @property
    def outliers(self):
        return self.outlier[: self.n].copy()


def database_view(inverted_file, bow_vectors, covisible, map_id, bad, map_bad, n_words):
    """orbd_database_view (KeyFrameDatabase flattened, S/KeyFrameDatabase.cc:594-761): inverted_file = {word: [kf, ...]} in
    insertion order, bow_vectors[kf] = (words ascending, values), covisible[kf] = GetBestCovisibilityKeyFrames(10) as indices."""
    K = len(bow_vectors)
    inv_start = np.zeros(n_words + 1, np.int32)
    for w, lst in inverted_file.items():
        inv_start[w + 1] = len(lst)
    inv_start = np.cumsum(inv_start).astype(np.int32)
    inv_kf = np.zeros(max(int(inv_start[-1]), 1), np.int32)
    for w, lst in inverted_file.items():
        inv_kf[inv_start[w]: inv_start[w] + len(lst)] = lst
    bow_start = np.cumsum([0] + [len(b[0]) for b in bow_vectors]).astype(np.int32)
    bow_word = _c(np.concatenate([b[0] for b in bow_vectors]) if K else np.zeros(1), np.int32)
    bow_value = _c(np.concatenate([b[1] for b in bow_vectors]) if K else np.zeros(1), np.float64)
    covis_start = np.cumsum([0] + [len(c) for c in covisible]).astype(np.int32)
    covis_kf = _c(np.concatenate([np.asarray(c, np.int32) for c in covisible] + [np.zeros(0, np.int32)]), np.int32)
    if covis_kf.size == 0:
        covis_kf = np.zeros(1, np.int32)
    arrs = [inv_start, inv_kf, bow_start, bow_word, bow_value, covis_start, covis_kf, _c(map_id, np.int32), _c(bad, np.uint8), _c(map_bad, np.uint8)]
    v = capi.DatabaseView(K, n_words, *[capi.ptr(a) for a in arrs])
    return v, arrs
