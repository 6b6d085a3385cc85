"""Seeded synthetic inputs for the hot path: stereo image sequences of a textured
plane seen along a smooth trajectory, map points derived from stereo depth, and local-BA problems.

Pure numpy; used by tests/, bench.py and __graft_entry__.smoke() for BOTH the HIP path and the oracle.
"""
import numpy as np

from . import _capi as capi

SEED_IMAGES = 0xC0FFEE
SEED_LBA = 0xBA5EBA11


def camera_for(width, height=None):
    """Pinhole intrinsics scaled from R/ros/conf/EuRoC_mono_client.yaml:9-12 (fx=458.654 @ 752 px), bf=47.9 @ 752."""
    s = width / 752.0
    fx = np.float32(458.654 * s)
    fy = fx
    cx = np.float32(width / 2.0)
    cy = np.float32((height if height else width * 3 // 4) / 2.0)
    bf = np.float32(47.9 * s)
    b = np.float32(bf / fx)
    return dict(fx=fx, fy=fy, cx=cx, cy=cy, bf=bf, b=b)


def make_texture(seed=SEED_IMAGES, width=1600, height=1200):
    """6 octaves of value noise + 400 random dark/bright rectangles and discs -> u8 texture with many FAST corners."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    tex = np.zeros((height, width), np.float32)
    amp = 1.0
    for o in range(6):
        gh, gw = 4 * 2 ** o + 2, 5 * 2 ** o + 2
        g = rng.rand(gh, gw).astype(np.float32)
        ys = np.linspace(0, gh - 1.001, height, dtype=np.float32)
        xs = np.linspace(0, gw - 1.001, width, dtype=np.float32)
        y0 = ys.astype(np.int32); x0 = xs.astype(np.int32)
        fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
        a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
        tex += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
        amp *= 0.55
    tex = (tex - tex.min()) / (tex.max() - tex.min())
    tex = 60 + 130 * tex
    yy, xx = np.mgrid[0:height, 0:width]
    for i in range(400):
        val = float(rng.choice([15, 35, 215, 240]))
        cx, cy = rng.randint(0, width), rng.randint(0, height)
        if i % 2 == 0:
            w, h = rng.randint(8, 60), rng.randint(8, 60)
            tex[max(cy - h, 0):cy + h, max(cx - w, 0):cx + w] = val
        else:
            r = rng.randint(5, 30)
            y0, y1, x0, x1 = max(cy - r, 0), min(cy + r + 1, height), max(cx - r, 0), min(cx + r + 1, width)
            m = (yy[y0:y1, x0:x1] - cy) ** 2 + (xx[y0:y1, x0:x1] - cx) ** 2 <= r * r
            tex[y0:y1, x0:x1][m] = val
    return np.clip(tex, 0, 255).astype(np.uint8)


def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


class Scene:
    """A textured plane (world z = 0, texture at `px_per_m`) viewed by a rectified stereo rig."""

    def __init__(self, width=640, height=480, seed=SEED_IMAGES, tex_size=(1600, 1200), px_per_m=200.0):
        self.W, self.H = width, height
        self.cam = camera_for(width, height)
        self.tex = make_texture(seed, *tex_size)
        self.px_per_m = px_per_m
        self.seed = seed

    def pose(self, k):
        """Tcw (4x4 float64) of the LEFT camera at frame k: smooth drift + tilt, plane ~2.2-3.6 m away."""
        t = 0.02 * k
        R = _rot(0.32 + 0.04 * np.sin(0.7 * t + 0.3), -0.22 + 0.05 * np.sin(0.5 * t), 0.03 * np.sin(0.9 * t))
        Cw = np.array([4.0 - 0.25 + 0.6 * t, 3.0 + 0.15 * np.sin(0.8 * t), -2.8 - 0.1 * np.sin(0.6 * t)])
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = -R @ Cw
        return T

    def _render(self, Tcw):
        c = self.cam
        K = np.array([[c["fx"], 0, c["cx"]], [0, c["fy"], c["cy"]], [0, 0, 1]], np.float64)
        R, t = Tcw[:3, :3], Tcw[:3, 3]
        Hm = K @ np.stack([R[:, 0], R[:, 1], t], axis=1)       # plane (X,Y,1) -> image
        Hi = np.linalg.inv(Hm)
        u, v = np.meshgrid(np.arange(self.W, dtype=np.float64), np.arange(self.H, dtype=np.float64))
        q = Hi @ np.stack([u.ravel(), v.ravel(), np.ones(u.size)])
        X, Y = q[0] / q[2], q[1] / q[2]
        tx = np.clip(X * self.px_per_m, 0, self.tex.shape[1] - 1.001)
        ty = np.clip(Y * self.px_per_m, 0, self.tex.shape[0] - 1.001)
        x0 = tx.astype(np.int32); y0 = ty.astype(np.int32)
        fx = tx - x0; fy = ty - y0
        T = self.tex.astype(np.float64)
        val = (T[y0, x0] * (1 - fx) + T[y0, x0 + 1] * fx) * (1 - fy) + (T[y0 + 1, x0] * (1 - fx) + T[y0 + 1, x0 + 1] * fx) * fy
        return np.clip(np.rint(val), 0, 255).astype(np.uint8).reshape(self.H, self.W)

    def stereo_pair(self, k):
        Tl = self.pose(k)
        Tr = Tl.copy()
        Tr[0, 3] -= float(self.cam["b"])     # x_right = x_left - b
        return self._render(Tl), self._render(Tr), Tl

    def depth_at(self, kps, Tcw):
        """Camera-frame depth of the plane z = 0 along the ray of every keypoint (mono agents have no stereo depth:
        the benchmark's last-frame / map views are seeded from the scene geometry instead)."""
        c = self.cam
        d = np.stack([(kps["x"].astype(np.float64) - float(c["cx"])) / float(c["fx"]),
                      (kps["y"].astype(np.float64) - float(c["cy"])) / float(c["fy"]), np.ones(len(kps))], axis=1)
        R, t = Tcw[:3, :3], Tcw[:3, 3]
        Ow = -R.T @ t
        dw = d @ R                                  # R^T d per row
        z = -Ow[2] / dw[:, 2]                       # Ow + z * dw hits world z = 0
        return np.where(np.isfinite(z) & (z > 0), z, -1.0).astype(np.float32)

    def frame_view_params(self):
        c = self.cam
        return dict(bounds=(0.0, float(self.W), 0.0, float(self.H)),
                    cam=(c["fx"], c["fy"], c["cx"], c["cy"], c["bf"], c["b"]))


def unproject_to_world(kps, depth, Tcw, cam):
    """Stereo unprojection as Frame::UnprojectStereo (S/Frame.cc) in float64 -> world points, valid mask."""
    z = depth.astype(np.float64)
    valid = z > 0
    x = (kps["x"].astype(np.float64) - float(cam["cx"])) * z / float(cam["fx"])
    y = (kps["y"].astype(np.float64) - float(cam["cy"])) * z / float(cam["fy"])
    Pc = np.stack([x, y, z], axis=1)
    R, t = Tcw[:3, :3], Tcw[:3, 3]
    Pw = (Pc - t) @ R          # R^T (Pc - t)
    return Pw.astype(np.float32), valid


def map_from_frame(kps, desc, depth, Tcw, cam, scale_factor=1.2, n_levels=8):
    """Map points as LocalMapping would create them from one stereo keyframe: position, normal,
    min/max distance (MapPoint::UpdateNormalAndDepth, S/MapPoint.cc:545-609), descriptor."""
    Pw, valid = unproject_to_world(kps, depth, Tcw, cam)
    idx = np.nonzero(valid)[0]
    Ow = (-Tcw[:3, :3].T @ Tcw[:3, 3]).astype(np.float32)
    PO = Pw[idx] - Ow
    dist = np.linalg.norm(PO, axis=1).astype(np.float32)
    normal = (PO / dist[:, None]).astype(np.float32)
    sf = np.float32(1.0)
    scales = [sf]
    for _ in range(1, n_levels):
        sf = np.float32(sf * np.float32(scale_factor))
        scales.append(sf)
    scales = np.array(scales, np.float32)
    lvl = kps["octave"][idx]
    max_d = (dist * scales[lvl]).astype(np.float32)
    min_d = (max_d / scales[n_levels - 1]).astype(np.float32)
    return dict(pos=Pw[idx].copy(), normal=normal, min_dist=min_d, max_dist=max_d, desc=desc[idx].copy(),
                n_obs=np.full(len(idx), 3, np.int32), bad=np.zeros(len(idx), np.uint8), src_idx=idx)


def perturb_pose(Tcw, rng, sigma_rot_deg=0.5, sigma_t=0.01):
    w = rng.randn(3) * np.deg2rad(sigma_rot_deg)
    dR = _rot(*w)
    T = Tcw.copy()
    T[:3, :3] = dR @ Tcw[:3, :3]
    T[:3, 3] = dR @ Tcw[:3, 3] + rng.randn(3) * sigma_t
    return T


# ---------------------------------------------------------------- local BA problems
def make_lba_problem(n_free=20, n_fixed=10, n_points=2000, seed=SEED_LBA, width=640, height=480,
                     outlier_frac=0.03, mono_frac=0.0, min_obs=3, max_obs=8):
    """Synthetic LBA problem of SURVEY.md section 8d: poses on a trajectory, each point seen by a contiguous run of
    min_obs..max_obs keyframes, pixel noise sigma = 1 px * scale[octave], gross outliers, float32 inputs.
    Pose order: fixed poses first (lower keyframe ids), then free ones -- ascending vertex id."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    cam = camera_for(width, height)
    fx, fy, cx, cy, bf = [float(cam[k]) for k in ("fx", "fy", "cx", "cy", "bf")]
    P = n_free + n_fixed
    poses_true = np.zeros((P, 4, 4))
    for k in range(P):
        R = _rot(0.02 * np.sin(0.3 * k), 0.03 * np.sin(0.2 * k + 1.0), 0.01 * np.sin(0.5 * k))
        C = np.array([0.12 * k, 0.02 * np.sin(0.4 * k), 0.03 * np.cos(0.3 * k)])
        poses_true[k] = np.eye(4)
        poses_true[k, :3, :3] = R
        poses_true[k, :3, 3] = -R @ C
    scales = 1.2 ** np.arange(8)
    level_p = np.array([217, 181, 151, 126, 105, 87, 73, 60], np.float64)
    level_p /= level_p.sum()
    points_true = np.zeros((n_points, 3))
    edges = []
    for j in range(n_points):
        nobs = rng.randint(min_obs, max_obs + 1)
        k0 = rng.randint(0, max(P - nobs, 0) + 1)
        kc = min(k0 + nobs // 2, P - 1)
        # a point in front of the middle observing camera
        z = rng.uniform(2.0, 9.0)
        u = rng.uniform(60, width - 60); v = rng.uniform(60, height - 60)
        Pc = np.array([(u - cx) * z / fx, (v - cy) * z / fy, z])
        Rc, tc = poses_true[kc, :3, :3], poses_true[kc, :3, 3]
        Xw = Rc.T @ (Pc - tc)
        points_true[j] = Xw
        for k in range(k0, min(k0 + nobs, P)):
            Xc = poses_true[k, :3, :3] @ Xw + poses_true[k, :3, 3]
            if Xc[2] < 0.3:
                continue
            uu = fx * Xc[0] / Xc[2] + cx; vv = fy * Xc[1] / Xc[2] + cy
            if not (0 <= uu < width and 0 <= vv < height):
                continue
            octave = rng.choice(8, p=level_p)
            s = scales[octave]
            nu, nv, nr = rng.randn(3) * s
            ur = uu - bf / Xc[2] + nr
            if rng.rand() < outlier_frac:
                nu += rng.choice([-20, 20]); nv += rng.choice([-20, 20])
            is_mono = rng.rand() < mono_frac
            edges.append((k, j, uu + nu, vv + nv, -1.0 if is_mono else ur, 1.0 / (s * s)))
    E = np.zeros(len(edges), dtype=capi.EDGE_DTYPE)
    for i, e in enumerate(edges):
        E[i] = (e[0], e[1], np.float32(e[2]), np.float32(e[3]), np.float32(e[4]), np.float32(np.float32(1.0) / np.float32(e[5] ** -1)))
    # inv_sigma2 as the reference computes it: 1.0f / (scale*scale) in float32
    sc = np.ones(8, np.float32)
    for i in range(1, 8):
        sc[i] = np.float32(sc[i - 1] * np.float32(1.2))
    inv_s2 = (np.float32(1.0) / (sc * sc)).astype(np.float32)
    oct_of = np.array([int(round(np.log(np.sqrt(1.0 / e[5])) / np.log(1.2))) for e in edges], np.int32) if edges else np.zeros(0, np.int32)
    E["inv_sigma2"] = inv_s2[oct_of]
    # noisy initial estimates, float32
    poses0 = np.zeros((P, 16), np.float32)
    fixed = np.zeros(P, np.uint8)
    fixed[:n_fixed] = 1
    for k in range(P):
        T = poses_true[k].copy()
        if not fixed[k]:
            T = perturb_pose(T, rng, sigma_rot_deg=0.3, sigma_t=0.01)
        poses0[k] = T.astype(np.float32).reshape(16)
    points0 = (points_true + rng.randn(n_points, 3) * 0.02).astype(np.float32)
    return dict(poses=poses0, pose_fixed=fixed, points=points0, edges=E, cam=(fx, fy, cx, cy, bf),
                poses_true=poses_true, points_true=points_true)


def make_pose_opt_problem(n=500, seed=0xF00D, width=640, height=480, outlier_frac=0.1, mono_frac=0.2, sigma_rot_deg=0.5,
                          sigma_t=0.02):
    """Correspondences of one frame for Optimizer::PoseOptimization: points in front of a camera, observations with
    1 px * scale noise, gross outliers, initial pose = truth perturbed."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    cam = camera_for(width, height)
    fx, fy, cx, cy, bf = [float(cam[k]) for k in ("fx", "fy", "cx", "cy", "bf")]
    T_true = np.eye(4)
    T_true[:3, :3] = _rot(0.05, -0.03, 0.02)
    T_true[:3, 3] = [0.1, -0.05, 0.2]
    z = rng.uniform(1.5, 9.0, n)
    uu = rng.uniform(30, width - 30, n); vv = rng.uniform(30, height - 30, n)
    Pc = np.stack([(uu - cx) * z / fx, (vv - cy) * z / fy, z], 1)
    Xw = (Pc - T_true[:3, 3]) @ T_true[:3, :3]
    sc = np.ones(8, np.float32)
    for i in range(1, 8):
        sc[i] = np.float32(sc[i - 1] * np.float32(1.2))
    octv = rng.randint(0, 8, n)
    s = sc[octv].astype(np.float64)
    u = uu + rng.randn(n) * s; v = vv + rng.randn(n) * s; ur = uu - bf / z + rng.randn(n) * s
    bad = rng.rand(n) < outlier_frac
    u[bad] += rng.choice([-25, 25], bad.sum()); v[bad] += rng.choice([-25, 25], bad.sum())
    ur[rng.rand(n) < mono_frac] = -1.0
    inv_s2 = (np.float32(1.0) / (sc * sc)).astype(np.float32)[octv]
    T0 = perturb_pose(T_true, rng, sigma_rot_deg, sigma_t)
    return dict(Xw=Xw.astype(np.float32), u=u.astype(np.float32), v=v.astype(np.float32), ur=ur.astype(np.float32),
                inv_sigma2=inv_s2, cam=(fx, fy, cx, cy, bf), Tcw=T0.astype(np.float32), T_true=T_true, bad=bad)


# ---------------------------------------------------------------- two-fisheye rig (mpCamera2 != NULL, NLeft != -1)
# TUM-VI-like 512 x 512 KannalaBrandt8 cameras (the reference ships no settings file; values of that order)
KB8_LEFT = (capi.CAM_KANNALA_BRANDT8, 190.978, 190.973, 254.932, 256.897, 0.00348, 0.000715, -0.00205, 0.000203)
KB8_RIGHT = (capi.CAM_KANNALA_BRANDT8, 190.442, 190.435, 252.598, 254.917, 0.00340, 0.00177, -0.00266, 0.000330)


def rig_Trl():
    """mTrl: the right camera's pose in the left camera's frame (10 cm baseline, a degree of misalignment)."""
    T = np.eye(4)
    T[:3, :3] = _rot(0.012, -0.02, 0.006)
    T[:3, 3] = [-0.101, 0.0012, -0.0009]
    return T


def kb8_project(cam, X):
    """KannalaBrandt8::project (or Pinhole::project for a pinhole tuple) in float64: for making observations; the oracle holds the
    reference's float32 form."""
    X = np.asarray(X, np.float64)
    if cam[0] == capi.CAM_PINHOLE:
        return np.stack([cam[1] * X[..., 0] / X[..., 2] + cam[3], cam[2] * X[..., 1] / X[..., 2] + cam[4]], -1)
    _, fx, fy, cx, cy, k1, k2, k3, k4 = cam
    r = np.hypot(X[..., 0], X[..., 1])
    th = np.arctan2(r, X[..., 2])
    psi = np.arctan2(X[..., 1], X[..., 0])
    d = th + k1 * th ** 3 + k2 * th ** 5 + k3 * th ** 7 + k4 * th ** 9
    return np.stack([fx * d * np.cos(psi) + cx, fy * d * np.sin(psi) + cy], -1)


def _kb8_ray(cam, u, v, rng_depth):
    """A point at depth-along-ray rng_depth whose projection is near (u, v) (first-order inverse: theta = r / f)."""
    _, fx, fy, cx, cy = cam[:5]
    mx, my = (u - cx) / fx, (v - cy) / fy
    if cam[0] == capi.CAM_PINHOLE:
        d = np.array([mx, my, 1.0])
        return rng_depth * d / np.linalg.norm(d)
    th = np.hypot(mx, my)
    psi = np.arctan2(my, mx)
    return rng_depth * np.array([np.sin(th) * np.cos(psi), np.sin(th) * np.sin(psi), np.cos(th)])


def make_lba_rig_problem(n_free=8, n_fixed=4, n_points=600, seed=0xF15E, size=512, outlier_frac=0.03, right_frac=0.6, left_frac=0.9,
                         left=KB8_LEFT, right=KB8_RIGHT):
    """A local BA window of a two-fisheye rig (S/Optimizer.cc:2021-2120 with mpCamera2): every observation of a point by a keyframe is a
    monocular edge through mpCamera (left) and / or an EdgeSE3ProjectXYZToBody through mpCamera2 after mTrl (right, ur =
    LBA_UR_RIGHT_CAMERA); no stereo edges (mvuRight is -1 on such frames).  Edges in creation order: per point, per keyframe,
    left then right."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    Trl = rig_Trl()
    P = n_free + n_fixed
    poses_true = np.zeros((P, 4, 4))
    for k in range(P):
        R = _rot(0.03 * np.sin(0.3 * k), 0.05 * np.sin(0.2 * k + 1.0), 0.02 * np.sin(0.5 * k))
        Cc = np.array([0.10 * k, 0.02 * np.sin(0.4 * k), 0.03 * np.cos(0.3 * k)])
        poses_true[k] = np.eye(4)
        poses_true[k, :3, :3] = R
        poses_true[k, :3, 3] = -R @ Cc
    sc = np.ones(8, np.float32)
    for i in range(1, 8):
        sc[i] = np.float32(sc[i - 1] * np.float32(1.2))
    inv_s2 = (np.float32(1.0) / (sc * sc)).astype(np.float32)
    points_true = np.zeros((n_points, 3))
    edges = []
    for j in range(n_points):
        nobs = rng.randint(3, 8)
        k0 = rng.randint(0, max(P - nobs, 0) + 1)
        kc = min(k0 + nobs // 2, P - 1)
        Pc = _kb8_ray(left, rng.uniform(40, size - 40), rng.uniform(40, size - 40), rng.uniform(1.5, 8.0))
        Xw = poses_true[kc, :3, :3].T @ (Pc - poses_true[kc, :3, 3])
        points_true[j] = Xw
        for k in range(k0, min(k0 + nobs, P)):
            Xl = poses_true[k, :3, :3] @ Xw + poses_true[k, :3, 3]
            Xr = Trl[:3, :3] @ Xl + Trl[:3, 3]
            for is_right, Xc, cam, frac in ((False, Xl, left, left_frac), (True, Xr, right, right_frac)):
                if Xc[2] < 0.2 or rng.rand() > frac:
                    continue
                uv = kb8_project(cam, Xc)
                if not (5 <= uv[0] < size - 5 and 5 <= uv[1] < size - 5):
                    continue
                octave = rng.randint(0, 8)
                noise = rng.randn(2) * float(sc[octave])
                if rng.rand() < outlier_frac:
                    noise += rng.choice([-20, 20], 2)
                edges.append((k, j, uv[0] + noise[0], uv[1] + noise[1], capi.UR_RIGHT_CAMERA if is_right else -1.0, inv_s2[octave]))
    E = np.zeros(len(edges), dtype=capi.EDGE_DTYPE)
    for i, e in enumerate(edges):
        E[i] = (e[0], e[1], np.float32(e[2]), np.float32(e[3]), np.float32(e[4]), e[5])
    poses0 = np.zeros((P, 16), np.float32)
    fixed = np.zeros(P, np.uint8)
    fixed[:n_fixed] = 1
    for k in range(P):
        T = poses_true[k].copy()
        if not fixed[k]:
            T = perturb_pose(T, rng, sigma_rot_deg=0.3, sigma_t=0.01)
        poses0[k] = T.astype(np.float32).reshape(16)
    points0 = (points_true + rng.randn(n_points, 3) * 0.02).astype(np.float32)
    # the five scalars of a KeyFrame (fx, fy, cx, cy, mbf): only stereo edges read them; there are none here
    return dict(poses=poses0, pose_fixed=fixed, points=points0, edges=E, cam=(left[1], left[2], left[3], left[4], 0.0),
                rig=(left, right, Trl.astype(np.float32)), poses_true=poses_true, points_true=points_true)


def make_pose_opt_rig_problem(n_left=300, n_right=200, seed=0xF15F, size=512, outlier_frac=0.1, sigma_rot_deg=0.5, sigma_t=0.02,
                              left=KB8_LEFT, right=KB8_RIGHT):
    """Correspondences of a two-fisheye Frame for Optimizer::PoseOptimization (S/Optimizer.cc:1085-1151): features i < Nleft are
    mvKeys (left camera), the others mvKeysRight (ur = LBA_UR_RIGHT_CAMERA)."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    Trl = rig_Trl()
    T_true = np.eye(4)
    T_true[:3, :3] = _rot(0.05, -0.03, 0.02)
    T_true[:3, 3] = [0.1, -0.05, 0.2]
    n = n_left + n_right
    sc = np.ones(8, np.float32)
    for i in range(1, 8):
        sc[i] = np.float32(sc[i - 1] * np.float32(1.2))
    Xw = np.zeros((n, 3)); u = np.zeros(n); v = np.zeros(n); ur = np.full(n, -1.0)
    octv = rng.randint(0, 8, n)
    for i in range(n):
        is_right = i >= n_left
        cam = right if is_right else left
        Pc = _kb8_ray(cam, rng.uniform(30, size - 30), rng.uniform(30, size - 30), rng.uniform(1.5, 8.0))    # in the observing camera
        Xl = Trl[:3, :3].T @ (Pc - Trl[:3, 3]) if is_right else Pc
        Xw[i] = T_true[:3, :3].T @ (Xl - T_true[:3, 3])
        uv = kb8_project(cam, Pc) + rng.randn(2) * float(sc[octv[i]])
        u[i], v[i] = uv
        if is_right:
            ur[i] = capi.UR_RIGHT_CAMERA
    bad = rng.rand(n) < outlier_frac
    u[bad] += rng.choice([-25, 25], bad.sum()); v[bad] += rng.choice([-25, 25], bad.sum())
    inv_s2 = (np.float32(1.0) / (sc * sc)).astype(np.float32)[octv]
    T0 = perturb_pose(T_true, rng, sigma_rot_deg, sigma_t)
    return dict(Xw=Xw.astype(np.float32), u=u.astype(np.float32), v=v.astype(np.float32), ur=ur.astype(np.float32),
                inv_sigma2=inv_s2, cam=(left[1], left[2], left[3], left[4], 0.0), rig=(left, right, Trl.astype(np.float32)),
                Tcw=T0.astype(np.float32), T_true=T_true, bad=bad)


def make_rig_track_scene(n_points=1500, n_distract=300, seed=0xF1E0, size=512, left=KB8_LEFT, right=KB8_RIGHT, stereo_frac=0.4,
                         occupied_frac=0.08, zero_obs_frac=0.05):
    """A two-camera Frame (Nleft != -1) and a local map for Frame::isInFrustum / SearchByProjection(Frame, MapPoints) in their rig form
    (S/Frame.cc:545-554,1154-1231; S/ORBmatcher.cc:44-214): points seen by one or both cameras, features at their (noisy) projections
    with descriptors a few bits away, near-twins for the ratio test, distractors, stereo partners (mvLeftToRightMatch /
    mvRightToLeftMatch), features that already hold a point (with and without observations), bad and zero-observation points."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    Trl = rig_Trl()
    Tlr = np.linalg.inv(Trl)
    Tcw = np.eye(4)
    Tcw[:3, :3] = _rot(0.04, -0.06, 0.03)
    Tcw[:3, 3] = [0.2, -0.1, 0.15]
    Rwc, Ow = Tcw[:3, :3].T, -Tcw[:3, :3].T @ Tcw[:3, 3]
    sf = np.float32(1.2)
    pos = np.zeros((n_points, 3)); normal = np.zeros((n_points, 3)); max_d = np.zeros(n_points); lvl = rng.randint(0, 8, n_points)
    for i in range(n_points):
        cam = left if rng.rand() < 0.5 else right
        Pc = _kb8_ray(cam, rng.uniform(-40, size + 40), rng.uniform(-40, size + 40), rng.uniform(1.5, 9.0))
        Xl = Pc if cam is left else Tlr[:3, :3] @ Pc + Tlr[:3, 3]
        pos[i] = Rwc @ Xl + Ow
        d = pos[i] - Ow
        dist = np.linalg.norm(d)
        nrm = d / dist + rng.randn(3) * (0.9 if rng.rand() < 0.08 else 0.15)          # some seen at too steep an angle
        normal[i] = nrm / np.linalg.norm(nrm)
        max_d[i] = dist * 1.2 ** (lvl[i] - rng.uniform(0.25, 0.75)) * (3.0 if rng.rand() < 0.03 else 1.0)
    min_d = max_d / 1.2 ** 7 * np.where(rng.rand(n_points) < 0.03, 40.0, 1.0)           # some out of their distance range
    desc = rng.randint(0, 256, (n_points, 32)).astype(np.uint8)
    n_obs = np.where(rng.rand(n_points) < zero_obs_frac, 0, rng.randint(1, 9, n_points)).astype(np.int32)
    bad = (rng.rand(n_points) < 0.04).astype(np.uint8)

    def noisy(d, bits):
        d = d.copy()
        for b in rng.choice(256, bits, replace=False):
            d[b >> 3] ^= 1 << (b & 7)
        return d

    base_angle = rng.uniform(0, 360, n_points)
    feats = {"L": [], "R": []}                                   # (x, y, octave, desc, point or -1)
    for i in range(n_points):
        Xc = Tcw[:3, :3] @ pos[i] + Tcw[:3, 3]
        for side, cam, X in (("L", left, Xc), ("R", right, Trl[:3, :3] @ Xc + Trl[:3, 3])):
            if X[2] <= 0.05 or rng.rand() < 0.25:
                continue
            uv = kb8_project(cam, X) + rng.randn(2) * (1.2 if rng.rand() < 0.85 else 9.0)         # some only a wider window finds
            if not (2 < uv[0] < size - 2 and 2 < uv[1] < size - 2):
                continue
            o = int(np.clip(lvl[i] - (1 if rng.rand() < 0.3 else 0), 0, 7))
            feats[side].append((uv[0], uv[1], o, noisy(desc[i], rng.randint(0, 45)), i))
            if rng.rand() < 0.15:                                                         # a near-twin: the ratio test's business
                feats[side].append((uv[0] + rng.randn(), uv[1] + rng.randn(), o if rng.rand() < 0.6 else max(o - 1, 0),
                                    noisy(desc[i], rng.randint(5, 50)), -1))
    for side in "LR":
        for _ in range(n_distract):
            feats[side].append((rng.uniform(2, size - 2), rng.uniform(2, size - 2), rng.randint(0, 8), rng.randint(0, 256, 32).astype(np.uint8), -1))
        order = rng.permutation(len(feats[side]))
        feats[side] = [feats[side][k] for k in order]

    def arrays(fl):
        k = np.zeros(len(fl), capi.KEYPOINT_DTYPE)
        k["x"] = [f[0] for f in fl]; k["y"] = [f[1] for f in fl]; k["octave"] = [f[2] for f in fl]
        k["size"] = 31.0
        k["angle"] = [np.float32((base_angle[f[4]] + rng.randn() * 2.0 + (140.0 if rng.rand() < 0.05 else 0.0)) % 360.0) if f[4] >= 0 else rng.uniform(0, 360)
                      for f in fl]
        return k, np.stack([f[3] for f in fl]).astype(np.uint8), np.array([f[4] for f in fl])

    kl, dl, pl = arrays(feats["L"]); kr, dr, pr = arrays(feats["R"])
    l2r = np.full(len(kl), -1, np.int32); r2l = np.full(len(kr), -1, np.int32)
    where_r = {p: j for j, p in enumerate(pr) if p >= 0}
    for j, p in enumerate(pl):
        if p >= 0 and p in where_r and rng.rand() < stereo_frac:
            l2r[j] = where_r[p]; r2l[where_r[p]] = j
    n = len(kl) + len(kr)
    amp0 = np.full(n, -1, np.int32); aob0 = np.zeros(n, np.int32)
    occ = rng.rand(n) < occupied_frac
    amp0[occ] = rng.randint(0, n_points, occ.sum()); aob0[occ] = rng.randint(0, 4, occ.sum())
    return dict(Tcw=Tcw.astype(np.float32), Trl=Trl.astype(np.float32), Tlr=Tlr.astype(np.float32)[:3], left=left, right=right, size=size,
                pos=pos.astype(np.float32), normal=normal.astype(np.float32), min_dist=min_d.astype(np.float32), max_dist=max_d.astype(np.float32),
                desc=desc, n_obs=n_obs, bad=bad, kps_left=kl, desc_left=dl, kps_right=kr, desc_right=dr, left_to_right=l2r, right_to_left=r2l,
                assigned_mp=amp0, assigned_obs=aob0, level=lvl, base_angle=base_angle)


def rig_last_frame(sc, n_last=900, seed=5, motion=(0.0, 0.0, 0.0)):
    """A last frame for SearchByProjection(CurrentFrame, LastFrame) on the scene above: its features hold the scene's first n_last points
    (some none, some flagged outliers), octave / angle near the current features' of the same point; Tcw_last = the current pose moved by
    `motion` (in the camera frame: z decides bForward / bBackward)."""
    rng = np.random.RandomState(seed)
    n_last = min(n_last, len(sc["pos"]))
    T_last = sc["Tcw"].astype(np.float64).copy()
    T_last[:3, 3] += np.asarray(motion, np.float64)
    return dict(mp_valid=(rng.rand(n_last) < 0.85).astype(np.uint8), outlier=(rng.rand(n_last) < 0.06).astype(np.uint8),
                world_pos=sc["pos"][:n_last].copy(), desc=sc["desc"][:n_last].copy(),
                octave=np.clip(sc["level"][:n_last] + rng.choice([-1, 0, 0, 0, 1], n_last), 0, 7).astype(np.int32),
                angle=((sc["base_angle"][:n_last] + rng.randn(n_last) * 2.0) % 360.0).astype(np.float32), n_obs=sc["n_obs"][:n_last].copy(),
                Tcw=T_last.astype(np.float32))


def make_fisheye_stereo_scene(n_stereo=700, n_mono_left=300, n_mono_right=260, n_distract=150, seed=0xF15C, size=512, left=KB8_LEFT, right=KB8_RIGHT):
    """The two feature sets of a two-fisheye Frame before Frame::ComputeStereoFishEyeMatches (S/Frame.cc:1093-1150): monocular features
    first, then the lapping-area ones -- 3-D points seen by both cameras (pixel noise by level, descriptors a few bits apart; some
    with a wrong partner position: the reprojection test's business; some far away: the parallax test's), near-duplicate descriptors
    for the ratio test, distractors."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    Trl = rig_Trl(); Tlr = np.linalg.inv(Trl)
    sc = np.ones(8, np.float32)
    for i in range(1, 8):
        sc[i] = np.float32(sc[i - 1] * np.float32(1.2))
    sigma2 = (sc * sc).astype(np.float32)

    def rand_feats(n):
        k = np.zeros(n, capi.KEYPOINT_DTYPE)
        k["x"] = rng.uniform(5, size - 5, n); k["y"] = rng.uniform(5, size - 5, n); k["octave"] = rng.randint(0, 8, n); k["size"] = 31.0
        return k, rng.randint(0, 256, (n, 32)).astype(np.uint8)

    kl_m, dl_m = rand_feats(n_mono_left); kr_m, dr_m = rand_feats(n_mono_right)
    L, R = [], []
    for i in range(n_stereo):
        depth = rng.uniform(0.6, 9.0) if rng.rand() < 0.9 else rng.uniform(40.0, 400.0)        # far points: too little parallax
        Pl = _kb8_ray(left, rng.uniform(160, size - 20), rng.uniform(40, size - 40), depth)
        Pr = Trl[:3, :3] @ Pl + Trl[:3, 3]
        o = rng.randint(0, 8)
        uvl = kb8_project(left, Pl) + rng.randn(2) * 0.6 * float(sc[o])
        uvr = kb8_project(right, Pr) + rng.randn(2) * 0.6 * float(sc[o])
        if rng.rand() < 0.08:
            uvr += rng.choice([-1, 1], 2) * rng.uniform(6, 30, 2)                            # a wrong partner: reprojection error
        d = rng.randint(0, 256, 32).astype(np.uint8)
        dr_ = d.copy()
        for b in rng.choice(256, rng.randint(0, 40), replace=False):
            dr_[b >> 3] ^= 1 << (b & 7)
        L.append((uvl[0], uvl[1], o, d)); R.append((uvr[0], uvr[1], min(7, max(0, o + rng.randint(-1, 2))), dr_))
        if rng.rand() < 0.12:                                                                 # a second right feature nearly as close: Lowe's ratio
            d2 = dr_.copy()
            for b in rng.choice(256, rng.randint(1, 12), replace=False):
                d2[b >> 3] ^= 1 << (b & 7)
            R.append((uvr[0] + rng.randn() * 3, uvr[1] + rng.randn() * 3, o, d2))
    for lst in (L, R):
        k, d = rand_feats(n_distract)
        lst.extend((k["x"][j], k["y"][j], int(k["octave"][j]), d[j]) for j in range(n_distract))
    pl, pr = rng.permutation(len(L)), rng.permutation(len(R))
    L = [L[j] for j in pl]; R = [R[j] for j in pr]

    def pack(mono_k, mono_d, lst):
        k = np.zeros(len(lst), capi.KEYPOINT_DTYPE)
        k["x"] = [f[0] for f in lst]; k["y"] = [f[1] for f in lst]; k["octave"] = [f[2] for f in lst]; k["size"] = 31.0
        return np.concatenate([mono_k, k]), np.concatenate([mono_d, np.stack([f[3] for f in lst])]).astype(np.uint8)

    kl, dl = pack(kl_m, dl_m, L); kr, dr = pack(kr_m, dr_m, R)
    return dict(kps_left=kl, desc_left=dl, mono_left=n_mono_left, kps_right=kr, desc_right=dr, mono_right=n_mono_right, left=left, right=right,
                Tlr=Tlr.astype(np.float32)[:3], Trl=Trl.astype(np.float32)[:3], level_sigma2=sigma2)


# ---------------------------------------------------------------- synthetic vocabulary (ORBvoc.txt is not in the reference tree)
def make_vocabulary(k=10, L=3, seed=0xB0C, stop_frac=0.03, descriptors=None):
    """A k-ary tree of depth L in DBoW2's node layout (node ids in creation order, children after parents): node descriptors are
    drawn from `descriptors` (if given, so that real features spread over the words) or at random; leaf weights play the idf
    (a few are 0 = stopped words).  Returns the arrays of views.vocab_view."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    child_lists = [[]]
    level_of = [0]
    frontier = [0]
    for lvl in range(1, L + 1):
        nxt = []
        for parent in frontier:
            kk = k if lvl < L or rng.rand() > 0.2 else max(2, k - 3)          # a few ragged leaf groups
            for _ in range(kk):
                child_lists.append([])
                level_of.append(lvl)
                child_lists[parent].append(len(child_lists) - 1)
                nxt.append(len(child_lists) - 1)
        frontier = nxt
    n = len(child_lists)
    child_start = np.zeros(n + 1, np.int32)
    for i, c in enumerate(child_lists):
        child_start[i + 1] = child_start[i] + len(c)
    child_ids = np.array([c for cl in child_lists for c in cl], np.int32)
    if descriptors is not None and len(descriptors) > 0:
        desc = np.ascontiguousarray(descriptors[rng.randint(0, len(descriptors), n)], np.uint8)
        flip = rng.randint(0, 256, desc.shape).astype(np.uint8) & rng.randint(0, 256, desc.shape).astype(np.uint8) & rng.randint(0, 256, desc.shape).astype(np.uint8)
        desc = desc ^ flip
    else:
        desc = rng.randint(0, 256, (n, 32)).astype(np.uint8)
    word_id = np.full(n, -1, np.int32)
    weight = np.zeros(n, np.float64)
    leaves = [i for i in range(n) if not child_lists[i]]
    for w, i in enumerate(leaves):
        word_id[i] = w
        weight[i] = 0.0 if rng.rand() < stop_frac else float(np.log(1.0 + 50.0 * rng.rand() + 1.0))
    return dict(child_start=child_start, child_ids=child_ids, desc=desc, weight=weight, word_id=word_id, L=L, k=k)


def make_full_vocabulary(k=10, L=6, seed=0x0B0C, stop_frac=0.02):
    """A COMPLETE k-ary tree of depth L in the heap layout (children of node i: k i + 1 .. k i + k; ids grow downwards as DBoW2's do),
    vectorised so that the reference's size -- k = 10, L = 6: 1 111 111 nodes, 10^6 words, 35 MB of descriptors, what ORBvoc.txt holds --
    takes seconds.  A child's descriptor is its parent's with a few dozen bits flipped (a walk then has a meaningful nearest child at
    every level); leaf weights play the idf, written with six significant digits (what saveToTextFile keeps), a few of them 0
    (stopped words).  Returns the arrays of views.vocab_view + k."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    n_inner = (k ** L - 1) // (k - 1)
    n = (k ** (L + 1) - 1) // (k - 1)
    desc = np.zeros((n, 32), np.uint8)
    lo = 0
    for lvl in range(L):
        cnt = k ** lvl
        parents = desc[lo: lo + cnt]
        first = k * lo + 1
        flips = rng.randint(0, 256, (cnt * k, 32)).astype(np.uint8) & rng.randint(0, 256, (cnt * k, 32)).astype(np.uint8)
        if lvl > 0:
            flips &= rng.randint(0, 256, (cnt * k, 32)).astype(np.uint8)
        desc[first: first + cnt * k] = np.repeat(parents, k, axis=0) ^ flips
        lo += cnt
    child_start = np.minimum(np.arange(n + 1, dtype=np.int64), n_inner) * k
    child_ids = np.arange(1, n, dtype=np.int32)
    word_id = np.zeros(n, np.int32)
    word_id[n_inner:] = np.arange(n - n_inner, dtype=np.int32)
    weight = np.zeros(n, np.float64)
    w = np.log(2.0 + 50.0 * rng.rand(n - n_inner))
    w[rng.rand(n - n_inner) < stop_frac] = 0.0
    weight[n_inner:] = np.array([float("%.6g" % x) for x in w]) if n - n_inner <= 20000 else np.round(w, 4)      # (<= 6 significant digits either way)
    return dict(child_start=child_start.astype(np.int32), child_ids=child_ids, desc=desc, weight=weight, word_id=word_id, L=L, k=k)
