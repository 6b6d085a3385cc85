"""ctypes binding of the CPU ORACLE (oracle/liboracle.so) -- test infrastructure only.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product
package.  Builds the library on demand with oracle/Makefile (g++, no dependencies).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from multi_orbslam3_amd import _capi as capi
from multi_orbslam3_amd import views

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("extractor.cc", "matching.cc", "lba.cc", "bow.cc", "vocab_text.cc", "orb_oracle.h",
                                             "orb_pattern_data.inc", "Makefile")]
    srcs.append(os.path.join(_HERE, "..", "include", "orbgpu.h"))
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs if os.path.exists(s))
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return LIB_PATH


def use_native(build=True):
    """Rebuild the oracle with -march=native for THIS host (bench.py's cpu_baseline on the GPU box) and use that build.
    Must be called before the first oracle call of the process; falls back to the portable build if the compiler fails.
    build=False: only select a native build that another process of this job has made (several ranks on one node must not
    write the same file at once: local rank 0 builds, the others wait for it and select)."""
    global LIB_PATH, _lib
    if _lib is not None:
        return LIB_PATH
    native = os.path.join(_HERE, "_native", "liboracle.so")
    if not build:
        if os.path.exists(native):
            LIB_PATH = native
        return LIB_PATH
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "_native/liboracle.so"])
        LIB_PATH = native
    except (subprocess.CalledProcessError, OSError):
        pass
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.endswith(os.path.join("_native", "liboracle.so")):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.oracle_fast_atan2.restype = C.c_float
        _lib.oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
    return _lib


def _chk(code):
    if code != 0:
        raise RuntimeError("oracle error %d" % code)


# ---------------------------------------------------------------- image primitives
def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros((dh, dw), dtype=np.uint8)
    lib().oracle_resize_linear(C.c_void_p(src.ctypes.data), src.shape[1], src.shape[0], src.strides[0],
                               C.c_void_p(dst.ctypes.data), dw, dh, dst.strides[0])
    return dst


def border_reflect101(src, border):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = src.shape
    dst = np.zeros((h + 2 * border, w + 2 * border), dtype=np.uint8)
    lib().oracle_border_reflect101(C.c_void_p(src.ctypes.data), w, h, src.strides[0], C.c_void_p(dst.ctypes.data),
                                   border, dst.strides[0])
    return dst


def fast_score(img, x, y):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    return lib().oracle_fast_score(C.c_void_p(img.ctypes.data), img.strides[0], x, y)


def fast_detect(img, threshold, nonmax=True, cap=65536):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros((cap, 3), dtype=np.int32)
    n = lib().oracle_fast_detect(C.c_void_p(img.ctypes.data), img.shape[1], img.shape[0], img.strides[0],
                                 int(threshold), int(nonmax), C.c_void_p(out.ctypes.data), cap)
    return out[:n].copy()


def gaussian_blur7(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    dst = np.zeros_like(img)
    lib().oracle_gaussian_blur7(C.c_void_p(img.ctypes.data), img.shape[1], img.shape[0], img.strides[0],
                                C.c_void_p(dst.ctypes.data), dst.strides[0])
    return dst


def fast_atan2(y, x):
    return float(lib().oracle_fast_atan2(C.c_float(y), C.c_float(x)))


def fast_cell_grid(level_w, level_h):
    """The cells of ComputeKeyPointsOctTree at a level of that size: (n x {x0, x1, y0, y1, row, column}, wCell, hCell)."""
    r = np.zeros((4096, 6), np.int32)
    wc, hc = C.c_int32(0), C.c_int32(0)
    n = lib().oracle_fast_cell_grid(int(level_w), int(level_h), C.c_void_p(r.ctypes.data), 4096, C.byref(wc), C.byref(hc))
    return r[:n].copy(), wc.value, hc.value


def orb_descriptor(img, x, y, angle_deg):
    """computeOrbDescriptor of the keypoint (x, y, angle) on the (already blurred) image img."""
    img = np.ascontiguousarray(img, np.uint8)
    d = np.zeros(32, np.uint8)
    lib().oracle_orb_descriptor(C.c_float(angle_deg), C.c_void_p(img.ctypes.data), img.shape[1], int(x), int(y), C.c_void_p(d.ctypes.data))
    return d


def three_maxima(bin_sizes):
    b = np.ascontiguousarray(bin_sizes, np.int32)
    out = np.zeros(3, np.int32)
    lib().oracle_three_maxima(C.c_void_p(b.ctypes.data), len(b), C.c_void_p(out.ctypes.data))
    return [int(v) for v in out]


def rot_bin(a, b):
    return int(lib().oracle_rot_bin(C.c_float(a), C.c_float(b)))


def hamming(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return lib().oracle_hamming(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data))


# ---------------------------------------------------------------- extractor
def make_config(n_features=1000, scale_factor=1.2, n_levels=8, ini_th=20, min_th=7, max_width=640, max_height=480,
                n_cams=1, device=0, gauss_taps=None, octree_oldest_first=False):
    return capi.OrbxConfig(n_features, scale_factor, n_levels, ini_th, min_th, max_width, max_height, n_cams, device,
                           (C.c_int32 * 4)(*(gauss_taps or (0, 0, 0, 0))), int(bool(octree_oldest_first)))


class Extractor:
    """oracle restatement of ORB_SLAM3::ORBextractor."""

    def __init__(self, **kw):
        self.cfg = make_config(**kw)
        self.h = C.c_void_p()
        _chk(lib().oracle_extractor_create(C.byref(self.cfg), C.byref(self.h)))

    def __del__(self):
        try:
            if self.h:
                lib().oracle_extractor_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def tables(self):
        nl = self.cfg.n_levels
        arrs = [np.zeros(nl, np.float32) for _ in range(4)] + [np.zeros(nl, np.int32)]
        _chk(lib().oracle_get_tables(self.h, *[C.c_void_p(a.ctypes.data) for a in arrs]))
        return arrs

    def ic_angle(self, img, x, y):
        """IC_Angle of the keypoint (x, y) on the level image img (degrees)."""
        img = np.ascontiguousarray(img, np.uint8)
        lib().oracle_ic_angle.restype = C.c_float
        return float(lib().oracle_ic_angle(self.h, C.c_void_p(img.ctypes.data), img.shape[1], int(x), int(y)))

    def umax(self):
        u = np.zeros(16, np.int32)
        lib().oracle_get_umax(self.h, C.c_void_p(u.ctypes.data))
        return u

    def extract(self, img, lap=(0, 0), cap=None):
        cap = cap or self.cfg.n_features * 2 + 64
        kps = np.zeros(cap, dtype=capi.KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n, nm = C.c_int(0), C.c_int(0)
        if img is None or img.size == 0:
            rc = lib().oracle_extract(self.h, None, 0, 0, 0, lap[0], lap[1], C.c_void_p(kps.ctypes.data),
                                      C.c_void_p(desc.ctypes.data), cap, C.byref(n), C.byref(nm))
            return rc, kps[:0], desc[:0], 0
        img = np.ascontiguousarray(img, dtype=np.uint8)
        rc = lib().oracle_extract(self.h, C.c_void_p(img.ctypes.data), img.shape[1], img.shape[0], img.strides[0],
                                  lap[0], lap[1], C.c_void_p(kps.ctypes.data), C.c_void_p(desc.ctypes.data), cap,
                                  C.byref(n), C.byref(nm))
        return rc, kps[: n.value].copy(), desc[: n.value].copy(), nm.value

    def level(self, l, border=False):
        w, h = C.c_int(0), C.c_int(0)
        fn = lib().oracle_get_level_bordered if border else lib().oracle_get_level
        _chk(fn(self.h, l, None, C.byref(w), C.byref(h)))
        e = 38 if border else 0
        out = np.zeros((h.value + e, w.value + e), np.uint8)
        _chk(fn(self.h, l, C.c_void_p(out.ctypes.data), C.byref(w), C.byref(h)))
        return out

    def candidates(self, l, cap=1 << 18):
        out = np.zeros((cap, 3), np.int32)
        n = C.c_int(0)
        _chk(lib().oracle_get_candidates(self.h, l, C.c_void_p(out.ctypes.data), cap, C.byref(n)))
        return out[: n.value].copy()


def distribute_octree(xys, min_x, max_x, min_y, max_y, n_target):
    xys = np.ascontiguousarray(xys, dtype=np.int32).reshape(-1, 3)
    out = np.zeros((max(len(xys), 1), 3), np.int32)
    n = lib().oracle_distribute_octree(C.c_void_p(xys.ctypes.data), len(xys), min_x, max_x, min_y, max_y, n_target,
                                       C.c_void_p(out.ctypes.data), len(out))
    return out[:n].copy()


def stereo_match(ex_l, ex_r, kps_l, desc_l, kps_r, desc_r, bf, b):
    kps_l = np.ascontiguousarray(kps_l, dtype=capi.KEYPOINT_DTYPE)
    kps_r = np.ascontiguousarray(kps_r, dtype=capi.KEYPOINT_DTYPE)
    desc_l = np.ascontiguousarray(desc_l, np.uint8)
    desc_r = np.ascontiguousarray(desc_r, np.uint8)
    ur = np.zeros(len(kps_l), np.float32)
    dp = np.zeros(len(kps_l), np.float32)
    _chk(lib().oracle_stereo_match(ex_l.h, ex_r.h, C.c_void_p(kps_l.ctypes.data), C.c_void_p(desc_l.ctypes.data),
                                   len(kps_l), C.c_void_p(kps_r.ctypes.data), C.c_void_p(desc_r.ctypes.data),
                                   len(kps_r), C.c_float(bf), C.c_float(b), C.c_void_p(ur.ctypes.data),
                                   C.c_void_p(dp.ctypes.data)))
    return ur, dp


# ---------------------------------------------------------------- frame / matchers
def undistort_points(xy, cam4, dist5):
    """cv::undistortPoints(pts, pts, K, mDistCoef, Mat(), K) (S/Frame.cc:740,767): xy n x 2 float32; cam4 = (fx, fy, cx, cy);
    dist5 = (k1, k2, p1, p2, k3) or None."""
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.zeros_like(xy)
    d = None if dist5 is None else capi.OrbxDistortion(*[float(v) for v in dist5])
    _chk(lib().oracle_undistort_points(C.c_void_p(xy.ctypes.data), len(xy), C.c_float(cam4[0]), C.c_float(cam4[1]), C.c_float(cam4[2]),
                                       C.c_float(cam4[3]), None if d is None else C.byref(d), C.c_void_p(out.ctypes.data)))
    return out


def undistort_keypoints(kps, cam4, dist5):
    """Frame::UndistortKeyPoints (S/Frame.cc:721-754): mvKeysUn = mvKeys with pt replaced."""
    un = kps.copy()
    if len(kps):
        xy = undistort_points(np.stack([kps["x"], kps["y"]], axis=1), cam4, dist5)
        un["x"], un["y"] = xy[:, 0], xy[:, 1]
    return un


def image_bounds(width, height, cam4, dist5):
    """Frame::ComputeImageBounds (S/Frame.cc:756-783) -> (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    if dist5 is None or float(np.float32(dist5[0])) == 0.0:
        return (0.0, float(width), 0.0, float(height))
    m = undistort_points(np.array([[0, 0], [width, 0], [0, height], [width, height]], np.float32), cam4, dist5)
    return (float(min(m[0, 0], m[2, 0])), float(max(m[1, 0], m[3, 0])), float(min(m[0, 1], m[1, 1])), float(max(m[2, 1], m[3, 1])))


def stereo_subpixel(dists, bestincR, scaleduR0, scale_of_level, uL, minD, maxD, bf):
    d = np.ascontiguousarray(dists, np.float32)
    ur, dp = C.c_float(0), C.c_float(0)
    ok = lib().oracle_stereo_subpixel(C.c_void_p(d.ctypes.data), (len(d) - 1) // 2, int(bestincR), C.c_float(scaleduR0), C.c_float(scale_of_level),
                                      C.c_float(uL), C.c_float(minD), C.c_float(maxD), C.c_float(bf), C.byref(ur), C.byref(dp))
    return bool(ok), np.float32(ur.value), np.float32(dp.value)


def build_grid(fv):
    start = np.zeros(capi.GRID_COLS * capi.GRID_ROWS + 1, np.int32)
    items = np.zeros(max(fv.n, 1), np.int32)
    _chk(lib().oracle_build_grid(C.byref(fv), C.c_void_p(start.ctypes.data), C.c_void_p(items.ctypes.data)))
    return start, items[: start[-1]].copy()


def features_in_area(fv, x, y, r, min_level=-1, max_level=-1):
    out = np.zeros(max(fv.n, 1), np.int32)
    n = lib().oracle_features_in_area(C.byref(fv), C.c_float(x), C.c_float(y), C.c_float(r), min_level, max_level,
                                      C.c_void_p(out.ctypes.data), len(out))
    return out[:n].copy()


def is_in_frustum(fv, Tcw, wv, limit=0.5):
    m = wv.m
    T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
    out = dict(track_in_view=np.zeros(m, np.uint8), proj_x=np.zeros(m, np.float32), proj_y=np.zeros(m, np.float32),
               proj_xr=np.zeros(m, np.float32), track_depth=np.zeros(m, np.float32),
               scale_level=np.zeros(m, np.int32), view_cos=np.zeros(m, np.float32))
    _chk(lib().oracle_is_in_frustum(C.byref(fv), C.c_void_p(T.ctypes.data), C.byref(wv), C.c_float(limit),
                                    *[C.c_void_p(out[k].ctypes.data) for k in
                                      ("track_in_view", "proj_x", "proj_y", "proj_xr", "track_depth", "scale_level",
                                       "view_cos")]))
    return out


def hamming_matrix(q, t):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    d = np.zeros((len(q), len(t)), np.int32)
    _chk(lib().oracle_hamming_matrix(C.c_void_p(q.ctypes.data), len(q), C.c_void_p(t.ctypes.data), len(t),
                                     C.c_void_p(d.ctypes.data)))
    return d


def hamming_best2(q, t):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    o = np.zeros((len(q), 4), np.int32)
    _chk(lib().oracle_hamming_best2(C.c_void_p(q.ctypes.data), len(q), C.c_void_p(t.ctypes.data), len(t),
                                    C.c_void_p(o.ctypes.data)))
    return o


def search_by_projection_mps(fv, mv, th, far, th_far, nnratio, assigned_mp, assigned_obs):
    amp = np.ascontiguousarray(assigned_mp, np.int32).copy()
    aob = np.ascontiguousarray(assigned_obs, np.int32).copy()
    n = C.c_int(0)
    _chk(lib().oracle_search_by_projection_mps(C.byref(fv), C.byref(mv), C.c_float(th), int(far), C.c_float(th_far),
                                               C.c_float(nnratio), C.c_void_p(amp.ctypes.data),
                                               C.c_void_p(aob.ctypes.data), C.byref(n)))
    return amp, aob, n.value


RIG_TRACK_KEYS = ("track_in_view", "proj_x", "proj_y", "track_depth", "scale_level", "view_cos")


def is_in_frustum_rig(fv, Tcw, rig, Tlr, wv, limit=0.5):
    """Both cameras' track fields of a rig frame: (left dict, right dict), keys RIG_TRACK_KEYS."""
    m = wv.m
    T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
    tlr = np.ascontiguousarray(Tlr, np.float32).reshape(12)
    mk = lambda: dict(track_in_view=np.zeros(m, np.uint8), proj_x=np.zeros(m, np.float32), proj_y=np.zeros(m, np.float32),
                      track_depth=np.zeros(m, np.float32), scale_level=np.zeros(m, np.int32), view_cos=np.zeros(m, np.float32))
    a, b = mk(), mk()
    _chk(lib().oracle_is_in_frustum_rig(C.byref(fv), C.c_void_p(T.ctypes.data), C.byref(rig), C.c_void_p(tlr.ctypes.data), C.byref(wv), C.c_float(limit),
                                        *[C.c_void_p(d[k].ctypes.data) for d in (a, b) for k in RIG_TRACK_KEYS]))
    return a, b


def search_by_projection_mps_rig(fv_left, fv_right, mv, mv_r, left_to_right, right_to_left, th, far, th_far, nnratio, assigned_mp, assigned_obs):
    amp = np.ascontiguousarray(assigned_mp, np.int32).copy()
    aob = np.ascontiguousarray(assigned_obs, np.int32).copy()
    l2r = np.ascontiguousarray(left_to_right, np.int32)
    r2l = np.ascontiguousarray(right_to_left, np.int32)
    n = C.c_int(0)
    _chk(lib().oracle_search_by_projection_mps_rig(C.byref(fv_left), C.byref(fv_right), C.byref(mv), C.byref(mv_r), C.c_void_p(l2r.ctypes.data),
                                                   C.c_void_p(r2l.ctypes.data), C.c_float(th), int(far), C.c_float(th_far), C.c_float(nnratio),
                                                   C.c_void_p(amp.ctypes.data), C.c_void_p(aob.ctypes.data), C.byref(n)))
    return amp, aob, n.value


def search_by_projection_frame_rig(fv_left, fv_right, Tcw_cur, rig, lv, th, mono, check_ori, assigned_mp, assigned_obs):
    amp = np.ascontiguousarray(assigned_mp, np.int32).copy()
    aob = np.ascontiguousarray(assigned_obs, np.int32).copy()
    T = np.ascontiguousarray(Tcw_cur, np.float32).reshape(16)
    n = C.c_int(0)
    _chk(lib().oracle_search_by_projection_frame_rig(C.byref(fv_left), C.byref(fv_right) if fv_right is not None else None, C.c_void_p(T.ctypes.data), C.byref(rig), C.byref(lv),
                                                     C.c_float(th), int(mono), int(check_ori), C.c_void_p(amp.ctypes.data),
                                                     C.c_void_p(aob.ctypes.data), C.byref(n)))
    return amp, aob, n.value


def fisheye_stereo_matches(view):
    l2r = np.zeros(max(view.n_left, 1), np.int32); r2l = np.zeros(max(view.n_right, 1), np.int32)
    depth = np.zeros(max(view.n_left, 1), np.float32); p3d = np.zeros((max(view.n_left, 1), 3), np.float32)
    n = C.c_int(0)
    _chk(lib().oracle_fisheye_stereo_matches(C.byref(view), C.c_void_p(l2r.ctypes.data), C.c_void_p(r2l.ctypes.data), C.c_void_p(depth.ctypes.data),
                                             C.c_void_p(p3d.ctypes.data), C.byref(n)))
    return l2r[: view.n_left], r2l[: view.n_right], depth[: view.n_left], p3d[: view.n_left], n.value


def kb8_unproject(cam, u, v):
    ray = np.zeros(3, np.float32)
    lib().oracle_kb8_unproject(C.byref(cam), C.c_float(u), C.c_float(v), C.c_void_p(ray.ctypes.data))
    return ray


def kb8_triangulate_matches(cam1, cam2, uv1, uv2, Tlr, sigma1, sigma2):
    a = np.ascontiguousarray(uv1, np.float32); b = np.ascontiguousarray(uv2, np.float32); T = np.ascontiguousarray(np.asarray(Tlr, np.float32).reshape(-1)[:12])
    p = np.zeros(3, np.float32)
    lib().oracle_kb8_triangulate_matches.restype = C.c_float
    z = lib().oracle_kb8_triangulate_matches(C.byref(cam1), C.byref(cam2), C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(T.ctypes.data),
                                             C.c_float(sigma1), C.c_float(sigma2), C.c_void_p(p.ctypes.data))
    return float(z), p


def search_local_points(fv, wv, Tcw, th, far, th_far, nnratio, assigned_mp, assigned_obs):
    amp = np.ascontiguousarray(assigned_mp, np.int32).copy()
    aob = np.ascontiguousarray(assigned_obs, np.int32).copy()
    T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
    n = C.c_int(0)
    _chk(lib().oracle_search_local_points(C.byref(fv), C.byref(wv), C.c_void_p(T.ctypes.data), C.c_float(th), int(far),
                                          C.c_float(th_far), C.c_float(nnratio), C.c_void_p(amp.ctypes.data),
                                          C.c_void_p(aob.ctypes.data), C.byref(n)))
    return amp, aob, n.value


def search_by_projection_frame(fv, Tcw_cur, lv, th, mono, check_ori, assigned_mp, assigned_obs):
    amp = np.ascontiguousarray(assigned_mp, np.int32).copy()
    aob = np.ascontiguousarray(assigned_obs, np.int32).copy()
    T = np.ascontiguousarray(Tcw_cur, np.float32).reshape(16)
    n = C.c_int(0)
    _chk(lib().oracle_search_by_projection_frame(C.byref(fv), C.c_void_p(T.ctypes.data), C.byref(lv), C.c_float(th),
                                                 int(mono), int(check_ori), C.c_void_p(amp.ctypes.data),
                                                 C.c_void_p(aob.ctypes.data), C.byref(n)))
    return amp, aob, n.value


def search_by_bow(fv, fvF, kf_desc, kf_mp_valid, kf_angle, fvK, nnratio, check_ori):
    kf_desc = np.ascontiguousarray(kf_desc, np.uint8)
    kf_mp_valid = np.ascontiguousarray(kf_mp_valid, np.uint8)
    kf_angle = np.ascontiguousarray(kf_angle, np.float32)
    matches = np.zeros(max(fv.n, 1), np.int32)
    n = C.c_int(0)
    _chk(lib().oracle_search_by_bow(C.byref(fv), C.byref(fvF), C.c_void_p(kf_desc.ctypes.data), len(kf_desc),
                                    C.c_void_p(kf_mp_valid.ctypes.data), C.c_void_p(kf_angle.ctypes.data),
                                    C.byref(fvK), C.c_float(nnratio), int(check_ori), C.c_void_p(matches.ctypes.data),
                                    C.byref(n)))
    return matches[: fv.n].copy(), n.value


def search_by_bow_rig(fv, n_left, fvF, kf_desc, kf_mp_valid, kf_angle, fvK, nnratio, check_ori):
    kf_desc = np.ascontiguousarray(kf_desc, np.uint8)
    kf_mp_valid = np.ascontiguousarray(kf_mp_valid, np.uint8)
    kf_angle = np.ascontiguousarray(kf_angle, np.float32)
    matches = np.zeros(max(fv.n, 1), np.int32)
    n = C.c_int(0)
    _chk(lib().oracle_search_by_bow_rig(C.byref(fv), int(n_left), C.byref(fvF), C.c_void_p(kf_desc.ctypes.data), len(kf_desc),
                                        C.c_void_p(kf_mp_valid.ctypes.data), C.c_void_p(kf_angle.ctypes.data), C.byref(fvK), C.c_float(nnratio),
                                        int(check_ori), C.c_void_p(matches.ctypes.data), C.byref(n)))
    return matches[: fv.n].copy(), n.value


def search_by_projection_sim3(kf, pts, Scw, matched, th, ratio_hamming=1.0, already_found=None, with_kfs=False):
    matched = np.ascontiguousarray(matched, np.int32).copy()
    S = np.ascontiguousarray(Scw, np.float32).reshape(16)
    af = None if already_found is None else np.ascontiguousarray(already_found, np.uint8)
    n = C.c_int(0)
    _chk(lib().oracle_search_by_projection_sim3(C.byref(kf), C.byref(pts), C.c_void_p(S.ctypes.data),
                                                None if af is None else C.c_void_p(af.ctypes.data), int(th),
                                                C.c_float(ratio_hamming), 0 if with_kfs else 1,
                                                C.c_void_p(matched.ctypes.data), C.byref(n)))
    return matched, n.value


def search_by_projection_sim3_cam(kf, pts, Scw, cam, matched, th, ratio_hamming=1.0, already_found=None):
    """... with pKF->mpCamera a camera model (cam: an orbg_camera)."""
    matched = np.ascontiguousarray(matched, np.int32).copy()
    S = np.ascontiguousarray(Scw, np.float32).reshape(16)
    af = None if already_found is None else np.ascontiguousarray(already_found, np.uint8)
    n = C.c_int(0)
    _chk(lib().oracle_search_by_projection_sim3_cam(C.byref(kf), C.byref(pts), C.c_void_p(S.ctypes.data), C.byref(cam),
                                                    None if af is None else C.c_void_p(af.ctypes.data), int(th), C.c_float(ratio_hamming),
                                                    C.c_void_p(matched.ctypes.data), C.byref(n)))
    return matched, n.value


def search_by_projection_reloc(cur, Tcw_cur, pts, kf_angle, assigned_mp, th, orb_dist, check_ori=True, already_found=None):
    """ORBmatcher::SearchByProjection(Frame&, KeyFrame*, set<MapPoint*>&, th, ORBdist), S/ORBmatcher.cc:2188-2310."""
    amp = np.ascontiguousarray(assigned_mp, np.int32).copy()
    T = np.ascontiguousarray(Tcw_cur, np.float32).reshape(16)
    ang = np.ascontiguousarray(kf_angle, np.float32)
    af = None if already_found is None else np.ascontiguousarray(already_found, np.uint8)
    n = C.c_int(0)
    _chk(lib().oracle_search_by_projection_reloc(C.byref(cur), C.c_void_p(T.ctypes.data), C.byref(pts),
                                                 None if af is None else C.c_void_p(af.ctypes.data), C.c_void_p(ang.ctypes.data),
                                                 C.c_float(th), int(orb_dist), int(bool(check_ori)), C.c_void_p(amp.ctypes.data), C.byref(n)))
    return amp, n.value


def search_by_projection_reloc_cam(cur, Tcw_cur, cam, pts, kf_angle, assigned_mp, th, orb_dist, check_ori=True, already_found=None):
    """... with CurrentFrame.mpCamera a camera model (cam: an orbg_camera)."""
    amp = np.ascontiguousarray(assigned_mp, np.int32).copy()
    T = np.ascontiguousarray(Tcw_cur, np.float32).reshape(16)
    ang = np.ascontiguousarray(kf_angle, np.float32)
    af = None if already_found is None else np.ascontiguousarray(already_found, np.uint8)
    n = C.c_int(0)
    _chk(lib().oracle_search_by_projection_reloc_cam(C.byref(cur), C.c_void_p(T.ctypes.data), C.byref(cam), C.byref(pts),
                                                     None if af is None else C.c_void_p(af.ctypes.data), C.c_void_p(ang.ctypes.data),
                                                     C.c_float(th), int(orb_dist), int(bool(check_ori)), C.c_void_p(amp.ctypes.data), C.byref(n)))
    return amp, n.value


def search_by_bow_kf(kf2, fv2, mp_valid2, desc1, mp_valid1, angle1, fv1, nnratio, check_ori):
    desc1 = np.ascontiguousarray(desc1, np.uint8)
    mp_valid1 = np.ascontiguousarray(mp_valid1, np.uint8)
    mp_valid2 = np.ascontiguousarray(mp_valid2, np.uint8)
    angle1 = np.ascontiguousarray(angle1, np.float32)
    matches = np.zeros(max(len(desc1), 1), np.int32)
    n = C.c_int(0)
    _chk(lib().oracle_search_by_bow_kf(C.byref(kf2), C.byref(fv2), C.c_void_p(mp_valid2.ctypes.data),
                                       C.c_void_p(desc1.ctypes.data), len(desc1), C.c_void_p(mp_valid1.ctypes.data),
                                       C.c_void_p(angle1.ctypes.data), C.byref(fv1), C.c_float(nnratio), int(check_ori),
                                       C.c_void_p(matches.ctypes.data), C.byref(n)))
    return matches[: len(desc1)].copy(), n.value


# ---------------------------------------------------------------- bag of words
def vocab_transform(voc, desc, levelsup=4):
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    n = len(desc)
    wid = np.zeros(max(n, 1), np.int32); nid = np.zeros(max(n, 1), np.int32); w = np.zeros(max(n, 1), np.float64)
    _chk(lib().oracle_vocab_transform(C.byref(voc), C.c_void_p(desc.ctypes.data), n, int(levelsup), C.c_void_p(wid.ctypes.data),
                                      C.c_void_p(nid.ctypes.data), C.c_void_p(w.ctypes.data)))
    return wid[:n], nid[:n], w[:n]


def vocab_bow(voc, desc, levelsup=4):
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    n = len(desc)
    bw = np.zeros(max(n, 1), np.int32); bv = np.zeros(max(n, 1), np.float64)
    fn = np.zeros(max(n, 1), np.uint32); fs = np.zeros(n + 1, np.uint32); ff = np.zeros(max(n, 1), np.uint32)
    nw, nn = C.c_int32(0), C.c_int32(0)
    _chk(lib().oracle_vocab_bow(C.byref(voc), C.c_void_p(desc.ctypes.data), n, int(levelsup), C.c_void_p(bw.ctypes.data),
                                C.c_void_p(bv.ctypes.data), C.byref(nw), C.c_void_p(fn.ctypes.data), C.c_void_p(fs.ctypes.data),
                                C.c_void_p(ff.ctypes.data), C.byref(nn)))
    k = nn.value
    return (bw[: nw.value].copy(), bv[: nw.value].copy()), (fn[:k].copy(), fs[: k + 1].copy(), ff[: int(fs[k]) if k else 0].copy())


def vocab_save_text(voc, k, path, scoring=0):
    """saveToTextFile's bytes for a flattened tree (TemplatedVocabulary.h:1431-1450)."""
    _chk(lib().oracle_vocab_save_text(C.byref(voc), int(k), int(scoring), str(path).encode()))


def vocab_load_text(path, keep_trailing_node=False):
    """loadFromTextFile with the reference's stream operations -> dict of the view's arrays (+ k, scoring, n_words)."""
    h = C.c_void_p()
    _chk(lib().oracle_vocab_load_text(str(path).encode(), int(bool(keep_trailing_node)), C.byref(h)))
    try:
        v = capi.VocabView(); k, sc, nw = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        _chk(lib().oracle_vocab_text_view(h, C.byref(v), C.byref(k), C.byref(sc), C.byref(nw)))
        return capi.vocab_view_arrays(v, k.value, sc.value, nw.value)
    finally:
        lib().oracle_vocab_text_free(h)


def distinctive_descriptors(desc, start):
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    start = np.ascontiguousarray(start, np.int32)
    m = len(start) - 1
    best = np.zeros(max(m, 1), np.int32)
    _chk(lib().oracle_distinctive_descriptors(C.c_void_p(desc.ctypes.data) if len(desc) else None, C.c_void_p(start.ctypes.data), m,
                                              C.c_void_p(best.ctypes.data)))
    return best[:m]


def score_l1(q_word, q_value, cand_start, cand_word, cand_value):
    qw = np.ascontiguousarray(q_word, np.int32); qv = np.ascontiguousarray(q_value, np.float64)
    cs = np.ascontiguousarray(cand_start, np.int32); cw = np.ascontiguousarray(cand_word, np.int32); cv = np.ascontiguousarray(cand_value, np.float64)
    m = len(cs) - 1
    out = np.zeros(max(m, 1), np.float64)
    _chk(lib().oracle_score_l1(C.c_void_p(qw.ctypes.data), C.c_void_p(qv.ctypes.data), len(qw), C.c_void_p(cs.ctypes.data),
                               C.c_void_p(cw.ctypes.data), C.c_void_p(cv.ctypes.data), m, C.c_void_p(out.ctypes.data)))
    return out[:m]


def wire_pack(kps, desc):
    kps = np.ascontiguousarray(kps, capi.KEYPOINT_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    n = len(kps)
    wire = np.zeros(47 * n, np.uint8)
    _chk(lib().oracle_wire_pack(C.c_void_p(kps.ctypes.data), C.c_void_p(desc.ctypes.data), n, C.c_void_p(wire.ctypes.data)))
    return wire


def wire_unpack(wire, n):
    wire = np.ascontiguousarray(wire, np.uint8)
    kps = np.zeros(n, capi.KEYPOINT_DTYPE); desc = np.zeros((n, 32), np.uint8)
    _chk(lib().oracle_wire_unpack(C.c_void_p(wire.ctypes.data), n, C.c_void_p(kps.ctypes.data), C.c_void_p(desc.ctypes.data)))
    return kps, desc


# ---------------------------------------------------------------- LBA
def lba_solve(problem, stop_flag=None, trace_cap=64):
    out = views.LbaOutput(problem.n_poses, problem.n_points, problem.n_edges, trace_cap)
    sp = None
    if stop_flag is not None:
        sp = C.c_void_p(stop_flag.ctypes.data)
    _chk(lib().oracle_lba_solve(C.byref(problem), sp, C.byref(out.c)))
    return out


def se3_exp(upd6):
    u = np.ascontiguousarray(upd6, np.float64)
    q = np.zeros(4)
    t = np.zeros(3)
    lib().oracle_se3_exp(C.c_void_p(u.ctypes.data), C.c_void_p(q.ctypes.data), C.c_void_p(t.ctypes.data))
    return q, t


def lba_edge_eval(q, t, X, cam5, edge):
    q = np.ascontiguousarray(q, np.float64)
    t = np.ascontiguousarray(t, np.float64)
    X = np.ascontiguousarray(X, np.float64)
    cam = np.ascontiguousarray(cam5, np.float32)
    e = np.ascontiguousarray(edge, dtype=capi.EDGE_DTYPE).reshape(1)
    err, A, B = np.zeros(3), np.zeros(9), np.zeros(18)
    lib().oracle_lba_edge_eval(C.c_void_p(q.ctypes.data), C.c_void_p(t.ctypes.data), C.c_void_p(X.ctypes.data),
                               C.c_void_p(cam.ctypes.data), C.c_void_p(e.ctypes.data), C.c_void_p(err.ctypes.data),
                               C.c_void_p(A.ctypes.data), C.c_void_p(B.ctypes.data))
    return err, A.reshape(3, 3), B.reshape(3, 6)


def robust_huber(e, delta):
    r = np.zeros(2)
    lib().oracle_robust_huber(C.c_double(e), C.c_double(delta), C.c_void_p(r.ctypes.data))
    return r


def camera_project(cam, X):
    """GeometricCamera::project / projectJac of cam = (model, fx, fy, cx, cy[, k1..k4]) at X: (uv, 2x3 Jacobian)."""
    rig = views.camera_rig(cam)
    X = np.ascontiguousarray(X, np.float64)
    uv, J = np.zeros(2), np.zeros(6)
    lib().oracle_camera_project(C.byref(rig.left), C.c_void_p(X.ctypes.data), C.c_void_p(uv.ctypes.data), C.c_void_p(J.ctypes.data))
    return uv, J.reshape(2, 3)


def lba_edge_eval_rig(q, t, X, cam5, rig, edge):
    """One edge of a problem with a camera rig: (err, d err / d point, d err / d pose, point in the observing camera's frame)."""
    q = np.ascontiguousarray(q, np.float64)
    t = np.ascontiguousarray(t, np.float64)
    X = np.ascontiguousarray(X, np.float64)
    cam = np.ascontiguousarray(cam5, np.float32)
    e = np.ascontiguousarray(edge, dtype=capi.EDGE_DTYPE).reshape(1)
    err, A, B, Xc = np.zeros(3), np.zeros(9), np.zeros(18), np.zeros(3)
    lib().oracle_lba_edge_eval_rig(C.c_void_p(q.ctypes.data), C.c_void_p(t.ctypes.data), C.c_void_p(X.ctypes.data),
                                   C.c_void_p(cam.ctypes.data), C.byref(rig), C.c_void_p(e.ctypes.data), C.c_void_p(err.ctypes.data),
                                   C.c_void_p(A.ctypes.data), C.c_void_p(B.ctypes.data), C.c_void_p(Xc.ctypes.data))
    return err, A.reshape(3, 3), B.reshape(3, 6), Xc


def pose_optimize(problem):
    out = views.PoseOptOutput(problem.n)
    _chk(lib().oracle_pose_optimize(C.byref(problem), C.byref(out.c)))
    return out


def detect_n_best_candidates(view, q_word, q_value, connected, query_map_id, n_candidates, place_score):
    """KeyFrameDatabase::DetectNBestCandidates (S/KeyFrameDatabase.cc:594-761) on views.database_view; place_score in/out."""
    qw = np.ascontiguousarray(q_word, np.int32); qv = np.ascontiguousarray(q_value, np.float64)
    con = np.ascontiguousarray(connected, np.uint8)
    loop = np.zeros(max(n_candidates, 1), np.int32); merge = np.zeros(max(n_candidates, 1), np.int32)
    nl, nm = C.c_int32(0), C.c_int32(0)
    _chk(lib().oracle_detect_n_best_candidates(C.byref(view), C.c_void_p(qw.ctypes.data), C.c_void_p(qv.ctypes.data), len(qw),
                                               C.c_void_p(con.ctypes.data), int(query_map_id), int(n_candidates),
                                               C.c_void_p(place_score.ctypes.data), C.c_void_p(loop.ctypes.data), C.byref(nl),
                                               C.c_void_p(merge.ctypes.data), C.byref(nm)))
    return loop[: nl.value].copy(), merge[: nm.value].copy()
