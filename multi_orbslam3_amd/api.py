"""Python host mirror of the reference's interface for the hot path, over the C-ABI of liborbgpu.so.

Class / method names follow the reference (ORBextractor::operator(), Frame::ComputeStereoMatches,
ORBmatcher::SearchByProjection / SearchByBoW, Optimizer::LocalBundleAdjustment); arguments are the flattened
numpy views of SURVEY.md Appendix E.  Every method raises OrbGpuError on a non-zero status -- in particular
ORBG_NO_DEVICE when no MI355X is visible: there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _capi as capi
from . import views


def _vp(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class ORBextractor:
    """ORB_SLAM3::ORBextractor (I/ORBextractor.h:47-113).  n_cams=2 gives the batched stereo rig."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, max_width=640,
                 max_height=480, n_cams=1, device=0, gauss_taps=None, octree_oldest_first=False):
        """gauss_taps / octree_oldest_first: the two deployment variants of orbx_config (OpenCV >= 4.5: (18, 34, 48, 56))."""
        self.lib = capi.load()
        self.cfg = capi.OrbxConfig(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_width, max_height,
                                   n_cams, device, (C.c_int32 * 4)(*(gauss_taps or (0, 0, 0, 0))), int(bool(octree_oldest_first)))
        self.h = C.c_void_p()
        capi.check(self.lib.orbx_create(C.byref(self.cfg), C.byref(self.h)), "orbx_create")
        self.cap = 2 * nfeatures + 256

    def close(self):
        if getattr(self, "h", None):
            self.lib.orbx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # getters of I/ORBextractor.h:65-85
    def tables(self):
        nl = self.cfg.n_levels
        arrs = [np.zeros(nl, np.float32) for _ in range(4)] + [np.zeros(nl, np.int32)]
        capi.check(self.lib.orbx_get_tables(self.h, *[_vp(a) for a in arrs]))
        return arrs

    def GetScaleFactors(self):
        return self.tables()[0]

    def GetInverseScaleSigmaSquares(self):
        return self.tables()[3]

    def __call__(self, image, vLappingArea=(0, 0)):
        """-> (monoIndex or -1, keypoints, descriptors): S/ORBextractor.cc:1068-1150."""
        kps = np.zeros(self.cap, dtype=capi.KEYPOINT_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n, nm = C.c_int(0), C.c_int(0)
        if image is None or image.size == 0:
            rc = self.lib.orbx_extract(self.h, 0, None, 0, 0, 0, 0, 0, _vp(kps), _vp(desc), self.cap, C.byref(n), C.byref(nm))
            assert rc == capi.ORBG_EMPTY
            return -1, kps[:0], desc[:0]
        image = np.ascontiguousarray(image, np.uint8)
        rc = self.lib.orbx_extract(self.h, 0, _vp(image), image.shape[1], image.shape[0], image.strides[0],
                                   int(vLappingArea[0]), int(vLappingArea[1]), _vp(kps), _vp(desc), self.cap,
                                   C.byref(n), C.byref(nm))
        capi.check(rc, "orbx_extract")
        return nm.value, kps[: n.value].copy(), desc[: n.value].copy()

    def extract_stereo(self, im_left, im_right, download=True):
        """Both ExtractORB calls of the stereo Frame ctor (S/Frame.cc:92-95) in one batched submission."""
        im_left = np.ascontiguousarray(im_left, np.uint8)
        im_right = np.ascontiguousarray(im_right, np.uint8)
        assert im_left.shape == im_right.shape and im_left.strides == im_right.strides
        nl, nr = C.c_int(0), C.c_int(0)
        if download:
            kl = np.zeros(self.cap, capi.KEYPOINT_DTYPE); dl = np.zeros((self.cap, 32), np.uint8)
            kr = np.zeros(self.cap, capi.KEYPOINT_DTYPE); dr = np.zeros((self.cap, 32), np.uint8)
        else:
            kl = dl = kr = dr = None
        rc = self.lib.orbx_extract_stereo(self.h, _vp(im_left), _vp(im_right), im_left.shape[1], im_left.shape[0],
                                          im_left.strides[0], _vp(kl), _vp(dl), self.cap, C.byref(nl), _vp(kr), _vp(dr),
                                          self.cap, C.byref(nr))
        capi.check(rc, "orbx_extract_stereo")
        if not download:
            return nl.value, nr.value
        return (kl[: nl.value].copy(), dl[: nl.value].copy()), (kr[: nr.value].copy(), dr[: nr.value].copy())

    def extract_stereo_dev(self, d_left, d_right, width, height, stride, download_left=False):
        """Images already in HBM (raw device pointers, e.g. torch tensor .data_ptr()).  Right-camera features always
        stay on the device; the left ones are also copied to the host when download_left is set."""
        nl, nr = C.c_int(0), C.c_int(0)
        kl = dl = None
        if download_left:
            if not hasattr(self, "_kl"):
                self._kl = np.zeros(self.cap, capi.KEYPOINT_DTYPE)
                self._dl = np.zeros((self.cap, 32), np.uint8)
            kl, dl = self._kl, self._dl
        rc = self.lib.orbx_extract_stereo_dev(self.h, C.c_void_p(d_left), C.c_void_p(d_right), width, height, stride,
                                              _vp(kl), _vp(dl), self.cap, C.byref(nl), None, None, 0, C.byref(nr))
        capi.check(rc, "orbx_extract_stereo_dev")
        if download_left:
            return nl.value, nr.value, kl[: nl.value], dl[: nl.value]
        return nl.value, nr.value

    def frame_stereo(self, frame, fv, im_left, im_right, bf, b, download=True):
        """Host-image variant of frame_stereo_dev (the actual Frame ctor hands over cv::Mat images)."""
        im_left = np.ascontiguousarray(im_left, np.uint8)
        im_right = np.ascontiguousarray(im_right, np.uint8)
        self._host_imgs = (im_left, im_right)
        return self.frame_stereo_dev(frame, fv, im_left.ctypes.data, im_right.ctypes.data, im_left.shape[1], im_left.shape[0],
                                     im_left.strides[0], bf, b, download, _fn=self.lib.orbx_frame_stereo)

    def frame_stereo_dev_submit(self, frame, fv, d_left, d_right, width, height, stride, bf, b):
        """First half of frame_stereo_dev: enqueue the Frame constructor and return; other handles / frames may be used
        until frame_stereo_dev_wait (the chain overlaps with their kernels on the GPU)."""
        self._pending = (frame, fv)                    # keep the view alive
        rc = self.lib.orbx_frame_stereo_dev_submit(self.h, frame.h if frame is not None else None, C.byref(fv), C.c_void_p(d_left),
                                                   C.c_void_p(d_right), width, height, stride, C.c_float(bf), C.c_float(b))
        capi.check(rc, "orbx_frame_stereo_dev_submit")

    def frame_stereo_submit(self, frame, fv, im_left, im_right, bf, b, async_ingest=False):
        """First half of the Frame constructor with HOST images (orbx_frame_stereo_submit): rows packed into the handle's
        pinned staging slot, one copy kernel + the constructor chain enqueued; collect with frame_stereo_dev_wait().  With
        async_ingest the packing and the launches run on the library's ingest thread and the call returns at once (the
        images are kept alive here until the wait)."""
        assert im_left.dtype == np.uint8 and im_right.dtype == np.uint8 and im_left.shape == im_right.shape
        assert im_left.strides[1] == 1 and im_right.strides == im_left.strides
        self._pending = (frame, fv, im_left, im_right)
        rc = self.lib.orbx_frame_stereo_submit(self.h, frame.h if frame is not None else None, C.byref(fv), C.c_void_p(im_left.ctypes.data),
                                               C.c_void_p(im_right.ctypes.data), im_left.shape[1], im_left.shape[0], im_left.strides[0],
                                               C.c_float(bf), C.c_float(b), 1 if async_ingest else 0)
        capi.check(rc, "orbx_frame_stereo_submit")

    def set_frame_outputs(self, cap):
        """orbx_set_frame_outputs: the two-halves constructor delivers mvKeys / mDescriptors / mvuRight / mvDepth of the left image
        into host arrays owned by this object (returned) by the time frame_stereo_dev_wait() returns; cap = 0 switches it off."""
        if cap <= 0:
            capi.check(self.lib.orbx_set_frame_outputs(self.h, None, None, None, None, 0), "orbx_set_frame_outputs")
            self._outputs = None
            return None
        out = dict(kps=np.zeros(cap, capi.KEYPOINT_DTYPE), desc=np.zeros((cap, 32), np.uint8), uright=np.zeros(cap, np.float32),
                   depth=np.zeros(cap, np.float32), kps_un=np.zeros(cap, capi.KEYPOINT_DTYPE))
        capi.check(self.lib.orbx_set_frame_outputs(self.h, C.c_void_p(capi.ptr(out["kps"])), C.c_void_p(capi.ptr(out["desc"])),
                                                   C.c_void_p(capi.ptr(out["uright"])), C.c_void_p(capi.ptr(out["depth"])), int(cap)),
                   "orbx_set_frame_outputs")
        # (mvKeysUn: delivered by the monocular constructor only)
        capi.check(self.lib.orbx_set_frame_outputs_un(self.h, C.c_void_p(capi.ptr(out["kps_un"]))), "orbx_set_frame_outputs_un")
        self._outputs = out
        return out

    # ---- the monocular Frame constructor (S/Frame.cc:260-358)
    @staticmethod
    def _dist(dist):
        """(k1, k2, p1, p2[, k3]) / capi.OrbxDistortion / None -> (keep-alive object, ctypes argument)"""
        if dist is None:
            return None, None
        d = dist if isinstance(dist, capi.OrbxDistortion) else capi.OrbxDistortion(*([float(v) for v in dist] + [0.0] * (5 - len(dist))))
        return d, C.byref(d)

    def frame_mono(self, frame, fv, image, dist=None, download=True, device_ptr=None, size=None):
        """Frame::Frame(mono): ExtractORB(0, im, 0, 1000) + UndistortKeyPoints + grid in ONE submission (orbx_frame_mono);
        -> (n, mvKeys, mvKeysUn, mDescriptors) or n.  device_ptr/size=(w, h, stride): the image is resident in HBM."""
        n = C.c_int(0)
        kps = kun = desc = None
        if download:
            kps = np.zeros(self.cap, capi.KEYPOINT_DTYPE); kun = np.zeros(self.cap, capi.KEYPOINT_DTYPE); desc = np.zeros((self.cap, 32), np.uint8)
        dk, darg = self._dist(dist)
        if device_ptr is not None:
            w, h, stride = size
            rc = self.lib.orbx_frame_mono_dev(self.h, frame.h if frame is not None else None, C.byref(fv), darg, C.c_void_p(device_ptr), w, h, stride,
                                              _vp(kps), _vp(kun), _vp(desc), self.cap, C.byref(n))
        else:
            image = np.ascontiguousarray(image, np.uint8)
            rc = self.lib.orbx_frame_mono(self.h, frame.h if frame is not None else None, C.byref(fv), darg, _vp(image), image.shape[1],
                                          image.shape[0], image.strides[0], _vp(kps), _vp(kun), _vp(desc), self.cap, C.byref(n))
        capi.check(rc, "orbx_frame_mono")
        if frame is not None:
            frame.n = n.value
        if download:
            return n.value, kps[: n.value].copy(), kun[: n.value].copy(), desc[: n.value].copy()
        return n.value

    def frame_mono_submit(self, frame, fv, image, dist=None, async_ingest=False, device_ptr=None, size=None):
        """First half of the monocular constructor (orbx_frame_mono_submit / _dev_submit); collect with frame_mono_wait()."""
        dk, darg = self._dist(dist)
        if device_ptr is not None:
            w, h, stride = size
            self._pending = (frame, fv, dk)
            rc = self.lib.orbx_frame_mono_dev_submit(self.h, frame.h if frame is not None else None, C.byref(fv), darg, C.c_void_p(device_ptr), w, h, stride)
        else:
            assert image.dtype == np.uint8 and image.strides[1] == 1
            self._pending = (frame, fv, dk, image)
            rc = self.lib.orbx_frame_mono_submit(self.h, frame.h if frame is not None else None, C.byref(fv), darg, C.c_void_p(image.ctypes.data),
                                                 image.shape[1], image.shape[0], image.strides[0], 1 if async_ingest else 0)
        capi.check(rc, "orbx_frame_mono_submit")

    def frame_mono_wait(self):
        return self.frame_stereo_dev_wait()[0]

    def frame_stereo_dev_wait(self):
        nl, nr = C.c_int(0), C.c_int(0)
        capi.check(self.lib.orbx_frame_stereo_dev_wait(self.h, C.byref(nl), C.byref(nr)), "orbx_frame_stereo_dev_wait")
        frame = self._pending[0] if getattr(self, "_pending", None) else None
        if frame is not None:
            frame.n = nl.value
        self._pending = None
        return nl.value, nr.value

    def frame_stereo_dev(self, frame, fv, d_left, d_right, width, height, stride, bf, b, download=False, _fn=None):
        """Frame::Frame(stereo) (S/Frame.cc:71-172) in one submission: extract L+R, ComputeStereoMatches and the
        feature grid, one final sync; `frame` views the left features on the device afterwards."""
        nl, nr = C.c_int(0), C.c_int(0)
        kl = dl = ur = dp = None
        if download:
            if not hasattr(self, "_kl"):
                self._kl = np.zeros(self.cap, capi.KEYPOINT_DTYPE)
                self._dl = np.zeros((self.cap, 32), np.uint8)
            if not hasattr(self, "_ur"):
                self._ur = np.zeros(self.cap, np.float32)
                self._dp = np.zeros(self.cap, np.float32)
            kl, dl, ur, dp = self._kl, self._dl, self._ur, self._dp
        fn = _fn if _fn is not None else self.lib.orbx_frame_stereo_dev
        rc = fn(self.h, frame.h if frame is not None else None, C.byref(fv), C.c_void_p(d_left),
                C.c_void_p(d_right), width, height, stride, C.c_float(bf), C.c_float(b), _vp(kl),
                _vp(dl), _vp(ur), _vp(dp), self.cap, C.byref(nl), C.byref(nr))
        capi.check(rc, "orbx_frame_stereo_dev")
        if frame is not None:
            frame.n = nl.value
        if download:
            n = nl.value
            return n, nr.value, kl[:n], dl[:n], ur[:n], dp[:n]
        return nl.value, nr.value

    def level(self, cam, level, border=False):
        """mvImagePyramid[level] (I/ORBextractor.h:87) of the last extraction; border=True includes the 19-px border."""
        w, h = C.c_int(0), C.c_int(0)
        fn = self.lib.orbx_get_level_bordered if border else self.lib.orbx_get_level
        capi.check(fn(self.h, cam, level, None, C.byref(w), C.byref(h)))
        e = 38 if border else 0
        out = np.zeros((h.value + e, w.value + e), np.uint8)
        capi.check(fn(self.h, cam, level, _vp(out), C.byref(w), C.byref(h)))
        return out

    def candidates(self, cam, level, cap=1 << 18):
        out = np.zeros((cap, 3), np.int32)
        n = C.c_int(0)
        capi.check(self.lib.orbx_get_candidates(self.h, cam, level, _vp(out), cap, C.byref(n)))
        return out[: n.value].copy()

    def ComputeStereoMatches(self, bf, b, n_left=None, download=True):
        """Frame::ComputeStereoMatches (S/Frame.cc:785-963) on the device-resident stereo extraction."""
        if not download:
            capi.check(self.lib.orbx_stereo_match(self.h, C.c_float(bf), C.c_float(b), None, None))
            return None
        ur = np.zeros(max(n_left if n_left is not None else self.cap, 1), np.float32)
        dp = np.zeros_like(ur)
        capi.check(self.lib.orbx_stereo_match(self.h, C.c_float(bf), C.c_float(b), _vp(ur), _vp(dp)), "orbx_stereo_match")
        if n_left is not None:
            return ur[:n_left], dp[:n_left]
        return ur, dp

    def ctor_timeline(self, reset=False):
        """Host-side timeline of the last Frame constructors of this handle: array [n, 5] of microseconds
        (queue, pack, enqueue, wait, latency), oldest first (orbx_get_ctor_timeline)."""
        out = np.zeros((512, 5), np.float32)
        n = C.c_int(0)
        capi.check(self.lib.orbx_get_ctor_timeline(self.h, _vp(out), 512, C.byref(n), int(bool(reset))), "orbx_get_ctor_timeline")
        return out[: n.value].copy()

    def set_profiling(self, level):
        capi.check(self.lib.orbx_set_profiling(self.h, int(level)))

    def set_profile_interval(self, interval, reset=True):
        capi.check(self.lib.orbx_set_profile_interval(self.h, int(interval), int(bool(reset))), "orbx_set_profile_interval")

    PROF_KERNELS = {"fast_cells_kernel": 0, "octree_kernel": 1, "orient_desc_gpu_kernel": 2, "pyr_tower_kernel": 3}

    def set_profile_kernel(self, name):
        """Which kernel of the constructor chain the level-1 event pair brackets (resets the accumulated times)."""
        capi.check(self.lib.orbx_set_profile_kernel(self.h, self.PROF_KERNELS[name]), "orbx_set_profile_kernel")

    def fast_kernel_stats(self):
        s, n = C.c_double(0.0), C.c_int64(0)
        capi.check(self.lib.orbx_get_fast_kernel_stats(self.h, C.byref(s), C.byref(n)), "orbx_get_fast_kernel_stats")
        return s.value, n.value

    def event_overhead_ms(self, reps=50):
        ms = C.c_float(0.0)
        capi.check(self.lib.orbx_event_overhead(self.h, int(reps), C.byref(ms)), "orbx_event_overhead")
        return ms.value

    def timings(self):
        t = np.zeros(8, np.float32)
        capi.check(self.lib.orbx_get_timings(self.h, _vp(t)))
        return dict(pyramid_ms=float(t[0]), fast_ms=float(t[1]), octree_host_ms=float(t[2]), desc_ms=float(t[3]),
                    stereo_ms=float(t[4]), fast_kernel_ms=float(t[5]))


def undistort_points(xy, cam4, dist, device=0):
    """cv::undistortPoints(pts, pts, K, mDistCoef, Mat(), K) on the device (orbx_undistort_points): xy n x 2 float32."""
    lib = capi.load()
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.zeros_like(xy)
    dk, darg = ORBextractor._dist(dist)
    capi.check(lib.orbx_undistort_points(int(device), _vp(xy), len(xy), C.c_float(cam4[0]), C.c_float(cam4[1]), C.c_float(cam4[2]),
                                         C.c_float(cam4[3]), darg, _vp(out)), "orbx_undistort_points")
    return out


def image_bounds(width, height, cam4, dist, device=0):
    """Frame::ComputeImageBounds (S/Frame.cc:756-783) -> (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    if dist is None or float(np.float32(dist[0] if not isinstance(dist, capi.OrbxDistortion) else dist.k1)) == 0.0:
        return (0.0, float(width), 0.0, float(height))
    m = undistort_points(np.array([[0, 0], [width, 0], [0, height], [width, height]], np.float32), cam4, dist, device)
    return (float(min(m[0, 0], m[2, 0])), float(max(m[1, 0], m[3, 0])), float(min(m[0, 1], m[1, 1])), float(max(m[2, 1], m[3, 1])))


class Frame:
    """Device-resident frame view (features + 64x48 grid) the matchers work on."""

    def __init__(self, cap_features=4096, device=0):
        self.lib = capi.load()
        self.h = C.c_void_p()
        capi.check(self.lib.orbm_frame_create(device, cap_features, C.byref(self.h)), "orbm_frame_create")
        self.n = 0
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.orbm_frame_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, fv, keep=None):
        self._keep = keep
        capi.check(self.lib.orbm_frame_upload(self.h, C.byref(fv)), "orbm_frame_upload")
        self.n = fv.n
        return self

    def from_extractor(self, extractor, fv, n_left):
        """Features, uRight and depth are taken device-to-device from the extractor's left camera."""
        fv.n = int(n_left)
        capi.check(self.lib.orbm_frame_from_extractor(self.h, extractor.h, C.byref(fv)), "orbm_frame_from_extractor")
        self.n = int(n_left)
        return self

    # ---- KeyFrame wire blocks (orb_slam3_ros/KF: N x CvKeyPoint 15 B + N x Descriptor 32 B)
    def pack_wire(self, device_ptr=None):
        """Features -> wire block; into device memory at device_ptr (for the RCCL exchange) or returned as a numpy array."""
        if device_ptr is not None:
            capi.check(self.lib.orbk_pack_frame(self.h, C.c_void_p(int(device_ptr)), 1), "orbk_pack_frame")
            return None
        wire = np.zeros(max(47 * self.n, 1), np.uint8)
        capi.check(self.lib.orbk_pack_frame(self.h, _vp(wire), 0), "orbk_pack_frame")
        return wire[: 47 * self.n]

    def from_wire(self, fv, wire=None, n=0, device_ptr=None):
        """KeyFrame received as a wire block (numpy bytes or device pointer) -> device-resident frame with its grid."""
        if device_ptr is not None:
            capi.check(self.lib.orbk_frame_from_wire(self.h, C.byref(fv), C.c_void_p(int(device_ptr)), int(n), 1), "orbk_frame_from_wire")
        else:
            wire = np.ascontiguousarray(wire, np.uint8)
            capi.check(self.lib.orbk_frame_from_wire(self.h, C.byref(fv), _vp(wire), int(n), 0), "orbk_frame_from_wire")
        self.n = int(n)
        return self

    def download(self):
        kps = np.zeros(max(self.n, 1), capi.KEYPOINT_DTYPE); desc = np.zeros((max(self.n, 1), 32), np.uint8)
        capi.check(self.lib.orbm_frame_download(self.h, _vp(kps), _vp(desc)), "orbm_frame_download")
        return kps[: self.n], desc[: self.n]

    def grid(self):
        start = np.zeros(capi.GRID_COLS * capi.GRID_ROWS + 1, np.int32)
        items = np.zeros(max(self.n, 1), np.int32)
        capi.check(self.lib.orbm_frame_get_grid(self.h, _vp(start), _vp(items)))
        return start, items[: start[-1]].copy()

    def isInFrustum(self, Tcw, wv, viewingCosLimit=0.5):
        m = wv.m
        T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        out = dict(track_in_view=np.zeros(m, np.uint8), proj_x=np.zeros(m, np.float32), proj_y=np.zeros(m, np.float32),
                   proj_xr=np.zeros(m, np.float32), track_depth=np.zeros(m, np.float32),
                   scale_level=np.zeros(m, np.int32), view_cos=np.zeros(m, np.float32))
        capi.check(self.lib.orbm_is_in_frustum(self.h, _vp(T), C.byref(wv), C.c_float(viewingCosLimit),
                                               *[_vp(out[k]) for k in ("track_in_view", "proj_x", "proj_y", "proj_xr",
                                                                         "track_depth", "scale_level", "view_cos")]),
                   "orbm_is_in_frustum")
        return out

    def isInFrustumRig(self, Tcw, rig, Tlr, wv, viewingCosLimit=0.5):
        """Frame::isInFrustum of a two-camera frame (S/Frame.cc:545-554,1154-1231); self is the LEFT camera's frame.  Returns the two
        cameras' track fields (left dict, right dict)."""
        m = wv.m
        T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        tlr = np.ascontiguousarray(np.asarray(Tlr, np.float32).reshape(-1)[:12])
        keys = ("track_in_view", "proj_x", "proj_y", "track_depth", "scale_level", "view_cos")
        mk = lambda: dict(track_in_view=np.zeros(m, np.uint8), proj_x=np.zeros(m, np.float32), proj_y=np.zeros(m, np.float32),
                          track_depth=np.zeros(m, np.float32), scale_level=np.zeros(m, np.int32), view_cos=np.zeros(m, np.float32))
        a, b = mk(), mk()
        capi.check(self.lib.orbm_is_in_frustum_rig(self.h, _vp(T), C.byref(rig), _vp(tlr), C.byref(wv), C.c_float(viewingCosLimit),
                                                   *[_vp(d[k]) for d in (a, b) for k in keys]), "orbm_is_in_frustum_rig")
        return a, b


def ComputeStereoFishEyeMatches(view, device=0):
    """Frame::ComputeStereoFishEyeMatches (S/Frame.cc:1093-1150) on an orbx_fisheye_stereo_view: (mvLeftToRightMatch, mvRightToLeftMatch,
    mvDepth, mvStereo3Dpoints as (Nleft, 3), nMatches)."""
    l2r = np.zeros(max(view.n_left, 1), np.int32); r2l = np.zeros(max(view.n_right, 1), np.int32)
    depth = np.zeros(max(view.n_left, 1), np.float32); p3d = np.zeros((max(view.n_left, 1), 3), np.float32)
    n = C.c_int(0)
    capi.check(capi.load().orbx_fisheye_stereo_matches(int(device), C.byref(view), _vp(l2r), _vp(r2l), _vp(depth), _vp(p3d), C.byref(n)),
               "orbx_fisheye_stereo_matches")
    return l2r[: view.n_left], r2l[: view.n_right], depth[: view.n_left], p3d[: view.n_left], n.value


class LocalMap:
    """Device-resident local map points (positions, normals, distances, descriptors)."""

    def __init__(self, cap_points=8192, device=0):
        self.lib = capi.load()
        self.h = C.c_void_p()
        capi.check(self.lib.orbm_map_create(device, cap_points, C.byref(self.h)), "orbm_map_create")

    def close(self):
        if getattr(self, "h", None):
            self.lib.orbm_map_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, wv):
        capi.check(self.lib.orbm_map_upload(self.h, C.byref(wv)), "orbm_map_upload")
        self.m = wv.m
        return self


class LastFrameOnDevice:
    """mLastFrame's view for SearchByProjection(Current, Last), resident on the device (orbm_lastview_*): uploaded when the tracking
    of that frame has finished, read from HBM by the next frame's search."""

    def __init__(self, cap_features=4096, device=0):
        self.lib = capi.load()
        self.h = C.c_void_p()
        capi.check(self.lib.orbm_lastview_create(device, cap_features, C.byref(self.h)), "orbm_lastview_create")

    def close(self):
        if getattr(self, "h", None):
            self.lib.orbm_lastview_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, lv):
        capi.check(self.lib.orbm_lastview_upload(self.h, C.byref(lv)), "orbm_lastview_upload")
        return self


class ORBmatcher:
    """ORB_SLAM3::ORBmatcher (I/ORBmatcher.h:35-108), hot-path searches only."""

    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self.lib = capi.load()
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self.device = device

    def DescriptorDistance(self, q, t):
        """Dense Hamming matrix (S/ORBmatcher.cc:2358-2374 for every pair)."""
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        d = np.zeros((len(q), len(t)), np.int32)
        capi.check(self.lib.orbm_hamming_matrix(self.device, _vp(q), len(q), _vp(t), len(t), _vp(d)), "orbm_hamming_matrix")
        return d

    def best2(self, q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        o = np.zeros((len(q), 4), np.int32)
        capi.check(self.lib.orbm_hamming_best2(self.device, _vp(q), len(q), _vp(t), len(t), _vp(o)), "orbm_hamming_best2")
        return o

    def SearchByProjection(self, F, mv, th=1.0, bFarPoints=False, thFarPoints=50.0, assigned_mp=None, assigned_obs=None):
        """(Frame&, vector<MapPoint*>&, th, bFarPoints, thFarPoints): S/ORBmatcher.cc:44-214."""
        amp = np.ascontiguousarray(assigned_mp, np.int32).copy()
        aob = np.ascontiguousarray(assigned_obs, np.int32).copy()
        n = C.c_int(0)
        capi.check(self.lib.orbm_search_by_projection_mps(F.h, C.byref(mv), C.c_float(th), int(bFarPoints),
                                                          C.c_float(thFarPoints), C.c_float(self.mfNNratio), _vp(amp),
                                                          _vp(aob), C.byref(n)), "orbm_search_by_projection_mps")
        return amp, aob, n.value

    def SearchByProjectionRig(self, FL, FR, mv, mv_r, left_to_right, right_to_left, th=1.0, bFarPoints=False, thFarPoints=50.0,
                              assigned_mp=None, assigned_obs=None):
        """SearchByProjection(Frame&, vector<MapPoint*>&, ...) on a two-camera frame (S/ORBmatcher.cc:44-214 with :145-211)."""
        amp = np.ascontiguousarray(assigned_mp, np.int32).copy()
        aob = np.ascontiguousarray(assigned_obs, np.int32).copy()
        l2r = np.ascontiguousarray(left_to_right, np.int32); r2l = np.ascontiguousarray(right_to_left, np.int32)
        n = C.c_int(0)
        capi.check(self.lib.orbm_search_by_projection_mps_rig(FL.h, FR.h, C.byref(mv), C.byref(mv_r), _vp(l2r), _vp(r2l), C.c_float(th),
                                                              int(bFarPoints), C.c_float(thFarPoints), C.c_float(self.mfNNratio), _vp(amp),
                                                              _vp(aob), C.byref(n)), "orbm_search_by_projection_mps_rig")
        return amp, aob, n.value

    def SearchByProjectionFrameRig(self, FL, FR, Tcw_cur, rig, lv, th, bMono=False, assigned_mp=None, assigned_obs=None):
        """SearchByProjection(CurrentFrame, LastFrame, th, bMono) on a two-camera current frame (S/ORBmatcher.cc:1970-2186 with :2092-2160);
        FR = None with a rig that has no right camera: one camera behind a model (a monocular fisheye frame)."""
        amp = np.ascontiguousarray(assigned_mp, np.int32).copy()
        aob = np.ascontiguousarray(assigned_obs, np.int32).copy()
        T = np.ascontiguousarray(Tcw_cur, np.float32).reshape(16)
        n = C.c_int(0)
        capi.check(self.lib.orbm_search_by_projection_frame_rig(FL.h, FR.h if FR is not None else None, _vp(T), C.byref(rig), C.byref(lv), C.c_float(th), int(bMono),
                                                                int(self.mbCheckOrientation), _vp(amp), _vp(aob), C.byref(n)),
                   "orbm_search_by_projection_frame_rig")
        return amp, aob, n.value

    def SearchLocalPoints(self, F, local_map, Tcw, th=1.0, bFarPoints=False, thFarPoints=50.0, assigned_mp=None,
                          assigned_obs=None, skip=None, inplace=False, in_frustum=None):
        """Fused Tracking::SearchLocalPoints body (S/Tracking.cc:3111-3153).  in_frustum: optional uint8[m] output, 1 where
        isInFrustum() returned true (the points the reference calls IncreaseVisible() for)."""
        amp, aob = self._state(assigned_mp, assigned_obs, inplace)
        T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        sk = None if skip is None else np.ascontiguousarray(skip, np.uint8)
        n = C.c_int(0)
        if in_frustum is not None:
            assert in_frustum.dtype == np.uint8 and in_frustum.flags["C_CONTIGUOUS"]
        capi.check(self.lib.orbm_search_local_points_vis(F.h, local_map.h, _vp(T), _vp(sk), C.c_float(th), int(bFarPoints),
                                                         C.c_float(thFarPoints), C.c_float(self.mfNNratio), _vp(amp), _vp(aob),
                                                         C.byref(n), _vp(in_frustum)), "orbm_search_local_points")
        return amp, aob, n.value

    @staticmethod
    def _state(assigned_mp, assigned_obs, inplace):
        """F.mvpMapPoints flattened; inplace=True updates the caller's int32 arrays (as the C ABI does) instead of copies."""
        if inplace:
            assert assigned_mp.dtype == np.int32 and assigned_obs.dtype == np.int32
            return assigned_mp, assigned_obs
        return np.ascontiguousarray(assigned_mp, np.int32).copy(), np.ascontiguousarray(assigned_obs, np.int32).copy()

    def SearchByProjectionFrame(self, CurrentFrame, Tcw_cur, lv, th, bMono, assigned_mp, assigned_obs, inplace=False):
        """(Frame &CurrentFrame, const Frame &LastFrame, th, bMono): S/ORBmatcher.cc:1970-2186."""
        amp, aob = self._state(assigned_mp, assigned_obs, inplace)
        T = np.ascontiguousarray(Tcw_cur, np.float32).reshape(16)
        n = C.c_int(0)
        capi.check(self.lib.orbm_search_by_projection_frame(CurrentFrame.h, _vp(T), C.byref(lv), C.c_float(th), int(bMono),
                                                            int(self.mbCheckOrientation), _vp(amp), _vp(aob), C.byref(n)),
                   "orbm_search_by_projection_frame")
        return amp, aob, n.value

    def SearchByProjectionFrameResident(self, CurrentFrame, Tcw_cur, last_on_device, th, bMono, assigned_mp, assigned_obs):
        """SearchByProjection(Current, Last) on a LastFrameOnDevice (orbm_search_by_projection_frame_resident)."""
        amp, aob = self._state(assigned_mp, assigned_obs, False)
        T = np.ascontiguousarray(Tcw_cur, np.float32).reshape(16)
        n = C.c_int(0)
        capi.check(self.lib.orbm_search_by_projection_frame_resident(CurrentFrame.h, _vp(T), last_on_device.h, C.c_float(th), int(bool(bMono)),
                                                                     int(self.mbCheckOrientation), _vp(amp), _vp(aob), C.byref(n)),
                   "orbm_search_by_projection_frame_resident")
        return amp, aob, n.value

    def SearchByBoW(self, F, fvF, kf_desc, kf_mp_valid, kf_angle, fvK):
        """(KeyFrame*, Frame&, vector<MapPoint*>&): S/ORBmatcher.cc:269-471."""
        kf_desc = np.ascontiguousarray(kf_desc, np.uint8)
        kf_mp_valid = np.ascontiguousarray(kf_mp_valid, np.uint8)
        kf_angle = np.ascontiguousarray(kf_angle, np.float32)
        matches = np.zeros(max(F.n, 1), np.int32)
        n = C.c_int(0)
        capi.check(self.lib.orbm_search_by_bow(F.h, C.byref(fvF), _vp(kf_desc), len(kf_desc), _vp(kf_mp_valid), _vp(kf_angle),
                                               C.byref(fvK), C.c_float(self.mfNNratio), int(self.mbCheckOrientation),
                                               _vp(matches), C.byref(n)), "orbm_search_by_bow")
        return matches[: F.n].copy(), n.value


    def SearchByBoWRig(self, F, n_left, fvF, kf_desc, kf_mp_valid, kf_angle, fvK):
        """SearchByBoW(KeyFrame*, Frame&, ...) on a two-camera Frame (S/ORBmatcher.cc:342-430); F holds all Nleft + Nright features."""
        kf_desc = np.ascontiguousarray(kf_desc, np.uint8)
        kf_mp_valid = np.ascontiguousarray(kf_mp_valid, np.uint8)
        kf_angle = np.ascontiguousarray(kf_angle, np.float32)
        matches = np.zeros(max(F.n, 1), np.int32)
        n = C.c_int(0)
        capi.check(self.lib.orbm_search_by_bow_rig(F.h, int(n_left), C.byref(fvF), _vp(kf_desc), len(kf_desc), _vp(kf_mp_valid), _vp(kf_angle),
                                                   C.byref(fvK), C.c_float(self.mfNNratio), int(self.mbCheckOrientation), _vp(matches), C.byref(n)),
                   "orbm_search_by_bow_rig")
        return matches[: F.n].copy(), n.value

    def SearchByProjectionSim3(self, pKF, Scw, points, vpMatched, th, ratioHamming=1.0, already_found=None, with_kfs=False, camera=None):
        """(KeyFrame*, Scw, vpPoints[, vpPointsKFs], vpMatched[, vpMatchedKF], th, ratioHamming): S/ORBmatcher.cc:473-587
        (with_kfs=False) and :589-700 (with_kfs=True).  points: LocalMap resident on the device."""
        matched = np.ascontiguousarray(vpMatched, np.int32).copy()
        S = np.ascontiguousarray(Scw, np.float32).reshape(16)
        af = None if already_found is None else np.ascontiguousarray(already_found, np.uint8)
        n = C.c_int(0)
        if camera is not None:                              # pKF->mpCamera is a camera model (an orbg_camera): a fisheye keyframe
            capi.check(self.lib.orbm_search_by_projection_sim3_cam(pKF.h, points.h, _vp(S), C.byref(camera), _vp(af), int(th), C.c_float(ratioHamming),
                                                                   _vp(matched), C.byref(n)), "orbm_search_by_projection_sim3_cam")
            return matched, n.value
        capi.check(self.lib.orbm_search_by_projection_sim3(pKF.h, points.h, _vp(S), _vp(af), int(th), C.c_float(ratioHamming),
                                                           0 if with_kfs else 1, _vp(matched), C.byref(n)),
                   "orbm_search_by_projection_sim3")
        return matched, n.value

    def SearchByProjectionReloc(self, CurrentFrame, Tcw, kf_points, kf_angle, assigned_mp, th, ORBdist, already_found=None, camera=None):
        """(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist): the relocalisation overload,
        S/ORBmatcher.cc:2188-2310.  kf_points: the keyframe's map point matches, feature by feature, as a LocalMap resident on the
        device (bad = no point / isBad()); kf_angle: pKF->mvKeysUn[i].angle; assigned_mp >= 0 where the frame already holds a point."""
        amp = np.ascontiguousarray(assigned_mp, np.int32).copy()
        T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        ang = np.ascontiguousarray(kf_angle, np.float32)
        af = None if already_found is None else np.ascontiguousarray(already_found, np.uint8)
        n = C.c_int(0)
        if camera is not None:                              # CurrentFrame.mpCamera is a camera model (an orbg_camera): a monocular fisheye frame
            capi.check(self.lib.orbm_search_by_projection_reloc_cam(CurrentFrame.h, kf_points.h, _vp(T), C.byref(camera), _vp(af), _vp(ang), C.c_float(th),
                                                                    int(ORBdist), int(self.mbCheckOrientation), _vp(amp), C.byref(n)),
                       "orbm_search_by_projection_reloc_cam")
            return amp, n.value
        capi.check(self.lib.orbm_search_by_projection_reloc(CurrentFrame.h, kf_points.h, _vp(T), _vp(af), _vp(ang), C.c_float(th), int(ORBdist),
                                                            int(self.mbCheckOrientation), _vp(amp), C.byref(n)),
                   "orbm_search_by_projection_reloc")
        return amp, n.value

    def SearchByBoWKF(self, pKF2, fv2, mp_valid2, desc1, mp_valid1, angle1, fv1):
        """(KeyFrame* pKF1, KeyFrame* pKF2, vpMatches12): S/ORBmatcher.cc:819-959; returns matches12 (indices into pKF2)."""
        desc1 = np.ascontiguousarray(desc1, np.uint8)
        mp_valid1 = np.ascontiguousarray(mp_valid1, np.uint8)
        mp_valid2 = np.ascontiguousarray(mp_valid2, np.uint8)
        angle1 = np.ascontiguousarray(angle1, np.float32)
        matches = np.zeros(max(len(desc1), 1), np.int32)
        n = C.c_int(0)
        capi.check(self.lib.orbm_search_by_bow_kf(pKF2.h, C.byref(fv2), _vp(mp_valid2), _vp(desc1), len(desc1), _vp(mp_valid1),
                                                  _vp(angle1), C.byref(fv1), C.c_float(self.mfNNratio),
                                                  int(self.mbCheckOrientation), _vp(matches), C.byref(n)),
                   "orbm_search_by_bow_kf")
        return matches[: len(desc1)].copy(), n.value


class ORBVocabulary:
    """DBoW2 ORBVocabulary (I/ORBVocabulary.h:30) on the device: transform() of descriptors into BowVector / FeatureVector."""

    def __init__(self, view, keep=None, device=0):
        self.lib = capi.load()
        self.h = C.c_void_p()
        self._keep = keep
        if view is not None:
            capi.check(self.lib.orbv_vocab_create(device, C.byref(view), C.byref(self.h)), "orbv_vocab_create")

    @classmethod
    def loadFromTextFile(cls, path, device=0, keep_trailing_node=False):
        """ORBVocabulary::loadFromTextFile (TemplatedVocabulary.h:1338-1427): the reference's ORBvoc.txt straight onto the device."""
        voc = cls(None, None, device)
        flags = capi.ORBV_TEXT_KEEP_TRAILING_NODE if keep_trailing_node else 0
        capi.check(voc.lib.orbv_vocab_from_text(int(device), str(path).encode(), flags, C.byref(voc.h)), "orbv_vocab_from_text")
        return voc

    def close(self):
        if getattr(self, "h", None):
            self.lib.orbv_vocab_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transform_features(self, desc=None, levelsup=4, frame=None):
        """Per feature (word_id, node_id, weight): transform(feature, id, w, &nid, levelsup)."""
        if frame is not None:
            n = frame.n
        else:
            desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
            n = len(desc)
        wid = np.zeros(max(n, 1), np.int32); nid = np.zeros(max(n, 1), np.int32); w = np.zeros(max(n, 1), np.float64)
        if frame is not None:
            capi.check(self.lib.orbv_transform_frame(self.h, frame.h, int(levelsup), _vp(wid), _vp(nid), _vp(w)), "orbv_transform_frame")
        else:
            capi.check(self.lib.orbv_transform(self.h, _vp(desc), n, int(levelsup), _vp(wid), _vp(nid), _vp(w)), "orbv_transform")
        return wid[:n], nid[:n], w[:n]

    def transform(self, desc=None, levelsup=4, frame=None):
        """transform(features, BowVector&, FeatureVector&, levelsup): ((words, values), (node_id, start, feat_idx))."""
        wid, nid, w = self.transform_features(desc, levelsup, frame)
        n = len(wid)
        bw = np.zeros(max(n, 1), np.int32); bv = np.zeros(max(n, 1), np.float64)
        fn = np.zeros(max(n, 1), np.uint32); fs = np.zeros(n + 1, np.uint32); ff = np.zeros(max(n, 1), np.uint32)
        nw, nn = C.c_int32(0), C.c_int32(0)
        capi.check(self.lib.orbv_bow_assemble(self.h, _vp(wid), _vp(nid), _vp(w), n, _vp(bw), _vp(bv), C.byref(nw), _vp(fn), _vp(fs),
                                              _vp(ff), C.byref(nn)), "orbv_bow_assemble")
        k = nn.value
        return (bw[: nw.value].copy(), bv[: nw.value].copy()), (fn[:k].copy(), fs[: k + 1].copy(), ff[: int(fs[k]) if k else 0].copy())


def load_text_vocabulary(path, keep_trailing_node=False):
    """The host-only half of the loader (no GPU): dict of the flattened tree's arrays + k, scoring, n_words."""
    lib = capi.load()
    h = C.c_void_p()
    capi.check(lib.orbv_text_load(str(path).encode(), capi.ORBV_TEXT_KEEP_TRAILING_NODE if keep_trailing_node else 0, C.byref(h)), "orbv_text_load")
    try:
        v = capi.VocabView(); k, sc, nw = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        capi.check(lib.orbv_text_view(h, C.byref(v), C.byref(k), C.byref(sc), C.byref(nw)), "orbv_text_view")
        return capi.vocab_view_arrays(v, k.value, sc.value, nw.value)
    finally:
        lib.orbv_text_free(h)


def BowScoreL1(q_word, q_value, cand_start, cand_word, cand_value, device=0):
    """L1Scoring::score of one query BowVector against m candidate BowVectors (CSR): DetectNBestCandidates' inner loop."""
    lib = capi.load()
    qw = np.ascontiguousarray(q_word, np.int32); qv = np.ascontiguousarray(q_value, np.float64)
    cs = np.ascontiguousarray(cand_start, np.int32); cw = np.ascontiguousarray(cand_word, np.int32); cv = np.ascontiguousarray(cand_value, np.float64)
    m = len(cs) - 1
    out = np.zeros(max(m, 1), np.float64)
    capi.check(lib.orbv_score_l1(int(device), _vp(qw), _vp(qv), len(qw), _vp(cs), _vp(cw), _vp(cv), m, _vp(out)), "orbv_score_l1")
    return out[:m]


def ComputeDistinctiveDescriptors(desc, start, device=0):
    """MapPoint::ComputeDistinctiveDescriptors (S/MapPoint.cc:448-522) for a batch of map points (CSR lists of descriptors)."""
    lib = capi.load()
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    start = np.ascontiguousarray(start, np.int32)
    m = len(start) - 1
    best = np.zeros(max(m, 1), np.int32)
    capi.check(lib.orbm_distinctive_descriptors(int(device), _vp(desc) if len(desc) else None, _vp(start), m, _vp(best)),
               "orbm_distinctive_descriptors")
    return best[:m]


class Optimizer:
    """ORB_SLAM3::Optimizer (I/Optimizer.h:30-113): LocalBundleAdjustment numerical core."""

    def __init__(self, device=0):
        self.lib = capi.load()
        self.h = C.c_void_p()
        capi.check(self.lib.lba_create(device, 0, 0, 0, C.byref(self.h)), "lba_create")
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.lba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def LocalBundleAdjustmentAsync(self, problem, out, pbStopFlag=None):
        """Submit the solve to the handle's own LocalMapping thread (lba_solve_async); collect with wait()."""
        # a refused submission (one solve in flight per handle -> ORBG_BAD_ARG) must not drop the references that keep the
        # in-flight problem / result arrays alive: they are replaced only once the library has accepted the new job
        sp = None if pbStopFlag is None else C.c_void_p(pbStopFlag.ctypes.data)
        if getattr(self, "_async_keep", None) is None:
            out.c.trace_len = 0
        fn = self.lib.lba_solve_async_b if self._is_bool_flag(pbStopFlag) else self.lib.lba_solve_async
        capi.check(fn(self.h, C.byref(problem), sp, C.byref(out.c)), "lba_solve_async")
        self._async_keep = (problem, out, pbStopFlag)

    @staticmethod
    def _is_bool_flag(flag):
        """np.bool_ / np.uint8 flag = the reference's `bool* pbStopFlag` (polled as one byte, lba_solve_hb); np.int32 = the
        C-ABI's int32 flag with its deterministic test forms."""
        if flag is None:
            return False
        if flag.dtype in (np.bool_, np.uint8):
            return True
        assert flag.dtype == np.int32, "pbStopFlag must be a bool / uint8 / int32 array of one element"
        return False

    def set_profiling(self, on=True, reset=True):
        """Bracket one LDL^T launch per solve with a HIP event pair on the handle's stream (bench.py roofline)."""
        capi.check(self.lib.lba_set_profiling(self.h, int(bool(on)), int(bool(reset))), "lba_set_profiling")

    def solver_stats(self):
        """(sum of bracket times in ms, brackets, unknowns of the system, solved on the FP64 matrix cores?)"""
        s, n, nu, mc = C.c_double(0.0), C.c_int64(0), C.c_int32(0), C.c_int32(0)
        capi.check(self.lib.lba_get_solver_stats(self.h, C.byref(s), C.byref(n), C.byref(nu), C.byref(mc)), "lba_get_solver_stats")
        return s.value, n.value, nu.value, bool(mc.value)

    def watchdog_count(self):
        """Launches of the eight-workgroup LDL^T that timed out waiting for a participant (each re-solved on the one-workgroup kernels)."""
        n = C.c_int64(0)
        capi.check(self.lib.lba_get_watchdog_count(self.h, C.byref(n)), "lba_get_watchdog_count")
        return n.value

    def event_overhead_ms(self, reps=100):
        ms = C.c_float(0.0)
        capi.check(self.lib.lba_event_overhead(self.h, int(reps), C.byref(ms)), "lba_event_overhead")
        return ms.value

    def wait(self):
        ms = C.c_double(0.0)
        capi.check(self.lib.lba_wait(self.h, C.byref(ms)), "lba_wait")
        self.last_solve_ms = ms.value
        keep, self._async_keep = getattr(self, "_async_keep", None), None
        return keep[1] if keep else None

    def PoseOptimization(self, problem):
        """int Optimizer::PoseOptimization(Frame*) (S/Optimizer.cc:964-1278); problem: views.pose_opt_problem(...)[0]."""
        out = views.PoseOptOutput(problem.n)
        capi.check(self.lib.pose_optimize(C.byref(problem), C.byref(out.c)), "pose_optimize")
        return out

    def LocalBundleAdjustment(self, problem, pbStopFlag=None, trace_cap=64, out=None):
        """problem: views.lba_problem(...)[0]; pbStopFlag: np.bool_[1] (the reference's bool) or np.int32[1], polled between LM
        iterations / trials.
        out: a views.LbaOutput of matching size to reuse (the result arrays are the caller's, as in the C ABI)."""
        if out is None:
            out = views.LbaOutput(problem.n_poses, problem.n_points, problem.n_edges, trace_cap)
        else:
            out.c.trace_len = 0
        sp = None if pbStopFlag is None else C.c_void_p(pbStopFlag.ctypes.data)
        fn = self.lib.lba_solve_hb if self._is_bool_flag(pbStopFlag) else self.lib.lba_solve_h
        capi.check(fn(self.h, C.byref(problem), sp, C.byref(out.c)), "lba_solve_h")
        return out


class KeyFrameDatabase:
    """Device-resident place-recognition database (inverted file, BowVectors, covisibility lists): the compute side of
    KeyFrameDatabase::DetectNBestCandidates (S/KeyFrameDatabase.cc:594-761).  view: views.database_view(...)."""

    def __init__(self, view, keep=None, device=0):
        self.lib = capi.load()
        self.h = C.c_void_p()
        self._keep = keep
        self.n_kfs = view.n_kfs
        capi.check(self.lib.orbd_database_create(device, C.byref(view), C.byref(self.h)), "orbd_database_create")

    def close(self):
        if getattr(self, "h", None):
            self.lib.orbd_database_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def DetectNBestCandidates(self, q_word, q_value, connected, query_map_id, nNumCandidates, place_score):
        """-> (vpLoopCand, vpMergeCand) as keyframe indices; place_score (float32[n_kfs]) is updated in place."""
        qw = np.ascontiguousarray(q_word, np.int32); qv = np.ascontiguousarray(q_value, np.float64)
        con = np.ascontiguousarray(connected, np.uint8)
        assert place_score.dtype == np.float32 and place_score.flags["C_CONTIGUOUS"] and len(place_score) == self.n_kfs
        loop = np.zeros(max(nNumCandidates, 1), np.int32); merge = np.zeros(max(nNumCandidates, 1), np.int32)
        nl, nm = C.c_int32(0), C.c_int32(0)
        capi.check(self.lib.orbd_detect_n_best_candidates(self.h, _vp(qw), _vp(qv), len(qw), _vp(con), int(query_map_id), int(nNumCandidates),
                                                          _vp(place_score), _vp(loop), C.byref(nl), _vp(merge), C.byref(nm)),
                   "orbd_detect_n_best_candidates")
        return loop[: nl.value].copy(), merge[: nm.value].copy()
