"""ctypes mirror of include/orbgpu.h and loader of the HIP library (liborbgpu.so).

The library is the product; there is NO CPU fallback: if the shared object is missing or cannot be
loaded, `load()` raises, and every entry point returns ORBG_NO_DEVICE when no HIP device is usable.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORBG_LIB") or os.path.join(_HERE, "liborbgpu.so")   # ORBG_LIB: an experimental build (tools/micro/variants)

ORBG_OK, ORBG_EMPTY, ORBG_BAD_ARG, ORBG_CAP_EXCEEDED, ORBG_HIP_ERROR, ORBG_NO_DEVICE, ORBG_INTERNAL = 0, -1, -2, -3, -4, -5, -6
LBA_APPLIED, LBA_ABORTED_BEFORE_OPT, LBA_REJECTED_OUTLIERS = 0, 1, 2
GRID_COLS, GRID_ROWS = 64, 48

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4")])
EDGE_DTYPE = np.dtype([("pose", "<i4"), ("point", "<i4"), ("u", "<f4"), ("v", "<f4"), ("ur", "<f4"),
                       ("inv_sigma2", "<f4")])
assert KEYPOINT_DTYPE.itemsize == 24 and EDGE_DTYPE.itemsize == 24


class OrbxConfig(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("scale_factor", C.c_float), ("n_levels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("max_width", C.c_int32),
                ("max_height", C.c_int32), ("n_cams", C.c_int32), ("device", C.c_int32),
                ("gauss_taps", C.c_int32 * 4), ("octree_oldest_first", C.c_int32)]


class OrbxDistortion(C.Structure):
    """orbx_distortion: mDistCoef {k1, k2, p1, p2, k3} (S/Tracking.cc:71-81)."""
    _fields_ = [("k1", C.c_float), ("k2", C.c_float), ("p1", C.c_float), ("p2", C.c_float), ("k3", C.c_float)]


class FrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("kps", C.c_void_p), ("desc", C.c_void_p), ("uright", C.c_void_p),
                ("depth", C.c_void_p), ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float),
                ("max_y", C.c_float), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("bf", C.c_float), ("b", C.c_float), ("n_levels", C.c_int32),
                ("scale_factor", C.c_float)]


class MapPointsView(C.Structure):
    _fields_ = [("m", C.c_int32), ("track_in_view", C.c_void_p), ("bad", C.c_void_p), ("proj_x", C.c_void_p),
                ("proj_y", C.c_void_p), ("proj_xr", C.c_void_p), ("track_depth", C.c_void_p),
                ("scale_level", C.c_void_p), ("view_cos", C.c_void_p), ("desc", C.c_void_p),
                ("n_obs", C.c_void_p)]


class WorldPointsView(C.Structure):
    _fields_ = [("m", C.c_int32), ("pos", C.c_void_p), ("normal", C.c_void_p), ("min_dist", C.c_void_p),
                ("max_dist", C.c_void_p), ("desc", C.c_void_p), ("n_obs", C.c_void_p), ("bad", C.c_void_p),
                ("skip", C.c_void_p)]


class LastFrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("mp_valid", C.c_void_p), ("outlier", C.c_void_p), ("world_pos", C.c_void_p),
                ("desc", C.c_void_p), ("octave", C.c_void_p), ("angle", C.c_void_p), ("n_obs", C.c_void_p),
                ("Tcw", C.c_float * 16)]


class FeatVecView(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("node_id", C.c_void_p), ("start", C.c_void_p), ("feat_idx", C.c_void_p)]


class VocabView(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("L", C.c_int32), ("weighting", C.c_int32), ("scoring_norm", C.c_int32),
                ("child_start", C.c_void_p), ("child_ids", C.c_void_p), ("desc", C.c_void_p), ("weight", C.c_void_p),
                ("word_id", C.c_void_p)]


ORBV_TF_IDF, ORBV_TF, ORBV_IDF, ORBV_BINARY = 0, 1, 2, 3
ORBV_NORM_NONE, ORBV_NORM_L1, ORBV_NORM_L2 = 0, 1, 2
ORBV_TEXT_KEEP_TRAILING_NODE = 1


def vocab_view_arrays(v, k=None, scoring=None, n_words=None):
    """Copies of the arrays an orbv_vocab_view points at (a view handed out by a text loader dies with its handle)."""
    import numpy as np
    n = int(v.n_nodes)

    def arr(p, count, dt):
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(dt)), shape=(count,)).copy() if count else np.zeros(0, dt)
    cs = arr(v.child_start, n + 1, C.c_int32)
    out = dict(child_start=cs, child_ids=arr(v.child_ids, int(cs[-1]), C.c_int32), desc=arr(v.desc, n * 32, C.c_uint8).reshape(n, 32),
               weight=arr(v.weight, n, C.c_double), word_id=arr(v.word_id, n, C.c_int32), L=int(v.L), weighting=int(v.weighting),
               scoring_norm=int(v.scoring_norm))
    if k is not None:
        out.update(k=int(k), scoring=int(scoring), n_words=int(n_words))
    return out


class Camera(C.Structure):
    """orbg_camera: model (0 pinhole, 1 KannalaBrandt8), fx fy cx cy, k1..k4."""
    _fields_ = [("model", C.c_int32), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("k", C.c_float * 4)]


class CameraRig(C.Structure):
    """orbg_camera_rig: mpCamera, mpCamera2 (has_right), mTrl (3 x 4 row-major)."""
    _fields_ = [("left", Camera), ("has_right", C.c_int32), ("right", Camera), ("Trl", C.c_float * 12)]


class FisheyeStereoView(C.Structure):
    """orbx_fisheye_stereo_view: the inputs of Frame::ComputeStereoFishEyeMatches."""
    _fields_ = [("n_left", C.c_int32), ("n_right", C.c_int32), ("mono_left", C.c_int32), ("mono_right", C.c_int32),
                ("kps_left", C.c_void_p), ("kps_right", C.c_void_p), ("desc_left", C.c_void_p), ("desc_right", C.c_void_p),
                ("level_sigma2", C.c_void_p), ("n_levels", C.c_int32), ("left", Camera), ("right", Camera), ("Tlr", C.c_float * 12)]


CAM_PINHOLE, CAM_KANNALA_BRANDT8 = 0, 1
UR_RIGHT_CAMERA = -2.0


class LbaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32),
                ("poses", C.c_void_p), ("pose_fixed", C.c_void_p), ("points", C.c_void_p), ("edges", C.c_void_p),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float),
                ("lambda_init", C.c_double), ("its_round1", C.c_int32), ("its_round2", C.c_int32),
                ("device", C.c_int32), ("rig", C.POINTER(CameraRig))]


class LbaResult(C.Structure):
    _fields_ = [("poses", C.c_void_p), ("points", C.c_void_p), ("edge_chi2", C.c_void_p),
                ("edge_depth_pos", C.c_void_p), ("edge_outlier", C.c_void_p), ("status", C.c_int32),
                ("iters_round1", C.c_int32), ("iters_round2", C.c_int32), ("n_outliers", C.c_int32),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("trace", C.c_void_p),
                ("trace_cap", C.c_int32), ("trace_len", C.c_int32)]


class PoseOptProblem(C.Structure):
    _fields_ = [("n", C.c_int32), ("Xw", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("ur", C.c_void_p),
                ("inv_sigma2", C.c_void_p), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("bf", C.c_float), ("Tcw", C.c_float * 16), ("device", C.c_int32), ("rig", C.POINTER(CameraRig))]


class DatabaseView(C.Structure):
    _fields_ = [("n_kfs", C.c_int32), ("n_words", C.c_int32), ("inv_start", C.c_void_p), ("inv_kf", C.c_void_p),
                ("bow_start", C.c_void_p), ("bow_word", C.c_void_p), ("bow_value", C.c_void_p), ("covis_start", C.c_void_p),
                ("covis_kf", C.c_void_p), ("map_id", C.c_void_p), ("bad", C.c_void_p), ("map_bad", C.c_void_p)]


class PoseOptResult(C.Structure):
    _fields_ = [("Tcw", C.c_float * 16), ("outlier", C.c_void_p), ("n_inliers", C.c_int32), ("n_bad", C.c_int32),
                ("iters", C.c_int32 * 4), ("chi2", C.c_double * 4)]


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data


# Every symbol include/orbgpu.h declares (tests assert that the library exports all of them).
EXPORTED_SYMBOLS = [
    "orbx_create", "orbx_destroy", "orbx_get_tables", "orbx_extract", "orbx_extract_stereo",
    "orbx_extract_stereo_dev", "orbx_get_level", "orbx_get_level_bordered", "orbx_get_candidates", "orbx_stereo_match",
    "orbm_frame_create", "orbm_frame_destroy", "orbm_frame_upload", "orbm_frame_from_extractor", "orbx_frame_stereo_dev", "orbx_frame_stereo", "orbx_frame_stereo_dev_submit", "orbx_frame_stereo_dev_wait", "orbx_frame_stereo_submit", "orbx_frame_stereo_wait", "orbx_set_frame_outputs",
    "orbm_frame_get_grid", "orbm_hamming_matrix", "orbm_hamming_best2", "orbm_is_in_frustum",
    "orbm_search_by_projection_mps", "orbm_is_in_frustum_rig", "orbm_search_by_projection_mps_rig", "orbm_search_by_projection_frame_rig", "orbm_search_by_bow_rig", "orbx_fisheye_stereo_matches", "orbm_map_create", "orbm_map_destroy", "orbm_map_upload",
    "orbm_search_local_points", "orbm_search_local_points_vis", "orbm_search_by_projection_frame", "orbm_search_by_bow",
    "orbm_search_by_projection_sim3", "orbm_search_by_bow_kf",
    "orbv_vocab_create", "orbv_vocab_destroy", "orbv_text_load", "orbv_text_view", "orbv_text_free", "orbv_vocab_from_text", "orbv_transform", "orbv_transform_frame", "orbv_bow_assemble",
    "orbm_distinctive_descriptors", "orbv_score_l1", "orbk_wire_bytes", "orbk_pack_frame", "orbk_frame_from_wire",
    "orbm_frame_download",
    "lba_solve", "lba_create", "lba_destroy", "lba_solve_h", "lba_solve_async", "lba_wait", "pose_optimize",
    "lba_solve_b", "lba_solve_hb", "lba_solve_async_b",
    "lba_set_profiling", "lba_get_solver_stats", "lba_event_overhead", "lba_get_watchdog_count",
    "orbd_database_create", "orbd_database_destroy", "orbd_detect_n_best_candidates",
    "orbx_set_stream", "orbm_frame_set_stream", "orbm_map_set_stream", "lba_set_stream", "orbv_vocab_set_stream", "orbd_database_set_stream",
    "pose_opt_set_stream", "orbx_get_ctor_timeline", "orbm_map_set_observations", "orbm_search_by_projection_reloc", "orbm_search_by_projection_reloc_cam", "orbm_search_by_projection_sim3_cam", "orbm_lastview_create", "orbm_lastview_destroy", "orbm_lastview_upload",
    "orbm_search_by_projection_frame_resident", "orbg_quiesce", "orbg_set_wait_policy", "orbg_get_wait_policy",
    "orbx_frame_mono", "orbx_frame_mono_dev", "orbx_frame_mono_submit", "orbx_frame_mono_dev_submit", "orbx_frame_mono_wait",
    "orbx_set_frame_outputs_un", "orbx_undistort_points",
    "orbg_version", "orbg_strerror", "orbg_device_count", "orbx_get_timings", "orbx_event_overhead", "orbx_set_profile_interval", "orbx_set_profile_kernel", "orbx_get_fast_kernel_stats", "orbx_set_profiling",
]

_lib = None


class OrbGpuError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        msg = "orbgpu error %d" % code
        try:
            msg += " (%s)" % load().orbg_strerror(code).decode()
        except Exception:
            pass
        super().__init__(msg + (" in " + where if where else ""))


def load():
    """Load liborbgpu.so; raise loudly if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("liborbgpu.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(expected at %s). There is no CPU fallback." % LIB_PATH)
    # ONE HIP / HSA runtime per process: PyTorch-ROCm ships its own libamdhip64 + libhsa-runtime64 and loads them by path even when
    # the system's are mapped already (the library's own dependency resolves to /opt/rocm); the second runtime then finds no
    # GPU ("No HIP GPUs are available" from the first torch.cuda call of a process that used this library before importing
    # torch).  With torch imported first the library's NEEDED entry resolves to the runtime torch mapped.  A C++ deployment
    # (no torch in the process) has the system runtime only.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.orbg_version.restype = C.c_char_p
    lib.orbg_strerror.restype = C.c_char_p
    lib.orbg_strerror.argtypes = [C.c_int]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("orbg_version", "orbg_strerror"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def check(code, where=""):
    if code != ORBG_OK:
        raise OrbGpuError(code, where)
