"""ctypes binding of libagentloop.so (csrc/agent_loop.cpp): the Tracking thread of one client as a C++ loop above the C-ABI.
bench.py prepares the handles and per-frame views once, then times AgentLoop.run()."""
import ctypes as C
import os

import numpy as np

from . import _capi as capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libagentloop.so")


class FrameIn(C.Structure):
    _fields_ = [("host_left", C.c_void_p), ("host_right", C.c_void_p), ("dev_left", C.c_void_p), ("dev_right", C.c_void_p),
                ("Tcw_guess", C.c_void_p), ("last_view", C.c_void_p)]


class Cfg(C.Structure):
    _fields_ = [("ex", C.c_void_p * 4), ("fr", C.c_void_p * 4), ("local_map", C.c_void_p), ("lba", C.c_void_p),
                ("frame_view", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("stride", C.c_int), ("bf", C.c_float), ("b", C.c_float),
                ("frames", C.c_void_p), ("n_frames", C.c_int), ("seq", C.c_void_p), ("n_seq", C.c_int),
                ("kf_maps", C.c_void_p), ("n_kf_maps", C.c_int), ("lba_prob", C.c_void_p), ("lba_out", C.c_void_p),
                ("po", C.c_void_p * 2), ("po_out", C.c_void_p * 2), ("frames_per_kf", C.c_int),
                ("pipelined", C.c_int), ("host_images", C.c_int), ("ingest_async", C.c_int), ("submit_first", C.c_int),
                ("lba_async", C.c_int), ("pose_opt", C.c_int), ("th_frame", C.c_float), ("mono", C.c_int), ("nn_frame", C.c_float),
                ("nn_map", C.c_float), ("amp", C.c_void_p), ("aob", C.c_void_p), ("cap", C.c_int), ("in_flight", C.c_int32 * 4),
                ("ahead", C.c_int32), ("ring", C.c_int32), ("lba_in_flight", C.c_int32), ("last_view_dev", C.c_void_p), ("last_view_frame", C.c_int32),
                ("amp_after_frame", C.c_void_p), ("mono_agent", C.c_int32), ("dist", C.c_void_p),
                ("sync_kps", C.c_void_p), ("sync_kps_un", C.c_void_p), ("sync_desc", C.c_void_p), ("sync_uright", C.c_void_p),
                ("sync_depth", C.c_void_p), ("sync_cap", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("stage_s", C.c_double * 8), ("lba_s", C.c_double), ("lba_calls", C.c_int64), ("lba_iters", C.c_int64),
                ("kp", C.c_int64), ("m_frame", C.c_int64), ("m_map", C.c_int64), ("error", C.c_int32), ("error_step", C.c_int32),
                ("worst_step_s", C.c_double), ("worst_stage_s", C.c_double * 8), ("worst_step_index", C.c_int64),
                ("last_nl", C.c_int32), ("last_nr", C.c_int32), ("last_n1", C.c_int32), ("last_n2", C.c_int32)]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libagentloop.so not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        capi.load()                                     # liborbgpu first
        lib = C.CDLL(LIB_PATH)
        lib.agent_run.restype = C.c_int
        lib.agent_run.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.agent_sizeof.restype = C.c_int
        assert (lib.agent_sizeof(0), lib.agent_sizeof(1), lib.agent_sizeof(2)) == (C.sizeof(Cfg), C.sizeof(Stats), C.sizeof(FrameIn)), \
            "agent.py structures are out of step with csrc/agent_loop.cpp"
        lib.agent_drain.restype = C.c_int
        lib.agent_drain.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _lib = lib
    return _lib


class AgentLoop:
    """Holds the agent_cfg (and everything it points to) of one client."""

    def __init__(self, exs, frs, local_map, opt, frame_view, width, height, stride, bf, b, frames, seq, kf_map_views, lba_prob, lba_out,
                 po_probs, frames_per_kf, cap, th_frame, mono, nn_map=0.8, last_view=None, mono_agent=False, dist=None):
        """mono_agent: a monocular client (frames carry one host / device image; Frame::Frame(mono) with lapping area {0, 1000});
        dist: its capi.OrbxDistortion (mDistCoef) or None."""
        self.lib = load()
        self.keep = [exs, frs, local_map, opt, frame_view, frames, kf_map_views, lba_prob, lba_out, po_probs]
        c = Cfg()
        for j in range(4):
            c.ex[j] = exs[min(j, len(exs) - 1)].h.value
            c.fr[j] = frs[min(j, len(frs) - 1)].h.value
        c.ring = max(min(len(exs), 4), 2)                  # ring of (handle, frame) pairs; frames handed over ahead: configure(ahead=)
        c.ahead = 1
        c.local_map, c.lba = local_map.h.value, opt.h.value
        c.frame_view = C.addressof(frame_view)
        c.width, c.height, c.stride, c.bf, c.b = width, height, stride, bf, b
        self.frames_in = (FrameIn * len(frames))()
        for k, f in enumerate(frames):
            fi = self.frames_in[k]
            fi.host_left = f["host"][0].ctypes.data
            fi.host_right = None if f["host"][1] is None else f["host"][1].ctypes.data
            fi.dev_left, fi.dev_right = f["dev"]
            fi.Tcw_guess = f["guess"].ctypes.data
            fi.last_view = C.addressof(f["last_view"])
        c.frames, c.n_frames = C.addressof(self.frames_in), len(frames)
        self.seq = np.ascontiguousarray(seq, np.int32)
        c.seq, c.n_seq = self.seq.ctypes.data, len(self.seq)
        self.kf_maps = (C.c_void_p * len(kf_map_views))(*[C.addressof(v) for v in kf_map_views])
        c.kf_maps, c.n_kf_maps = C.addressof(self.kf_maps), len(kf_map_views)
        c.lba_prob, c.lba_out = C.addressof(lba_prob), C.addressof(lba_out.c)
        self.po_out = []
        for j, p in enumerate(po_probs):
            out = capi.PoseOptResult()
            flags = np.zeros(max(p.n, 1), np.uint8)
            out.outlier = flags.ctypes.data
            self.po_out.append((out, flags))
            c.po[j] = C.addressof(p)
            c.po_out[j] = C.addressof(out)
        c.frames_per_kf = frames_per_kf
        self.amp = np.full(cap, -1, np.int32)
        self.aob = np.zeros(cap, np.int32)
        c.amp, c.aob, c.cap = self.amp.ctypes.data, self.aob.ctypes.data, cap
        c.th_frame, c.mono, c.nn_frame, c.nn_map = th_frame, int(mono), 0.9, nn_map
        self.keep.append(last_view)
        c.last_view_dev = last_view.h.value if last_view is not None else None      # api.LastFrameOnDevice: the last frame's view resident on the device
        c.last_view_frame = -1
        c.amp_after_frame = None
        self.amp_after_frame = None
        c.mono_agent = int(bool(mono_agent))
        self.keep.append(dist)
        c.dist = None if dist is None else C.addressof(dist)
        self.c = c

    def set_sync_outputs(self, cap):
        """Host arrays the SYNCHRONOUS constructor of the loop delivers mvKeys / mvKeysUn / mDescriptors / mvuRight / mvDepth into
        (cap = 0: off again).  Returns them as a dict (as ORBextractor.set_frame_outputs does for the two-halves constructor)."""
        c = self.c
        if cap <= 0:
            c.sync_kps = c.sync_kps_un = c.sync_desc = c.sync_uright = c.sync_depth = None
            c.sync_cap = 0
            self._sync_out = None
            return None
        out = dict(kps=np.zeros(cap, capi.KEYPOINT_DTYPE), kps_un=np.zeros(cap, capi.KEYPOINT_DTYPE), desc=np.zeros((cap, 32), np.uint8),
                   uright=np.zeros(cap, np.float32), depth=np.zeros(cap, np.float32))
        c.sync_kps, c.sync_kps_un, c.sync_desc = out["kps"].ctypes.data, out["kps_un"].ctypes.data, out["desc"].ctypes.data
        c.sync_uright, c.sync_depth, c.sync_cap = out["uright"].ctypes.data, out["depth"].ctypes.data, int(cap)
        self._sync_out = out
        return out

    def capture_first_search(self, on=True):
        """From now on every step also copies F.mvpMapPoints as SearchByProjection(Current, Last) left it into
        self.amp_after_frame (bench.py's parity gate; off: no copy)."""
        if on and self.amp_after_frame is None:
            self.amp_after_frame = np.full(len(self.amp), -1, np.int32)
        self.c.amp_after_frame = self.amp_after_frame.ctypes.data if on else None

    def configure(self, pipelined, host_images, ingest_async, submit_first, lba_async, pose_opt, ahead=1):
        c = self.c
        c.ahead = max(1, min(int(ahead), c.ring - 1))
        c.pipelined, c.host_images, c.ingest_async, c.submit_first = int(pipelined), int(host_images), int(ingest_async), int(submit_first)
        c.lba_async, c.pose_opt = int(lba_async), int(pose_opt)

    def run(self, first_step, n_steps, last_is_final=False, timed=False, step_s=None, stats=None):
        st = stats if stats is not None else Stats()
        rc = self.lib.agent_run(C.addressof(self.c), int(first_step), int(n_steps), int(last_is_final), int(timed),
                                None if step_s is None else step_s.ctypes.data, C.addressof(st))
        capi.check(rc, "agent_run (step %d)" % st.error_step)
        return st

    def drain(self, stats=None, timed=False):
        rc = self.lib.agent_drain(C.addressof(self.c), None if stats is None else C.addressof(stats), int(timed))
        capi.check(rc, "agent_drain")
