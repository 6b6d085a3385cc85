import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from multi_orbslam3_amd import api, synth, views, _capi as capi
sc = synth.Scene(640, 480)
cam = sc.cam
L, R, Tcw = sc.stereo_pair(0)
dL = torch.from_numpy(L).cuda(); dR = torch.from_numpy(R).cuda()
ex = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=2)
lib = ex.lib
nl, nr = C.c_int(0), C.c_int(0)
def raw_extract():
    lib.orbx_extract_stereo_dev(ex.h, C.c_void_p(dL.data_ptr()), C.c_void_p(dR.data_ptr()), 640, 480, 640, None, None, 0, C.byref(nl), None, None, 0, C.byref(nr))
def raw_stereo():
    lib.orbx_stereo_match(ex.h, C.c_float(float(cam["bf"])), C.c_float(float(cam["b"])), None, None)
for f in (raw_extract, raw_stereo):
    for _ in range(20): f()
def t(f, n=200):
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("raw extract us", t(raw_extract), ex.timings())
print("raw stereo us", t(raw_stereo))
print("api extract us", t(lambda: ex.extract_stereo_dev(dL.data_ptr(), dR.data_ptr(), 640, 480, 640)))
print("api stereo us", t(lambda: ex.ComputeStereoMatches(float(cam["bf"]), float(cam["b"]), download=False)))
p = sc.frame_view_params()
fv, keep = views.frame_view(np.zeros(1, capi.KEYPOINT_DTYPE), np.zeros((1, 32), np.uint8), None, None, p["bounds"], p["cam"], 8, 1.2)
F = api.Frame(4096)
print("api from_extractor us", t(lambda: F.from_extractor(ex, fv, nl.value)))
# last-frame view from this frame itself
n, nr2, kl, dl = ex.extract_stereo_dev(dL.data_ptr(), dR.data_ptr(), 640, 480, 640, download_left=True)
ur, dp = ex.ComputeStereoMatches(float(cam["bf"]), float(cam["b"]), n_left=n)
kl, dl = kl.copy(), dl.copy()
Pw, valid = synth.unproject_to_world(kl, dp, Tcw, cam)
lv, keep2 = views.lastframe_view(valid.astype(np.uint8), np.zeros(n, np.uint8), Pw, dl, kl["octave"], kl["angle"], np.full(n, 3, np.int32), Tcw.astype(np.float32))
F.from_extractor(ex, fv, n)
m = api.ORBmatcher(0.9, True)
amp = np.full(n, -1, np.int32); aob = np.zeros(n, np.int32)
T = Tcw.astype(np.float32).reshape(16).copy()
print("api search_frame us", t(lambda: m.SearchByProjectionFrame(F, T, lv, 7.0, False, amp, aob)))
cnt = C.c_int(0)
a2, b2 = amp.copy(), aob.copy()
def raw_sf():
    a2[:] = -1; b2[:] = 0
    lib.orbm_search_by_projection_frame(F.h, C.c_void_p(T.ctypes.data), C.byref(lv), C.c_float(7.0), 0, 1, C.c_void_p(a2.ctypes.data), C.c_void_p(b2.ctypes.data), C.byref(cnt))
print("raw search_frame us", t(raw_sf), cnt.value)
chunk = synth.map_from_frame(kl, dl, dp, Tcw, cam)
mp = {k: np.concatenate([chunk[k]] * 6) for k in chunk}
wv, keep3 = views.worldpoints_view(mp["pos"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], mp["n_obs"], mp["bad"])
LM = api.LocalMap(16384).upload(wv)
m2 = api.ORBmatcher(0.8)
print("api search_local us", t(lambda: m2.SearchLocalPoints(F, LM, T, 1.0, False, 0.0, amp, aob)), wv.m)
def raw_sl():
    a2[:] = -1; b2[:] = 0
    lib.orbm_search_local_points(F.h, LM.h, C.c_void_p(T.ctypes.data), None, C.c_float(1.0), 0, C.c_float(0.0), C.c_float(0.8), C.c_void_p(a2.ctypes.data), C.c_void_p(b2.ctypes.data), C.byref(cnt))
print("raw search_local us", t(raw_sl), cnt.value)
