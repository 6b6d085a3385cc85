"""Ad-hoc timing of the C4 configuration (1280x720 stereo, 2000 features, 50-KF local BA) -- not the benchmark metric."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from multi_orbslam3_amd import api, synth, views, _capi as capi
sc = synth.Scene(1280, 720)
L, R, Tcw = sc.stereo_pair(0)
ex = api.ORBextractor(2000, 1.2, 8, 20, 7, 1280, 720, n_cams=2)
p = sc.frame_view_params()
fv, keep = views.frame_view(np.zeros(1, capi.KEYPOINT_DTYPE), np.zeros((1, 32), np.uint8), None, None, p["bounds"], p["cam"], 8, 1.2)
F = api.Frame(8192)
bf, b = float(sc.cam["bf"]), float(sc.cam["b"])
for _ in range(5):
    r = ex.frame_stereo(F, fv, L, R, bf, b, download=False)
t0 = time.perf_counter()
for _ in range(50):
    r = ex.frame_stereo(F, fv, L, R, bf, b, download=False)
print("C4 frame ctor (host images) ms", (time.perf_counter() - t0) / 50 * 1e3, "kps", r)
prob = synth.make_lba_problem(n_free=50, n_fixed=20, n_points=8000, width=1280, height=720)
lp, k2 = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
opt = api.Optimizer()
out = views.LbaOutput(lp.n_poses, lp.n_points, lp.n_edges)
for _ in range(2):
    opt.LocalBundleAdjustment(lp, out=out)
t0 = time.perf_counter()
for _ in range(5):
    o = opt.LocalBundleAdjustment(lp, out=out)
print("C4 LBA ms", (time.perf_counter() - t0) / 5 * 1e3, "iters", o.iters, "edges", lp.n_edges)
