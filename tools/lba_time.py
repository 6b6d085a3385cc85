"""Wall time of the C2 / C4 local BA on an otherwise idle GPU: python tools/lba_time.py [C2|C4] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multi_orbslam3_amd import api, synth, views
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
nfree, nfix, npts, W, H = (20, 10, 2000, 640, 480) if cfg == "C2" else (50, 20, 8000, 1280, 720)
prob = synth.make_lba_problem(n_free=nfree, n_fixed=nfix, n_points=npts, width=W, height=H)
p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
opt = api.Optimizer()
out = views.LbaOutput(p.n_poses, p.n_points, p.n_edges)
for _ in range(10):
    opt.LocalBundleAdjustment(p, out=out)
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); opt.LocalBundleAdjustment(p, out=out); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("%s local BA: median %.4f ms, mean %.4f, min %.4f (iters %s) %s" % (cfg, np.median(ts), ts.mean(), ts.min(), out.iters,
      " ".join("%s=%s" % (k, os.environ[k]) for k in os.environ if k.startswith("ORBG_"))))
