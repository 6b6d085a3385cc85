"""Local BA beyond the 50 free poses of the largest benchmark window: which solve kernel runs, agreement with the oracle, time.
Run on the GPU box: python tools/lba_large_windows.py [sizes ...]   (ORBG_LDLT_WIDE=1 forces the many-workgroup kernels)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_orbslam3_amd import api, synth, views
from oracle import binding as ob
for nf in ([int(a) for a in sys.argv[1:]] or (51, 56, 64, 65, 80, 120)):
    prob = synth.make_lba_problem(n_free=nf, n_fixed=3, n_points=30 * nf, mono_frac=0.2, seed=300 + nf)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    t0 = time.perf_counter(); o = ob.lba_solve(p); t1 = time.perf_counter()
    opt = api.Optimizer()
    try:
        g = opt.LocalBundleAdjustment(p)
        t2 = time.perf_counter(); g = opt.LocalBundleAdjustment(p); t3 = time.perf_counter()
    except Exception as e:
        print(nf, "error:", e); continue
    tg, to = g.trace_rows(), o.trace_rows()
    print(nf, "status", g.status == o.status, "iters", g.iters == o.iters, "pose err %.1e" % np.abs(g.poses - o.poses).max(),
          "chi2 rel %.1e" % (np.max(np.abs(tg[:, 1] - to[:, 1]) / to[:, 1]) if tg.shape == to.shape else -1),
          "flags", np.array_equal(g.edge_outlier, o.edge_outlier), "gpu %.2f ms oracle %.1f ms" % (1e3 * (t3 - t2), 1e3 * (t1 - t0)))
