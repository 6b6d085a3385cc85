"""Wall time of a local BA per free-pose count on an otherwise idle GPU, with the LDL^T kernel choices lba.hip offers for 35 .. 50 free
poses (ORBG_LDLT_XCD=0: one workgroup; =all: eight workgroups of one XCD from 14 tile rows on):  python tools/lba_sizes.py [reps]
Run once per setting of the switch (it is read when the handle is created)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multi_orbslam3_amd import api, synth, views
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for nfree in (21, 24, 26, 29, 31, 34, 36, 38, 40, 42, 45, 47, 50):
    prob = synth.make_lba_problem(n_free=nfree, n_fixed=nfree // 3, n_points=150 * nfree, width=1280, height=720, seed=11)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    opt = api.Optimizer()
    out = views.LbaOutput(p.n_poses, p.n_points, p.n_edges)
    for _ in range(5):
        opt.LocalBundleAdjustment(p, out=out)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); opt.LocalBundleAdjustment(p, out=out); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    T = ((((6 * nfree + 3) & ~3) + 1) + 15) // 16
    print("%2d free poses (%2d tile rows): median %.4f ms, min %.4f, LM iterations %s  ORBG_LDLT_XCD=%s" % (
        nfree, T, np.median(ts), ts.min(), out.iters, os.environ.get("ORBG_LDLT_XCD", "(default)")))
