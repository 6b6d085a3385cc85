import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from multi_orbslam3_amd import api, synth, views
opt = api.Optimizer()
for n in (450, 650, 830):
    pr = synth.make_pose_opt_problem(n=n, seed=77)
    p, keep = views.pose_opt_problem(pr["Xw"], pr["u"], pr["v"], pr["ur"], pr["inv_sigma2"], pr["cam"], pr["Tcw"])
    for _ in range(10): opt.PoseOptimization(p)
    t0 = time.perf_counter()
    for _ in range(100): g = opt.PoseOptimization(p)
    print("PoseOptimization n=%d: %.1f us/call, iters %s" % (n, 1e4 * (time.perf_counter() - t0), list(g.c.iters)))
