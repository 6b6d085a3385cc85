"""Wall time of the two optimisers on problems of the two-fisheye rig (KannalaBrandt8 models, right-camera ToBody edges) next to the
pinhole problems of the same size and the CPU oracle: python tools/rig_time.py  (on a GPU box)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_orbslam3_amd import api, synth, views  # noqa: E402
from oracle import binding as ob  # noqa: E402


def timed(fn, reps):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


def main():
    opt = api.Optimizer()
    rows = []
    for nf, nx, npts in ((20, 10, 2000), (50, 20, 8000)):
        pr = synth.make_lba_rig_problem(n_free=nf, n_fixed=nx, n_points=npts, seed=0xF15E + nf)
        rig = views.camera_rig(*pr["rig"])
        p, keep = views.lba_problem(pr["poses"], pr["pose_fixed"], pr["points"], pr["edges"], pr["cam"], rig=rig)
        out = views.LbaOutput(p.n_poses, p.n_points, p.n_edges)
        g = timed(lambda: opt.LocalBundleAdjustment(p, out=out), 30)
        o = timed(lambda: ob.lba_solve(p), 3)
        q = synth.make_lba_problem(n_free=nf, n_fixed=nx, n_points=npts)
        p2, keep2 = views.lba_problem(q["poses"], q["pose_fixed"], q["points"], q["edges"], q["cam"])
        out2 = views.LbaOutput(p2.n_poses, p2.n_points, p2.n_edges)
        g2 = timed(lambda: opt.LocalBundleAdjustment(p2, out=out2), 30)
        rows.append(f"local BA, {nf}+{nx} keyframes, {npts} points: rig {p.n_edges} edges ({int((pr['edges']['ur'] <= -1.5).sum())} of the right camera) "
                    f"{g:.3f} ms (iterations {out.iters}), oracle {o:.1f} ms; pinhole stereo {p2.n_edges} edges {g2:.3f} ms (iterations {out2.iters})")
    for nl, nr in ((300, 200), (600, 400), (1500, 1200)):
        pr = synth.make_pose_opt_rig_problem(n_left=nl, n_right=nr)
        p, keep = views.pose_opt_problem(pr["Xw"], pr["u"], pr["v"], pr["ur"], pr["inv_sigma2"], pr["cam"], pr["Tcw"], rig=views.camera_rig(*pr["rig"]))
        g = timed(lambda: opt.PoseOptimization(p), 50)
        o = timed(lambda: ob.pose_optimize(p), 5)
        q = synth.make_pose_opt_problem(n=nl + nr)
        p2, keep2 = views.pose_opt_problem(q["Xw"], q["u"], q["v"], q["ur"], q["inv_sigma2"], q["cam"], q["Tcw"])
        g2 = timed(lambda: opt.PoseOptimization(p2), 50)
        rows.append(f"PoseOptimization, {nl} left + {nr} right features: rig {g * 1e3:.0f} us, oracle {o * 1e3:.0f} us; pinhole {nl + nr} features {g2 * 1e3:.0f} us")
    print("\n".join(rows))


if __name__ == "__main__":
    main()
