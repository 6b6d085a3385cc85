#!/usr/bin/env python3
"""Generate the committed golden fixtures (small, seeded) with the CPU oracle.

The reference ships no golden vectors and cannot be built here, so these fixtures are produced
by this repo's oracle; they freeze its outputs so that (a) an accidental change of the oracle is caught on CPU and
(b) the HIP path is checked against committed data on the GPU box.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from multi_orbslam3_amd import synth, views  # noqa: E402
from oracle import binding as ob  # noqa: E402
import helpers  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    sc = synth.Scene(160, 120, tex_size=(400, 300), px_per_m=50.0)
    L, R, Tcw = sc.stereo_pair(0)
    exL = ob.Extractor(n_features=300, n_levels=4, max_width=160, max_height=120)
    exR = ob.Extractor(n_features=300, n_levels=4, max_width=160, max_height=120)
    rc, kl, dl, _ = exL.extract(L)
    rc, kr, dr, _ = exR.extract(R)
    ur, dp = ob.stereo_match(exL, exR, kl, dl, kr, dr, float(sc.cam["bf"]), float(sc.cam["b"]))
    cands = [exL.candidates(l) for l in range(4)]
    np.savez_compressed(os.path.join(OUT, "extract_160x120.npz"), L=L, R=R, kps_l=kl, desc_l=dl, kps_r=kr, desc_r=dr,
                        uright=ur, depth=dp, level1=exL.level(1), level3=exL.level(3),
                        cand0=cands[0], cand1=cands[1], cand2=cands[2], cand3=cands[3])
    rng = np.random.RandomState(42)
    q = rng.randint(0, 256, (24, 32)).astype(np.uint8)
    t = rng.randint(0, 256, (40, 32)).astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "hamming_24x40.npz"), q=q, t=t, dist=ob.hamming_matrix(q, t), best2=ob.hamming_best2(q, t))
    prob = synth.make_lba_problem(n_free=5, n_fixed=2, n_points=60, width=320, height=240, mono_frac=0.15, seed=99)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    o = ob.lba_solve(p)
    np.savez_compressed(os.path.join(OUT, "lba_5p2_60.npz"), poses=prob["poses"], pose_fixed=prob["pose_fixed"],
                        points=prob["points"], edges=prob["edges"], cam=np.array(prob["cam"], np.float32),
                        out_poses=o.poses, out_points=o.points, out_outlier=o.edge_outlier, out_chi2=o.edge_chi2,
                        trace=o.trace_rows(), iters=np.array(o.iters), status=np.array([o.status]))
    print("kps", len(kl), len(kr), "stereo", int((ur > 0).sum()), "lba iters", o.iters, "edges", p.n_edges)


if __name__ == "__main__":
    main()
