"""One local BA window of a given size, three solves (for a kernel trace): python tools/lba_one.py <free poses> [<points per pose>]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_orbslam3_amd import api, synth, views
nf = int(sys.argv[1]); ppp = int(sys.argv[2]) if len(sys.argv) > 2 else 30
prob = synth.make_lba_problem(n_free=nf, n_fixed=3, n_points=ppp * nf, mono_frac=0.2, seed=300 + nf)
p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
opt = api.Optimizer()
for _ in range(3):
    g = opt.LocalBundleAdjustment(p)
print(nf, g.status, g.iters)
