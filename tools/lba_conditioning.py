"""How far the chi^2 trace of a local BA moves when only the ROUNDING of the reduced-camera solve changes (matrix-core LDL^T,
the two vector-ALU kernels, the oracle's dense Cholesky), on well- and ill-conditioned windows.  Run on the GPU box:
    python tools/lba_conditioning.py
Measured (round 2): 3..8 observations per landmark: every solver within 2e-14 of the oracle; two observations per landmark or
single-observation landmarks: 6e-9 .. 1e-7 between ANY two of them -- conditioning, not a kernel: the parity tests on such
windows (tests/test_gpu_parity.py::test_lba_covisibility_structures) compare decisions exactly and chi^2 to 1e-6."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_orbslam3_amd import api, synth, views
from oracle import binding as ob
for (nf, mn, mx, mono, outl) in [(24, 2, 2, 0.0, 0.03), (5, 2, 2, 0.5, 0.3), (5, 1, 3, 0.5, 0.3), (24, 3, 8, 0.2, 0.03)]:
    prob = synth.make_lba_problem(n_free=nf, n_fixed=2, n_points=15 * nf + 50, mono_frac=mono, outlier_frac=outl, min_obs=mn, max_obs=mx,
                                  seed=9000 + 100 * mn + mx + nf)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    o = ob.lba_solve(p).trace_rows()[:, 1]
    res = {}
    for name, env in (("mfma", {}), ("valu", {"ORBG_LDLT_VALU": "1"}), ("valu_rows", {"ORBG_LDLT_VALU": "1", "ORBG_LDLT_ROWS": "1"})):
        for k in ("ORBG_LDLT_VALU", "ORBG_LDLT_ROWS"): os.environ.pop(k, None)
        os.environ.update(env)
        res[name] = api.Optimizer().LocalBundleAdjustment(p).trace_rows()[:, 1]
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.abs(b)))
    print((nf, mn, mx, mono, outl), "mfma-oracle %.1e  valu-oracle %.1e  rows-oracle %.1e  mfma-valu %.1e" %
          (rel(res["mfma"], o), rel(res["valu"], o), rel(res["valu_rows"], o), rel(res["mfma"], res["valu"])))
