"""Workgroup timelines of the three per-LM-step kernels next to the LDL^T (library built with -DLBA_PROFILE as
tools/micro/variants/liborbgpu_lbaprof.so): for the LAST launch of each kernel in a C2 (or C4) local BA, when every workgroup ran
(100 MHz wall clock, common to all compute units), by role.   python3 tools/micro/lba_prof.py [C2|C4]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["ORBG_LIB"] = os.path.join(ROOT, "tools", "micro", "variants", "liborbgpu_lbaprof.so")
from multi_orbslam3_amd import api, synth, views, _capi
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
nfree, nfix, npts, W, H = (20, 10, 2000, 640, 480) if cfg == "C2" else (50, 20, 8000, 1280, 720)
prob = synth.make_lba_problem(n_free=nfree, n_fixed=nfix, n_points=npts, width=W, height=H)
p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
opt = api.Optimizer()
out = views.LbaOutput(p.n_poses, p.n_points, p.n_edges)
lib = _capi.load()
N = 8 + 2 * 1020
buf = (C.c_longlong * (3 * N))()
lib.pose_opt_debug_prof if False else None
for _ in range(5):
    opt.LocalBundleAdjustment(p, out=out)
lib.lba_debug_prof(buf, 1)
opt.LocalBundleAdjustment(p, out=out)
lib.lba_debug_prof(buf, 0)
a = np.frombuffer(buf, dtype=np.int64).reshape(3, N)
NE = p.n_edges
nP = nfree
n_eb, n_lb = (NE + 255) // 256, (4 * npts + 255) // 256
roles = {0: [("pose workgroups (one per free pose)", 0, nP), ("edge workgroups (256 edges each)", nP, nP + n_eb), ("landmark workgroups (64 landmarks x 4 lanes)", nP + n_eb, nP + n_eb + n_lb),
             ("publisher workgroup", nP + n_eb + n_lb, nP + n_eb + n_lb + 1)],
         1: [("pose-pair workgroups", 0, nP * (nP + 1) // 2)], 2: [("workgroups of 64 landmarks / poses", 0, (npts + nfree + nfix + 63) // 64)]}
names = ["k_errlin", "k_schur", "k_update"]
print("%s local BA, %d edges, %d free poses, %d landmarks; last launch of each kernel; times in us after the kernel's first workgroup started" % (cfg, NE, nP, npts))
for k in range(3):
    st, en = a[k, 8::2], a[k, 9::2]
    used = st > 0
    if not used.any():
        continue
    t0 = st[used].min()
    print("%s: %d workgroups, first start 0.00, last start %.2f, last end %.2f" % (names[k], int(used.sum()), (st[used].max() - t0) * 0.01, (en[used].max() - t0) * 0.01))
    for nm, lo, hi in roles[k]:
        hi = min(hi, 1020)
        s_, e_ = st[lo:hi], en[lo:hi]
        ok = s_ > 0
        if not ok.any():
            continue
        d = (e_[ok] - s_[ok]) * 0.01
        print("   %-40s %4d: start %.2f .. %.2f, duration median %.2f max %.2f, end max %.2f" % (nm, int(ok.sum()), (s_[ok].min() - t0) * 0.01, (s_[ok].max() - t0) * 0.01,
                                                                                             float(np.median(d)), d.max(), (e_[ok].max() - t0) * 0.01))
    if k == 1:
        print("   workgroup 0 (pair (0, 0), diagonal): item loop done at %.2f, block sums at %.2f" % (a[k, 0] * 0.01, a[k, 1] * 0.01))
    if k == 2:
        print("   workgroup 0: landmarks updated at %.2f" % (a[k, 0] * 0.01))
