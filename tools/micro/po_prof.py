"""Phase breakdown of pose_opt_kernel (library built with -DPO_PROFILE as tools/micro/variants/liborbgpu_poprof.so)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["ORBG_LIB"] = os.path.join(ROOT, "tools", "micro", "variants", "liborbgpu_poprof.so")
from multi_orbslam3_amd import api, synth, views, _capi
opt = api.Optimizer(0)
pr = synth.make_pose_opt_problem(n=450, seed=77)
po, keep = views.pose_opt_problem(pr["Xw"], pr["u"], pr["v"], pr["ur"], pr["inv_sigma2"], pr["cam"], pr["Tcw"], device=0)
lib = _capi.load()
buf = (C.c_longlong * 16)()
lib.pose_opt_debug_prof.argtypes = [C.c_void_p, C.c_int]
for _ in range(3):
    opt.PoseOptimization(po)
lib.pose_opt_debug_prof(buf, 1)
r = opt.PoseOptimization(po)
lib.pose_opt_debug_prof(buf, 0)
names = ["buildSystem", "reduce28", "solve6+oplus", "trial eval", "trial sum", "control/rest"]
tot = sum(buf[i] for i in range(6))
print("iters", r.iters, "total cycles", tot)
for i, nm in enumerate(names):
    print("%-14s %8d cycles  %5.1f %%" % (nm, buf[i], 100.0 * buf[i] / tot))
print("inside reduce28 (thread 0): fold32 %d, fold16 %d, LDS stores %d, barrier %d, column sums %d, barrier %d" % tuple(buf[8:14]))
print("inside solve6+oplus (thread 0 of wavefront 0): po_solve6 %d, pose_oplus_series %d, candidate to LDS + barrier %d, candidate back from LDS %d" % (buf[6], buf[7], buf[14], buf[15]))
