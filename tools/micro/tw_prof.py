"""Phase timestamps of pyr_tower_kernel (library built with -DTW_PROFILE as tools/micro/variants/liborbgpu_twprof.so)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["ORBG_LIB"] = os.path.join(ROOT, "tools", "micro", "variants", "liborbgpu_twprof.so")
from multi_orbslam3_amd import api, synth, _capi
sc = synth.Scene(640, 480)
L, R, _ = sc.stereo_pair(5)
ex = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=2)
for _ in range(3):
    ex.extract_stereo(L, R)
lib = _capi.load()
buf = (C.c_longlong * 32)()
lib.orbx_debug_tw_prof.argtypes = [C.c_void_p]
assert lib.orbx_debug_tw_prof(buf) == 0
p = np.frombuffer(buf, dtype=np.int64)
names = ["tile ranges + geometry", "image loads issued, taps built", "level 0 to LDS + pyramid"] + ["level %d" % l for l in range(1, 8)]
for i, nm in enumerate(names):
    print("%-34s %7d cycles" % (nm, p[i + 1] - p[i]))
print("total %d cycles" % (p[10] - p[0]))
