"""Per-pass cycle counts of octree_kernel (library built with -DOCT_PROFILE as tools/micro/variants/liborbgpu_octprof.so)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["ORBG_LIB"] = os.path.join(ROOT, "tools", "micro", "variants", "liborbgpu_octprof.so")
from multi_orbslam3_amd import api, synth
sc = synth.Scene(640, 480)
L, R, _ = sc.stereo_pair(5)
ex = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=2)
for _ in range(3):
    ex.extract_stereo(L, R)
from multi_orbslam3_amd import _capi
lib = _capi.load()
buf = (C.c_longlong * (32 * 48))()
lib.orbx_debug_oct_prof.argtypes = [C.c_void_p]
assert lib.orbx_debug_oct_prof(buf) == 0
p = np.frombuffer(buf, dtype=np.int64).reshape(32, 48)
for b in range(16):
    t0 = p[b, 0]
    nk, n = p[b, 46] >> 16, p[b, 46] & 0xFFFF
    line = "blk %2d (cam %d lvl %d) nk=%4d n=%4d  setup %5d loop %6d tail %5d | " % (b, b % 2, b // 2, nk, n, p[b, 1] - t0, p[b, 2] - p[b, 1], p[b, 3] - p[b, 2])
    k = 0
    while 4 + 2 * k < 46 and p[b, 4 + 2 * k] > t0 and (k == 0 or p[b, 4 + 2 * k] > p[b, 2 + 2 * k]):
        nxt = p[b, 6 + 2 * k] if (6 + 2 * k < 46 and p[b, 6 + 2 * k] > p[b, 4 + 2 * k]) else p[b, 2]
        line += "[n=%d m%d %d] " % (p[b, 5 + 2 * k] >> 2, p[b, 5 + 2 * k] & 3, nxt - p[b, 4 + 2 * k])
        k += 1
    print(line)

for b in (0, 2):
    if p[b, 27] > p[b, 20] > 0:
        print("blk %d jump start: before %d | codes + histogram %d | barrier %d | masks %d | barrier %d | decisions + nodes %d | key positions %d | barrier %d" %
              (b, p[b,20]-p[b,0], p[b,21]-p[b,20], p[b,22]-p[b,21], p[b,23]-p[b,22], p[b,24]-p[b,23], p[b,25]-p[b,24], p[b,26]-p[b,25], p[b,27]-p[b,26]))
for b in (0, 2):
    print("blk %d pass 2 (mode 1): keys %d | barrier %d | scan %d | children %d | barrier %d | follow %d" % (b, p[b,30]-p[b,8], p[b,31]-p[b,30], p[b,32]-p[b,31], p[b,33]-p[b,32], p[b,34]-p[b,33], p[b,35]-p[b,34]))
b = 0
print("blk 0 pass 4 (mode 2): keys+rankkeys %d | barrier %d | rank %d | barrier %d | order scan+cut %d | (barrier+C+surv scan) %d | children %d | barrier+follow %d" % (p[b,36]-p[b,12], p[b,37]-p[b,36], p[b,38]-p[b,37], p[b,39]-p[b,38], p[b,40]-p[b,39], p[b,41]-p[b,40], p[b,42]-p[b,41], p[b,43]-p[b,42]))
