"""SearchLocalPoints on local maps far larger than the benchmark's (each point of a small map repeated): agreement with the
oracle and time per call.  Run on the GPU box: python tools/search_large_map.py [repeats ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from multi_orbslam3_amd import api, synth
from oracle import binding as ob
import helpers
scene = synth.Scene(640, 480)
rng = np.random.RandomState(5)
frs = [helpers.oracle_stereo_frame(scene, k) for k in (2, 6)]
cur = helpers.oracle_stereo_frame(scene, 4)
mp = helpers.local_map_from(scene, frs, rng)
fv, keep = helpers.frame_view_of(scene, cur)
n = len(cur["kps"])
T = synth.perturb_pose(cur["Tcw"], rng).astype(np.float32)
F = api.Frame().upload(fv, keep)
m = api.ORBmatcher(0.8)
# "spread": only the first copy is in view, the others lie elsewhere in the world (a merged server map); otherwise every copy is
spread = os.environ.get("SPREAD", "1") != "0"
for rep in ([int(a) for a in sys.argv[1:]] or (1, 8, 64, 256)):
    big = {k: np.concatenate([v] * rep) for k, v in mp.items()}
    big["pos"] = big["pos"] + rng.randn(*big["pos"].shape) * 0.01
    if spread:
        npt = len(mp["pos"])
        for c in range(1, rep):
            big["pos"][c * npt:(c + 1) * npt] += np.array([30.0 * c, -20.0 * c, 2.0 * c], np.float32)
    wv, keep2 = helpers.world_view_of(big)
    amp0 = np.full(n, -1, np.int32); aob0 = np.zeros(n, np.int32)
    try:
        LM = api.LocalMap(len(big["pos"]) + 64).upload(wv)
        g = m.SearchLocalPoints(F, LM, T, 3.0, False, 0.0, amp0, aob0, None)
        t0 = time.perf_counter()
        for _ in range(5):
            g = m.SearchLocalPoints(F, LM, T, 3.0, False, 0.0, amp0, aob0, None)
        dt = (time.perf_counter() - t0) / 5
    except Exception as e:
        print(len(big["pos"]), "points: error", e); continue
    t1 = time.perf_counter(); o = ob.search_local_points(fv, wv, T, 3.0, False, 0.0, 0.8, amp0, aob0); t2 = time.perf_counter()
    print("spread" if spread else "all in view", len(big["pos"]), "points: %.1f ns per point, matches" % (1e9 * dt / len(big["pos"])), g[2], o[2], "equal", g[2] == o[2] and np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1]),
          "gpu %.3f ms oracle %.1f ms" % (1e3 * dt, 1e3 * (t2 - t1)))
