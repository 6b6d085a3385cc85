import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch
from multi_orbslam3_amd import api, synth, views, _capi as capi
from tests import helpers
sc = synth.Scene(640, 480)
fr = helpers.oracle_stereo_frame(sc, 3)
fv, keep = helpers.frame_view_of(sc, fr)
F = api.Frame(8192).upload(fv, keep)
n = len(fr["kps"])
w = torch.zeros(47 * n, dtype=torch.uint8, device="cuda:0")
F.pack_wire(device_ptr=w.data_ptr())
torch.cuda.synchronize()
K = api.Frame(8192)
mp = helpers.local_map_from(sc, [helpers.oracle_stereo_frame(sc, k) for k in (0, 1, 2)])
wv, keep3 = helpers.world_view_of(mp)
LM = api.LocalMap(16384); LM.upload(wv)
m = api.ORBmatcher(0.75, True)
free = np.full(n, -1, np.int32)
T = fr["Tcw"].astype(np.float32)
for _ in range(20):
    K.from_wire(fv, n=n, device_ptr=w.data_ptr()); m.SearchByProjectionSim3(K, T, LM, free, 4, 1.5)
t0 = time.perf_counter()
for _ in range(200): K.from_wire(fv, n=n, device_ptr=w.data_ptr())
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(200): m.SearchByProjectionSim3(K, T, LM, free, 4, 1.5)
t2 = time.perf_counter()
for _ in range(200):
    K.from_wire(fv, n=n, device_ptr=w.data_ptr()); m.SearchByProjectionSim3(K, T, LM, free, 4, 1.5)
t3 = time.perf_counter()
print("from_wire %.1f us, search_sim3 %.1f us, both %.1f us (n=%d, m=%d)" % (5e3 * (t1 - t0), 5e3 * (t2 - t1), 5e3 * (t3 - t2), n, wv.m))
# the same through the one-collective tick (1-rank RCCL group)
from multi_orbslam3_amd import harness
grp = harness.AgentGroup("nccl", force_group=True)
grp.open_data_plane()
bufs = grp.tick_buffers(max_features=2048, device="cuda:0", max_blocks=8)
blocks = [(n, w)] * 8
def tick(nb, do_search=True, do_wire=True):
    got = grp.all_gather_keyframe_blocks(bufs, blocks[:nb])
    for lst in got:
        for (nr, blk) in lst:
            if do_wire: K.from_wire(fv, n=nr, device_ptr=blk.data_ptr())
            if do_search: m.SearchByProjectionSim3(K, T, LM, free[:nr], 4, 1.5)
for nb in (2, 8):
    for ds, dw in ((False, False), (False, True), (True, True)):
        for _ in range(5): tick(nb, ds, dw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): tick(nb, ds, dw)
        torch.cuda.synchronize(); print("tick %d blocks, wire %s search %s: %.1f us" % (nb, dw, ds, 2e4 * (time.perf_counter() - t0)))
grp.close()
