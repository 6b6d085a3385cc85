"""What do kernel BOUNDARIES on other streams cost the local BA?  Times the C2 solve alone and next to 1 / 2 / 3 host threads that each
launch tiny kernels (one element, ~5 us apart) on a stream of their own -- no memory traffic, no fences inside: only dispatches."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from multi_orbslam3_amd import api, synth, views

prob = synth.make_lba_problem(n_free=20, n_fixed=10, n_points=2000, width=640, height=480)
p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
opt = api.Optimizer()
out = views.LbaOutput(p.n_poses, p.n_points, p.n_edges)
stop = False
counts = [0, 0, 0]

def spam(i, big):
    s = torch.cuda.Stream()
    x = torch.zeros(1 if not big else 1 << 20, device="cuda")
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(64):
                x.add_(1.0)
            counts[i] += 64
            s.synchronize()

def measure(tag, reps=150):
    for _ in range(10):
        opt.LocalBundleAdjustment(p, out=out)
    c0 = sum(counts); t00 = time.perf_counter()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); opt.LocalBundleAdjustment(p, out=out); ts.append(time.perf_counter() - t0)
    dt = time.perf_counter() - t00
    print("%-40s local BA median %.4f ms (min %.4f), neighbour dispatches %.0f k/s" % (tag, np.median(ts) * 1e3, min(ts) * 1e3, (sum(counts) - c0) / dt / 1e3))

measure("alone")
for big in (False, True):
    threads = []
    for i in range(3):
        stop = False
        t = threading.Thread(target=spam, args=(i, big)); t.start(); threads.append(t)
        time.sleep(0.3)
        measure("%d thread(s) of %s kernels" % (i + 1, "4 MB add" if big else "1-element"))
    stop = True
    for t in threads:
        t.join()
