import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from multi_orbslam3_amd import api, synth
from oracle import binding as ob
g = np.load("tests/golden/extract_160x120.npz")
def run(host, img, nf, nl, w, h):
    if host: os.environ["ORBG_HOST_OCTREE"] = "1"
    else: os.environ.pop("ORBG_HOST_OCTREE", None)
    ex = api.ORBextractor(nf, 1.2, nl, 20, 7, w, h, n_cams=1)
    nm, k, d = ex(img)
    return k, d, ex
for name, img, nf, nl in (("golden", g["L"], 300, 4), ("scene", synth.Scene(640, 480).stereo_pair(0)[0], 1000, 8)):
    h_, w_ = img.shape
    kh, dh, exh = run(True, img, nf, nl, w_, h_)
    kg, dg, exg = run(False, img, nf, nl, w_, h_)
    print(name, "host n", len(kh), "gpu n", len(kg))
    for l in range(nl):
        a = kh[kh["octave"] == l]; b = kg[kg["octave"] == l]
        same = len(a) == len(b) and np.array_equal(a, b)
        sa = set(map(tuple, np.stack([a["x"], a["y"]], 1).tolist())); sb = set(map(tuple, np.stack([b["x"], b["y"]], 1).tolist()))
        print("  level", l, "host", len(a), "gpu", len(b), "same order", same, "same set", sa == sb, "only host", len(sa - sb), "only gpu", len(sb - sa))
        if not same and len(a) and len(b):
            n = min(len(a), len(b))
            diff = [i for i in range(n) if a[i] != b[i]]
            print("    first diffs", diff[:5], a[diff[:2]] if diff else "", b[diff[:2]] if diff else "")
