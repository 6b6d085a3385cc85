"""Python model of the closed-form quad-tree used by octree_kernel, checked against the oracle on CPU."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import binding as ob

def model(c, minX, maxX, minY, maxY, N, verbose=False):
    if len(c) == 0: return []
    W, H = maxX - minX, maxY - minY
    nIni = max(int(np.floor(np.float32(W) / np.float32(H) + np.float32(0.5))), 1)
    hX = np.float32(W) / np.float32(nIni)
    node_of = np.array([min(int(np.float32(x) / hX), nIni - 1) for x in c[:, 0]])
    nodes = []   # dict(box, cnt, seq)
    remap = {}
    for r in range(nIni):
        cnt = int((node_of == r).sum())
        if cnt:
            remap[r] = len(nodes)
            nodes.append(dict(x0=int(hX * np.float32(r)), x1=int(hX * np.float32(r + 1)), y0=0, y1=H, cnt=cnt, seq=len(nodes)))
    node_of = np.array([remap[r] for r in node_of])
    mode = 1
    for it in range(64):
        n = len(nodes); prev = n
        exp = [i for i in range(n) if nodes[i]["cnt"] > 1]
        if not exp: break
        if mode == 1: order = exp
        else: order = sorted(exp, key=lambda i: (nodes[i]["cnt"], nodes[i]["seq"]), reverse=True)
        # children counts
        kids = {}
        for i in exp:
            nd = nodes[i]
            mx = nd["x0"] + ((nd["x1"] - nd["x0"] + 1) >> 1); my = nd["y0"] + ((nd["y1"] - nd["y0"] + 1) >> 1)
            keys = np.nonzero(node_of == i)[0]
            q = (c[keys, 0] >= mx).astype(int) + 2 * (c[keys, 1] >= my).astype(int)
            kids[i] = (mx, my, keys, q)
        cut = len(order) - 1
        if mode == 2:
            size = n
            for o, i in enumerate(order):
                k4 = len(set(kids[i][3].tolist()))
                size += k4 - 1
                if size >= N: cut = o; break
        P = order[:cut + 1]
        children = []
        child_of = {}
        nexp = 0
        for i in P:
            mx, my, keys, q = kids[i]; nd = nodes[i]
            for qq in range(4):
                cnt = int((q == qq).sum())
                if cnt == 0: continue
                child_of[(i, qq)] = len(children)
                children.append(dict(x0=mx if qq & 1 else nd["x0"], x1=nd["x1"] if qq & 1 else mx, y0=my if qq & 2 else nd["y0"],
                                     y1=nd["y1"] if qq & 2 else my, cnt=cnt, seq=len(children)))
                nexp += cnt > 1
        C = len(children)
        Pset = set(P)
        surv = [i for i in range(n) if i not in Pset]
        new_nodes = list(reversed(children)) + [nodes[i] for i in surv]
        new_of = node_of.copy()
        for i in P:
            mx, my, keys, q = kids[i]
            for k, qq in zip(keys, q):
                new_of[k] = C - 1 - child_of[(i, int(qq))]
        for r, i in enumerate(surv):
            new_of[node_of == i] = C + r
        nodes, node_of = new_nodes, new_of
        n = len(nodes)
        if verbose: print("  pass", it, "mode", mode, "n", prev, "->", n, "nexp", nexp, "cut", cut, "of", len(order))
        if n >= N or n == prev: break
        if mode == 1 and n + 3 * nexp > N: mode = 2
    out = []
    for i in range(len(nodes)):
        keys = np.nonzero(node_of == i)[0]
        best = keys[0]
        for k in keys[1:]:
            if c[k, 2] > c[best, 2]: best = k
        out.append(tuple(c[best]))
    return out

if __name__ == "__main__":
    g = np.load("tests/golden/extract_160x120.npz")
    ex = ob.Extractor(n_features=300, n_levels=4, max_width=160, max_height=120)
    ex.extract(g["L"])
    fpl = ex.tables()[4]
    for l in range(4):
        c = ex.candidates(l)
        w, h = ex.level(l).shape[::-1]
        ref = [tuple(r) for r in ob.distribute_octree(c, 16, w - 16, 16, h - 16, int(fpl[l]))]
        mod = model(c, 16, w - 16, 16, h - 16, int(fpl[l]), verbose=(l >= 2))
        print("level", l, "N", fpl[l], "cands", len(c), "oracle", len(ref), "model", len(mod), "equal", ref == mod)
    rng = np.random.RandomState(0)
    bad = 0
    for t in range(300):
        n = rng.randint(1, 400); N = rng.randint(1, 300)
        xs = rng.randint(0, 300, n); ys = rng.randint(0, 200, n)
        _, first = np.unique(xs * 1000 + ys, return_index=True)
        c = np.stack([xs, ys, rng.randint(7, 255, n)], 1)[np.sort(first)].astype(np.int32)
        ref = [tuple(r) for r in ob.distribute_octree(c, 16, 316, 16, 216, N)]
        mod = model(c, 16, 316, 16, 216, N)
        bad += ref != mod
    print("random trials mismatching:", bad)
