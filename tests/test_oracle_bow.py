"""Known-answer tests that pin the oracle's bag-of-words restatement against independent,
straightforward Python models of DBoW2's transform and MapPoint::ComputeDistinctiveDescriptors."""
import numpy as np
import pytest

from multi_orbslam3_amd import _capi as capi
from multi_orbslam3_amd import synth, views
from oracle import binding as ob


def _ham(a, b):
    return int(np.unpackbits(a ^ b).sum())


def _py_transform(v, f, levelsup):
    nid_level = v["L"] - levelsup
    nid, node, level = 0, 0, 0
    cs, ci = v["child_start"], v["child_ids"]
    while cs[node + 1] > cs[node]:
        level += 1
        kids = ci[cs[node]:cs[node + 1]]
        d = [_ham(f, v["desc"][k]) for k in kids]
        node = int(kids[int(np.argmin(d))])              # np.argmin returns the FIRST minimum = strict '<' scan
        if level == nid_level:
            nid = node
    return int(v["word_id"][node]), nid, float(v["weight"][node])


@pytest.mark.parametrize("k,L,levelsup", [(10, 3, 2), (4, 5, 4), (10, 2, 4), (3, 4, 0)])
def test_transform_walk_vs_python(k, L, levelsup):
    rng = np.random.RandomState(10 * k + L)
    v = synth.make_vocabulary(k=k, L=L, seed=k * 100 + L)
    vv, keep = views.vocab_view(v["child_start"], v["child_ids"], v["desc"], v["weight"], v["word_id"], L)
    feats = rng.randint(0, 256, (200, 32)).astype(np.uint8)
    feats[:20] = v["desc"][rng.randint(1, len(v["desc"]), 20)]        # exact hits
    wid, nid, w = ob.vocab_transform(vv, feats, levelsup)
    for i in range(len(feats)):
        assert (int(wid[i]), int(nid[i]), float(w[i])) == _py_transform(v, feats[i], levelsup), i
    if L - levelsup <= 0:
        assert (nid == 0).all()                                        # root (TemplatedVocabulary.h:1224)
    # a feature equal to a leaf descriptor whose ancestors are also nearest reaches that leaf: distance 0 can only be beaten by an earlier 0
    leaves = np.nonzero(v["word_id"] >= 0)[0]
    assert set(np.unique(wid)) <= set(v["word_id"][leaves])


def test_bow_and_feature_vector_bookkeeping():
    rng = np.random.RandomState(3)
    v = synth.make_vocabulary(k=6, L=3, seed=5, stop_frac=0.2)
    feats = rng.randint(0, 256, (400, 32)).astype(np.uint8)
    for weighting, norm in [(capi.ORBV_TF_IDF, capi.ORBV_NORM_L1), (capi.ORBV_TF, capi.ORBV_NORM_NONE), (capi.ORBV_IDF, capi.ORBV_NORM_L2),
                            (capi.ORBV_BINARY, capi.ORBV_NORM_L1)]:
        vv, keep = views.vocab_view(v["child_start"], v["child_ids"], v["desc"], v["weight"], v["word_id"], 3, weighting, norm)
        (bw, bv), (fn, fs, ff) = ob.vocab_bow(vv, feats, 1)
        wid, nid, w = ob.vocab_transform(vv, feats, 1)
        live = w > 0
        assert (~live).sum() > 10                                       # stopped words exist and are dropped everywhere
        assert np.array_equal(bw, np.unique(wid[live]))                 # ascending word ids
        # values: python model with a dict, same accumulation order
        vals = {}
        for i in np.nonzero(live)[0]:
            if weighting in (capi.ORBV_TF_IDF, capi.ORBV_TF):
                vals[int(wid[i])] = vals.get(int(wid[i]), 0.0) + float(w[i])
            else:
                vals.setdefault(int(wid[i]), float(w[i]))
        ref = np.array([vals[int(k2)] for k2 in bw])
        if weighting in (capi.ORBV_TF_IDF, capi.ORBV_TF) and norm == capi.ORBV_NORM_NONE:
            ref = ref / float(len(ref))
        if norm == capi.ORBV_NORM_L1:
            s = 0.0
            for x in ref:
                s += abs(x)
            ref = ref / s
        elif norm == capi.ORBV_NORM_L2:
            s = 0.0
            for x in ref:
                s += x * x
            ref = ref / np.sqrt(s)
        assert np.array_equal(bv, ref)
        # feature vector: ascending node ids, ascending feature indices inside a node, exactly the live features
        assert np.array_equal(fn, np.unique(nid[live]).astype(np.uint32))
        assert np.array_equal(np.sort(ff), np.nonzero(live)[0].astype(np.uint32))
        for a in range(len(fn)):
            seg = ff[fs[a]:fs[a + 1]]
            assert (np.diff(seg.astype(np.int64)) > 0).all() and (nid[seg] == fn[a]).all()


def test_distinctive_descriptor_vs_python():
    rng = np.random.RandomState(9)
    base = rng.randint(0, 256, (40, 32)).astype(np.uint8)
    lists, start = [], [0]
    for p in range(40):
        N = [0, 1, 2, 3, 4, 7, 8, 20, 65][p % 9]
        d = np.repeat(base[p][None], N, 0) ^ (rng.randint(0, 256, (N, 32)).astype(np.uint8) & rng.randint(0, 256, (N, 32)).astype(np.uint8)
                                              & rng.randint(0, 256, (N, 32)).astype(np.uint8))
        if N >= 4:
            d[1] = d[0]                                                  # ties: first index must win
        lists.append(d); start.append(start[-1] + N)
    desc = np.concatenate(lists) if start[-1] else np.zeros((0, 32), np.uint8)
    best = ob.distinctive_descriptors(desc, start)
    for p in range(40):
        d = lists[p]; N = len(d)
        if N == 0:
            assert best[p] == -1
            continue
        D = np.array([[_ham(d[i], d[j]) for j in range(N)] for i in range(N)])
        med = [int(np.sort(D[i])[int(0.5 * (N - 1))]) for i in range(N)]
        assert best[p] == int(np.argmin(med)), p


def test_wire_block_layout_and_conversions():
    """KF wire block = N x {f32 x, f32 y, u8 size, f32 angle, u8 response, i8 octave} + N x u8[32] (KF.msg / CvKeyPoint.msg)."""
    import struct
    rng = np.random.RandomState(2)
    n = 7
    kps = np.zeros(n, capi.KEYPOINT_DTYPE)
    kps["x"] = rng.rand(n) * 600; kps["y"] = rng.rand(n) * 400; kps["size"] = 31 * 1.2 ** rng.randint(0, 8, n)
    kps["angle"] = rng.rand(n) * 360; kps["response"] = rng.randint(7, 256, n) + 0.0; kps["octave"] = rng.randint(0, 8, n)
    desc = rng.randint(0, 256, (n, 32)).astype(np.uint8)
    wire = ob.wire_pack(kps, desc)
    assert wire.shape == (47 * n,)
    for i in range(n):
        x, y, size, angle, resp, octv = struct.unpack_from("<ffBfBb", wire.tobytes(), 15 * i)
        assert (np.float32(x), np.float32(y), np.float32(angle)) == (kps["x"][i], kps["y"][i], kps["angle"][i])
        assert size == int(kps["size"][i]) and resp == int(kps["response"][i]) and octv == kps["octave"][i]   # (u_int8_t) casts truncate
        assert np.array_equal(wire[15 * n + 32 * i: 15 * n + 32 * (i + 1)], desc[i])
    k2, d2 = ob.wire_unpack(wire, n)
    assert np.array_equal(d2, desc)
    for f in ("x", "y", "angle", "octave"):
        assert np.array_equal(k2[f], kps[f])
    assert np.array_equal(k2["size"], np.floor(kps["size"])) and np.array_equal(k2["response"], kps["response"])


def test_l1_score_vs_python():
    rng = np.random.RandomState(6)
    def bow(nw):
        w = np.sort(rng.choice(5000, nw, replace=False)).astype(np.int32)
        v = rng.rand(nw); v /= v.sum()
        return w, v
    qw, qv = bow(300)
    cands = [bow(n) for n in (0, 1, 50, 300, 800)] + [(qw.copy(), qv.copy())]
    cs = np.cumsum([0] + [len(c[0]) for c in cands]).astype(np.int32)
    cw = np.concatenate([c[0] for c in cands]); cv = np.concatenate([c[1] for c in cands])
    s = ob.score_l1(qw, qv, cs, cw, cv)
    for c, (w, v) in enumerate(cands):
        common = np.intersect1d(qw, w)
        acc = 0.0
        for word in common:                                          # ascending word order, as the merge walk visits them
            vi = float(qv[np.searchsorted(qw, word)]); wi = float(v[np.searchsorted(w, word)])
            acc += abs(vi - wi) - abs(vi) - abs(wi)
        assert s[c] == -acc / 2.0
    assert s[0] == 0.0 and abs(s[-1] - 1.0) < 1e-12                  # nothing in common -> 0, identical vectors -> 1
