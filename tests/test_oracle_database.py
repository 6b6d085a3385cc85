"""Known-answer tests for the oracle's DetectNBestCandidates (S/KeyFrameDatabase.cc:594-761): a hand-computed case and a pure-Python
model with the reference's per-keyframe state, so that the oracle the GPU is compared with is itself pinned to first principles."""
import numpy as np

from multi_orbslam3_amd import views
from oracle import binding as ob
import helpers


def _l1(qw, qv, w, v):
    a = dict(zip(qw.tolist(), qv.tolist()))
    s = 0.0
    for x, y in zip(w.tolist(), v.tolist()):
        if x in a:
            s += abs(a[x] - y) - abs(a[x]) - abs(y)
    return -s / 2.0


def _model(db, qw, qv, connected, qmap, n, place):
    """Pure-Python transcription of the algorithm on index arrays (lists instead of std::list, sorted() is stable)."""
    K = len(db["bows"])
    queried, words, order = [False] * K, [0] * K, []
    for w in qw.tolist():
        for kf in db["inv"].get(int(w), []):
            if not queried[kf]:
                words[kf] = 0
                if not connected[kf]:
                    queried[kf] = True
                    order.append(kf)
            words[kf] += 1
    if not order:
        return [], []
    mx = max(words[k] for k in order)
    mn = int(np.float32(mx) * np.float32(0.8))
    scored = []
    for kf in order:
        if words[kf] > mn:
            si = np.float32(_l1(qw, qv, *db["bows"][kf]))
            place[kf] = si
            scored.append((si, kf))
    acc = []
    for si, kf in scored:
        best, bs, a = kf, si, np.float32(si)
        for k2 in db["covis"][kf]:
            if not queried[k2]:
                continue
            a = np.float32(a + place[k2])
            if place[k2] > bs:
                best, bs = k2, place[k2]
        acc.append((a, best))
    acc = sorted(acc, key=lambda t: -t[0])
    loop, merge, added = [], [], set()
    for a, kf in acc:
        if not (len(loop) < n or len(merge) < n):
            break
        if db["bad"][kf] or kf in added:
            continue
        if db["map_id"][kf] == qmap and len(loop) < n:
            loop.append(kf)
        elif db["map_id"][kf] != qmap and len(merge) < n and not db["map_bad"][kf]:
            merge.append(kf)
        added.add(kf)
    return loop, merge


def test_hand_computed_case():
    # 4 keyframes, 6 words.  Query = words {0,1,2} at 1/3 each.  KF0 shares 3 words, KF1 shares 2 (< 0.8*3 -> not scored, int(2.4)=2, needs > 2),
    # KF2 shares 3 words and is in another map, KF3 is connected to the query (excluded although it shares everything).
    bows = [(np.array([0, 1, 2], np.int32), np.array([1 / 3, 1 / 3, 1 / 3])), (np.array([0, 1, 5], np.int32), np.array([0.25, 0.25, 0.5])),
            (np.array([0, 1, 2, 4], np.int32), np.array([0.25, 0.25, 0.25, 0.25])), (np.array([0, 1, 2], np.int32), np.array([1 / 3, 1 / 3, 1 / 3]))]
    inv = {0: [0, 1, 2, 3], 1: [0, 1, 2, 3], 2: [0, 2, 3], 4: [2], 5: [1]}
    covis = [[1], [0], [], []]
    v, keep = views.database_view(inv, bows, covis, [0, 0, 1, 0], [0, 0, 0, 0], [0, 0, 0, 0], 6)
    place = np.zeros(4, np.float32)
    place[1] = 0.5                                  # stale score of an earlier query: KF1 shares words, so KF0's group adds it
    qw, qv = np.array([0, 1, 2], np.int32), np.array([1 / 3, 1 / 3, 1 / 3])
    loop, merge = ob.detect_n_best_candidates(v, qw, qv, [0, 0, 0, 1], 0, 3, place)
    assert loop.tolist() == [0] and merge.tolist() == [2]
    assert abs(place[0] - 1.0) < 1e-6 and abs(place[2] - 0.75) < 1e-6 and place[1] == 0.5 and place[3] == 0.0


def test_oracle_matches_the_python_model():
    rng = np.random.RandomState(5)
    db = helpers.random_database(rng)
    v, keep = views.database_view(db["inv"], db["bows"], db["covis"], db["map_id"], db["bad"], db["map_bad"], db["n_words"])
    K = len(db["bows"])
    place_o = np.zeros(K, np.float32)
    place_m = np.zeros(K, np.float32)
    hits = 0
    for q in range(12):
        kq = rng.randint(K)
        qw, qv = db["bows"][kq]
        con = np.zeros(K, np.uint8)
        con[max(kq - 3, 0): kq + 4] = 1            # the query keyframe and its neighbours are "connected"
        loop, merge = ob.detect_n_best_candidates(v, qw, qv, con, int(db["map_id"][kq]), 3, place_o)
        ml, mm = _model(db, qw, qv, con, int(db["map_id"][kq]), 3, place_m)
        assert loop.tolist() == ml and merge.tolist() == mm, q
        assert np.array_equal(place_o, place_m)
        hits += len(ml) + len(mm)
    assert hits > 12
