"""The GPU quad-tree (octree_kernel) relies on a closed form of the reference's std::list bookkeeping:
    new list = reverse(children of the processed nodes, in processing order)  ++  untouched nodes (order kept),
phase 2 = descending (key count, creation order) with a prefix-sum cut at N.  tests/octree_model.py is a direct Python
model of exactly that formulation; here it is checked against the oracle's list-based restatement of
DistributeOctTree (S/ORBextractor.cc:537-761) on random candidate sets, including N above / below what is reachable."""
import numpy as np

from oracle import binding as ob
from octree_model import model


def test_closed_form_equals_list_based_quadtree():
    rng = np.random.RandomState(123)
    checked_phase2 = 0
    for trial in range(120):
        n = rng.randint(1, 500)
        N = rng.randint(1, 320)
        w, h = [(300, 200), (608, 448), (150, 300), (79, 51)][rng.randint(4)]
        xs = rng.randint(0, w, n); ys = rng.randint(0, h, n)
        _, first = np.unique(xs * 10000 + ys, return_index=True)
        c = np.stack([xs, ys, rng.randint(7, 255, n)], 1)[np.sort(first)].astype(np.int32)
        ref = [tuple(r) for r in ob.distribute_octree(c, 16, 16 + w, 16, 16 + h, N)]
        assert model(c, 16, 16 + w, 16, 16 + h, N) == ref, (trial, n, N, w, h)
        checked_phase2 += len(ref) >= N
    assert checked_phase2 > 20
