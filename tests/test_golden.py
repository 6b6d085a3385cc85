"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py with the oracle):
the oracle must keep reproducing them on CPU; the HIP path must reproduce them on the GPU box."""
import os

import numpy as np
import pytest

from multi_orbslam3_amd import synth, views
from oracle import binding as ob

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def test_oracle_reproduces_golden_extraction():
    g = _load("extract_160x120.npz")
    sc = synth.Scene(160, 120, tex_size=(400, 300), px_per_m=50.0)
    exL = ob.Extractor(n_features=300, n_levels=4, max_width=160, max_height=120)
    exR = ob.Extractor(n_features=300, n_levels=4, max_width=160, max_height=120)
    rc, kl, dl, _ = exL.extract(g["L"])
    rc, kr, dr, _ = exR.extract(g["R"])
    assert np.array_equal(kl, g["kps_l"]) and np.array_equal(dl, g["desc_l"])
    assert np.array_equal(kr, g["kps_r"]) and np.array_equal(dr, g["desc_r"])
    assert np.array_equal(exL.level(1), g["level1"]) and np.array_equal(exL.level(3), g["level3"])
    for l in range(4):
        assert np.array_equal(exL.candidates(l), g["cand%d" % l])
    ur, dp = ob.stereo_match(exL, exR, kl, dl, kr, dr, float(sc.cam["bf"]), float(sc.cam["b"]))
    assert np.array_equal(ur, g["uright"]) and np.array_equal(dp, g["depth"])
    assert len(kl) > 100 and (ur > 0).sum() > 20


def test_oracle_reproduces_golden_hamming_and_lba():
    g = _load("hamming_24x40.npz")
    assert np.array_equal(ob.hamming_matrix(g["q"], g["t"]), g["dist"])
    assert np.array_equal(ob.hamming_best2(g["q"], g["t"]), g["best2"])
    assert np.array_equal(g["dist"], np.unpackbits(g["q"][:, None, :] ^ g["t"][None, :, :], axis=2).sum(axis=2))
    b = _load("lba_5p2_60.npz")
    p, keep = views.lba_problem(b["poses"], b["pose_fixed"], b["points"], b["edges"], tuple(b["cam"]))
    o = ob.lba_solve(p)
    assert o.status == int(b["status"][0]) and tuple(o.iters) == tuple(b["iters"])
    assert np.allclose(o.poses, b["out_poses"], atol=1e-6) and np.allclose(o.points, b["out_points"], atol=1e-6)
    assert np.array_equal(o.edge_outlier, b["out_outlier"])


@pytest.mark.gpu
def test_gpu_reproduces_golden():
    from multi_orbslam3_amd import api
    g = _load("extract_160x120.npz")
    ex = api.ORBextractor(300, 1.2, 4, 20, 7, 160, 120, n_cams=2)
    (kl, dl), (kr, dr) = ex.extract_stereo(g["L"], g["R"])
    assert np.array_equal(kl, g["kps_l"]) and np.array_equal(dl, g["desc_l"])
    assert np.array_equal(kr, g["kps_r"]) and np.array_equal(dr, g["desc_r"])
    assert np.array_equal(ex.level(0, 1), g["level1"]) and np.array_equal(ex.level(0, 3), g["level3"])
    for l in range(4):
        assert np.array_equal(ex.candidates(0, l), g["cand%d" % l])
    sc = synth.Scene(160, 120, tex_size=(400, 300), px_per_m=50.0)
    ur, dp = ex.ComputeStereoMatches(float(sc.cam["bf"]), float(sc.cam["b"]), n_left=len(kl))
    assert np.array_equal(ur, g["uright"]) and np.array_equal(dp, g["depth"])
    h = _load("hamming_24x40.npz")
    m = api.ORBmatcher()
    assert np.array_equal(m.DescriptorDistance(h["q"], h["t"]), h["dist"]) and np.array_equal(m.best2(h["q"], h["t"]), h["best2"])
    b = _load("lba_5p2_60.npz")
    p, keep = views.lba_problem(b["poses"], b["pose_fixed"], b["points"], b["edges"], tuple(b["cam"]))
    o = api.Optimizer().LocalBundleAdjustment(p)
    assert o.status == int(b["status"][0]) and tuple(o.iters) == tuple(b["iters"])
    assert np.abs(o.poses - b["out_poses"]).max() <= 1e-4 and np.abs(o.points - b["out_points"]).max() <= 1e-4
    assert np.array_equal(o.edge_outlier, b["out_outlier"])
