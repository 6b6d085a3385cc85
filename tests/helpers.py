"""Shared builders of seeded test inputs (tests only; uses the CPU oracle to produce features)."""
import numpy as np

from multi_orbslam3_amd import synth, views
from oracle import binding as ob


def oracle_stereo_frame(scene, k, n_features=1000):
    """Extract + stereo-match frame k with the oracle: dict(kps, desc, uright, depth, Tcw, exL, exR, L, R)."""
    L, R, Tcw = scene.stereo_pair(k)
    exL = ob.Extractor(n_features=n_features, max_width=scene.W, max_height=scene.H)
    exR = ob.Extractor(n_features=n_features, max_width=scene.W, max_height=scene.H)
    rc, kl, dl, _ = exL.extract(L)
    assert rc == 0
    rc, kr, dr, _ = exR.extract(R)
    assert rc == 0
    ur, dp = ob.stereo_match(exL, exR, kl, dl, kr, dr, float(scene.cam["bf"]), float(scene.cam["b"]))
    return dict(kps=kl, desc=dl, kps_r=kr, desc_r=dr, uright=ur, depth=dp, Tcw=Tcw, exL=exL, exR=exR, L=L, R=R)


def frame_view_of(scene, fr, with_stereo=True):
    p = scene.frame_view_params()
    return views.frame_view(fr["kps"], fr["desc"], fr["uright"] if with_stereo else None,
                            fr["depth"] if with_stereo else None, p["bounds"], p["cam"], 8, 1.2)


def world_view_of(mp, skip=None):
    return views.worldpoints_view(mp["pos"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], mp["n_obs"],
                                  mp["bad"], skip)


def local_map_from(scene, frames, rng=None):
    """Concatenate the map points created from several oracle frames (a small 'local map')."""
    parts = [synth.map_from_frame(f["kps"], f["desc"], f["depth"], f["Tcw"], scene.cam) for f in frames]
    out = {}
    for k in parts[0]:
        out[k] = np.concatenate([p[k] for p in parts])
    if rng is not None:   # shuffle so that vector order != spatial order, sprinkle bad / zero-observation points
        perm = rng.permutation(len(out["pos"]))
        for k in out:
            out[k] = np.ascontiguousarray(out[k][perm])
        out["bad"][rng.rand(len(perm)) < 0.02] = 1
        out["n_obs"][rng.rand(len(perm)) < 0.05] = 0
    return out


def lastframe_view_of(fr, rng=None):
    """LastFrame view: every stereo-matched feature carries a map point at its unprojected position."""
    from multi_orbslam3_amd import synth as s
    return fr


def make_lastframe(scene, fr, rng):
    Pw, valid = synth.unproject_to_world(fr["kps"], fr["depth"], fr["Tcw"], scene.cam)
    n = len(fr["kps"])
    outlier = (rng.rand(n) < 0.03).astype(np.uint8)
    n_obs = np.full(n, 2, np.int32)
    n_obs[rng.rand(n) < 0.05] = 0
    lv, keep = views.lastframe_view(valid.astype(np.uint8), outlier, Pw, fr["desc"], fr["kps"]["octave"], fr["kps"]["angle"],
                                    n_obs, fr["Tcw"].astype(np.float32))
    return lv, keep
