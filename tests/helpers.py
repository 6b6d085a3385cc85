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


def make_lastframe(scene, fr, rng):
    Pw, valid = synth.unproject_to_world(fr["kps"], fr["depth"], fr["Tcw"], scene.cam)
    n = len(fr["kps"])
    outlier = (rng.rand(n) < 0.03).astype(np.uint8)
    n_obs = np.full(n, 2, np.int32)
    n_obs[rng.rand(n) < 0.05] = 0
    lv, keep = views.lastframe_view(valid.astype(np.uint8), outlier, Pw, fr["desc"], fr["kps"]["octave"], fr["kps"]["angle"],
                                    n_obs, fr["Tcw"].astype(np.float32))
    return lv, keep


def random_database(rng, n_kfs=300, n_words=4000, words_per_kf=(60, 260), n_maps=2, bad_frac=0.05):
    """A synthetic KeyFrameDatabase: keyframes cluster around a few "places" (shared word pools) so that queries find groups of
    similar keyframes; inverted file in keyframe-insertion order, L1-normalised BowVectors, covisibility = neighbours in id."""
    import numpy as np
    places = [rng.choice(n_words, 500, replace=False) for _ in range(12)]
    bows, inv = [], {}
    for k in range(n_kfs):
        pool = places[(k // 7) % len(places)]
        nw = rng.randint(*words_per_kf)
        w = np.unique(np.concatenate([rng.choice(pool, int(nw * 0.8)), rng.choice(n_words, nw - int(nw * 0.8))])).astype(np.int32)
        v = rng.rand(len(w)) + 0.05
        v /= v.sum()
        bows.append((w, v))
        for x in w:
            inv.setdefault(int(x), []).append(k)
    covis = []
    for k in range(n_kfs):
        c = [j for j in (k - 1, k + 1, k - 2, k + 2, k - 3, k + 7, k - 7, k + 14) if 0 <= j < n_kfs]
        rng.shuffle(c)
        covis.append(c[: rng.randint(0, 9)])
    map_id = (np.arange(n_kfs) * n_maps // n_kfs).astype(np.int32)
    bad = (rng.rand(n_kfs) < bad_frac).astype(np.uint8)
    map_bad = np.zeros(n_kfs, np.uint8)
    return dict(inv=inv, bows=bows, covis=covis, map_id=map_id, bad=bad, map_bad=map_bad, n_words=n_words)


def rig_track_views(sc):
    """Views of synth.make_rig_track_scene: (left frame view, right frame view, world points view, camera rig, keep-alive list)."""
    from multi_orbslam3_amd import views
    bounds = (0, sc["size"], 0, sc["size"])
    cam = (sc["left"][1], sc["left"][2], sc["left"][3], sc["left"][4], 0.0, 0.1)                    # (mb: the forward / backward test of the frame search)
    fl, k1 = views.frame_view(sc["kps_left"], sc["desc_left"], None, None, bounds, cam)
    fr, k2 = views.frame_view(sc["kps_right"], sc["desc_right"], None, None, bounds, cam)
    wv, k3 = views.worldpoints_view(sc["pos"], sc["normal"], sc["min_dist"], sc["max_dist"], sc["desc"], sc["n_obs"], sc["bad"])
    rig = views.camera_rig(sc["left"], sc["right"], sc["Trl"])
    return fl, fr, wv, rig, [k1, k2, k3]


def rig_mappoint_views(sc, a, b):
    """orbm_mappoints_view pair (left camera's track fields, right camera's) from the two dicts of is_in_frustum_rig."""
    from multi_orbslam3_amd import views
    mv, k4 = views.mappoints_view(a["track_in_view"], sc["bad"], a["proj_x"], a["proj_y"], a["proj_x"], a["track_depth"], a["scale_level"],
                                  a["view_cos"], sc["desc"], sc["n_obs"])
    mvr, k5 = views.mappoints_view(b["track_in_view"], sc["bad"], b["proj_x"], b["proj_y"], b["proj_x"], b["track_depth"], b["scale_level"],
                                   b["view_cos"], sc["desc"], sc["n_obs"])
    return mv, mvr, [k4, k5]
