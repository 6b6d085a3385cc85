"""GPU parity of the monocular Frame constructor (orbx_frame_mono*, S/Frame.cc:260-358) against the oracle: extraction with the lapping
area {0, 1000} (reversed order), Frame::UndistortKeyPoints on the device for a distorted camera, the feature grid, the searches on the
frame it leaves on the device; synchronous, two-halves (host / device image, ingest thread) and libagentloop forms."""
import numpy as np
import pytest

from multi_orbslam3_amd import _capi as capi
from multi_orbslam3_amd import api, synth, views
from oracle import binding as ob
import helpers

pytestmark = pytest.mark.gpu

EUROC_DIST = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)       # R/ros/conf/EuRoC_mono_client.yaml
DISTS = {"none": None, "euroc": EUROC_DIST, "k1_zero": (0.0, 0.3, 0.01, 0.01, 0.1), "k3": (-0.25, 0.05, 0.001, -0.0007, 0.01)}


def _oracle_mono_frame(scene, k, dist, n_features=1000):
    """What Frame::Frame(mono) leaves behind, by the oracle: mvKeys (reversed order), mvKeysUn, descriptors, image bounds, grid."""
    L, R, Tcw = scene.stereo_pair(k)
    oe = ob.Extractor(n_features=n_features, max_width=scene.W, max_height=scene.H)
    rc, kps, desc, nmono = oe.extract(L, (0, 1000))
    assert nmono == 0
    p = scene.frame_view_params()
    cam4 = p["cam"][:4]
    kun = ob.undistort_keypoints(kps, cam4, dist)
    bounds = ob.image_bounds(scene.W, scene.H, cam4, dist)
    fv, keep = views.frame_view(kun, desc, None, None, bounds, p["cam"], 8, 1.2)
    return dict(L=np.ascontiguousarray(L), Tcw=Tcw, kps=kps, kps_un=kun, desc=desc, bounds=bounds, fv=fv, keep=keep)


def _empty_view(scene, bounds):
    p = scene.frame_view_params()
    return views.frame_view(np.zeros(1, capi.KEYPOINT_DTYPE), np.zeros((1, 32), np.uint8), None, None, bounds, p["cam"], 8, 1.2)


def test_undistort_points_on_the_device_equal_the_oracles(scene):
    rng = np.random.RandomState(1)
    cam4 = scene.frame_view_params()["cam"][:4]
    xy = np.concatenate([rng.uniform([0, 0], [640, 480], (5000, 2)), [[0, 0], [640, 0], [0, 480], [640, 480]]]).astype(np.float32)
    for name, dist in DISTS.items():
        g = api.undistort_points(xy, cam4, dist)
        o = ob.undistort_points(xy, cam4, dist)
        assert g.tobytes() == o.tobytes(), name
        assert api.image_bounds(640, 480, cam4, dist) == ob.image_bounds(640, 480, cam4, dist), name
    assert api.undistort_points(xy, cam4, None).tobytes() == xy.tobytes()
    assert len(api.undistort_points(np.zeros((0, 2), np.float32), cam4, EUROC_DIST)) == 0


@pytest.mark.parametrize("dist_name", sorted(DISTS))
@pytest.mark.parametrize("k", [2, 9])
def test_fused_mono_frame_constructor(scene, dist_name, k):
    """orbx_frame_mono: mvKeys, mvKeysUn, mDescriptors bit-equal to the oracle's, the device grid equal to the oracle's grid of
    mvKeysUn, and the frame it leaves on the device gives the oracle's matches in SearchByProjection(Current, Last) (mono, th 15)
    and SearchLocalPoints."""
    dist = DISTS[dist_name]
    o = _oracle_mono_frame(scene, k, dist)
    ex = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=1)
    F = api.Frame()
    fv0, keep0 = _empty_view(scene, o["bounds"])
    n, kps, kun, desc = ex.frame_mono(F, fv0, o["L"], dist)
    assert n == len(o["kps"]) > 900
    assert kps.tobytes() == o["kps"].tobytes() and desc.tobytes() == o["desc"].tobytes()
    assert kun.tobytes() == o["kps_un"].tobytes()
    if dist is not None and dist[0] != 0.0:
        assert np.abs(kun["x"] - kps["x"]).max() > 1.0                     # the lens model moved the keypoints near the corners
    else:
        assert kun.tobytes() == kps.tobytes()
    gs, gi = F.grid()
    os_, oi = ob.build_grid(o["fv"])
    assert np.array_equal(gs, os_) and np.array_equal(gi, oi)
    # the searches on the device frame (mono: no uRight gate, th = 15 as Tracking uses for monocular frames, S/Tracking.cc:2617)
    rng = np.random.RandomState(5)
    prev = helpers.oracle_stereo_frame(scene, k - 1)
    lv, keep_l = helpers.make_lastframe(scene, prev, rng)
    guess = synth.perturb_pose(o["Tcw"], rng).astype(np.float32)
    amp = np.full(n, -1, np.int32); aob = np.zeros(n, np.int32)
    g1 = api.ORBmatcher(0.9, True).SearchByProjectionFrame(F, guess, lv, 15.0, True, amp, aob)
    o1 = ob.search_by_projection_frame(o["fv"], guess, lv, 15.0, True, True, amp, aob)
    assert g1[2] == o1[2] > 100 and np.array_equal(g1[0], o1[0]) and np.array_equal(g1[1], o1[1])
    mp = helpers.local_map_from(scene, [prev, helpers.oracle_stereo_frame(scene, k + 3)], rng)
    wv, keep_w = helpers.world_view_of(mp)
    LM = api.LocalMap().upload(wv)
    g2 = api.ORBmatcher(0.8, True).SearchLocalPoints(F, LM, guess, 1.0, False, 0.0, g1[0], g1[1])
    o2 = ob.search_local_points(o["fv"], wv, guess, 1.0, False, 0.0, 0.8, o1[0], o1[1])
    assert g2[2] == o2[2] and np.array_equal(g2[0], o2[0]) and np.array_equal(g2[1], o2[1])


@pytest.mark.parametrize("mode", ["host", "host_async", "device"])
def test_mono_constructor_two_halves_over_a_ring_of_handles(scene, mode):
    """orbx_frame_mono_submit / _dev_submit + _wait: Frame(t+1) is handed over on the other handle before Frame(t) is collected; every
    frame's mvKeys / mvKeysUn / mDescriptors reach the host arrays of orbx_set_frame_outputs(+_un) and equal the oracle's."""
    import torch
    ids = [1, 2, 3, 4, 5]
    orc = [_oracle_mono_frame(scene, i, EUROC_DIST) for i in ids]
    ex = [api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=1) for _ in range(2)]
    Fr = [api.Frame(), api.Frame()]
    outs = [e.set_frame_outputs(2048) for e in ex]
    fv0, keep0 = _empty_view(scene, orc[0]["bounds"])
    dev = [torch.from_numpy(o["L"]).cuda() for o in orc]
    torch.cuda.synchronize()

    def submit(c, t):
        if mode == "device":
            ex[c].frame_mono_submit(Fr[c], fv0, None, EUROC_DIST, device_ptr=dev[t].data_ptr(), size=(640, 480, 640))
        else:
            ex[c].frame_mono_submit(Fr[c], fv0, orc[t]["L"], EUROC_DIST, async_ingest=(mode == "host_async"))

    submit(0, 0)
    for t in range(len(ids)):
        cur = t & 1
        if t + 1 < len(ids):
            submit(cur ^ 1, t + 1)
        n = ex[cur].frame_mono_wait()
        o, out = orc[t], outs[cur]
        assert n == len(o["kps"])
        assert out["kps"][:n].tobytes() == o["kps"].tobytes() and out["desc"][:n].tobytes() == o["desc"].tobytes(), t
        assert out["kps_un"][:n].tobytes() == o["kps_un"].tobytes(), t
        gs, gi = Fr[cur].grid()
        os_, oi = ob.build_grid(o["fv"])
        assert np.array_equal(gs, os_) and np.array_equal(gi, oi), t


def test_mono_constructor_argument_checks_and_host_quadtree_path(scene, monkeypatch):
    """What the header says is refused is refused: a distorted camera without a frame object, undistorted bounds that are not the image
    rectangle, an empty image (ORBG_EMPTY as the reference's early return); and the host quad-tree path (dense noise overflows the
    device lists) still undistorts and grids."""
    o = _oracle_mono_frame(scene, 4, EUROC_DIST)
    ex = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=1)
    F = api.Frame()
    fv_rect, k1 = _empty_view(scene, (0.0, 640.0, 0.0, 480.0))
    fv_dist, k2 = _empty_view(scene, o["bounds"])
    with pytest.raises(capi.OrbGpuError) as e:
        ex.frame_mono(None, fv_dist, o["L"], EUROC_DIST)
    assert e.value.code == capi.ORBG_BAD_ARG
    with pytest.raises(capi.OrbGpuError) as e:
        ex.frame_mono(F, fv_dist, o["L"], None)                    # no distortion, but bounds that are not the rectangle
    assert e.value.code == capi.ORBG_BAD_ARG
    with pytest.raises(capi.OrbGpuError) as e:
        ex.frame_mono(F, fv_rect, np.zeros((0, 0), np.uint8), None)
    assert e.value.code == capi.ORBG_EMPTY
    n, kps, kun, desc = ex.frame_mono(F, fv_dist, o["L"], EUROC_DIST)   # the handle is still usable
    assert n == len(o["kps"]) and kun.tobytes() == o["kps_un"].tobytes()
    # dense noise: more candidates than the LDS-resident quad-trees hold -> host quad-trees, same tail
    rng = np.random.RandomState(2)
    noise = rng.randint(0, 256, (480, 640)).astype(np.uint8)
    oe = ob.Extractor(n_features=1000)
    rc, okps, odesc, nm = oe.extract(noise, (0, 1000))
    cam4 = scene.frame_view_params()["cam"][:4]
    okun = ob.undistort_keypoints(okps, cam4, EUROC_DIST)
    n, kps, kun, desc = ex.frame_mono(F, fv_dist, noise, EUROC_DIST)
    assert n == len(okps) and kps.tobytes() == okps.tobytes() and desc.tobytes() == odesc.tobytes() and kun.tobytes() == okun.tobytes()
    fvn, kn = views.frame_view(okun, odesc, None, None, o["bounds"], scene.frame_view_params()["cam"], 8, 1.2)
    gs, gi = F.grid()
    os_, oi = ob.build_grid(fvn)
    assert np.array_equal(gs, os_) and np.array_equal(gi, oi)
