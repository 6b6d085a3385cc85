"""GPU parity tests: the HIP path (through the C-ABI of liborbgpu.so) against the CPU oracle on the same seeded
inputs.  Bit-exact for pyramid bytes, FAST candidates, keypoints, angles, descriptors, Hamming distances, stereo
matches and matcher outputs; <= 1e-4 for LBA poses / points after float32 write-back (tolerance from
BASELINE.json north_star)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from multi_orbslam3_amd import _capi as capi
from multi_orbslam3_amd import api, synth, views
from oracle import binding as ob
import helpers

pytestmark = pytest.mark.gpu


def _assert_extract_equal(g, o, what):
    gk, gd = g
    ok, od = o
    assert len(gk) == len(ok), "%s: count %d vs %d" % (what, len(gk), len(ok))
    for f in ("x", "y", "size", "angle", "response", "octave"):
        assert np.array_equal(gk[f], ok[f]), "%s: field %s differs" % (what, f)
    assert np.array_equal(gd, od), "%s: descriptors differ" % what


@pytest.mark.parametrize("k", [0, 7])
def test_extractor_mono_bit_exact(scene, k):
    L, R, _ = scene.stereo_pair(k)
    ex = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=1)
    oe = ob.Extractor(n_features=1000)
    nm, kps, desc = ex(L, (0, 0))
    rc, okps, odesc, onm = oe.extract(L)
    for l in range(8):
        assert np.array_equal(ex.level(0, l), oe.level(l)), "pyramid level %d" % l
        assert np.array_equal(ex.candidates(0, l), oe.candidates(l)), "FAST candidates level %d" % l
    _assert_extract_equal((kps, desc), (okps, odesc), "mono")
    assert nm == onm == len(kps)
    # mono Frame ctor path: lapping {0,1000} reverses the order (S/Frame.cc:289)
    nm2, k2, d2 = ex(L, (0, 1000))
    rc, ok2, od2, onm2 = oe.extract(L, lap=(0, 1000))
    assert nm2 == onm2 == 0
    _assert_extract_equal((k2, d2), (ok2, od2), "mono-lapping")


@pytest.mark.parametrize("shape,scale,levels", [((480, 640), 1.2, 8), ((241, 323), 1.2, 8), ((480, 640), 1.1, 12), ((300, 400), 2.0, 4)])
def test_pyramid_one_launch_and_per_level_paths(monkeypatch, shape, scale, levels):
    """Every pyramid level, border included: the one-launch tile tower, the per-level launches it falls back to when the
    halo does not fit (scale 2.0), odd sizes and other scale factors -- all against the oracle's cv::resize cascade."""
    rng = np.random.RandomState(shape[0] + levels)
    img = rng.randint(0, 256, shape).astype(np.uint8)
    img[::7, ::5] = 255
    H, W = shape
    oe = ob.Extractor(n_features=300, scale_factor=scale, n_levels=levels, max_width=W, max_height=H)
    oe.extract(img)
    for no_tower in ("", "1"):
        if no_tower:
            monkeypatch.setenv("ORBG_NO_TOWER", "1")
        else:
            monkeypatch.delenv("ORBG_NO_TOWER", raising=False)
        ex = api.ORBextractor(300, scale, levels, 20, 7, W, H, n_cams=1)
        ex(img, (0, 0))
        for l in range(levels):
            assert np.array_equal(ex.level(0, l, border=True), oe.level(l, border=True)), "level %d (no_tower=%r)" % (l, no_tower)


def test_extractor_dense_noise_overflows_to_host_trees(scene):
    """A noise image has more FAST candidates per level than the LDS-resident quad-tree (and the candidate buffer) holds:
    the device pass must flag the overflow without touching anything out of range -- recycled, non-zero device memory
    included -- and the host quad-trees redo the frame with the same result as the oracle."""
    L, R, _ = scene.stereo_pair(2)
    warm = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=2)
    warm.extract_stereo(L, R)
    del warm                                                            # its buffers are recycled by the next handle
    rng = np.random.RandomState(488)
    img = rng.randint(0, 256, (480, 640)).astype(np.uint8)
    oe = ob.Extractor(n_features=300, max_width=640, max_height=480)
    rc, okps, odesc, onm = oe.extract(img)
    assert sum(len(oe.candidates(l)) for l in range(8)) > 60000
    for n_cams in (1, 2):
        ex = api.ORBextractor(300, 1.2, 8, 20, 7, 640, 480, n_cams=n_cams)
        for _ in range(2):
            nm, kps, desc = ex(img, (0, 0))
            _assert_extract_equal((kps, desc), (okps, odesc), "dense noise")
    (kl, dl), (kr, dr) = ex.extract_stereo(img, img)
    _assert_extract_equal((kl, dl), (okps, odesc), "dense noise stereo L")
    _assert_extract_equal((kr, dr), (okps, odesc), "dense noise stereo R")


@pytest.mark.parametrize("variant", [dict(gauss_taps=(18, 34, 48, 56)), dict(octree_oldest_first=True), dict(gauss_taps=(16, 32, 48, 64)),
                                     dict(gauss_taps=(18, 34, 48, 56), octree_oldest_first=True)])
def test_extractor_deployment_variants(scene, variant):
    """The two places where the reference's output depends on its build (OpenCV's Gaussian taps, the heap-address tie-break of
    DistributeOctTree) are switchable in orbx_config: every variant is bit-exact against the oracle's same variant, on the device
    quad-tree path (stereo rig) and the host quad-tree path (mono with a partial lapping area), and differs from the default."""
    L, R, _ = scene.stereo_pair(5)
    ex = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=2, **variant)
    (kl, dl), (kr, dr) = ex.extract_stereo(L, R)
    ol, orr = ob.Extractor(n_features=1000, **variant), ob.Extractor(n_features=1000, **variant)
    rc, okl, odl, _ = ol.extract(L)
    rc, okr, odr, _ = orr.extract(R)
    _assert_extract_equal((kl, dl), (okl, odl), "variant %r left" % (variant,))
    _assert_extract_equal((kr, dr), (okr, odr), "variant %r right" % (variant,))
    exm = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=1, **variant)
    nm, km, dm = exm(L, (200, 400))                                   # partial lapping area -> host quad-trees
    rc, okm, odm, onm = ol.extract(L, lap=(200, 400))
    assert nm == onm
    _assert_extract_equal((km, dm), (okm, odm), "variant %r mono" % (variant,))
    rc, dk, dd, _ = ob.Extractor(n_features=1000).extract(L)          # the default build
    assert len(dk) != len(okl) or not np.array_equal(dd, odl)


@pytest.mark.parametrize("W,H,nf,kind", [(752, 480, 1200, "scene"), (1241, 376, 2000, "scene"), (323, 241, 500, "scene"),
                                          (131, 97, 300, "noise"), (70, 70, 100, "noise"), (641, 479, 1000, "blocks"),
                                          (1000, 333, 777, "blocks"), (320, 240, 3500, "noise")])
def test_extractor_image_sizes_and_feature_counts_bit_exact(W, H, nf, kind):
    """Image shapes whose pyramid levels have odd widths / heights and partial 30-pixel cells at the borders (EuRoC 752x480,
    KITTI 1241x376, shapes that are no multiple of anything), images barely larger than the 2 x 19-pixel border at the top
    levels, more requested features than the image has corners, three kinds of content -- keypoints and descriptors of every
    one against the oracle, mono with and without lapping."""
    rng = np.random.RandomState(W * 1000 + H)
    if kind == "scene":
        img = synth.Scene(W, H, tex_size=(max(2 * W, 800), max(2 * H, 600)), px_per_m=100.0).stereo_pair(2)[0]
    elif kind == "noise":
        img = rng.randint(0, 256, (H, W)).astype(np.uint8)
    else:
        # random rectangles on a ramp: strong isolated corners, large flat areas (cells that fall back to minThFAST or stay empty)
        img = np.tile((np.arange(W) * 40 // W + 60).astype(np.uint8), (H, 1))
        for _ in range(60):
            x0, y0 = rng.randint(0, W - 8), rng.randint(0, H - 8)
            img[y0:y0 + rng.randint(4, 60), x0:x0 + rng.randint(4, 80)] = rng.randint(0, 256)
    img = np.ascontiguousarray(img)
    ex = api.ORBextractor(nf, 1.2, 8, 20, 7, W, H, n_cams=1)
    oe = ob.Extractor(n_features=nf, max_width=W, max_height=H)
    for lap in ((0, 0), (0, 1000), (W // 3, 2 * W // 3)):
        nm, kps, desc = ex(img, lap)
        rc, okps, odesc, onm = oe.extract(img, lap=lap)
        _assert_extract_equal((kps, desc), (okps, odesc), "%dx%d %s lap %s" % (W, H, kind, lap))
        assert nm == onm
    for l in range(8):
        assert np.array_equal(ex.level(0, l), oe.level(l)), "pyramid level %d" % l


def test_extractor_empty_image():
    ex = api.ORBextractor(100, 1.2, 8, 20, 7, 320, 240)
    nm, kps, desc = ex(None)
    assert nm == -1 and len(kps) == 0


def test_extractor_refuses_what_the_header_says_it_refuses(small_scene):
    with pytest.raises(capi.OrbGpuError):                   # include/orbgpu.h: n_features 1 .. 3500
        api.ORBextractor(5000, 1.2, 8, 20, 7, 320, 240)
    with pytest.raises(capi.OrbGpuError):
        api.ORBextractor(1000, 1.0, 8, 20, 7, 320, 240)     # scale factor must exceed 1
    ex = api.ORBextractor(100, 1.2, 8, 20, 7, 320, 240)
    with pytest.raises(capi.OrbGpuError):                   # the top level of a 64 x 64 image is 18 pixels: inside the border
        ex(np.zeros((64, 64), np.uint8), (0, 0))
    assert len(ex(np.zeros((240, 320), np.uint8), (0, 0))[1]) == 0      # ... and the handle is still usable:
    L = small_scene.stereo_pair(1)[0]                                   # the refused size left nothing of its geometry behind
    nm, kps, desc = ex(L, (0, 0))
    oe = ob.Extractor(n_features=100)
    rc, okps, odesc, onm = oe.extract(L)
    _assert_extract_equal((kps, desc), (okps, odesc), "after a refused image size")
    for l in range(8):
        assert np.array_equal(ex.level(0, l), oe.level(l)), "pyramid level %d after a refused image size" % l


def test_extractor_stereo_batched_and_size_change(scene, small_scene):
    ex = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=2)
    for sc, nf in ((scene, 1000), (small_scene, 1000), (scene, 1000)):
        L, R, _ = sc.stereo_pair(3)
        (kl, dl), (kr, dr) = ex.extract_stereo(L, R)
        ol = ob.Extractor(n_features=nf, max_width=sc.W, max_height=sc.H)
        orr = ob.Extractor(n_features=nf, max_width=sc.W, max_height=sc.H)
        rc, okl, odl, _ = ol.extract(L)
        rc, okr, odr, _ = orr.extract(R)
        _assert_extract_equal((kl, dl), (okl, odl), "left %dx%d" % (sc.W, sc.H))
        _assert_extract_equal((kr, dr), (okr, odr), "right %dx%d" % (sc.W, sc.H))
        for l in (0, 4, 7):
            assert np.array_equal(ex.level(1, l), orr.level(l))


def test_extractor_low_texture_uses_min_threshold():
    """Cells without a single iniTh corner fall back to minTh (S/ORBextractor.cc:825-829)."""
    rng = np.random.RandomState(5)
    img = (120 + rng.randint(-6, 7, (240, 320))).astype(np.uint8)       # weak texture: only minTh corners
    img[60:180, 80:240] = (100 + rng.randint(0, 120, (120, 160))).astype(np.uint8)
    ex = api.ORBextractor(500, 1.2, 8, 20, 7, 320, 240)
    oe = ob.Extractor(n_features=500, max_width=320, max_height=240)
    nm, kps, desc = ex(img)
    rc, okps, odesc, _ = oe.extract(img)
    for l in range(8):
        assert np.array_equal(ex.candidates(0, l), oe.candidates(l)), "level %d" % l
    _assert_extract_equal((kps, desc), (okps, odesc), "low-texture")
    assert (okps["response"] < 20).any() and (okps["response"] >= 20).any()


def test_stereo_match_bit_exact(scene):
    fr = helpers.oracle_stereo_frame(scene, 2)
    ex = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=2)
    (kl, dl), (kr, dr) = ex.extract_stereo(fr["L"], fr["R"])
    assert np.array_equal(kl, fr["kps"]) and np.array_equal(dr, fr["desc_r"])
    ur, dp = ex.ComputeStereoMatches(float(scene.cam["bf"]), float(scene.cam["b"]), n_left=len(kl))
    assert (fr["uright"] > 0).sum() > 200
    assert np.array_equal(ur.view(np.uint32), fr["uright"].view(np.uint32))
    assert np.array_equal(dp.view(np.uint32), fr["depth"].view(np.uint32))


def test_fused_stereo_frame_constructor(scene):
    """orbx_frame_stereo_dev = Frame::Frame(stereo): extraction + ComputeStereoMatches + grid in one submission."""
    fr = helpers.oracle_stereo_frame(scene, 9)
    last = helpers.oracle_stereo_frame(scene, 8)
    ex = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=2)
    fv, keep = helpers.frame_view_of(scene, fr)
    F = api.Frame()
    cam = scene.cam
    for _ in range(2):          # second pass re-uses every buffer
        n, nr, kl, dl, ur, dp = ex.frame_stereo(F, fv, fr["L"], fr["R"], float(cam["bf"]), float(cam["b"]), download=True)
        assert n == len(fr["kps"]) and nr == len(fr["kps_r"])
        assert np.array_equal(kl, fr["kps"]) and np.array_equal(dl, fr["desc"])
        assert np.array_equal(ur.view(np.uint32), fr["uright"].view(np.uint32))
        assert np.array_equal(dp.view(np.uint32), fr["depth"].view(np.uint32))
        gs, gi = F.grid()
        os_, oi = ob.build_grid(fv)
        assert np.array_equal(gs, os_) and np.array_equal(gi, oi)
        rng = np.random.RandomState(2)
        lv, keep2 = helpers.make_lastframe(scene, last, rng)
        T = synth.perturb_pose(fr["Tcw"], rng).astype(np.float32)
        a0, b0 = np.full(n, -1, np.int32), np.zeros(n, np.int32)
        g = api.ORBmatcher(0.9, True).SearchByProjectionFrame(F, T, lv, 7.0, False, a0, b0)
        o = ob.search_by_projection_frame(fv, T, lv, 7.0, False, True, a0, b0)
        assert g[2] == o[2] and np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])


@pytest.mark.parametrize("W,H,nf", [(752, 480, 1200), (1241, 376, 2000), (323, 241, 600)])
def test_fused_stereo_frame_constructor_image_shapes(W, H, nf):
    """Frame::Frame(stereo) in one submission on EuRoC / KITTI / odd image shapes: both feature sets, the stereo matches
    (row bands and sliding windows clipped by a width that is no multiple of anything) and the grid against the oracle, twice
    on one handle."""
    sc = synth.Scene(W, H, tex_size=(max(2 * W, 800), max(2 * H, 600)), px_per_m=100.0)
    ex = api.ORBextractor(nf, 1.2, 8, 20, 7, W, H, n_cams=2)
    F = api.Frame()
    for k in (1, 6):
        fr = helpers.oracle_stereo_frame(sc, k, n_features=nf)
        fv, keep = helpers.frame_view_of(sc, fr)
        n, nr, kl, dl, ur, dp = ex.frame_stereo(F, fv, fr["L"], fr["R"], float(sc.cam["bf"]), float(sc.cam["b"]), download=True)
        assert n == len(fr["kps"]) and nr == len(fr["kps_r"]), (W, H, k)
        assert np.array_equal(kl, fr["kps"]) and np.array_equal(dl, fr["desc"]), (W, H, k)
        assert np.array_equal(ur.view(np.uint32), fr["uright"].view(np.uint32)), (W, H, k)
        assert np.array_equal(dp.view(np.uint32), fr["depth"].view(np.uint32)), (W, H, k)
        assert (fr["uright"] > 0).sum() > 50
        gs, gi = F.grid()
        os_, oi = ob.build_grid(fv)
        assert np.array_equal(gs, os_) and np.array_equal(gi, oi), (W, H, k)


def test_frame_constructor_submit_wait_pipelines_across_frames(scene):
    """orbx_frame_stereo_dev_submit / _wait: frame t+1 is constructed on one handle while frame t (other handle, other
    frame object) is being tracked; every frame's features, grid and matches equal the synchronous path / the oracle."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    cam = scene.cam
    bf, bb = float(cam["bf"]), float(cam["b"])
    ids = [3, 4, 5, 6]
    orc = [helpers.oracle_stereo_frame(scene, i) for i in ids]
    ex = [api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=2) for _ in range(2)]
    Fr = [api.Frame(), api.Frame()]
    dimg = []
    for fr in orc:
        pair = []
        for im in (fr["L"], fr["R"]):
            im = np.ascontiguousarray(im, np.uint8)
            d = ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(im.size)) == 0
            assert hip.hipMemcpy(d, ctypes.c_void_p(im.ctypes.data), ctypes.c_size_t(im.size), 1) == 0     # HostToDevice
            pair.append(d)
        dimg.append(pair)
    fvs = [helpers.frame_view_of(scene, fr) for fr in orc]
    m = api.ORBmatcher(0.9, True)
    rng = np.random.RandomState(4)
    with pytest.raises(Exception):
        ex[0].frame_stereo_dev_wait()                                   # nothing submitted
    ex[0].frame_stereo_dev_submit(Fr[0], fvs[0][0], dimg[0][0].value, dimg[0][1].value, 640, 480, 640, bf, bb)
    with pytest.raises(Exception):
        ex[0].extract_stereo(orc[0]["L"], orc[0]["R"])                  # the handle is busy until the wait
    with pytest.raises(Exception):
        ex[0].frame_stereo_dev_submit(Fr[0], fvs[0][0], dimg[0][0].value, dimg[0][1].value, 640, 480, 640, bf, bb)
    for t in range(len(ids)):
        cur = t & 1
        n, nr = ex[cur].frame_stereo_dev_wait()
        if t + 1 < len(ids):                                            # next frame's constructor runs during this frame's tracking
            ex[cur ^ 1].frame_stereo_dev_submit(Fr[cur ^ 1], fvs[t + 1][0], dimg[t + 1][0].value, dimg[t + 1][1].value, 640, 480, 640, bf, bb)
        fr, (fv, keep) = orc[t], fvs[t]
        assert n == len(fr["kps"]) and nr == len(fr["kps_r"])
        kd, dd = Fr[cur].download()[:2]
        assert np.array_equal(kd, fr["kps"]) and np.array_equal(dd, fr["desc"])
        gs, gi = Fr[cur].grid()
        os_, oi = ob.build_grid(fv)
        assert np.array_equal(gs, os_) and np.array_equal(gi, oi)
        lv, keep2 = helpers.make_lastframe(scene, helpers.oracle_stereo_frame(scene, ids[t] - 1), rng)
        T = synth.perturb_pose(fr["Tcw"], rng).astype(np.float32)
        a0, b0 = np.full(n, -1, np.int32), np.zeros(n, np.int32)
        g = m.SearchByProjectionFrame(Fr[cur], T, lv, 7.0, False, a0, b0)
        o = ob.search_by_projection_frame(fv, T, lv, 7.0, False, True, a0, b0)
        assert g[2] == o[2] and g[2] > 50 and np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])
    for pair in dimg:
        for d in pair:
            hip.hipFree(d)


@pytest.mark.parametrize("async_ingest", [False, True])
def test_frame_constructor_submit_with_host_images(scene, async_ingest):
    """orbx_frame_stereo_submit / _wait: the pipelined constructor fed with HOST images (what Tracking::GrabImageStereo holds,
    S/Tracking.cc:1014-1083) through the pinned staging slot + copy kernel -- on the calling thread and on the library's ingest
    thread.  Features, stereo matches and grid equal the oracle's for every frame; images with a row stride; the staging slot
    is private to a handle, so with flags == 0 the caller's image may be overwritten right after the submit."""
    cam = scene.cam
    bf, bb = float(cam["bf"]), float(cam["b"])
    ids = [3, 4, 5, 6, 7]
    orc = [helpers.oracle_stereo_frame(scene, i) for i in ids]
    ex = [api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=2) for _ in range(2)]
    Fr = [api.Frame(), api.Frame()]
    fvs = [helpers.frame_view_of(scene, fr) for fr in orc]

    def host_pair(t):
        fr = orc[t]
        if t % 2 == 1:                                   # odd frames: rows inside a wider buffer (cv::Mat ROI: stride > width)
            big = [np.full((480, 704), 77, np.uint8), np.full((480, 704), 78, np.uint8)]
            big[0][:, 32:672] = fr["L"]; big[1][:, 32:672] = fr["R"]
            return big[0][:, 32:672], big[1][:, 32:672]
        return np.array(fr["L"], np.uint8, order="C"), np.array(fr["R"], np.uint8, order="C")

    with pytest.raises(Exception):
        ex[0].frame_stereo_dev_wait()                                   # nothing submitted
    pair = host_pair(0)
    ex[0].frame_stereo_submit(Fr[0], fvs[0][0], pair[0], pair[1], bf, bb, async_ingest=async_ingest)
    with pytest.raises(Exception):
        ex[0].extract_stereo(orc[0]["L"], orc[0]["R"])                  # the handle is busy until the wait
    with pytest.raises(Exception):
        ex[0].frame_stereo_submit(Fr[0], fvs[0][0], pair[0], pair[1], bf, bb, async_ingest=async_ingest)
    for t in range(len(ids)):
        cur = t & 1
        n, nr = ex[cur].frame_stereo_dev_wait()
        if t + 1 < len(ids):
            pair = host_pair(t + 1)
            ex[cur ^ 1].frame_stereo_submit(Fr[cur ^ 1], fvs[t + 1][0], pair[0], pair[1], bf, bb, async_ingest=async_ingest)
            if not async_ingest:
                pair[0][...] = 0; pair[1][...] = 255                    # the rows were packed into the staging slot by the call
        fr, (fv, keep) = orc[t], fvs[t]
        assert n == len(fr["kps"]) and nr == len(fr["kps_r"]), (t, n, nr)
        kd, dd = Fr[cur].download()[:2]
        assert np.array_equal(kd, fr["kps"]) and np.array_equal(dd, fr["desc"])
        gs, gi = Fr[cur].grid()
        os_, oi = ob.build_grid(fv)
        assert np.array_equal(gs, os_) and np.array_equal(gi, oi)
    # the synchronous host-image constructor goes through the same staging path
    n, nr, kl, dl, ur, dp = ex[0].frame_stereo(Fr[0], fvs[2][0], orc[2]["L"], orc[2]["R"], bf, bb)
    assert np.array_equal(kl[:n], orc[2]["kps"]) and np.array_equal(dl[:n], orc[2]["desc"])
    assert np.array_equal(ur[:n], orc[2]["uright"]) and np.array_equal(dp[:n], orc[2]["depth"])


def test_parity_subset_with_poisoned_allocations():
    """ORBG_POISON=1 fills every new device / pinned buffer with 0xA5 (fresh HIP allocations are usually zero, so a kernel that consumes
    memory nobody wrote passes unnoticed): a constructor test, a search test and the local BA's window sizes in such a process."""
    env = dict(os.environ, ORBG_POISON="1")
    here = os.path.abspath(__file__)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", here + "::test_fused_stereo_frame_constructor",
                        here + "::test_frame_constructor_submit_wait_pipelines_across_frames", here + "::test_lba_window_sizes_cover_every_ldlt_kernel",
                        here + "::test_lba_every_window_size_up_to_50_free_poses", here + "::test_pose_optimization_parity"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=os.path.dirname(os.path.dirname(here)))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("async_ingest,device_images", [(False, False), (True, False), (False, True)])
def test_two_halves_constructor_delivers_features_to_host(scene, async_ingest, device_images):
    """orbx_set_frame_outputs: the pipelined constructor hands mvKeys / mDescriptors / mvuRight / mvDepth of the left image to host
    arrays by the time _wait returns (what an unchanged Tracking / KeyFrame reads every frame, S/Frame.cc:100-118) -- equal to the
    oracle's for every frame of a ping-pong over two handles, too small a capacity is refused by _wait, and switching the delivery
    off leaves the arrays alone."""
    import torch
    cam = scene.cam
    bf, bb = float(cam["bf"]), float(cam["b"])
    ids = [3, 4, 5, 6]
    orc = [helpers.oracle_stereo_frame(scene, i) for i in ids]
    ex = [api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=2) for _ in range(2)]
    Fr = [api.Frame(), api.Frame()]
    fvs = [helpers.frame_view_of(scene, fr) for fr in orc]
    outs = [e.set_frame_outputs(2048) for e in ex]
    dev = [(torch.from_numpy(np.ascontiguousarray(fr["L"])).cuda(), torch.from_numpy(np.ascontiguousarray(fr["R"])).cuda()) for fr in orc]
    torch.cuda.synchronize()

    def submit(c, t):
        if device_images:
            ex[c].frame_stereo_dev_submit(Fr[c], fvs[t][0], dev[t][0].data_ptr(), dev[t][1].data_ptr(), 640, 480, 640, bf, bb)
        else:
            ex[c].frame_stereo_submit(Fr[c], fvs[t][0], np.ascontiguousarray(orc[t]["L"]), np.ascontiguousarray(orc[t]["R"]), bf, bb, async_ingest=async_ingest)

    submit(0, 0)
    for t in range(len(ids)):
        cur = t & 1
        if t + 1 < len(ids):
            submit(cur ^ 1, t + 1)
        n, nr = ex[cur].frame_stereo_dev_wait()
        fr, o = orc[t], outs[cur]
        assert n == len(fr["kps"])
        assert np.array_equal(o["kps"][:n], fr["kps"]) and np.array_equal(o["desc"][:n], fr["desc"]), t
        assert np.array_equal(o["uright"][:n], fr["uright"]) and np.array_equal(o["depth"][:n], fr["depth"]), t
    # a capacity below the frame's feature count: refused at _wait, the handle is usable afterwards
    ex[0].set_frame_outputs(16)
    submit(0, 0)
    with pytest.raises(Exception):
        ex[0].frame_stereo_dev_wait()
    keep = ex[1]._outputs["kps"].copy()
    ex[1].set_frame_outputs(0)
    submit(1, 0)
    n, nr = ex[1].frame_stereo_dev_wait()
    assert n == len(orc[0]["kps"]) and np.array_equal(keep, outs[1]["kps"])            # delivery off: the old arrays are untouched
    ex[0].set_frame_outputs(0)
    submit(0, 1)
    assert ex[0].frame_stereo_dev_wait()[0] == len(orc[1]["kps"])


@pytest.mark.parametrize("switch", ["ORBG_OCT_NO_JUMP"])
def test_frame_constructor_chain_variants(switch):
    """ORBG_OCT_NO_JUMP=1 (read when a handle sets up its geometry): every quad-tree pass replayed one by one instead of the first (up
    to three) uniform passes taken from a three-level histogram of the keys -- the check of the jump start against the plain pass
    loop.  The constructor parity tests -- features, stereo matches, grid against the oracle; host images through the ingest thread
    on odd shapes -- in a process that has the switch."""
    env = dict(os.environ, **{switch: "1"})
    here = os.path.abspath(__file__)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", here + "::test_frame_constructor_submit_wait_pipelines_across_frames",
                        here + "::test_host_image_submit_on_other_image_shapes", here + "::test_fused_stereo_frame_constructor", here + "::test_frame_constructor_submit_with_host_images"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=os.path.dirname(os.path.dirname(here)))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])


def test_gpu_and_host_quadtree_paths_agree(scene, small_scene, monkeypatch):
    """The LDS-resident GPU quad-tree and the host implementation (ORBG_HOST_OCTREE=1; also the overflow / partial-
    lapping fallback) give identical keypoints; a partial lapping area forces the host path transparently."""
    for sc in (scene, small_scene):
        L, R, _ = sc.stereo_pair(5)
        monkeypatch.delenv("ORBG_HOST_OCTREE", raising=False)
        ex_gpu = api.ORBextractor(1000, 1.2, 8, 20, 7, sc.W, sc.H, n_cams=2)
        monkeypatch.setenv("ORBG_HOST_OCTREE", "1")
        ex_host = api.ORBextractor(1000, 1.2, 8, 20, 7, sc.W, sc.H, n_cams=2)
        monkeypatch.delenv("ORBG_HOST_OCTREE", raising=False)
        a = ex_gpu.extract_stereo(L, R)
        b = ex_host.extract_stereo(L, R)
        for cam in (0, 1):
            _assert_extract_equal(a[cam], b[cam], "gpu-vs-host quad-tree cam %d" % cam)
        for l in (0, 3, 7):
            assert np.array_equal(ex_gpu.candidates(1, l), ex_host.candidates(1, l))
    # partial lapping area (fisheye-stereo style): front part natural order, lapping part from the back
    ex = api.ORBextractor(500, 1.2, 8, 20, 7, 640, 480)
    oe = ob.Extractor(n_features=500)
    L, _, _ = scene.stereo_pair(6)
    nm, k, d = ex(L, (200, 400))
    rc, ok, od, onm = oe.extract(L, lap=(200, 400))
    assert nm == onm and 0 < nm < len(k)
    _assert_extract_equal((k, d), (ok, od), "partial lapping")


def test_hamming_kernels():
    rng = np.random.RandomState(1)
    q = rng.randint(0, 256, (301, 32)).astype(np.uint8)
    t = np.repeat(rng.randint(0, 256, (257, 32)).astype(np.uint8), 2, axis=0)[rng.permutation(514)]
    m = api.ORBmatcher()
    assert np.array_equal(m.DescriptorDistance(q, t), ob.hamming_matrix(q, t))
    assert np.array_equal(m.best2(q, t), ob.hamming_best2(q, t))
    z = np.zeros((1, 32), np.uint8); o = np.full((1, 32), 255, np.uint8)
    assert m.DescriptorDistance(z, o)[0, 0] == 256 and m.DescriptorDistance(z, z)[0, 0] == 0
    assert m.best2(q[:3], t[:1])[0, 3] == -1 and m.best2(q[:3], t[:1])[0, 2] == 256   # ragged: one candidate only


def test_grid_and_frustum(scene):
    fr = helpers.oracle_stereo_frame(scene, 1)
    fv, keep = helpers.frame_view_of(scene, fr)
    F = api.Frame().upload(fv, keep)
    gs, gi = F.grid()
    os_, oi = ob.build_grid(fv)
    assert np.array_equal(gs, os_) and np.array_equal(gi, oi)
    mp = helpers.local_map_from(scene, [fr, helpers.oracle_stereo_frame(scene, 6)], np.random.RandomState(3))
    wv, keep2 = helpers.world_view_of(mp)
    T = synth.perturb_pose(scene.pose(3), np.random.RandomState(4)).astype(np.float32)
    g = F.isInFrustum(T, wv)
    o = ob.is_in_frustum(fv, T, wv)
    assert o["track_in_view"].sum() > 100
    for k in o:
        a, b = g[k], o[k]
        if a.dtype == np.float32:
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), k
        else:
            assert np.array_equal(a, b), k


@pytest.mark.parametrize("th", [1.0, 3.0, 15.0])
def test_search_by_projection_map(scene, th):
    rng = np.random.RandomState(11)
    f0, f1, f2 = [helpers.oracle_stereo_frame(scene, k) for k in (0, 4, 8)]
    cur = helpers.oracle_stereo_frame(scene, 5)
    mp = helpers.local_map_from(scene, [f0, f1, f2], rng)
    fv, keep = helpers.frame_view_of(scene, cur)
    wv, keep2 = helpers.world_view_of(mp)
    T = synth.perturb_pose(cur["Tcw"], rng).astype(np.float32)
    n = len(cur["kps"])
    amp0 = np.full(n, -1, np.int32); aob0 = np.zeros(n, np.int32)
    pre = rng.rand(n) < 0.1                   # some features already matched by the motion model
    amp0[pre] = 7; aob0[pre] = rng.randint(0, 3, pre.sum())
    # (a) pre-projected view: exactly SearchByProjection(F, vpMapPoints, th, bFar, thFar)
    tr = ob.is_in_frustum(fv, T, wv)
    mv, keep3 = views.mappoints_view(tr["track_in_view"], mp["bad"], tr["proj_x"], tr["proj_y"], tr["proj_xr"], tr["track_depth"],
                                     tr["scale_level"], tr["view_cos"], mp["desc"], mp["n_obs"])
    F = api.Frame().upload(fv, keep)
    m = api.ORBmatcher(0.8)
    g = m.SearchByProjection(F, mv, th, True, 4.0, amp0, aob0)
    o = ob.search_by_projection_mps(fv, mv, th, True, 4.0, 0.8, amp0, aob0)
    assert o[2] > 50
    assert g[2] == o[2] and np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])
    # (b) fused isInFrustum + search with the map resident on the device
    skip = (rng.rand(len(mp["pos"])) < 0.05).astype(np.uint8)
    LM = api.LocalMap().upload(wv)
    g2 = m.SearchLocalPoints(F, LM, T, th, False, 0.0, amp0, aob0, skip)
    wv2, keep4 = helpers.world_view_of(mp, skip)
    o2 = ob.search_local_points(fv, wv2, T, th, False, 0.0, 0.8, amp0, aob0)
    assert g2[2] == o2[2] and np.array_equal(g2[0], o2[0]) and np.array_equal(g2[1], o2[1])


def test_projection_searches_on_empty_far_and_crowded_maps(scene):
    """The map side of TrackLocalMap at its edges: no map points at all; every point bad; a camera looking away from the map
    (nothing in the frustum); a crowded map (each point eight times, so that every feature has tied candidates and the
    best / second-best bookkeeping decides); a last frame without map points."""
    rng = np.random.RandomState(21)
    f0, f1 = [helpers.oracle_stereo_frame(scene, k) for k in (2, 6)]
    cur = helpers.oracle_stereo_frame(scene, 4)
    mp = helpers.local_map_from(scene, [f0, f1], rng)
    fv, keep = helpers.frame_view_of(scene, cur)
    n = len(cur["kps"])
    T = synth.perturb_pose(cur["Tcw"], rng).astype(np.float32)
    F = api.Frame().upload(fv, keep)
    m = api.ORBmatcher(0.8)

    def check(mpx, Tx, what, expect_some=None):
        amp0 = np.full(n, -1, np.int32); aob0 = np.zeros(n, np.int32)
        wv, keep2 = helpers.world_view_of(mpx)
        LM = api.LocalMap().upload(wv)
        g = m.SearchLocalPoints(F, LM, Tx, 3.0, False, 0.0, amp0, aob0, None)
        o = ob.search_local_points(fv, wv, Tx, 3.0, False, 0.0, 0.8, amp0, aob0)
        assert g[2] == o[2] and np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1]), what
        if expect_some is not None:
            assert (o[2] > 0) == expect_some, (what, o[2])

    check({k: v[:0] for k, v in mp.items()}, T, "empty map", False)
    allbad = dict(mp); allbad["bad"] = np.ones_like(mp["bad"])
    check(allbad, T, "all bad", False)
    Taway = T.copy(); Taway[:3, :3] = T[:3, :3] @ np.diag([-1.0, 1.0, -1.0]).astype(np.float32)     # half a turn about y
    check(mp, Taway, "looking away", False)
    crowded = {k: np.concatenate([v] * 8) for k, v in mp.items()}
    check(crowded, T, "crowded", True)
    # last frame without a single map point: SearchByProjection(CurrentFrame, LastFrame) finds nothing, touches nothing
    last = helpers.oracle_stereo_frame(scene, 3)
    Pw, valid = synth.unproject_to_world(last["kps"], last["depth"], last["Tcw"], scene.cam)
    nl = len(last["kps"])
    lv0, keep3 = views.lastframe_view(np.zeros(nl, np.uint8), np.zeros(nl, np.uint8), Pw, last["desc"], last["kps"]["octave"],
                                      last["kps"]["angle"], np.full(nl, 2, np.int32), last["Tcw"].astype(np.float32))
    lv1, keep4 = views.lastframe_view(valid.astype(np.uint8), np.ones(nl, np.uint8), Pw, last["desc"], last["kps"]["octave"],
                                      last["kps"]["angle"], np.full(nl, 2, np.int32), last["Tcw"].astype(np.float32))
    amp0 = np.full(n, -1, np.int32); aob0 = np.zeros(n, np.int32)
    for lvx in (lv0, lv1):                                 # no map points; every map point an outlier of the last frame
        g = api.ORBmatcher(0.9, True).SearchByProjectionFrame(F, T, lvx, 7.0, False, amp0, aob0)
        o = ob.search_by_projection_frame(fv, T, lvx, 7.0, False, True, amp0, aob0)
        assert o[2] == 0
        assert g[2] == o[2] and np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])


@pytest.mark.parametrize("th,mono", [(7.0, False), (15.0, True), (14.0, False)])
def test_search_by_projection_frame(scene, th, mono):
    rng = np.random.RandomState(13)
    last = helpers.oracle_stereo_frame(scene, 10)
    cur = helpers.oracle_stereo_frame(scene, 11)
    fv, keep = helpers.frame_view_of(scene, cur, with_stereo=not mono)
    lv, keep2 = helpers.make_lastframe(scene, last, rng)
    T = synth.perturb_pose(cur["Tcw"], rng).astype(np.float32)
    n = len(cur["kps"])
    amp0 = np.full(n, -1, np.int32); aob0 = np.zeros(n, np.int32)
    F = api.Frame().upload(fv, keep)
    m = api.ORBmatcher(0.9, True)
    g = m.SearchByProjectionFrame(F, T, lv, th, mono, amp0, aob0)
    o = ob.search_by_projection_frame(fv, T, lv, th, mono, True, amp0, aob0)
    assert o[2] > 100
    assert g[2] == o[2] and np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])


def test_search_by_bow(scene):
    kf = helpers.oracle_stereo_frame(scene, 20)
    cur = helpers.oracle_stereo_frame(scene, 21)
    fv, keep = helpers.frame_view_of(scene, cur)
    # stand-in vocabulary (ORBvoc.txt is absent from the reference tree): node id = first 5 descriptor bits + octave/4
    node = lambda d, k: (d[:, 0].astype(np.int64) >> 3) * 2 + (k["octave"] // 4)
    fvF, kF = views.featvec_view(*views.featvec_from_nodes(node(cur["desc"], cur["kps"])))
    fvK, kK = views.featvec_view(*views.featvec_from_nodes(node(kf["desc"], kf["kps"])))
    valid = (kf["depth"] > 0).astype(np.uint8)
    F = api.Frame().upload(fv, keep)
    m = api.ORBmatcher(0.7, True)
    g = m.SearchByBoW(F, fvF, kf["desc"], valid, kf["kps"]["angle"], fvK)
    o = ob.search_by_bow(fv, fvF, kf["desc"], valid, kf["kps"]["angle"], fvK, 0.7, True)
    assert o[1] > 20
    assert g[1] == o[1] and np.array_equal(g[0], o[0])


@pytest.mark.parametrize("case", ["disjoint_nodes", "no_valid_points", "one_node", "coarse_nodes", "no_orientation_check"])
def test_search_by_bow_edge_cases(scene, case):
    """SearchByBoW(KF, F) when the two feature vectors share no node; when the keyframe has no map point; when every feature
    falls into ONE node (every keyframe point is compared with every frame feature: the longest candidate lists); with 4 coarse
    nodes; without the rotation histogram."""
    kf = helpers.oracle_stereo_frame(scene, 14)
    cur = helpers.oracle_stereo_frame(scene, 15)
    fv, keep = helpers.frame_view_of(scene, cur)
    if case == "one_node":
        node = lambda d, k: np.zeros(len(d), np.int64)
    elif case == "coarse_nodes":
        node = lambda d, k: d[:, 0].astype(np.int64) >> 6
    else:
        node = lambda d, k: (d[:, 0].astype(np.int64) >> 3) * 2 + (k["octave"] // 4)
    nF, nK = node(cur["desc"], cur["kps"]), node(kf["desc"], kf["kps"])
    if case == "disjoint_nodes":
        nK = nK + 1000
    fvF, kF = views.featvec_view(*views.featvec_from_nodes(nF))
    fvK, kK = views.featvec_view(*views.featvec_from_nodes(nK))
    valid = (kf["depth"] > 0).astype(np.uint8)
    if case == "no_valid_points":
        valid[:] = 0
    check = case != "no_orientation_check"
    F = api.Frame().upload(fv, keep)
    g = api.ORBmatcher(0.7, check).SearchByBoW(F, fvF, kf["desc"], valid, kf["kps"]["angle"], fvK)
    o = ob.search_by_bow(fv, fvF, kf["desc"], valid, kf["kps"]["angle"], fvK, 0.7, check)
    assert g[1] == o[1] and np.array_equal(g[0], o[0]), case
    if case in ("disjoint_nodes", "no_valid_points"):
        assert o[1] == 0
    else:
        assert o[1] > 10


@pytest.mark.parametrize("th,ratio,with_kfs,scale", [(3, 1.5, False, 1.0), (8, 1.0, True, 1.7), (8, 1.5, False, 0.6)])
def test_search_by_projection_sim3(scene, th, ratio, with_kfs, scale):
    """Server-side SearchByProjection(KeyFrame*, Scw, ...), both overloads."""
    rng = np.random.RandomState(21)
    f0, f1 = [helpers.oracle_stereo_frame(scene, k) for k in (12, 16)]
    kf = helpers.oracle_stereo_frame(scene, 14)
    mp = helpers.local_map_from(scene, [f0, f1], rng)
    fv, keep = helpers.frame_view_of(scene, kf)
    T = synth.perturb_pose(kf["Tcw"], rng).astype(np.float32)
    S = T.copy()
    S[:3, :] *= np.float32(scale)                       # Scw = [s*R | s*t]
    n = len(kf["kps"])
    matched0 = np.full(n, -1, np.int32)
    pre = rng.rand(n) < 0.15
    matched0[pre] = 3
    found = (rng.rand(len(mp["pos"])) < 0.1).astype(np.uint8)
    skip = (rng.rand(len(mp["pos"])) < 0.05).astype(np.uint8)
    wv, keep2 = helpers.world_view_of(mp, skip)
    F = api.Frame().upload(fv, keep)
    LM = api.LocalMap().upload(wv)
    m = api.ORBmatcher(0.75, True)
    g = m.SearchByProjectionSim3(F, S, LM, matched0, th, ratio, found, with_kfs)
    o = ob.search_by_projection_sim3(fv, wv, S, matched0, th, ratio, found, with_kfs)
    assert o[1] > 50
    assert g[1] == o[1] and np.array_equal(g[0], o[0])
    assert np.array_equal(g[0][pre], matched0[pre])     # occupied features are never overwritten
    # no candidates at all
    g0 = m.SearchByProjectionSim3(F, S, LM, matched0, th, ratio, np.ones(len(mp["pos"]), np.uint8), with_kfs)
    assert g0[1] == 0 and np.array_equal(g0[0], matched0)


@pytest.mark.parametrize("th,mono", [(7.0, False), (15.0, True)])
def test_search_by_projection_frame_with_a_resident_last_frame_view(scene, th, mono):
    """orbm_lastview_upload + orbm_search_by_projection_frame_resident: the last frame's view on the device (uploaded a frame time
    before it is read) gives what the in-place form and the oracle give, across re-uploads of views of different sizes."""
    rng = np.random.RandomState(41)
    m = api.ORBmatcher(0.9, True)
    LV = api.LastFrameOnDevice(1024)                       # smaller than the views: the buffers grow
    for k in (3, 4, 5):
        last, cur = helpers.oracle_stereo_frame(scene, k), helpers.oracle_stereo_frame(scene, k + 1)
        lv, keep_lv = helpers.make_lastframe(scene, last, rng)
        fv, keep = helpers.frame_view_of(scene, cur, with_stereo=not mono)
        F = api.Frame().upload(fv, keep)
        n = len(cur["kps"])
        amp0 = np.full(n, -1, np.int32); aob0 = np.zeros(n, np.int32)
        amp0[::11] = 3; aob0[::11] = 1
        T = synth.perturb_pose(cur["Tcw"], rng).astype(np.float32)
        LV.upload(lv)
        g = m.SearchByProjectionFrameResident(F, T, LV, th, mono, amp0, aob0)
        h = m.SearchByProjectionFrame(F, T, lv, th, mono, amp0, aob0)
        o = ob.search_by_projection_frame(fv, T, lv, th, mono, True, amp0, aob0)
        assert o[2] > 100
        for a, b in ((g, o), (h, o)):
            assert a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), k


@pytest.mark.parametrize("th,orb_dist,check_ori", [(10.0, 100, True), (3.0, 64, True), (10.0, 100, False), (1.0, 50, True)])
def test_search_by_projection_relocalisation_overload(scene, th, orb_dist, check_ori):
    """SearchByProjection(Frame&, KeyFrame*, set<MapPoint*>&, th, ORBdist) (S/ORBmatcher.cc:2188-2310; Tracking::Relocalization calls
    it with 10 / 100 and 3 / 64): the keyframe's map points feature by feature (some features without a point, some points bad, some
    in sAlreadyFound), a current frame that already holds points (any of them blocks its feature), a perturbed pose."""
    rng = np.random.RandomState(23)
    kf = helpers.oracle_stereo_frame(scene, 18)
    cur = helpers.oracle_stereo_frame(scene, 20)
    mp = synth.map_from_frame(kf["kps"], kf["desc"], kf["depth"], kf["Tcw"], scene.cam)        # points the keyframe created (stereo features)
    nk = len(kf["kps"])
    # feature-by-feature view of pKF->GetMapPointMatches(): features without depth have no point (bad = 1)
    pos = np.zeros((nk, 3), np.float32); nrm = np.zeros((nk, 3), np.float32); dmin = np.zeros(nk, np.float32); dmax = np.ones(nk, np.float32)
    desc = np.zeros((nk, 32), np.uint8); bad = np.ones(nk, np.uint8)
    idx = mp["src_idx"]
    pos[idx] = mp["pos"]; nrm[idx] = mp["normal"]; dmin[idx] = mp["min_dist"]; dmax[idx] = mp["max_dist"]; desc[idx] = mp["desc"]; bad[idx] = 0
    bad[idx[rng.rand(len(idx)) < 0.05]] = 1                                                        # isBad()
    found = np.zeros(nk, np.uint8); found[idx[rng.rand(len(idx)) < 0.1]] = 1                      # sAlreadyFound
    wv, keep2 = views.worldpoints_view(pos, nrm, dmin, dmax, desc, np.full(nk, 2, np.int32), bad, None)
    fv, keep = helpers.frame_view_of(scene, cur)
    n = len(cur["kps"])
    amp0 = np.full(n, -1, np.int32)
    pre = rng.rand(n) < 0.1
    amp0[pre] = 2 ** 31 - 1
    T = synth.perturb_pose(cur["Tcw"], rng).astype(np.float32)
    F = api.Frame().upload(fv, keep)
    KP = api.LocalMap().upload(wv)
    m = api.ORBmatcher(0.75, check_ori)
    g = m.SearchByProjectionReloc(F, T, KP, kf["kps"]["angle"], amp0, th, orb_dist, found)
    o = ob.search_by_projection_reloc(fv, T, wv, kf["kps"]["angle"], amp0, th, orb_dist, check_ori, found)
    assert o[1] > (60 if th >= 3 else 10), o[1]
    assert g[1] == o[1] and np.array_equal(g[0], o[0])
    assert np.array_equal(g[0][pre], amp0[pre])                 # features that hold a point are never taken
    newly = (g[0] >= 0) & ~pre
    assert not found[g[0][newly]].any() and not bad[g[0][newly]].any()
    g0 = m.SearchByProjectionReloc(F, T, KP, kf["kps"]["angle"], amp0, th, orb_dist, np.ones(nk, np.uint8))
    assert g0[1] == 0 and np.array_equal(g0[0], amp0)


@pytest.mark.parametrize("check_ori", [True, False])
def test_search_by_bow_kf(scene, check_ori):
    """Server-side SearchByBoW(KeyFrame*, KeyFrame*)."""
    rng = np.random.RandomState(22)
    kf1 = helpers.oracle_stereo_frame(scene, 30)
    kf2 = helpers.oracle_stereo_frame(scene, 31)
    fv2v, keep = helpers.frame_view_of(scene, kf2)
    node = lambda d, k: (d[:, 0].astype(np.int64) >> 3) * 2 + (k["octave"] // 4)
    fv2, k2 = views.featvec_view(*views.featvec_from_nodes(node(kf2["desc"], kf2["kps"])))
    fv1, k1 = views.featvec_view(*views.featvec_from_nodes(node(kf1["desc"], kf1["kps"])))
    valid1 = ((kf1["depth"] > 0) & (rng.rand(len(kf1["kps"])) < 0.9)).astype(np.uint8)
    valid2 = ((kf2["depth"] > 0) & (rng.rand(len(kf2["kps"])) < 0.9)).astype(np.uint8)
    F2 = api.Frame().upload(fv2v, keep)
    m = api.ORBmatcher(0.8, check_ori)
    g = m.SearchByBoWKF(F2, fv2, valid2, kf1["desc"], valid1, kf1["kps"]["angle"], fv1)
    o = ob.search_by_bow_kf(fv2v, fv2, valid2, kf1["desc"], valid1, kf1["kps"]["angle"], fv1, 0.8, check_ori)
    assert o[1] > 20
    assert g[1] == o[1] and np.array_equal(g[0], o[0])
    sel = g[0] >= 0
    assert valid1[sel].all() and valid2[g[0][sel]].all()
    assert len(np.unique(g[0][sel])) == sel.sum()       # vbMatched2: a pKF2 feature is used at most once


@pytest.mark.parametrize("shape", [(4, 2, 60), (20, 10, 2000)])
def test_lba_parity(shape):
    nf, nx, npts = shape
    prob = synth.make_lba_problem(n_free=nf, n_fixed=nx, n_points=npts, mono_frac=0.1)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    g = api.Optimizer().LocalBundleAdjustment(p)
    o = ob.lba_solve(p)
    assert g.status == o.status == capi.LBA_APPLIED
    assert g.iters == o.iters
    assert np.abs(g.poses - o.poses).max() <= 1e-4          # tolerance stated by north_star
    assert np.abs(g.points - o.points).max() <= 1e-4
    assert np.array_equal(g.edge_outlier, o.edge_outlier)
    assert np.array_equal(g.edge_depth_pos, o.edge_depth_pos)
    tg, to = g.trace_rows(), o.trace_rows()
    assert tg.shape == to.shape
    assert np.allclose(tg[:, 1], to[:, 1], rtol=1e-9) and np.array_equal(tg[:, 2], to[:, 2])
    assert g.chi2[1] < g.chi2[0]
    # bit-reproducible run to run (fixed-order reductions)
    g2 = api.Optimizer().LocalBundleAdjustment(p)
    assert np.array_equal(g.poses, g2.poses) and np.array_equal(g.points, g2.points)


@pytest.mark.parametrize("nf", [1, 2, 3, 7, 13, 19, 20, 21, 22, 26, 31, 36, 40, 47, 50])
def test_lba_window_sizes_cover_every_ldlt_kernel(nf, monkeypatch):
    """Free-pose counts around every kernel boundary of the reduced-camera-system solve: the matrix-core column kernel
    (<= 20 poses, 1..8 tile columns), the four-wavefront tile kernel (9 tile rows: 21..23 poses), the 512-thread kernels with their
    tile store in the accumulation / high vector registers in their four instantiations (10 / 11..13 / 14..15 / 16..19 tile rows:
    <= 26 / 34 / 39 / 50 poses), the eight-workgroup kernel of one XCD (ldlt_xcd.hpp: the default from 9 tile rows = 21 free poses on;
    run here with its same-L2 hand-overs, with the agent-scope ones it falls back to, and switched off for the one-workgroup kernels), and the
    many-workgroup blocked kernels of windows with more than 50 free poses (forced here on every size by ORBG_LDLT_WIDE)."""
    prob = synth.make_lba_problem(n_free=nf, n_fixed=3, n_points=40 * nf + 60, mono_frac=0.2, seed=100 + nf)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    o = ob.lba_solve(p)
    variants = [{}, {"ORBG_LDLT_WIDE": "1"}]
    if nf >= 21:
        variants += [{"ORBG_LDLT_XCD": "safe"}, {"ORBG_LDLT_XCD": "0"}]
    for env in variants:
        for key in ("ORBG_LDLT_WIDE", "ORBG_LDLT_XCD"):
            monkeypatch.delenv(key, raising=False)
        for key, val in env.items():
            monkeypatch.setenv(key, val)
        g = api.Optimizer().LocalBundleAdjustment(p)
        assert g.status == o.status and g.iters == o.iters, (nf, env)
        assert np.abs(g.poses - o.poses).max() <= 1e-4 and np.abs(g.points - o.points).max() <= 1e-4, (nf, env)
        assert np.array_equal(g.edge_outlier, o.edge_outlier)
        tg, to = g.trace_rows(), o.trace_rows()
        assert tg.shape == to.shape and np.array_equal(tg[:, 2], to[:, 2]), (nf, env)
        assert np.allclose(tg[:, 1], to[:, 1], rtol=1e-9), (nf, env)


def test_lba_ldlt_timeout_is_not_taken_for_a_rejected_step(monkeypatch):
    """A launch of the eight-workgroup LDL^T whose participants are not all placed (ORBG_LDLT_XCD=short: the handle's first such launch
    is one participant short; its waits give up after 2 s of wall-clock time) reports TIMED OUT, not "not positive definite": the LM
    loop must not answer it with a rejected step (another trajectory than the oracle's).  The window is re-solved on the one-workgroup
    kernels -- same status, iterations, trial counts, outlier sets and state as the oracle -- the handle counts the event, and its next
    window is solved without another one."""
    prob = synth.make_lba_problem(n_free=30, n_fixed=3, n_points=1260, mono_frac=0.2, seed=130)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    o = ob.lba_solve(p)
    monkeypatch.setenv("ORBG_LDLT_XCD", "short")
    opt = api.Optimizer()
    monkeypatch.delenv("ORBG_LDLT_XCD")
    assert opt.watchdog_count() == 0
    import time
    t0 = time.time()
    g = opt.LocalBundleAdjustment(p)
    took = time.time() - t0
    assert opt.watchdog_count() == 1 and 1.5 < took < 10.0, (opt.watchdog_count(), took)
    for rep in range(2):
        assert g.status == o.status == capi.LBA_APPLIED and g.iters == o.iters
        assert np.abs(g.poses - o.poses).max() <= 1e-4 and np.abs(g.points - o.points).max() <= 1e-4
        assert np.array_equal(g.edge_outlier, o.edge_outlier)
        tg, to = g.trace_rows(), o.trace_rows()
        assert tg.shape == to.shape and np.array_equal(tg[:, 2], to[:, 2]) and np.allclose(tg[:, 1], to[:, 1], rtol=1e-9)
        g = opt.LocalBundleAdjustment(p)
    assert opt.watchdog_count() == 1
    opt.close()


def test_lba_large_windows_solved_concurrently_by_three_handles():
    """Three LocalMapping threads of one process (three handles) solve 38 / 45 / 50 free-pose windows at the same time, eight times
    over: their eight-workgroup LDL^T kernels (ldlt_xcd.hpp) spin on each other's hand-overs while the others' workgroups are being
    placed -- on different XCDs per handle -- and every result is the oracle's."""
    import threading
    probs = []
    for nf in (38, 45, 50):
        prob = synth.make_lba_problem(n_free=nf, n_fixed=4, n_points=30 * nf, mono_frac=0.2, seed=500 + nf)
        p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
        probs.append((p, keep, ob.lba_solve(p)))
    results = [[] for _ in probs]

    def work(k):
        opt = api.Optimizer()
        for _ in range(8):
            results[k].append(opt.LocalBundleAdjustment(probs[k][0]))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(probs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
        assert not t.is_alive(), "a concurrent solve did not finish"
    for k, (p, keep, o) in enumerate(probs):
        assert len(results[k]) == 8
        for g in results[k]:
            assert g.status == o.status and g.iters == o.iters
            assert np.abs(g.poses - o.poses).max() <= 1e-4 and np.abs(g.points - o.points).max() <= 1e-4
            assert np.array_equal(g.edge_outlier, o.edge_outlier)
            assert np.array_equal(g.poses, results[k][0].poses)          # ... and the same bits every time


@pytest.mark.parametrize("nf", [51, 52, 56, 64, 65, 83, 120])
def test_lba_windows_beyond_the_benchmark_sizes(nf):
    """More than 50 free poses (a covisibility window of ORB-SLAM3 is not bounded): from 51 on the reduced camera system is
    factorised by the many-workgroup blocked kernels (k_wide_*; 51 is also the last size of the row-pair kernel); beyond 64
    free poses the pair items come from the host (the pose masks no longer fit one word)."""
    prob = synth.make_lba_problem(n_free=nf, n_fixed=3, n_points=25 * nf, mono_frac=0.2, seed=300 + nf)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    o = ob.lba_solve(p)
    g = api.Optimizer().LocalBundleAdjustment(p)
    assert g.status == o.status and g.iters == o.iters
    assert np.abs(g.poses - o.poses).max() <= 1e-4 and np.abs(g.points - o.points).max() <= 1e-4
    assert np.array_equal(g.edge_outlier, o.edge_outlier)
    tg, to = g.trace_rows(), o.trace_rows()
    assert tg.shape == to.shape and np.array_equal(tg[:, 2], to[:, 2]) and np.allclose(tg[:, 1], to[:, 1], rtol=1e-9)


def test_lba_every_window_size_up_to_50_free_poses():
    """Default kernel choice at EVERY free-pose count 1..50 (the sizes the test above leaves out included): each count pads the
    reduced camera system's last tile row differently (6 n mod 16 takes all eight even residues), and the four-wavefront
    kernel's tile -> register map differs per tile-row count."""
    for nf in range(1, 51):
        prob = synth.make_lba_problem(n_free=nf, n_fixed=2, n_points=12 * nf + 40, mono_frac=0.25, seed=7000 + nf)
        p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
        o = ob.lba_solve(p)
        g = api.Optimizer().LocalBundleAdjustment(p)
        assert g.status == o.status and g.iters == o.iters, nf
        assert np.abs(g.poses - o.poses).max() <= 1e-4 and np.abs(g.points - o.points).max() <= 1e-4, nf
        assert np.array_equal(g.edge_outlier, o.edge_outlier), nf
        tg, to = g.trace_rows(), o.trace_rows()
        assert tg.shape == to.shape and np.array_equal(tg[:, 2], to[:, 2]) and np.allclose(tg[:, 1], to[:, 1], rtol=1e-9), nf


@pytest.mark.parametrize("min_obs,max_obs", [(2, 2), (2, 50), (1, 3), (6, 12)])
@pytest.mark.parametrize("mono_frac,outlier_frac", [(0.0, 0.03), (1.0, 0.0), (0.5, 0.3)])
def test_lba_covisibility_structures(min_obs, max_obs, mono_frac, outlier_frac):
    """Shapes of the reduced camera system: block-diagonal-ish (two observations per landmark), dense (a landmark seen by up to
    every keyframe), landmarks with a single observation (3 x 3 landmark blocks of rank 2 when that observation is monocular),
    all-stereo / all-monocular / heavily contaminated edge sets -- at a size of each solve kernel.
    Windows with one or two observations per landmark are badly conditioned: a change of ROUNDING in the solve alone (matrix-core
    LDL^T vs either vector-ALU kernel vs the oracle's Cholesky) moves their chi^2 trace by 6e-9 .. 1e-7, while every solver agrees
    to 2e-14 on 3..8 observations (tools/lba_conditioning.py).  So here: decisions (status, iteration counts, accepted / rejected
    steps, outlier flags) exactly, chi^2 to 1e-6; the 1e-9 bound stays on the well-conditioned windows of the tests above."""
    for nf in (5, 24, 45):
        prob = synth.make_lba_problem(n_free=nf, n_fixed=2, n_points=15 * nf + 50, mono_frac=mono_frac, outlier_frac=outlier_frac,
                                      min_obs=min_obs, max_obs=max_obs, seed=9000 + 100 * min_obs + max_obs + nf)
        p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
        o = ob.lba_solve(p)
        g = api.Optimizer().LocalBundleAdjustment(p)
        tag = (nf, min_obs, max_obs, mono_frac, outlier_frac)
        assert g.status == o.status and g.iters == o.iters, tag
        assert np.array_equal(np.isfinite(g.poses), np.isfinite(o.poses)) and np.array_equal(np.isfinite(g.points), np.isfinite(o.points)), tag
        fin = np.isfinite(o.points)
        assert np.nanmax(np.abs(g.poses - o.poses)) <= 1e-4 and np.abs(g.points[fin] - o.points[fin]).max() <= 1e-4, tag
        assert np.array_equal(g.edge_outlier, o.edge_outlier), tag
        tg, to = g.trace_rows(), o.trace_rows()
        assert tg.shape == to.shape and np.array_equal(tg[:, 2], to[:, 2]) and np.allclose(tg[:, 1], to[:, 1], rtol=1e-6 if min_obs < 3 else 1e-9, equal_nan=True), tag


@pytest.mark.parametrize("k_trials", [1, 2, 3, "round1", "round1+1", 100])
def test_lba_abort_at_a_given_trial_matches_the_oracle(k_trials):
    """*pbStopFlag raised by Tracking while the solve runs (S/LocalMapping.cc:381-386 -> G/core/optimization_algorithm_
    levenberg.cpp:149, G/core/sparse_optimizer.cpp:376, S/Optimizer.cc:2135-2139).  The deterministic form of the flag (-k:
    raised once k LM trials are evaluated) puts the product and the oracle at the same poll point: iteration counts, status
    and the written-back state must agree although the product had speculative launches in flight when it saw the flag."""
    prob = synth.make_lba_problem(n_free=8, n_fixed=3, n_points=400, mono_frac=0.1, seed=321)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    full = ob.lba_solve(p)
    trials_round1 = int(full.trace_rows()[: full.iters[0], 2].sum())
    k = {"round1": trials_round1, "round1+1": trials_round1 + 1}.get(k_trials, k_trials)
    stop_o, stop_g = np.array([-k], np.int32), np.array([-k], np.int32)
    o = ob.lba_solve(p, stop_flag=stop_o)
    opt = api.Optimizer()
    g = opt.LocalBundleAdjustment(p, pbStopFlag=stop_g)
    assert g.status == o.status and g.iters == o.iters, (k, g.iters, o.iters)
    if k < int(full.trace_rows()[:, 2].sum()):
        assert tuple(o.iters) != tuple(full.iters)             # the flag really cut the solve short
    assert np.abs(g.poses - o.poses).max() <= 1e-4 and np.abs(g.points - o.points).max() <= 1e-4
    assert np.array_equal(g.edge_outlier, o.edge_outlier) and np.array_equal(g.edge_depth_pos, o.edge_depth_pos)
    tg, to = g.trace_rows(), o.trace_rows()
    assert tg.shape == to.shape and np.array_equal(tg[:, 2], to[:, 2]) and np.allclose(tg[:, 1], to[:, 1], rtol=1e-9)
    # the handle is clean afterwards: the next (uninterrupted) solve is the full one, bit for bit as on a fresh handle
    again = opt.LocalBundleAdjustment(p)
    fresh = api.Optimizer().LocalBundleAdjustment(p)
    assert again.iters == fresh.iters == full.iters and np.array_equal(again.poses, fresh.poses) and np.array_equal(again.points, fresh.points)


def test_lba_abort_from_another_thread_leaves_the_results_alone():
    """A real asynchronous abort: Tracking raises the flag at an arbitrary moment while the library's LocalMapping thread
    solves (speculative linearisations / solves / exports may be in flight).  After lba_wait the caller-visible arrays must be
    final -- nothing launched on speculation may still write into them -- and the result must be one the solver can
    legitimately return: a prefix of the uninterrupted iteration sequence."""
    import threading
    import time as _time
    prob = synth.make_lba_problem(n_free=20, n_fixed=10, n_points=2000, seed=77)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    opt = api.Optimizer()
    full = opt.LocalBundleAdjustment(p)
    out = views.LbaOutput(p.n_poses, p.n_points, p.n_edges)
    seen = set()
    for rep in range(24):
        stop = np.zeros(1, np.int32)
        delay = 1e-6 * (20 + 45 * rep)                         # 20 us .. 1.1 ms: before, inside and after the solve (~0.8 ms)

        def raiser():
            t_end = _time.perf_counter() + delay
            while _time.perf_counter() < t_end:
                pass
            stop[0] = 1

        th = threading.Thread(target=raiser)
        opt.LocalBundleAdjustmentAsync(p, out, pbStopFlag=stop)
        th.start()
        got = opt.wait()
        th.join()
        snap = (out.poses.copy(), out.points.copy(), out.edge_outlier.copy(), out.edge_chi2.copy() if out.edge_chi2 is not None else None)
        assert got is out and out.status in (capi.LBA_APPLIED, capi.LBA_ABORTED_BEFORE_OPT, capi.LBA_REJECTED_OUTLIERS)
        it = tuple(out.iters)
        seen.add(it)
        assert it[0] <= full.iters[0] and it[1] <= full.iters[1] and (it[1] == 0 or it[0] == full.iters[0])
        assert np.isfinite(out.poses).all() and np.isfinite(out.points).all()
        if it == tuple(full.iters):
            assert np.array_equal(out.poses, full.poses) and np.array_equal(out.points, full.points)
        _time.sleep(0.003)                                     # anything still running on the device would show up now
        assert np.array_equal(out.poses, snap[0]) and np.array_equal(out.points, snap[1]) and np.array_equal(out.edge_outlier, snap[2])
    assert len(seen) >= 2, seen                                # the sweep hit the solve at different poll points
    again = opt.LocalBundleAdjustment(p)
    assert again.iters == full.iters and np.array_equal(again.poses, full.poses) and np.array_equal(again.points, full.points)



def test_lba_abort_through_the_references_bool_flag():
    """`bool* pbStopFlag` as the reference owns it (LocalMapping::mbAbortBA, one byte, raised by Tracking through InterruptBA while
    the solve runs: S/LocalMapping.cc:381-386): the library polls that byte (lba_solve_hb / lba_solve_async_b).  Wherever a solve
    saw the flag -- it reports the LM trials it had evaluated -- the oracle stopped at that poll point gives the same result."""
    import threading
    import time as _time
    prob = synth.make_lba_problem(n_free=20, n_fixed=10, n_points=2000, seed=78)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    opt = api.Optimizer()
    full = opt.LocalBundleAdjustment(p)
    flag = np.ones(1, np.bool_)                                   # raised before the call: S/Optimizer.cc:2127-2129
    g0 = opt.LocalBundleAdjustment(p, pbStopFlag=flag)
    assert g0.status == capi.LBA_ABORTED_BEFORE_OPT and g0.iters == (0, 0)
    out = views.LbaOutput(p.n_poses, p.n_points, p.n_edges)
    seen = set()
    for rep in range(20):
        flag = np.zeros(1, np.uint8)
        delay = 1e-6 * (30 + 40 * rep)

        def raiser():
            t_end = _time.perf_counter() + delay
            while _time.perf_counter() < t_end:
                pass
            flag[0] = 1

        th = threading.Thread(target=raiser)
        opt.LocalBundleAdjustmentAsync(p, out, pbStopFlag=flag)
        th.start()
        got = opt.wait()
        th.join()
        it = tuple(got.iters)
        seen.add(it)
        if got.status == capi.LBA_ABORTED_BEFORE_OPT:
            continue
        k = int(got.trace_rows()[:, 2].sum())
        stop_o = np.array([-k if k > 0 else np.iinfo(np.int32).min], np.int32)
        o = ob.lba_solve(p, stop_flag=stop_o)
        assert got.status == o.status and it == tuple(o.iters), (rep, k, it, o.iters)
        assert np.abs(got.poses - o.poses).max() <= 1e-4 and np.abs(got.points - o.points).max() <= 1e-4, (rep, k)
        assert np.array_equal(got.edge_outlier, o.edge_outlier), (rep, k)
    assert len(seen) >= 2, seen
    again = opt.LocalBundleAdjustment(p)
    assert again.iters == full.iters and np.array_equal(again.poses, full.poses)


def test_blocked_ldlt_is_deterministic_next_to_a_busy_gpu():
    """The many-workgroup blocked LDL^T (windows of more than 50 free poses) re-reads the diagonal block in every panel workgroup:
    a workgroup that is dispatched late must still see the unfactored block.  Other streams keep every compute unit busy while the
    solve runs; results must be the bits of the solve on an idle GPU (and the oracle's, to tolerance)."""
    import torch
    prob = synth.make_lba_problem(n_free=64, n_fixed=3, n_points=1600, mono_frac=0.2, seed=364)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    opt = api.Optimizer()
    idle = opt.LocalBundleAdjustment(p)
    o = ob.lba_solve(p)
    assert idle.iters == o.iters and np.abs(idle.poses - o.poses).max() <= 1e-4
    a = torch.randn(4096, 4096, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(3)]
    for rep in range(6):
        for st in streams:                                        # ~10 ms of large GEMMs on three streams
            with torch.cuda.stream(st):
                b = a
                for _ in range(6):
                    b = b @ a
        busy = opt.LocalBundleAdjustment(p)
        assert busy.iters == idle.iters, rep
        assert np.array_equal(busy.poses, idle.poses) and np.array_equal(busy.points, idle.points), rep
        assert np.array_equal(busy.trace_rows(), idle.trace_rows()), rep
    torch.cuda.synchronize()


def test_handles_run_on_caller_supplied_streams(scene):
    """*_set_stream (include/orbgpu.h, "Streams"): every handle type takes the caller's hipStream_t instead of the library's pooled
    stream, results are unchanged, the caller's stream is not destroyed with the handle, and NULL returns to the pool."""
    import ctypes as C
    import torch
    lib = capi.load()
    streams = [torch.cuda.Stream() for _ in range(4)]
    sp = [C.c_void_p(s.cuda_stream) for s in streams]
    L, R, _ = scene.stereo_pair(3)
    p = scene.frame_view_params()
    fv, keep = views.frame_view(np.zeros(1, capi.KEYPOINT_DTYPE), np.zeros((1, 32), np.uint8), None, None, p["bounds"], p["cam"], 8, 1.2)
    prob = synth.make_lba_problem(n_free=6, n_fixed=2, n_points=240, seed=31)
    lp, keep2 = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    pr = synth.make_pose_opt_problem(n=300, outlier_frac=0.1, mono_frac=0.2, seed=9)
    pp, keep3 = views.pose_opt_problem(pr["Xw"], pr["u"], pr["v"], pr["ur"], pr["inv_sigma2"], pr["cam"], pr["Tcw"])
    bf, b = float(scene.cam["bf"]), float(scene.cam["b"])

    def run(own):
        ex = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=2)
        F = api.Frame()
        opt = api.Optimizer()
        if own:
            capi.check(lib.orbx_set_stream(ex.h, sp[0]), "orbx_set_stream")
            capi.check(lib.orbm_frame_set_stream(F.h, sp[1]), "orbm_frame_set_stream")
            capi.check(lib.lba_set_stream(opt.h, sp[2]), "lba_set_stream")
            capi.check(lib.pose_opt_set_stream(0, sp[3]), "pose_opt_set_stream")
        res = ex.frame_stereo(F, fv, L, R, bf, b)
        kd, dd = F.download()[:2]
        lo = opt.LocalBundleAdjustment(lp)
        po = opt.PoseOptimization(pp)
        if own:
            # back to the pool and once more on the pool's streams
            capi.check(lib.orbx_set_stream(ex.h, None), "orbx_set_stream")
            capi.check(lib.lba_set_stream(opt.h, None), "lba_set_stream")
            capi.check(lib.pose_opt_set_stream(0, None), "pose_opt_set_stream")
            lo2 = opt.LocalBundleAdjustment(lp)
            assert np.array_equal(lo2.poses, lo.poses)
        out = ([np.asarray(x) for x in res], kd.copy(), dd.copy(), lo.poses.copy(), lo.iters, po.Tcw.copy(), np.asarray(po.outliers).copy())
        ex.close(); F.close(); opt.close()
        return out

    a, bres = run(False), run(True)
    for x, y in zip(a[0], bres[0]):
        assert np.array_equal(x, y)
    for i in (1, 2, 3, 5, 6):
        assert np.array_equal(a[i], bres[i]), i
    assert a[4] == bres[4]
    for s in streams:              # the caller's streams are alive and usable after the handles are gone
        with torch.cuda.stream(s):
            t = torch.ones(16, device="cuda") * 2
        s.synchronize()
        assert float(t.sum()) == 32.0


def test_fused_constructor_refuses_a_distorted_camera_view(scene):
    """orbx_frame_stereo* feed mvKeys straight into the grid (mDistCoef[0] == 0, S/Frame.cc:723-727): a view whose image bounds are
    not the image rectangle -- what ComputeImageBounds yields for a distorted camera, S/Frame.cc:753-773 -- is refused, not mis-gridded."""
    L, R, _ = scene.stereo_pair(3)
    p = scene.frame_view_params()
    ex = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=2)
    F = api.Frame()
    bf, b = float(scene.cam["bf"]), float(scene.cam["b"])
    bad_bounds = (-21.7, 655.2, -14.3, 497.9)          # EuRoC-like undistorted corners
    fv, keep = views.frame_view(np.zeros(1, capi.KEYPOINT_DTYPE), np.zeros((1, 32), np.uint8), None, None, bad_bounds, p["cam"], 8, 1.2)
    with pytest.raises(capi.OrbGpuError) as e:
        ex.frame_stereo(F, fv, L, R, bf, b)
    assert e.value.code == capi.ORBG_BAD_ARG
    with pytest.raises(capi.OrbGpuError) as e:
        ex.frame_stereo_submit(F, fv, L, R, bf, b, async_ingest=True)
    assert e.value.code == capi.ORBG_BAD_ARG
    fv2, keep2 = views.frame_view(np.zeros(1, capi.KEYPOINT_DTYPE), np.zeros((1, 32), np.uint8), None, None, p["bounds"], p["cam"], 8, 1.2)
    res = ex.frame_stereo(F, fv2, L, R, bf, b)          # the handle is still usable
    assert res is not None


@pytest.mark.parametrize("W,H,nf", [(752, 480, 1200), (323, 241, 600)])
def test_host_image_submit_on_other_image_shapes(W, H, nf):
    """orbx_frame_stereo_submit through the library's ingest thread on EuRoC / odd image shapes (an image size that is no multiple
    of 16 bytes, rows inside a wider buffer): features, stereo matches and grid equal the oracle's."""
    sc = synth.Scene(W, H, tex_size=(max(2 * W, 800), max(2 * H, 600)), px_per_m=100.0)
    ex = api.ORBextractor(nf, 1.2, 8, 20, 7, W, H, n_cams=2)
    F = api.Frame()
    for k in (2, 5):
        fr = helpers.oracle_stereo_frame(sc, k, n_features=nf)
        fv, keep = helpers.frame_view_of(sc, fr)
        big = [np.full((H, W + 37), 9, np.uint8), np.full((H, W + 37), 200, np.uint8)]
        big[0][:, 5:5 + W] = fr["L"]; big[1][:, 5:5 + W] = fr["R"]
        ex.frame_stereo_submit(F, fv, big[0][:, 5:5 + W], big[1][:, 5:5 + W], float(sc.cam["bf"]), float(sc.cam["b"]), async_ingest=True)
        n, nr = ex.frame_stereo_dev_wait()
        assert n == len(fr["kps"]) and nr == len(fr["kps_r"]), (W, H, k)
        kd, dd = F.download()[:2]
        assert np.array_equal(kd, fr["kps"]) and np.array_equal(dd, fr["desc"]), (W, H, k)
        gs, gi = F.grid()
        os_, oi = ob.build_grid(fv)
        assert np.array_equal(gs, os_) and np.array_equal(gi, oi), (W, H, k)


def test_lba_structure_passes_on_grouped_and_scattered_edge_lists():
    """The host's structure passes (csrc/lba.hip, lba_solve_impl): the reference adds a landmark's edges consecutively, and the passes
    exploit that (rotating packed counters, list positions carried along a run); an edge list in ANY other order -- shuffled, two runs
    per landmark, by pose -- takes the per-landmark cursors.  Every order against the oracle on the same list (sums follow the edge
    order)."""
    code = (
        "import hashlib, numpy as np\n"
        "from multi_orbslam3_amd import api, synth, views\n"
        "from oracle import binding as ob\n"
        "dig = hashlib.sha256()\n"
        "for nf, nfix, npts, seed in [(20, 10, 2000, 3), (7, 3, 300, 4), (1, 2, 40, 5), (12, 0, 500, 6)]:\n"
        "    prob = synth.make_lba_problem(n_free=nf, n_fixed=nfix, n_points=npts, mono_frac=0.3, seed=seed)\n"
        "    E = prob['edges']\n"
        "    rng = np.random.RandomState(seed)\n"
        "    half = np.concatenate([np.arange(0, len(E), 2), np.arange(1, len(E), 2)])          # two runs per landmark\n"
        "    orders = [np.arange(len(E)), rng.permutation(len(E)), half, np.argsort(E['pose'], kind='stable')]\n"
        "    for order in orders:\n"
        "        pr = dict(prob, edges=np.ascontiguousarray(E[order]))\n"
        "        p, keep = views.lba_problem(pr['poses'], pr['pose_fixed'], pr['points'], pr['edges'], pr['cam'])\n"
        "        o = ob.lba_solve(p)\n"
        "        g = api.Optimizer().LocalBundleAdjustment(p)\n"
        "        assert g.status == o.status and g.iters == o.iters, (nf, g.iters, o.iters)\n"
        "        assert np.abs(g.poses - o.poses).max() <= 1e-4 and np.abs(g.points - o.points).max() <= 1e-4, nf\n"
        "        assert np.array_equal(g.edge_outlier, o.edge_outlier), nf\n"
        "        dig.update(np.ascontiguousarray(g.poses).tobytes()); dig.update(np.ascontiguousarray(g.points).tobytes())\n"
        "        dig.update(np.ascontiguousarray(g.edge_chi2).tobytes())\n"
        "print('structure ok', dig.hexdigest())\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ),
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "structure ok" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.parametrize("seed,noise,lam", [(10, 3.0, 0.0), (9, 3.0, 0.0), (11, 3.0, 0.0), (10, 1.0, 1e-12)])
def test_lba_rejected_trials_discard_the_speculative_linearisation(seed, noise, lam):
    """Far-off initial estimates make g2o's LM reject trials (qmax up to 10 in the trace, rounds that end on qmax == 10):
    the linearisation that was launched for a rejected trial state must be dropped, the retry must use the old one, and
    the next round must start from freshly computed residuals."""
    prob = synth.make_lba_problem(n_free=6, n_fixed=2, n_points=150, outlier_frac=0.1, seed=seed)
    rng = np.random.RandomState(seed)
    poses = prob["poses"].copy().reshape(-1, 4, 4)
    for i in range(len(poses)):
        if not prob["pose_fixed"][i]:
            poses[i][:3, 3] += (noise * rng.randn(3)).astype(np.float32)
    poses = poses.reshape(prob["poses"].shape)
    pts = prob["points"] + (0.5 * noise * rng.randn(*prob["points"].shape)).astype(np.float32)
    p, keep = views.lba_problem(poses, prob["pose_fixed"], pts, prob["edges"], prob["cam"], lambda_init=lam)
    o = ob.lba_solve(p)
    g = api.Optimizer().LocalBundleAdjustment(p)
    tg, to = g.trace_rows(), o.trace_rows()
    assert (to[:, 2] > 1).any(), "no LM trial was rejected: the case does not exercise the discard path"
    assert g.status == o.status and g.iters == o.iters
    assert tg.shape == to.shape and np.array_equal(tg[:, 2], to[:, 2])
    # these start several metres off: the discrete LM flow (accept / reject sequence) must be identical; lambda, chi2 and
    # the poorly constrained far points are compared relatively (the reciprocal-based LDL^T differs from the oracle's
    # divisions in the last bits and the rejected trials amplify that)
    assert np.allclose(tg[:, 0], to[:, 0], rtol=1e-5) and np.allclose(tg[:, 1], to[:, 1], rtol=1e-6)
    assert np.abs(g.poses - o.poses).max() <= 1e-4
    assert np.allclose(g.points, o.points, rtol=1e-4, atol=1e-4)


@pytest.mark.timeout(120)
def test_lba_degenerate_inputs_end_like_the_oracle():
    """A free pose with a single observation (rank-deficient pose block) and a NaN landmark: the barrier-free LDL^T must
    not hang on a bad pivot, and status / iteration counts must be the oracle's."""
    prob = synth.make_lba_problem(n_free=5, n_fixed=2, n_points=60, seed=3)
    edges = prob["edges"]
    keep = np.ones(len(edges), bool)
    idx = np.where(edges["pose"] == edges["pose"].max())[0]
    keep[idx[1:]] = False
    e2 = edges[keep]
    pts_nan = prob["points"].copy()
    pts_nan.reshape(-1)[5] = np.nan
    for pts in (prob["points"], pts_nan):
        p, k = views.lba_problem(prob["poses"], prob["pose_fixed"], pts, e2, prob["cam"])
        o = ob.lba_solve(p)
        g = api.Optimizer().LocalBundleAdjustment(p)
        assert g.status == o.status and g.iters == o.iters


def test_lba_stop_flag_and_outlier_rejection():
    prob = synth.make_lba_problem(n_free=4, n_fixed=2, n_points=80)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    stop = np.ones(1, np.int32)
    g = api.Optimizer().LocalBundleAdjustment(p, stop)
    assert g.status == capi.LBA_ABORTED_BEFORE_OPT and np.array_equal(g.poses, prob["poses"])
    bad = synth.make_lba_problem(n_free=4, n_fixed=2, n_points=80, outlier_frac=0.9)
    p2, keep2 = views.lba_problem(bad["poses"], bad["pose_fixed"], bad["points"], bad["edges"], bad["cam"])
    g2 = api.Optimizer().LocalBundleAdjustment(p2)
    o2 = ob.lba_solve(p2)
    assert g2.status == o2.status


def test_c4_sizes_1280x720_2000_features_and_50kf_lba():
    """BASELINE.json configs[3] sizes: 1280x720, 2000 features/frame, 50 free + 20 fixed keyframes, 8000 points."""
    sc = synth.Scene(1280, 720, tex_size=(3200, 1800), px_per_m=400.0)
    L, R, _ = sc.stereo_pair(0)
    ex = api.ORBextractor(2000, 1.2, 8, 20, 7, 1280, 720, n_cams=2)
    (kl, dl), (kr, dr) = ex.extract_stereo(L, R)
    ol = ob.Extractor(n_features=2000, max_width=1280, max_height=720)
    orr = ob.Extractor(n_features=2000, max_width=1280, max_height=720)
    rc, okl, odl, _ = ol.extract(L)
    rc, okr, odr, _ = orr.extract(R)
    assert len(okl) >= 1900
    _assert_extract_equal((kl, dl), (okl, odl), "C4 left")
    _assert_extract_equal((kr, dr), (okr, odr), "C4 right")
    ur, dp = ex.ComputeStereoMatches(float(sc.cam["bf"]), float(sc.cam["b"]), n_left=len(kl))
    our, odp = ob.stereo_match(ol, orr, okl, odl, okr, odr, float(sc.cam["bf"]), float(sc.cam["b"]))
    assert np.array_equal(ur.view(np.uint32), our.view(np.uint32)) and np.array_equal(dp.view(np.uint32), odp.view(np.uint32))
    prob = synth.make_lba_problem(n_free=50, n_fixed=20, n_points=8000, width=1280, height=720)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    g = api.Optimizer().LocalBundleAdjustment(p)
    o = ob.lba_solve(p)
    assert g.status == o.status == capi.LBA_APPLIED and g.iters == o.iters
    assert np.abs(g.poses - o.poses).max() <= 1e-4 and np.abs(g.points - o.points).max() <= 1e-4
    assert np.array_equal(g.edge_outlier, o.edge_outlier)


@pytest.mark.parametrize("far", [False, True])
def test_search_local_points_on_a_large_merged_map(scene, far):
    """A server-size map (about 30 k points: eight copies of a local map shifted out of view, one jittered copy in view, bad and
    zero-observation points, skip flags) takes the large-map path of SearchLocalPoints -- frustum test per thread, ordered
    compaction, window search per surviving point -- and must give what the per-point path and the oracle give: identical
    assignments, identical in-frustum flags."""
    rng = np.random.RandomState(71)
    frs = [helpers.oracle_stereo_frame(scene, k) for k in (2, 6, 9)]
    cur = helpers.oracle_stereo_frame(scene, 4)
    base = helpers.local_map_from(scene, frs, rng)
    parts = []
    for c in range(17):
        p = {k: v.copy() for k, v in base.items()}
        if c % 2 == 0:
            p["pos"] = p["pos"] + np.array([40.0 * (c + 1), -25.0 * c, 3.0 * c], np.float32)      # far outside the frustum
        else:
            p["pos"] = (p["pos"] + rng.randn(*p["pos"].shape) * 0.01).astype(np.float32)          # jittered copy in view
        parts.append(p)
    big = {k: np.concatenate([p[k] for p in parts]) for k in base}
    m = len(big["pos"])
    assert m > 25000
    skip = (rng.rand(m) < 0.1).astype(np.uint8)
    wv, keep2 = helpers.world_view_of(big, skip)
    fv, keep = helpers.frame_view_of(scene, cur)
    F = api.Frame().upload(fv, keep)
    n = len(cur["kps"])
    T = synth.perturb_pose(cur["Tcw"], rng).astype(np.float32)
    amp0 = np.full(n, -1, np.int32); aob0 = np.zeros(n, np.int32)
    amp0[::7] = 5; aob0[::7] = 2                                     # some features already hold a point
    LM = api.LocalMap(65536).upload(wv)
    vis = np.zeros(m, np.uint8)
    g = api.ORBmatcher(0.8, True).SearchLocalPoints(F, LM, T, 3.0, far, 4.5, amp0, aob0, None, in_frustum=vis)
    o = ob.search_local_points(fv, wv, T, 3.0, far, 4.5, 0.8, amp0, aob0)
    assert o[2] > 100 and g[2] == o[2] and np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])
    otrk = ob.is_in_frustum(fv, T, wv, 0.5)
    want_vis = np.asarray(otrk["track_in_view"]).astype(np.uint8) * (1 - skip) * (1 - big["bad"])
    assert np.array_equal(vis, want_vis) and 1000 < int(vis.sum()) < m // 2


def test_search_local_points_on_a_large_map_dense_in_view(scene):
    """The other end of the large-map path: a merged map seen from INSIDE -- 24 jittered copies of a local map, about 43 k points, every
    one of them in the frustum (nothing for the cull to drop: every point becomes a window search, every feature has dozens of
    competing candidates, the candidate lists overflow their slots into the shared region).  Same assignments and in-frustum flags
    as the oracle."""
    rng = np.random.RandomState(72)
    frs = [helpers.oracle_stereo_frame(scene, k) for k in (2, 6, 9)]
    cur = helpers.oracle_stereo_frame(scene, 4)
    base = helpers.local_map_from(scene, frs, rng)
    parts = []
    for c in range(24):
        p = {k: v.copy() for k, v in base.items()}
        p["pos"] = (p["pos"] + rng.randn(*p["pos"].shape) * 0.004).astype(np.float32)
        parts.append(p)
    big = {k: np.concatenate([p[k] for p in parts]) for k in base}
    m = len(big["pos"])
    assert m > 40000
    wv, keep2 = helpers.world_view_of(big, None)
    fv, keep = helpers.frame_view_of(scene, cur)
    F = api.Frame().upload(fv, keep)
    n = len(cur["kps"])
    T = synth.perturb_pose(cur["Tcw"], rng).astype(np.float32)
    amp0 = np.full(n, -1, np.int32); aob0 = np.zeros(n, np.int32)
    LM = api.LocalMap(65536).upload(wv)
    vis = np.zeros(m, np.uint8)
    g = api.ORBmatcher(0.8, True).SearchLocalPoints(F, LM, T, 3.0, False, 0.0, amp0, aob0, None, in_frustum=vis)
    o = ob.search_local_points(fv, wv, T, 3.0, False, 0.0, 0.8, amp0, aob0)
    assert o[2] > 100 and g[2] == o[2] and np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])
    otrk = ob.is_in_frustum(fv, T, wv, 0.5)
    want_vis = np.asarray(otrk["track_in_view"]).astype(np.uint8) * (1 - big["bad"])
    assert np.array_equal(vis, want_vis) and int(vis.sum()) > 0.7 * m


def test_c4_sizes_matchers_2000_features_8k_map_points():
    """The tracking searches at BASELINE.json configs[3] sizes: 1280x720 frames of ~2000 features against a local map of ~8 k map
    points (seven keyframes' worth, shuffled, with bad / zero-observation points): SearchLocalPoints (fused isInFrustum +
    SearchByProjection(F, MPs)), the stand-alone isInFrustum + SearchByProjection(F, MPs) pair, SearchByProjection(Cur, Last) and
    SearchByBoW(KF, F) -- match arrays and counts identical to the oracle's."""
    sc = synth.Scene(1280, 720, tex_size=(3200, 1800), px_per_m=400.0)
    rng = np.random.RandomState(44)
    frs = [helpers.oracle_stereo_frame(sc, k, n_features=2000) for k in (0, 2, 4, 6, 8, 10, 12, 13, 14)]
    cur, last = frs[-1], frs[-2]
    assert len(cur["kps"]) >= 1900
    mp = helpers.local_map_from(sc, frs[:7], rng)
    assert 7000 <= len(mp["pos"]) <= 12000, len(mp["pos"])
    fv, keep = helpers.frame_view_of(sc, cur)
    F = api.Frame(8192).upload(fv, keep)
    n = len(cur["kps"])
    T = synth.perturb_pose(cur["Tcw"], rng).astype(np.float32)
    # ---- SearchByProjection(Cur, Last), then SearchLocalPoints on top of its assignments (as Tracking does)
    lv, keep2 = helpers.make_lastframe(sc, last, rng)
    a0, b0 = np.full(n, -1, np.int32), np.zeros(n, np.int32)
    g1 = api.ORBmatcher(0.9, True).SearchByProjectionFrame(F, T, lv, 7.0, False, a0, b0)
    o1 = ob.search_by_projection_frame(fv, T, lv, 7.0, False, True, a0, b0)
    assert o1[2] > 300 and g1[2] == o1[2] and np.array_equal(g1[0], o1[0]) and np.array_equal(g1[1], o1[1])
    wv, keep3 = helpers.world_view_of(mp)
    LM = api.LocalMap(16384)
    LM.upload(wv)
    g2 = api.ORBmatcher(0.8, True).SearchLocalPoints(F, LM, T, 1.0, False, 0.0, o1[0].copy(), o1[1].copy(), None)
    o2 = ob.search_local_points(fv, wv, T, 1.0, False, 0.0, 0.8, o1[0].copy(), o1[1].copy())
    assert o2[2] > 300 and g2[2] == o2[2] and np.array_equal(g2[0], o2[0]) and np.array_equal(g2[1], o2[1])
    # ---- the two-call form: isInFrustum for every point, then SearchByProjection(F, MPs) on the stored track fields
    trk = F.isInFrustum(T, wv, 0.5)
    otrk = ob.is_in_frustum(fv, T, wv, 0.5)
    for key in otrk:
        assert np.array_equal(np.asarray(trk[key]).view(np.uint8), np.asarray(otrk[key]).view(np.uint8)), key
    assert int(np.asarray(otrk["track_in_view"]).sum()) > 1500
    mv, keep4 = views.mappoints_view(otrk["track_in_view"], mp["bad"], otrk["proj_x"], otrk["proj_y"], otrk["proj_xr"], otrk["track_depth"],
                                     otrk["scale_level"], otrk["view_cos"], mp["desc"], mp["n_obs"])
    g3 = api.ORBmatcher(0.8, True).SearchByProjection(F, mv, 3.0, True, 40.0, np.full(n, -1, np.int32), np.zeros(n, np.int32))
    o3 = ob.search_by_projection_mps(fv, mv, 3.0, True, 40.0, 0.8, np.full(n, -1, np.int32), np.zeros(n, np.int32))
    assert o3[2] > 300 and g3[2] == o3[2] and np.array_equal(g3[0], o3[0]) and np.array_equal(g3[1], o3[1])
    # ---- SearchByBoW(KF, F) with 2000-feature sides
    node = lambda d, k: (d[:, 0].astype(np.int64) >> 2) * 2 + (k["octave"] // 4)
    fvF, kF = views.featvec_view(*views.featvec_from_nodes(node(cur["desc"], cur["kps"])))
    fvK, kK = views.featvec_view(*views.featvec_from_nodes(node(last["desc"], last["kps"])))
    valid = (last["depth"] > 0).astype(np.uint8)
    g4 = api.ORBmatcher(0.7, True).SearchByBoW(F, fvF, last["desc"], valid, last["kps"]["angle"], fvK)
    o4 = ob.search_by_bow(fv, fvF, last["desc"], valid, last["kps"]["angle"], fvK, 0.7, True)
    assert o4[1] > 100 and g4[1] == o4[1] and np.array_equal(g4[0], o4[0])


@pytest.mark.parametrize("shape", [(8, 4, 600, "kb8"), (20, 10, 2000, "kb8"), (30, 6, 1500, "kb8"), (6, 3, 300, "pinhole_left"),
                                   (10, 4, 500, "left_only")])
def test_lba_with_the_two_fisheye_rig(shape):
    """LocalBundleAdjustment of keyframes with mpCamera2 (S/Optimizer.cc:2021-2120): monocular edges through KannalaBrandt8::project /
    projectJac, the right camera's observations as EdgeSE3ProjectXYZToBody after mTrl -- vs. the oracle.  theta and psi of
    KannalaBrandt8::project are float32 values of atan2f / sqrtf (S/CameraModels/KannalaBrandt8.cpp:52-56): the device rounds the
    double atan2 to float, the oracle calls this host's atan2f -- a last-bit difference there moves a residual by ~1e-5 px, hence the
    looser tolerance on chi2 than for the pinhole problems."""
    nf, nx, npts, kind = shape
    kw = {}
    if kind == "pinhole_left":
        kw = dict(left=(capi.CAM_PINHOLE, 190.978, 190.973, 254.932, 256.897))
    if kind == "left_only":
        kw = dict(right_frac=0.0)
    prob = synth.make_lba_rig_problem(n_free=nf, n_fixed=nx, n_points=npts, seed=0xF15E + nf, **kw)
    left, right, Trl = prob["rig"]
    rig = views.camera_rig(left, None, None) if kind == "left_only" else views.camera_rig(left, right, Trl)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"], rig=rig)
    n_right = int((prob["edges"]["ur"] <= -1.5).sum())
    assert (n_right == 0) == (kind == "left_only") and len(prob["edges"]) > 4 * npts // 2
    g = api.Optimizer().LocalBundleAdjustment(p)
    o = ob.lba_solve(p)
    assert g.status == o.status == capi.LBA_APPLIED
    assert g.iters == o.iters
    # (left camera only = a monocular window: a landmark seen three times over 20 cm of baseline has a depth that the last bits of the
    # residuals move by millimetres; the poses, which all landmarks constrain, do not move)
    assert np.abs(g.poses - o.poses).max() <= 1e-4 and np.abs(g.points - o.points).max() <= (5e-3 if kind == "left_only" else 1e-4)
    assert (g.edge_outlier != o.edge_outlier).sum() <= 1 and np.array_equal(g.edge_depth_pos, o.edge_depth_pos)
    tg, to = g.trace_rows(), o.trace_rows()
    assert tg.shape == to.shape and np.array_equal(tg[:, 2], to[:, 2])
    assert np.allclose(tg[:, 1], to[:, 1], rtol=1e-6)
    assert g.chi2[1] < 0.8 * g.chi2[0]
    g2 = api.Optimizer().LocalBundleAdjustment(p)
    assert np.array_equal(g.poses, g2.poses) and np.array_equal(g.points, g2.points)


def test_lba_rig_windows_50_percent_rule_counts_the_left_cameras_edges_only():
    """S/Optimizer.cc:2256: the local BA returns without writing when vToErase holds at least half as many observations as there are
    monocular + stereo edges -- the right camera's edges are in vToErase but not in that sum.  A window whose right-camera observations
    are off by 30 px has fewer outliers than half of ALL its edges and still is refused; the product and the oracle agree (and the
    reference's own text, run in tests/test_reference_formulas.py, says the same)."""
    pr = synth.make_lba_rig_problem(n_free=4, n_fixed=2, n_points=90, seed=113, outlier_frac=0.0)
    right = pr["edges"]["ur"] <= -1.5
    sel = np.sort(np.concatenate([np.nonzero(right)[0], np.nonzero(~right)[0][::3]]))
    E = pr["edges"][sel].copy()
    E["u"][E["ur"] <= -1.5] += 30.0
    p, keep = views.lba_problem(pr["poses"], pr["pose_fixed"], pr["points"], E, pr["cam"], rig=views.camera_rig(*pr["rig"]))
    g = api.Optimizer().LocalBundleAdjustment(p)
    o = ob.lba_solve(p)
    n_left = int((E["ur"] > -1.5).sum())
    assert g.status == o.status == capi.LBA_REJECTED_OUTLIERS
    assert g.n_outliers == o.n_outliers and 0.5 * n_left <= g.n_outliers < 0.5 * len(E)


def test_lba_rig_with_a_pinhole_left_camera_and_no_right_one_is_the_pinhole_problem():
    """A rig that only restates mpCamera = Pinhole{fx, fy, cx, cy} must give the result of the five scalars (the monocular edge
    through Pinhole::project / projectJac is the edge the scalar path writes out)."""
    prob = synth.make_lba_problem(n_free=8, n_fixed=4, n_points=500, mono_frac=0.5, seed=77)
    fx, fy, cx, cy, bf = prob["cam"]
    p0, k0 = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    p1, k1 = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"],
                               rig=views.camera_rig((capi.CAM_PINHOLE, fx, fy, cx, cy)))
    g0 = api.Optimizer().LocalBundleAdjustment(p0)
    g1 = api.Optimizer().LocalBundleAdjustment(p1)
    assert g0.iters == g1.iters and np.array_equal(g0.edge_outlier, g1.edge_outlier)
    assert np.abs(g0.poses - g1.poses).max() <= 1e-6 and np.abs(g0.points - g1.points).max() <= 1e-6
    o1 = ob.lba_solve(p1)
    assert o1.iters == g1.iters and np.abs(o1.poses - g1.poses).max() <= 1e-4


@pytest.mark.parametrize("kind", ["pinhole_stereo", "rig"])
def test_lba_first_lm_step_is_the_dense_gauss_newton_step(kind):
    """The PRODUCT against first principles, not against the oracle's solver: the state after the first accepted LM trial (k_errlin,
    k_schur, the matrix-core LDL^T, k_update) against the dense normal equations over all unknowns solved with numpy (tests/dense_lm.py;
    only the per-edge residuals / Jacobians come from the oracle, and those are pinned against central differences on the CPU)."""
    import sys as _sys
    _sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from dense_lm import dense_first_step, first_step_of
    if kind == "rig":
        pr = synth.make_lba_rig_problem(n_free=6, n_fixed=3, n_points=150, seed=31, outlier_frac=0.02)
        rig = views.camera_rig(*pr["rig"])
    else:
        pr = synth.make_lba_problem(n_free=6, n_fixed=3, n_points=150, seed=32, mono_frac=0.3)
        rig = None
    g = first_step_of(lambda p, stop: api.Optimizer().LocalBundleAdjustment(p, pbStopFlag=stop), pr, rig)
    poses, points, dx = dense_first_step(pr, rig)
    assert np.abs(poses - g.poses.reshape(-1, 4, 4)[:, :3, :]).max() < 2e-6 and np.abs(points - g.points).max() < 5e-6
    assert np.abs(dx).max() > 1e-3


@pytest.mark.parametrize("nl,nr,of", [(300, 200, 0.1), (600, 400, 0.3), (40, 0, 0.0), (0, 60, 0.0), (1500, 1200, 0.2), (2200, 1800, 0.05),
                                      (5, 4, 0.0)])
def test_pose_optimization_with_the_two_fisheye_rig(nl, nr, of):
    """PoseOptimization of a Frame with Nleft != -1 (S/Optimizer.cc:1085-1151): features of the left camera through
    EdgeSE3ProjectXYZOnlyPose with KannalaBrandt8, those of the right camera through EdgeSE3ProjectXYZOnlyPoseToBody."""
    pr = synth.make_pose_opt_rig_problem(n_left=nl, n_right=nr, outlier_frac=of, seed=0xF15F + nl)
    rig = views.camera_rig(*pr["rig"])
    p, keep = views.pose_opt_problem(pr["Xw"], pr["u"], pr["v"], pr["ur"], pr["inv_sigma2"], pr["cam"], pr["Tcw"], rig=rig)
    g = api.Optimizer().PoseOptimization(p)
    o = ob.pose_optimize(p)
    # a correspondence whose chi2 sits within the float32-atan2 difference of the 5.991 threshold may fall on either side
    assert abs(g.n_inliers - o.n_inliers) <= 1 and (g.outliers != o.outliers).sum() <= 1
    assert np.abs(g.Tcw.astype(np.float64) - o.Tcw.astype(np.float64)).max() <= 1e-5
    # near the optimum "did chi2 improve" is decided 1e-9 relative to chi2 -- below the 1e-8 the float32 atan2 leaves between the two
    # sides -- so a round may stop an iteration or two apart (measured both ways round), a hair further along the same valley
    assert all(abs(a - b) <= 2 for a, b in zip(g.iters, o.iters)), (g.iters, o.iters)
    assert np.allclose(g.chi2, o.chi2, rtol=5e-3, atol=1e-6)
    if nl + nr >= 100:
        assert np.abs(g.Tcw - pr["T_true"]).max() < np.abs(pr["Tcw"] - pr["T_true"]).max()
    g2 = api.Optimizer().PoseOptimization(p)
    assert np.array_equal(g.Tcw, g2.Tcw) and g.iters == g2.iters


@pytest.mark.parametrize("n,of,mono", [(500, 0.1, 0.2), (900, 0.3, 0.0), (40, 0.0, 1.0), (8, 0.0, 0.0), (2, 0.0, 0.0), (1, 0.0, 0.0),
                                       (2000, 0.2, 0.1), (3500, 0.05, 0.5), (257, 0.6, 0.3), (513, 0.0, 1.0)])
def test_pose_optimization_parity(n, of, mono):
    """Optimizer::PoseOptimization (row f-2): whole solve in one launch vs. the oracle."""
    pr = synth.make_pose_opt_problem(n=n, outlier_frac=of, mono_frac=mono, seed=100 + n)
    p, keep = views.pose_opt_problem(pr["Xw"], pr["u"], pr["v"], pr["ur"], pr["inv_sigma2"], pr["cam"], pr["Tcw"])
    g = api.Optimizer().PoseOptimization(p)
    o = ob.pose_optimize(p)
    # what the caller sees: the outlier flags, the inlier count, the pose.  The kernel adds the per-correspondence terms in a tree
    # order (round 6: lane folds), g2o / the oracle serially; "chi2 improved by less than 1e-3" is a threshold on sums that differ
    # in their last bits, so a round may run one LM iteration more or less (the sweep below counts how often: a few per cent) --
    # the round then ends a hair further down the same valley
    assert g.n_inliers == o.n_inliers
    assert np.array_equal(g.outliers, o.outliers)
    assert np.abs(g.Tcw.astype(np.float64) - o.Tcw.astype(np.float64)).max() <= 1e-6
    assert all(abs(a - b) <= 1 for a, b in zip(g.iters, o.iters)), (g.iters, o.iters)
    if g.iters == o.iters:
        assert np.allclose(g.chi2, o.chi2, rtol=1e-8, atol=1e-9)
    else:
        assert np.allclose(g.chi2, o.chi2, rtol=1e-4, atol=1e-6)
    g2 = api.Optimizer().PoseOptimization(p)
    assert np.array_equal(g.Tcw, g2.Tcw) and g.iters == g2.iters        # bit-reproducible run to run


def test_lba_async_matches_blocking_call():
    """lba_solve_async / lba_wait (the library's LocalMapping thread) give the blocking call's result bit for bit."""
    prob = synth.make_lba_problem(n_free=6, n_fixed=3, n_points=300)
    p, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    opt = api.Optimizer()
    ref = opt.LocalBundleAdjustment(p)
    out = views.LbaOutput(p.n_poses, p.n_points, p.n_edges)
    for _ in range(3):                                       # the worker is reused across submissions
        opt.LocalBundleAdjustmentAsync(p, out)
        with pytest.raises(capi.OrbGpuError):
            opt.LocalBundleAdjustmentAsync(p, out)           # one solve in flight per handle
        got = opt.wait()
        assert got is out and out.status == capi.LBA_APPLIED and opt.last_solve_ms > 0
        assert np.array_equal(out.poses, ref.poses) and np.array_equal(out.points, ref.points)
        assert np.array_equal(out.edge_outlier, ref.edge_outlier) and out.iters == ref.iters
    assert opt.wait() is None                                # nothing in flight


@pytest.mark.parametrize("k,L,levelsup", [(10, 3, 2), (10, 4, 4), (5, 6, 4)])
def test_vocabulary_transform_parity(scene, k, L, levelsup):
    """DBoW2 transform (row f-3): per-feature walk on the device, BowVector / FeatureVector bookkeeping vs the oracle."""
    fr = helpers.oracle_stereo_frame(scene, 2)
    v = synth.make_vocabulary(k=k, L=L, seed=31 + L, descriptors=fr["desc"])
    vv, keep = views.vocab_view(v["child_start"], v["child_ids"], v["desc"], v["weight"], v["word_id"], L)
    voc = api.ORBVocabulary(vv, keep)
    g = voc.transform_features(fr["desc"], levelsup)
    o = ob.vocab_transform(vv, fr["desc"], levelsup)
    for a, b in zip(g, o):
        assert np.array_equal(a, b)
    assert len(np.unique(o[0])) > 20
    (gw, gv), (gn, gs, gf) = voc.transform(fr["desc"], levelsup)
    (ow, ov), (on, os_, of) = ob.vocab_bow(vv, fr["desc"], levelsup)
    assert np.array_equal(gw, ow) and np.array_equal(gv, ov)            # same accumulation order -> identical doubles
    assert np.array_equal(gn, on) and np.array_equal(gs, os_) and np.array_equal(gf, of)
    # features that are already resident on the device (no descriptor upload)
    fv, keep2 = helpers.frame_view_of(scene, fr)
    F = api.Frame().upload(fv, keep2)
    g2 = voc.transform_features(frame=F, levelsup=levelsup)
    for a, b in zip(g2, o):
        assert np.array_equal(a, b)
    assert voc.transform_features(np.zeros((0, 32), np.uint8), levelsup)[0].shape == (0,)


def test_vocabulary_from_text_at_the_references_size(scene, tmp_path):
    """ORBvoc.txt's size through ORBvoc.txt's format: a k = 10, L = 6 tree (1 111 111 nodes, 10^6 words, 35 MB of descriptors) written
    as saveToTextFile writes it, loaded onto the device by orbv_vocab_from_text (the product's loadFromTextFile: no reference-side
    access to m_nodes) and by the oracle's stream-based reader.  transform() of real frames' descriptors -- word, node at four levels up,
    weight per feature; BowVector and FeatureVector -- bit-equal; then KeyFrameDatabase::DetectNBestCandidates over a database of 400
    keyframes whose BowVectors come from that vocabulary (an inverted file of 10^6 words): candidate lists and scores identical."""
    v = synth.make_full_vocabulary(10, 6)
    vv, keep = views.vocab_view(v["child_start"], v["child_ids"], v["desc"], v["weight"], v["word_id"], v["L"])
    path = str(tmp_path / "voc6.txt")
    ob.vocab_save_text(vv, 10, path)
    voc = api.ORBVocabulary.loadFromTextFile(path)
    o_arr = ob.vocab_load_text(path)
    os.remove(path)
    for key in ("child_start", "child_ids", "desc", "weight", "word_id"):
        assert np.array_equal(o_arr[key], v[key]), key
    ov, okeep = views.vocab_view(o_arr["child_start"], o_arr["child_ids"], o_arr["desc"], o_arr["weight"], o_arr["word_id"], o_arr["L"])
    frames = [helpers.oracle_stereo_frame(scene, t) for t in (2, 9)]
    for fr in frames:
        for levelsup in (4, 2):
            g = voc.transform_features(fr["desc"], levelsup)
            o = ob.vocab_transform(ov, fr["desc"], levelsup)
            for a, b in zip(g, o):
                assert np.array_equal(a, b)
            assert len(np.unique(o[0])) > 0.9 * len(fr["desc"])        # a million words: nearly every feature its own
        (gw, gv), (gn, gs, gf) = voc.transform(fr["desc"], 4)
        (ow, ovv), (on, os_, of) = ob.vocab_bow(ov, fr["desc"], 4)
        assert np.array_equal(gw, ow) and np.array_equal(gv, ovv) and np.array_equal(gn, on) and np.array_equal(gs, os_) and np.array_equal(gf, of)
        assert 10 <= len(gn) <= 100                                    # FeatureVector nodes: level L - 4 = 2 of the tree
    # features resident on the device
    fv, keep2 = helpers.frame_view_of(scene, frames[0])
    F = api.Frame().upload(fv, keep2)
    g2 = voc.transform_features(frame=F, levelsup=4)
    for a, b in zip(g2, ob.vocab_transform(ov, frames[0]["desc"], 4)):
        assert np.array_equal(a, b)
    # ---- a database over the million words: 400 keyframes = noisy copies of 40 "places" (descriptor sets), BowVectors through the oracle
    rng = np.random.RandomState(66)
    base = np.concatenate([fr["desc"] for fr in frames])
    places = [base[rng.choice(len(base), 700, replace=False)] for _ in range(40)]
    bows, inv = [], {}
    for kf in range(400):
        d = places[(kf // 5) % 40].copy()
        flip = rng.rand(len(d)) < 0.3                                   # a third of the features change a few bits: other words
        d[flip] ^= (rng.randint(0, 256, (int(flip.sum()), 32)).astype(np.uint8) & rng.randint(0, 256, (int(flip.sum()), 32)).astype(np.uint8)
                    & rng.randint(0, 256, (int(flip.sum()), 32)).astype(np.uint8))
        (bw, bv), _fvec = ob.vocab_bow(ov, d, 4)
        bows.append((bw, bv))
        for w in bw:
            inv.setdefault(int(w), []).append(kf)
    covis = [[j for j in (kf - 1, kf + 1, kf - 2, kf + 2) if 0 <= j < 400] for kf in range(400)]
    map_id = (np.arange(400) // 200).astype(np.int32)
    dv, dkeep = views.database_view(inv, bows, covis, map_id, np.zeros(400, np.uint8), np.zeros(400, np.uint8), 10 ** 6)
    D = api.KeyFrameDatabase(dv, dkeep)
    pg, po = np.zeros(400, np.float32), np.zeros(400, np.float32)
    total = 0
    for q in range(12):
        kq = int(rng.randint(400))
        (qw, qv), _f = voc.transform(places[(kq // 5) % 40], 4)        # the query's BowVector from the DEVICE vocabulary
        con = np.zeros(400, np.uint8)
        con[max(kq - 2, 0): kq + 3] = 1
        gl, gm = D.DetectNBestCandidates(qw, qv, con, int(map_id[kq]), 3, pg)
        ol, om = ob.detect_n_best_candidates(dv, qw, qv, con, int(map_id[kq]), 3, po)
        assert np.array_equal(gl, ol) and np.array_equal(gm, om), (q, gl, ol, gm, om)
        assert np.array_equal(pg.view(np.uint32), po.view(np.uint32)), q
        total += len(gl) + len(gm)
    assert total > 12


def test_distinctive_descriptors_parity(scene):
    """MapPoint::ComputeDistinctiveDescriptors (row f-3) for a batch of map points, incl. empty, single, tied and >128 lists."""
    rng = np.random.RandomState(4)
    fr = helpers.oracle_stereo_frame(scene, 5)
    lists, start = [], [0]
    for p in range(300):
        N = [0, 1, 2, 3, 5, 9, 16, 40, 129, 7][p % 10]
        base = fr["desc"][rng.randint(0, len(fr["desc"]))]
        noise = rng.randint(0, 256, (N, 32)).astype(np.uint8) & rng.randint(0, 256, (N, 32)).astype(np.uint8) & rng.randint(0, 256, (N, 32)).astype(np.uint8)
        d = np.repeat(base[None], N, 0) ^ noise
        if N >= 3:
            d[2] = d[0]
        lists.append(d); start.append(start[-1] + N)
    desc = np.concatenate(lists)
    g = api.ComputeDistinctiveDescriptors(desc, start)
    o = ob.distinctive_descriptors(desc, start)
    assert np.array_equal(g, o)
    assert (g[::10] == -1).all() and (g[1::10] == 0).all()


def test_keyframe_wire_blocks_and_l1_score(scene):
    """Row f-4: device pack / unpack of KF wire blocks, a KeyFrame rebuilt from a block serves the KeyFrame matcher, L1 BoW score."""
    rng = np.random.RandomState(17)
    kf = helpers.oracle_stereo_frame(scene, 14)
    fv, keep = helpers.frame_view_of(scene, kf)
    F = api.Frame().upload(fv, keep)
    wire = F.pack_wire()
    assert np.array_equal(wire, ob.wire_pack(kf["kps"], kf["desc"]))
    n = len(kf["kps"])
    K = api.Frame().from_wire(fv, wire, n)
    k2, d2 = K.download()
    ok, od = ob.wire_unpack(wire, n)
    assert np.array_equal(d2, od) and all(np.array_equal(k2[f], ok[f]) for f in ("x", "y", "size", "angle", "response", "octave"))
    assert np.array_equal(K.grid()[0], F.grid()[0]) and np.array_equal(K.grid()[1], F.grid()[1])
    # the received KeyFrame gives the same Sim3 projection matches as the original one (size / response are not used by matching)
    mp = helpers.local_map_from(scene, [helpers.oracle_stereo_frame(scene, 12)], rng)
    wv, keep2 = helpers.world_view_of(mp)
    LM = api.LocalMap().upload(wv)
    m = api.ORBmatcher(0.75, True)
    free = np.full(n, -1, np.int32)
    T = kf["Tcw"].astype(np.float32)
    a = m.SearchByProjectionSim3(F, T, LM, free, 4, 1.5)
    b = m.SearchByProjectionSim3(K, T, LM, free, 4, 1.5)
    assert a[1] > 30 and a[1] == b[1] and np.array_equal(a[0], b[0])
    # device-to-device path (what the RCCL exchange uses)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")          # plain HIP runtime calls (torch must not initialise HIP after the library did)
    dptr = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dptr), ctypes.c_size_t(47 * n)) == 0
    F.pack_wire(device_ptr=dptr.value)
    back = np.zeros(47 * n, np.uint8)
    assert hip.hipMemcpy(ctypes.c_void_p(back.ctypes.data), dptr, ctypes.c_size_t(47 * n), 2) == 0     # hipMemcpyDeviceToHost
    assert np.array_equal(back, wire)
    K2 = api.Frame().from_wire(fv, n=n, device_ptr=dptr.value)
    assert np.array_equal(K2.download()[1], od)
    hip.hipFree(dptr)
    # L1 BoW score
    def bow(nw):
        w = np.sort(rng.choice(20000, nw, replace=False)).astype(np.int32)
        v = rng.rand(nw); v /= v.sum()
        return w, v
    qw, qv = bow(400)
    cands = [bow(k) for k in rng.randint(0, 900, 200)] + [(qw, qv)]
    cs = np.cumsum([0] + [len(c[0]) for c in cands]).astype(np.int32)
    cw = np.concatenate([c[0] for c in cands]); cv = np.concatenate([c[1] for c in cands])
    g = api.BowScoreL1(qw, qv, cs, cw, cv)
    o = ob.score_l1(qw, qv, cs, cw, cv)
    assert np.array_equal(g, o) and abs(g[-1] - 1.0) < 1e-12


def test_pose_optimization_sweep_decisions_that_matter_match_the_oracle():
    """160 problems of tools/po_sweep.py (n in [1, 3500], 0-60 % outliers, 0-100 % mono edges, pose errors up to 3 degrees / 15 cm).
    The kernel sums the per-correspondence terms in a tree order, g2o / the oracle serially: `nBadLM` ("chi2 improved by less than
    1e-3") and accept / reject are thresholds on sums that differ in their last bits, so an LM iteration more or less happens in a
    few per cent of the problems (3.6 % of 500, profiles/r4_*_pose_opt_sweep.txt).  What the caller sees must not move: the outlier
    flags, the inlier count the function returns, the pose (to 1e-6)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("po_sweep", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "po_sweep.py"))
    sw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sw)
    opt = api.Optimizer()
    n_iter_diff = 0
    for s in range(1000, 1160):
        pr, meta = sw.problem(s)
        p, keep = views.pose_opt_problem(pr["Xw"], pr["u"], pr["v"], pr["ur"], pr["inv_sigma2"], pr["cam"], pr["Tcw"])
        g, o = opt.PoseOptimization(p), ob.pose_optimize(p)
        assert np.array_equal(g.outliers, o.outliers), (s, meta)
        assert g.n_inliers == o.n_inliers, (s, meta)
        assert np.abs(g.Tcw.astype(np.float64) - o.Tcw.astype(np.float64)).max() <= 1e-6, (s, meta)
        n_iter_diff += g.iters != o.iters
    assert n_iter_diff <= 16, n_iter_diff                # <= 10 % of the problems


def test_completion_fallback_path_gives_same_results(tmp_path):
    """ORBG_NO_POLL=1 switches every completion wait back to the runtime's blocking waits; ORBG_NO_POLL=lba,ingest is the per-role
    policy of a host short of CPUs (the tracking thread spins, the local-BA worker and the ingest thread block).  The start-up
    policy is read once per process, so each form runs in a process of its own: results must not depend on the wait."""
    code = (
        "import sys, numpy as np\n"
        "from multi_orbslam3_amd import _capi as capi, api, synth, views\n"
        "scene = synth.Scene(640, 480)\n"
        "L, R, _ = scene.stereo_pair(4)\n"
        "p = scene.frame_view_params()\n"
        "fv, keep = views.frame_view(np.zeros(1, capi.KEYPOINT_DTYPE), np.zeros((1, 32), np.uint8), None, None, p['bounds'], p['cam'], 8, 1.2)\n"
        "prob = synth.make_lba_problem(n_free=5, n_fixed=2, n_points=200)\n"
        "lp, keep2 = views.lba_problem(prob['poses'], prob['pose_fixed'], prob['points'], prob['edges'], prob['cam'])\n"
        "pr = synth.make_pose_opt_problem(n=300, outlier_frac=0.1, mono_frac=0.2, seed=9)\n"
        "pp, keep3 = views.pose_opt_problem(pr['Xw'], pr['u'], pr['v'], pr['ur'], pr['inv_sigma2'], pr['cam'], pr['Tcw'])\n"
        "ex = api.ORBextractor(1000, 1.2, 8, 20, 7, 640, 480, n_cams=2)\n"
        "F = api.Frame()\n"
        "res = ex.frame_stereo(F, fv, L, R, float(scene.cam['bf']), float(scene.cam['b']))\n"
        "ex.frame_stereo_submit(F, fv, L, R, float(scene.cam['bf']), float(scene.cam['b']), async_ingest=True)\n"
        "n2 = ex.frame_stereo_dev_wait()\n"
        "opt = api.Optimizer()\n"
        "lo = opt.LocalBundleAdjustment(lp)\n"
        "opt.LocalBundleAdjustmentAsync(lp, views.LbaOutput(lp.n_poses, lp.n_points, lp.n_edges))\n"
        "lo2 = opt.wait()\n"
        "po = opt.PoseOptimization(pp)\n"
        "out = {'r%d' % i: np.asarray(x) for i, x in enumerate(res)}\n"
        "out.update(n2=np.asarray(n2), lba_poses=lo.poses, lba_iters=np.asarray(lo.iters), lba2_poses=lo2.poses, po_T=po.Tcw, po_out=np.asarray(po.outliers))\n"
        "np.savez(sys.argv[1], **out)\n"
        "print('ok')\n")
    outs = []
    for j, no_poll in enumerate((None, "1", "lba,ingest")):
        env = dict(os.environ)
        env.pop("ORBG_NO_POLL", None)
        if no_poll:
            env["ORBG_NO_POLL"] = no_poll
        f = str(tmp_path / ("poll%d.npz" % j))
        r = subprocess.run([sys.executable, "-c", code, f], capture_output=True, text=True, timeout=600, env=env,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
        outs.append(dict(np.load(f)))
    a = outs[0]
    for b in outs[1:]:
        assert sorted(a) == sorted(b)
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["lba_poses"], a["lba2_poses"])


def test_detect_n_best_candidates_parity():
    """Row f-4: KeyFrameDatabase::DetectNBestCandidates on the device-resident database (inverted-file walk, common-word counts,
    0.8 * max filter, L1 scores, covisibility accumulation) vs the oracle, over a sequence of queries that share the
    per-keyframe score state; candidate lists identical, scores bit-identical."""
    rng = np.random.RandomState(11)
    db = helpers.random_database(rng, n_kfs=600, n_words=6000)
    db["n_words"] += 1                                           # one word no keyframe contains
    v, keep = views.database_view(db["inv"], db["bows"], db["covis"], db["map_id"], db["bad"], db["map_bad"], db["n_words"])
    D = api.KeyFrameDatabase(v, keep)
    K = len(db["bows"])
    pg, po = np.zeros(K, np.float32), np.zeros(K, np.float32)
    total = 0
    for q in range(25):
        kq = rng.randint(K)
        qw, qv = db["bows"][kq]
        if q == 7:                                               # a query that shares no word with anybody
            unused = [w for w in range(db["n_words"]) if w not in db["inv"]]
            qw, qv = np.array(unused[:1], np.int32), np.array([1.0])
        con = np.zeros(K, np.uint8)
        con[max(kq - 3, 0): kq + 4] = 1
        n_c = int(rng.choice([1, 3, 5]))
        gl, gm = D.DetectNBestCandidates(qw, qv, con, int(db["map_id"][kq]), n_c, pg)
        ol, om = ob.detect_n_best_candidates(v, qw, qv, con, int(db["map_id"][kq]), n_c, po)
        assert np.array_equal(gl, ol) and np.array_equal(gm, om), (q, gl, ol, gm, om)
        assert np.array_equal(pg.view(np.uint32), po.view(np.uint32)), q
        total += len(gl) + len(gm)
    assert total > 40


@pytest.mark.parametrize("case", ["one_kf", "all_bad", "all_connected", "other_maps_bad", "large"])
def test_detect_n_best_candidates_edge_databases(case):
    """The database at its edges: a single keyframe; every keyframe bad; every keyframe connected to the query (nothing may be
    returned); every other map bad (no merge candidates); 3000 keyframes over 20 000 words."""
    rng = np.random.RandomState({"one_kf": 1, "all_bad": 2, "all_connected": 3, "other_maps_bad": 4, "large": 5}[case])
    if case == "one_kf":
        db = helpers.random_database(rng, n_kfs=1, n_words=500)
    elif case == "large":
        db = helpers.random_database(rng, n_kfs=3000, n_words=20000)
    else:
        db = helpers.random_database(rng, n_kfs=120, n_words=2000)
    if case == "all_bad":
        db["bad"][:] = 1
    if case == "other_maps_bad":
        db["map_bad"][1:] = 1
    v, keep = views.database_view(db["inv"], db["bows"], db["covis"], db["map_id"], db["bad"], db["map_bad"], db["n_words"])
    D = api.KeyFrameDatabase(v, keep)
    K = len(db["bows"])
    pg, po = np.zeros(K, np.float32), np.zeros(K, np.float32)
    for q in range(6):
        kq = rng.randint(K)
        qw, qv = db["bows"][kq]
        con = np.ones(K, np.uint8) if case == "all_connected" else np.zeros(K, np.uint8)
        con[kq] = 1
        gl, gm = D.DetectNBestCandidates(qw, qv, con, int(db["map_id"][kq]), 3, pg)
        ol, om = ob.detect_n_best_candidates(v, qw, qv, con, int(db["map_id"][kq]), 3, po)
        assert np.array_equal(gl, ol) and np.array_equal(gm, om), (case, q, gl, ol, gm, om)
        assert np.array_equal(pg.view(np.uint32), po.view(np.uint32)), (case, q)
        if case in ("one_kf", "all_bad", "all_connected"):
            assert len(ol) == 0 and len(om) == 0
        if case == "other_maps_bad":
            assert len(om) == 0


def test_library_and_torch_share_one_hip_runtime_whatever_the_import_order():
    """PyTorch-ROCm maps its own HIP / HSA runtime even when the system's is mapped already; a second runtime in the process
    finds no GPU.  `_capi.load()` therefore imports torch before it maps liborbgpu.so: a process that uses the library first and
    torch afterwards (this file does, test by test) must still get the GPU from torch, and only one libamdhip64 may be mapped."""
    code = ("from multi_orbslam3_amd import api\n"
            "opt = api.Optimizer()\n"
            "import torch\n"
            "a = torch.ones(8, device='cuda')\n"
            "assert float(a.sum().item()) == 8.0\n"
            "libs = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l))\n"
            "assert len(libs) == 1, libs\n"
            "print('one runtime:', libs[0])\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "one runtime:" in r.stdout


def test_one_stream_per_handle_is_still_a_working_configuration():
    """ORBG_STREAM_POOL=0: every handle creates a stream of its own (the layout before the library pooled its streams into three
    hardware-queue-sized ones, csrc/common.hpp).  The frame constructor, the matchers and the local BA agree with the oracle
    there as well."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ORBG_STREAM_POOL="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-k",
                        "test_fused_stereo_frame_constructor or test_search_by_projection or test_lba_parity or test_lba_async or test_pose_optimization_parity"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout


def test_ldlt_kernels_against_a_long_double_host_factorisation():
    """tools/micro/ldlt_mfma_test: every LDL^T kernel of csrc/ldlt_mfma.hpp (column kernel, 8-wavefront tile kernel, the four forms
    of the 4-wavefront kernel) and the eight-workgroup kernel of csrc/ldlt_xcd.hpp (from 9 tile rows on: after one launch, after
    210, and with its agent-scope hand-overs forced) on random SPD systems of 6 ... 300 unknowns against a long-double factorisation
    on the host (|dx| <= 1e-10 max|x|), the operand layout probe of v_mfma_f64_16x16x4_f64 and the zero-pivot flags.  Built
    without the in-kernel timelines (-DNO_PROFILE: the code the library ships)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tools", "micro", "ldlt_mfma_test.hip")
    exe = os.path.join(root, "tools", "micro", "ldlt_mfma_test_np")
    hdr = os.path.join(root, "multi_orbslam3_amd", "csrc", "ldlt_mfma.hpp")
    hdr2 = os.path.join(root, "multi_orbslam3_amd", "csrc", "ldlt_xcd.hpp")
    inc = os.path.join(root, "multi_orbslam3_amd", "csrc", "ldlt_jump_tables.inc")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(p) for p in (src, hdr, hdr2, inc)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-DNO_PROFILE", "-o", exe, src],
                              timeout=900)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL OK" in r.stdout and "FAIL" not in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]
    sizes = [ln for ln in r.stdout.splitlines() if ln.startswith("n=")]
    assert len(sizes) == 16, sizes
    assert sum("xcd (8 workgroups" in ln for ln in r.stdout.splitlines()) == 9 and "xcd zero pivot: ok=0 ok" in r.stdout, r.stdout[-3000:]
    # 20000 launches of the eight-workgroup kernel on two alternating systems, the participating XCD changing every 64 launches and the
    # launch counter wrapping half way: every sampled result bit-identical to the first of its size, no wait that does not end
    r = subprocess.run([exe, "stress", "20000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "stress: 20000 launches, ALL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-500:]
    # a participant that the dispatcher never places (here: a grid one participant short): every wait gives up after its bound (wall-clock
    # time since the wavefront started), the launch comes back as TIMED OUT (ok = -2: distinct from 0 = not positive definite, which
    # is an LM verdict) instead of never, and the context's next launch is a good one (-DXWATCHDOG: the bound is 0.25 s instead of 2 s,
    # and the wait is named)
    exe_wd = os.path.join(root, "tools", "micro", "ldlt_mfma_test_wd")
    if not os.path.exists(exe_wd) or os.path.getmtime(exe_wd) < max(os.path.getmtime(p) for p in (src, hdr, hdr2, inc)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-DNO_PROFILE", "-DXWATCHDOG", "-o", exe_wd, src],
                              timeout=900)
    r = subprocess.run([exe_wd], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "xcd with a participant missing: ok=-2" in r.stdout and "xcd after the timed-out launch: ok=1 ok" in r.stdout and "FAIL" not in r.stdout, r.stdout[-2000:] + r.stderr[-500:]


RIG_CAMERAS = {"two fisheyes": (synth.KB8_LEFT, synth.KB8_RIGHT),
               "two pinholes": ((capi.CAM_PINHOLE, 260.0, 259.0, 255.0, 257.0), (capi.CAM_PINHOLE, 261.0, 260.5, 253.0, 256.0))}


@pytest.mark.parametrize("cams", sorted(RIG_CAMERAS))
def test_is_in_frustum_of_a_two_camera_frame(cams):
    """Frame::isInFrustum with Nleft != -1 (S/Frame.cc:545-554): isInFrustumChecks through either camera (:1154-1231) -- flags and
    predicted levels equal the oracle's; projections within 3e-4 px (KannalaBrandt8::project goes through atan2f / cosf / sinf,
    which the device takes as the rounded float64 functions and the oracle from the host's libm), depths and viewing cosines bit-equal."""
    left, right = RIG_CAMERAS[cams]
    sc = synth.make_rig_track_scene(left=left, right=right)
    fl, fr, wv, rig, keep = helpers.rig_track_views(sc)
    F = api.Frame().upload(fl, keep[0])
    g = F.isInFrustumRig(sc["Tcw"], rig, sc["Tlr"], wv)
    o = ob.is_in_frustum_rig(fl, sc["Tcw"], rig, sc["Tlr"], wv)
    for side in (0, 1):
        assert 700 < int(o[side]["track_in_view"].sum()) < len(sc["pos"]) - 200
        for k in ob.RIG_TRACK_KEYS:
            a, b = g[side][k], o[side][k]
            if k in ("proj_x", "proj_y") and cams == "two fisheyes":
                assert np.abs(a.astype(np.float64) - b).max() <= 3e-4, (side, k)      # pixels (one ulp of psi moves f r cos(psi) by up to 1e-4)
            elif a.dtype == np.float32:
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (side, k)
            else:
                assert np.array_equal(a, b), (side, k)
    both = o[0]["track_in_view"] & o[1]["track_in_view"]
    assert 0 < int(both.sum()) < int((o[0]["track_in_view"] | o[1]["track_in_view"]).sum())     # points only one camera sees
    assert (o[0]["scale_level"][o[0]["track_in_view"] == 0] == -1).all()


@pytest.mark.parametrize("th,far,case", [(1.0, False, "plain"), (3.0, True, "plain"), (1.0, True, "freed"), (15.0, False, "crowded")])
def test_search_by_projection_map_points_on_a_two_camera_frame(th, far, case):
    """ORBmatcher::SearchByProjection(Frame, MapPoints) with Nleft != -1 (S/ORBmatcher.cc:44-214): the left camera's block with the
    stereo partner on the right written along, the right camera's block (:145-211: its own grid, no th, partner on the left written
    whatever it held), a failed ratio test on the left leaving the point altogether -- match arrays and counts equal the oracle's.
    "freed": most points have no observations and many features hold a point at entry, so that a partner write frees a feature the
    kernels had left out (the call repeats on unfiltered lists); "crowded": th = 15 on a frame where a fifth of the features is taken."""
    kw = dict(plain={}, freed=dict(zero_obs_frac=0.6, occupied_frac=0.35, stereo_frac=0.9, seed=0xF1E1), crowded=dict(occupied_frac=0.2, seed=0xF1E2))[case]
    sc = synth.make_rig_track_scene(**kw)
    fl, fr, wv, rig, keep = helpers.rig_track_views(sc)
    a, b = ob.is_in_frustum_rig(fl, sc["Tcw"], rig, sc["Tlr"], wv)
    mv, mvr, keep2 = helpers.rig_mappoint_views(sc, a, b)
    FL, FR = api.Frame().upload(fl, keep[0]), api.Frame().upload(fr, keep[1])
    m = api.ORBmatcher(0.8)
    g = m.SearchByProjectionRig(FL, FR, mv, mvr, sc["left_to_right"], sc["right_to_left"], th, far, 6.0, sc["assigned_mp"], sc["assigned_obs"])
    o = ob.search_by_projection_mps_rig(fl, fr, mv, mvr, sc["left_to_right"], sc["right_to_left"], th, far, 6.0, 0.8, sc["assigned_mp"], sc["assigned_obs"])
    nl = len(sc["kps_left"])
    changed = o[0] != sc["assigned_mp"]
    assert o[2] > 300 and changed[:nl].sum() > 100 and changed[nl:].sum() > 100
    assert g[2] == o[2] and np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])
    if case == "freed":                                       # the scene does contain the event the second pass exists for
        occ0 = (sc["assigned_mp"] >= 0) & (sc["assigned_obs"] > 0)
        assert int((occ0 & changed & (o[1] == 0)).sum()) > 0
    # a second call on the result: features taken by the first are not candidates any more
    g2 = m.SearchByProjectionRig(FL, FR, mv, mvr, sc["left_to_right"], sc["right_to_left"], th, far, 6.0, g[0], g[1])
    o2 = ob.search_by_projection_mps_rig(fl, fr, mv, mvr, sc["left_to_right"], sc["right_to_left"], th, far, 6.0, 0.8, o[0], o[1])
    assert g2[2] == o2[2] and np.array_equal(g2[0], o2[0]) and np.array_equal(g2[1], o2[1])


def test_rig_search_refuses_what_it_cannot_be():
    sc = synth.make_rig_track_scene(n_points=50, n_distract=20)
    fl, fr, wv, rig, keep = helpers.rig_track_views(sc)
    a, b = ob.is_in_frustum_rig(fl, sc["Tcw"], rig, sc["Tlr"], wv)
    mv, mvr, keep2 = helpers.rig_mappoint_views(sc, a, b)
    FL, FR = api.Frame().upload(fl, keep[0]), api.Frame().upload(fr, keep[1])
    m = api.ORBmatcher(0.8)
    with pytest.raises(capi.OrbGpuError):                     # one frame object for both cameras
        m.SearchByProjectionRig(FL, FL, mv, mvr, sc["left_to_right"], sc["right_to_left"], 1.0, False, 0.0, sc["assigned_mp"], sc["assigned_obs"])
    bad = sc["left_to_right"].copy(); bad[0] = len(sc["kps_right"])
    with pytest.raises(capi.OrbGpuError):                     # a partner index outside the right camera's features
        m.SearchByProjectionRig(FL, FR, mv, mvr, bad, sc["right_to_left"], 1.0, False, 0.0, sc["assigned_mp"], sc["assigned_obs"])
    odd = views.camera_rig(sc["left"], sc["right"], sc["Trl"]); odd.right.model = 7
    with pytest.raises(capi.OrbGpuError):                     # a camera model the library does not know
        FL.isInFrustumRig(sc["Tcw"], odd, sc["Tlr"], wv)


@pytest.mark.parametrize("case", ["sideways", "forward", "backward", "mono_flag", "no_orientation_check", "crowded", "pinholes"])
def test_search_by_projection_last_frame_on_a_two_camera_frame(case):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) with CurrentFrame.Nleft != -1 (S/ORBmatcher.cc:1970-2186): the
    left camera's search, the right camera's (:2092-2160, the point moved by mTrl and projected through mpCamera), a point whose left
    window is empty skipped on both sides, level windows by the direction of motion, the rotation histogram over both cameras'
    matches -- match arrays and counts equal the oracle's."""
    kw = dict(crowded=dict(occupied_frac=0.3, seed=0xF1E3), pinholes=dict(left=RIG_CAMERAS["two pinholes"][0], right=RIG_CAMERAS["two pinholes"][1])).get(case, {})
    sc = synth.make_rig_track_scene(**kw)
    fl, fr, wv, rig, keep = helpers.rig_track_views(sc)
    motion = dict(forward=(0.02, 0.0, 0.4), backward=(0.0, -0.03, -0.4)).get(case, (0.03, 0.01, 0.02))
    last = synth.rig_last_frame(sc, motion=motion)
    lv, keep2 = views.lastframe_view(last["mp_valid"], last["outlier"], last["world_pos"], last["desc"], last["octave"], last["angle"], last["n_obs"], last["Tcw"])
    FL, FR = api.Frame().upload(fl, keep[0]), api.Frame().upload(fr, keep[1])
    check = case != "no_orientation_check"
    mono = case == "mono_flag"
    m = api.ORBmatcher(0.9, check)
    nl = len(sc["kps_left"])
    for th in (7.0, 15.0):
        g = m.SearchByProjectionFrameRig(FL, FR, sc["Tcw"], rig, lv, th, mono, sc["assigned_mp"], sc["assigned_obs"])
        o = ob.search_by_projection_frame_rig(fl, fr, sc["Tcw"], rig, lv, th, mono, check, sc["assigned_mp"], sc["assigned_obs"])
        changed = o[0] != sc["assigned_mp"]
        assert o[2] > 250 and changed[:nl].sum() > 100 and changed[nl:].sum() > 100, (th, o[2])
        assert g[2] == o[2] and np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1]), th
    with pytest.raises(capi.OrbGpuError):
        m.SearchByProjectionFrameRig(FL, FL, sc["Tcw"], rig, lv, 7.0, mono, sc["assigned_mp"], sc["assigned_obs"])


def _rig_bow_scene(sc, n_kf=900, seed=9, shift=5):
    """A keyframe holding the scene's first n_kf points (descriptors a few bits off the points', angles near the frame features'), the
    two-camera frame's ALL-features view, and the two feature vectors over a stand-in vocabulary (node = leading descriptor bits)."""
    rng = np.random.RandomState(seed)
    kps = np.concatenate([sc["kps_left"], sc["kps_right"]]); desc = np.concatenate([sc["desc_left"], sc["desc_right"]])
    bounds = (0, sc["size"], 0, sc["size"])
    fv, keep = views.frame_view(kps, desc, None, None, bounds, (sc["left"][1], sc["left"][2], sc["left"][3], sc["left"][4], 0.0, 0.1))
    kf_desc = sc["desc"][:n_kf].copy()
    flips = rng.randint(0, 256, kf_desc.shape).astype(np.uint8) & rng.randint(0, 256, kf_desc.shape).astype(np.uint8) & rng.randint(0, 256, kf_desc.shape).astype(np.uint8) \
        & rng.randint(0, 256, kf_desc.shape).astype(np.uint8)
    flips[:, 0] &= 0x07                                   # (keep the node bits: the vocabulary puts near descriptors into one word)
    kf_desc ^= flips
    kf_angle = ((sc["base_angle"][:n_kf] + rng.randn(n_kf) * 2.0) % 360.0).astype(np.float32)
    valid = (rng.rand(n_kf) < 0.85).astype(np.uint8)
    node = lambda d: d[:, 0].astype(np.int64) >> shift
    fvF, kF = views.featvec_view(*views.featvec_from_nodes(node(desc)))
    fvK, kK = views.featvec_view(*views.featvec_from_nodes(node(kf_desc)))
    return fv, keep, fvF, fvK, kf_desc, kf_angle, valid, [kF, kK]


@pytest.mark.parametrize("case", ["fine_nodes", "coarse_nodes", "no_orientation_check", "no_right_features_in_the_vectors"])
def test_search_by_bow_on_a_two_camera_frame(case):
    """ORBmatcher::SearchByBoW(KeyFrame, Frame) with F.Nleft != -1 (S/ORBmatcher.cc:342-430): the best two of a bucket per camera, the
    left camera's best through TH_LOW and the ratio test, the right camera's through TH_LOW alone and only under the left's gate, both
    written into vpMapPointMatches and the rotation histogram -- matches and count equal the oracle's."""
    sc = synth.make_rig_track_scene()
    shift = 3 if case == "fine_nodes" else 5
    fv, keep, fvF, fvK, kf_desc, kf_angle, valid, keep2 = _rig_bow_scene(sc, shift=shift)
    nl = len(sc["kps_left"])
    n_left = fv.n if case == "no_right_features_in_the_vectors" else nl      # (every feature counts as the left camera's: the single-camera rule per bucket)
    check = case != "no_orientation_check"
    F = api.Frame().upload(fv, keep)
    g = api.ORBmatcher(0.7, check).SearchByBoWRig(F, n_left, fvF, kf_desc, valid, kf_angle, fvK)
    o = ob.search_by_bow_rig(fv, n_left, fvF, kf_desc, valid, kf_angle, fvK, 0.7, check)
    assert g[1] == o[1] and np.array_equal(g[0], o[0]), case
    if case == "no_right_features_in_the_vectors":
        s = ob.search_by_bow(fv, fvF, kf_desc, valid, kf_angle, fvK, 0.7, check)
        assert s[1] == o[1] and np.array_equal(s[0], o[0])
    else:
        assert o[1] > 150 and (o[0][:nl] >= 0).sum() > 60 and (o[0][nl:] >= 0).sum() > 60, (o[1], (o[0][:nl] >= 0).sum(), (o[0][nl:] >= 0).sum())


@pytest.mark.parametrize("case", ["regular", "few_right_features", "no_monocular_part", "many_twins"])
def test_fisheye_stereo_matches_of_the_frame_constructor(case):
    """Frame::ComputeStereoFishEyeMatches (S/Frame.cc:1093-1150): 2-nearest-neighbour Hamming match of the lapping-area features, Lowe's
    ratio, KannalaBrandt8::TriangulateMatches (parallax, depths, reprojection errors) -- mvLeftToRightMatch / mvRightToLeftMatch and
    the match count equal the oracle's; mvDepth and mvStereo3Dpoints to 1e-5 relative (both solve the 4 x 4 null vector by Jacobi
    rotations in float64; the float32 tan / atan2 / cos / sin differ in the last place)."""
    kw = dict(regular={}, few_right_features=dict(n_stereo=1, n_distract=0, n_mono_right=5), no_monocular_part=dict(n_mono_left=0, n_mono_right=0, seed=0xF15D),
              many_twins=dict(n_stereo=1200, n_distract=400, seed=0xF15E))[case]
    sc = synth.make_fisheye_stereo_scene(**kw)
    v, keep = views.fisheye_stereo_view(sc["kps_left"], sc["desc_left"], sc["mono_left"], sc["kps_right"], sc["desc_right"], sc["mono_right"], sc["left"],
                                        sc["right"], sc["Tlr"], sc["level_sigma2"])
    g = api.ComputeStereoFishEyeMatches(v)
    o = ob.fisheye_stereo_matches(v)
    assert g[4] == o[4] and np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1]), (case, g[4], o[4])
    hit = o[0] >= 0
    assert np.array_equal(g[2] < 0, o[2] < 0) and (o[2][~hit] == -1).all()
    if case == "few_right_features":
        assert o[4] <= 1
        return
    assert o[4] > 150 and (o[0][: sc["mono_left"]] == -1).all() and (o[1][: sc["mono_right"]] == -1).all()
    assert np.abs(g[2][hit] - o[2][hit]).max() <= 1e-5 * np.abs(o[2][hit]).max() and np.abs(g[3][hit] - o[3][hit]).max() <= 1e-5 * np.abs(o[3][hit]).max()
    # every accepted pair is mutual unless a later left feature took the right one over
    back = o[1][o[0][hit]]
    assert (back >= np.nonzero(hit)[0]).all() and (back == np.nonzero(hit)[0]).mean() > 0.9


def test_matcher_with_one_camera_behind_a_model():
    """A monocular fisheye frame (Nleft == -1, mpCamera a KannalaBrandt8, no second camera): Frame::isInFrustum's Nleft == -1 branch and
    SearchByProjection(CurrentFrame, LastFrame) project through mpCamera->project (S/Frame.cc:489, S/ORBmatcher.cc:2012).  The rig
    entry points with a rig that has no right camera: the frustum's left outputs are the two-camera call's left outputs, the frame
    search (right frame = NULL) equals the oracle's and differs from the pinhole entry point on the same inputs."""
    sc = synth.make_rig_track_scene()
    fl, fr, wv, rig, keep = helpers.rig_track_views(sc)
    mono = views.camera_rig(sc["left"])
    F = api.Frame().upload(fl, keep[0])
    g_two, _ = F.isInFrustumRig(sc["Tcw"], rig, sc["Tlr"], wv)
    g_one, g_none = F.isInFrustumRig(sc["Tcw"], mono, sc["Tlr"], wv)
    o_one, _ = ob.is_in_frustum_rig(fl, sc["Tcw"], mono, sc["Tlr"], wv)
    for k in ob.RIG_TRACK_KEYS:
        assert np.array_equal(g_one[k], g_two[k]), k
        if k in ("proj_x", "proj_y"):
            assert np.abs(g_one[k].astype(np.float64) - o_one[k]).max() <= 3e-4, k
        else:
            assert np.array_equal(g_one[k], o_one[k]), k
    assert not g_none["track_in_view"].any()
    last = synth.rig_last_frame(sc)
    lv, keep2 = views.lastframe_view(last["mp_valid"], last["outlier"], last["world_pos"], last["desc"], last["octave"], last["angle"], last["n_obs"], last["Tcw"])
    m = api.ORBmatcher(0.9, True)
    nl = len(sc["kps_left"])
    amp0, aob0 = sc["assigned_mp"][:nl], sc["assigned_obs"][:nl]
    g = m.SearchByProjectionFrameRig(F, None, sc["Tcw"], mono, lv, 15.0, True, amp0, aob0)
    o = ob.search_by_projection_frame_rig(fl, None, sc["Tcw"], mono, lv, 15.0, 1, 1, amp0, aob0)
    assert g[2] == o[2] > 150 and np.array_equal(g[0], o[0]) and np.array_equal(g[1], o[1])
    p = ob.search_by_projection_frame(fl, sc["Tcw"], lv, 15.0, 1, 1, amp0, aob0)          # the pinhole of the view: another projection
    assert not np.array_equal(p[0], o[0])
    with pytest.raises(capi.OrbGpuError):                     # a right frame without a right camera
        m.SearchByProjectionFrameRig(F, api.Frame().upload(fr, keep[1]), sc["Tcw"], mono, lv, 15.0, True, amp0, aob0)


@pytest.mark.parametrize("th,orb_dist", [(10.0, 100), (3.0, 64)])
def test_relocalisation_search_with_a_camera_model(th, orb_dist):
    """SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) on a monocular fisheye frame: S/ORBmatcher.cc:2217 projects through
    CurrentFrame.mpCamera, a KannalaBrandt8 here -- orbm_search_by_projection_reloc_cam against the oracle; with a PINHOLE handed in as
    the model it is the pinhole entry point."""
    sc = synth.make_rig_track_scene()
    fl, fr, wv, rig, keep = helpers.rig_track_views(sc)
    rng = np.random.RandomState(31)
    nk = 900                                                     # the candidate keyframe's features hold the scene's first 900 points
    bad = sc["bad"][:nk].copy(); bad[rng.rand(nk) < 0.1] = 1     # (features without a point)
    found = (rng.rand(nk) < 0.1).astype(np.uint8)
    kv, keep2 = views.worldpoints_view(sc["pos"][:nk], sc["normal"][:nk], sc["min_dist"][:nk], sc["max_dist"][:nk], sc["desc"][:nk], sc["n_obs"][:nk], bad, None)
    kf_angle = ((sc["base_angle"][:nk] + rng.randn(nk) * 2.0) % 360.0).astype(np.float32)
    nl = len(sc["kps_left"])
    amp0 = np.where(sc["assigned_mp"][:nl] >= 0, 2 ** 31 - 1, -1).astype(np.int32)
    F = api.Frame().upload(fl, keep[0])
    KP = api.LocalMap().upload(kv)
    m = api.ORBmatcher(0.75, True)
    cam = views.camera_rig(sc["left"]).left
    g = m.SearchByProjectionReloc(F, sc["Tcw"], KP, kf_angle, amp0, th, orb_dist, found, camera=cam)
    o = ob.search_by_projection_reloc_cam(fl, sc["Tcw"], cam, kv, kf_angle, amp0, th, orb_dist, True, found)
    assert o[1] > 150 and g[1] == o[1] and np.array_equal(g[0], o[0])
    pin = views.camera_rig((capi.CAM_PINHOLE, fl.fx, fl.fy, fl.cx, fl.cy)).left
    gp = m.SearchByProjectionReloc(F, sc["Tcw"], KP, kf_angle, amp0, th, orb_dist, found, camera=pin)
    g0 = m.SearchByProjectionReloc(F, sc["Tcw"], KP, kf_angle, amp0, th, orb_dist, found)
    assert gp[1] == g0[1] and np.array_equal(gp[0], g0[0]) and not np.array_equal(g0[0], g[0])


@pytest.mark.parametrize("th,ratio,scale", [(3, 1.5, 1.0), (8, 1.0, 1.4)])
def test_sim3_keyframe_search_with_a_camera_model(th, ratio, scale):
    """Server-side SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th, ratioHamming) on a fisheye keyframe: S/ORBmatcher.cc:515
    projects through pKF->mpCamera -- orbm_search_by_projection_sim3_cam against the oracle; with a PINHOLE handed in as the model it
    is orbm_search_by_projection_sim3 (camera_project = 1)."""
    sc = synth.make_rig_track_scene()
    fl, fr, wv, rig, keep = helpers.rig_track_views(sc)
    rng = np.random.RandomState(33)
    S = sc["Tcw"].copy(); S[:3, :] *= np.float32(scale)
    nl = len(sc["kps_left"])
    matched0 = np.where(rng.rand(nl) < 0.15, 3, -1).astype(np.int32)
    found = (rng.rand(wv.m) < 0.1).astype(np.uint8)
    F = api.Frame().upload(fl, keep[0])
    LM = api.LocalMap().upload(wv)
    m = api.ORBmatcher(0.75, True)
    cam = views.camera_rig(sc["left"]).left
    g = m.SearchByProjectionSim3(F, S, LM, matched0, th, ratio, found, camera=cam)
    o = ob.search_by_projection_sim3_cam(fl, wv, S, cam, matched0, th, ratio, found)
    assert o[1] > 100 and g[1] == o[1] and np.array_equal(g[0], o[0])
    pin = views.camera_rig((capi.CAM_PINHOLE, fl.fx, fl.fy, fl.cx, fl.cy)).left
    gp = m.SearchByProjectionSim3(F, S, LM, matched0, th, ratio, found, camera=pin)
    g0 = m.SearchByProjectionSim3(F, S, LM, matched0, th, ratio, found)
    assert gp[1] == g0[1] and np.array_equal(gp[0], g0[0]) and not np.array_equal(g0[0], g[0])


def test_two_camera_matchers_over_forty_scenes():
    """The two-camera forms over forty seeded scenes (sizes, occupancy, stereo partners, zero-observation points, motion and th vary with the
    seed): isInFrustum flags and levels, SearchByProjection(Frame, MapPoints), SearchByProjection(CurrentFrame, LastFrame), SearchByBoW and
    ComputeStereoFishEyeMatches -- every discrete output equal to the oracle's in every scene (the float32 atan2 / cos / sin / tan of
    KannalaBrandt8 differ in the last place between the device and the host's libm: a window edge or a threshold would have to sit within
    1e-4 px of a projection for a result to flip)."""
    m8, m7 = api.ORBmatcher(0.8, True), api.ORBmatcher(0.7, True)
    bad = []
    for seed in range(int(os.environ.get("ORBG_SWEEP_SCENES", "40"))):
        rng = np.random.RandomState(1000 + seed)
        sc = synth.make_rig_track_scene(n_points=int(rng.randint(300, 1600)), n_distract=int(rng.randint(20, 400)), seed=0x5000 + seed,
                                        stereo_frac=float(rng.uniform(0.1, 0.9)), occupied_frac=float(rng.uniform(0.0, 0.3)), zero_obs_frac=float(rng.uniform(0.0, 0.3)))
        fl, fr, wv, rig, keep = helpers.rig_track_views(sc)
        FL, FR = api.Frame().upload(fl, keep[0]), api.Frame().upload(fr, keep[1])
        g, o = FL.isInFrustumRig(sc["Tcw"], rig, sc["Tlr"], wv), ob.is_in_frustum_rig(fl, sc["Tcw"], rig, sc["Tlr"], wv)
        if not all(np.array_equal(g[s][k], o[s][k]) for s in (0, 1) for k in ("track_in_view", "scale_level")):
            bad.append((seed, "frustum"))
        mv, mvr, keep2 = helpers.rig_mappoint_views(sc, o[0], o[1])
        th = float(rng.choice([1.0, 3.0, 5.0, 15.0])); far = bool(rng.rand() < 0.5)
        a = m8.SearchByProjectionRig(FL, FR, mv, mvr, sc["left_to_right"], sc["right_to_left"], th, far, 6.0, sc["assigned_mp"], sc["assigned_obs"])
        b = ob.search_by_projection_mps_rig(fl, fr, mv, mvr, sc["left_to_right"], sc["right_to_left"], th, far, 6.0, 0.8, sc["assigned_mp"], sc["assigned_obs"])
        if not (a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])):
            bad.append((seed, "search map points"))
        last = synth.rig_last_frame(sc, n_last=min(900, len(sc["pos"])), seed=seed, motion=(0.03, 0.01, float(rng.choice([-0.4, 0.02, 0.4]))))
        lv, keep3 = views.lastframe_view(last["mp_valid"], last["outlier"], last["world_pos"], last["desc"], last["octave"], last["angle"], last["n_obs"], last["Tcw"])
        thf = float(rng.choice([7.0, 15.0]))
        a = m8.SearchByProjectionFrameRig(FL, FR, sc["Tcw"], rig, lv, thf, False, sc["assigned_mp"], sc["assigned_obs"])
        b = ob.search_by_projection_frame_rig(fl, fr, sc["Tcw"], rig, lv, thf, 0, 1, sc["assigned_mp"], sc["assigned_obs"])
        if not (a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])):
            bad.append((seed, "search last frame"))
        fv, keepa, fvF, fvK, kf_desc, kf_angle, valid, keepb = _rig_bow_scene(sc, n_kf=min(900, len(sc["pos"])), seed=seed, shift=int(rng.choice([3, 5])))
        FA = api.Frame().upload(fv, keepa)
        a = m7.SearchByBoWRig(FA, len(sc["kps_left"]), fvF, kf_desc, valid, kf_angle, fvK)
        b = ob.search_by_bow_rig(fv, len(sc["kps_left"]), fvF, kf_desc, valid, kf_angle, fvK, 0.7, True)
        if not (a[1] == b[1] and np.array_equal(a[0], b[0])):
            bad.append((seed, "bow"))
        fs = synth.make_fisheye_stereo_scene(n_stereo=int(rng.randint(100, 900)), n_mono_left=int(rng.randint(0, 300)), n_mono_right=int(rng.randint(0, 300)),
                                             n_distract=int(rng.randint(0, 200)), seed=0x6000 + seed)
        v, keep4 = views.fisheye_stereo_view(fs["kps_left"], fs["desc_left"], fs["mono_left"], fs["kps_right"], fs["desc_right"], fs["mono_right"], fs["left"],
                                             fs["right"], fs["Tlr"], fs["level_sigma2"])
        a, b = api.ComputeStereoFishEyeMatches(v), ob.fisheye_stereo_matches(v)
        if not (a[4] == b[4] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])):
            bad.append((seed, "fisheye stereo"))
    assert not bad, bad
