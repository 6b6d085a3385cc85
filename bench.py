#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric):
"tracking+localBA frames/sec per agent, 640x480 stereo, 1/2/4/8 agents".

One STEP = one frame of one agent through the hot path.  The frame's two images are HOST arrays (what Tracking::GrabImageStereo
holds, S/Tracking.cc:1014-1083): the pipelined Frame constructor packs them into pinned staging and copies them to HBM on
the extractor's stream while the previous frame is tracked (`value`; --device-images gives the accounting with images already
resident in HBM, reported as value_device_images):
  [Frame ctor: extract L+R (pyramid, FAST, quad-tree, angle, rBRIEF) -> ComputeStereoMatches -> feature grid]
  ->  SearchByProjection(cur, last)  ->  SearchLocalPoints (isInFrustum + SearchByProjection over the local map)
and, every FRAMES_PER_KF-th step (a keyframe), one Local Bundle Adjustment plus the upload of the refreshed local map.
By default (--lba-mode async) the LBA runs on the library's own worker thread and HIP stream (lba_solve_async /
lba_wait), concurrently with the frame loop, exactly as the reference runs LocalMapping next to Tracking
(S/ClientSystem.cc:105-106); every LBA triggered in the timed region is waited for before the clock stops.
`--lba-mode inline` gives the serial accounting fps = 1 / (t_frontend + t_LBA / FRAMES_PER_KF).

--config selects the workload: C2 (BASELINE.json configs[1], the metric's configuration, default), C4 (configs[3]'s
single-GPU part: 1280x720, 2000 features, 50-KF local BA) or mono (the monocular agents of configs[4]: host image ->
ORBextractor::operator() with the lapping area of S/Frame.cc:289 -> Frame upload -> the two searches -> mono-edge LBA).

Besides `value` the line carries value_device_images (images already in HBM), value_sync_ctor_host_images (the constructor an
UNCHANGED Tracking thread calls: synchronous, host images), value_with_pose_opt (the two PoseOptimization calls of Tracking
per frame included) and fps_formula (BASELINE.md's 1/(t_frontend + t_LBA/K)), each from a shorter timed region,
p50 / p95 of the per-step times, the measured device-copy bandwidth, and the host CPU.  An internal untimed pre-warm
(chunks of 100 steps until the step rate is stationary: >= 0.3 s, the last three chunks within 1.5 %, at most 3 s; `prewarm_steps`,
`prewarm_s`) precedes --warmup so that a short driver run is at steady state.

Agents shard one per GPU with no data-path collective -> weak scaling; `value` is the
aggregate over all ranks.  Launch for N>1:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
--server-tick adds the server-side exchange of configs[2]/[4] (not part of `value`): RCCL all-gather of KeyFrame wire
blocks -> KeyFrame rebuilt on the device -> SearchByProjection(KF, Scw, map points), microseconds per tick.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAMES_PER_KF = 5
HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md)
FP64_MATRIX_PEAK_TFLOPS = 78.6  # AMD MI355X spec, FP64 matrix (dense); measured issue rate of v_mfma_f64_16x16x4_f64:
#                                 one per 66 cycles per SIMD = 31 FLOP/clk/SIMD (tools/micro/mfma_f64_latency.hip) = 76 TF

CONFIGS = {
    "C2": dict(W=640, H=480, stereo=True, n_features=1000, lba=(20, 10, 2000), mono_frac=0.0, frame_cap=4096, map_cap=32768, local_kfs=20,
               label="C2: 1 client stereo 640x480 synthetic, 1000 ORB feat/frame, 20-KF local BA window "
                     "(20 free + 10 fixed KFs, 2000 points)"),
    "C4": dict(W=1280, H=720, stereo=True, n_features=2000, lba=(50, 20, 8000), mono_frac=0.0, frame_cap=8192, map_cap=131072, local_kfs=50,
               label="C4 (single-GPU part of configs[3]): 1 client stereo 1280x720 synthetic, 2000 ORB feat/frame, 50-KF local BA "
                     "window (50 free + 20 fixed KFs, 8000 points); the VISUAL local BA, i.e. what a stereo-inertial client runs before its IMU is "
                     "initialised (Optimizer::LocalBundleAdjustment; LocalInertialBA and the IMU types are out of SURVEY section 8's scope)"),
    "mono": dict(W=640, H=480, stereo=False, n_features=1000, lba=(20, 10, 2000), mono_frac=1.0, frame_cap=4096, map_cap=32768, local_kfs=20,
                 label="mono agent of configs[4]: 1 client mono 640x480 synthetic (Frame::Frame(mono), S/Frame.cc:260-358: host image -> "
                       "ExtractORB with lapping area {0,1000} -> grid, one fused submission), 1000 ORB feat/frame, 20-KF local BA of monocular edges"),
    "mono_dist": dict(W=640, H=480, stereo=False, n_features=1000, lba=(20, 10, 2000), mono_frac=1.0, frame_cap=4096, map_cap=32768, local_kfs=20,
                      dist=(-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0),
                      label="mono agent with a DISTORTED camera (EuRoC's radial-tangential coefficients, R/ros/conf/EuRoC_mono_client.yaml, on the "
                            "synthetic pinhole images: Frame::UndistortKeyPoints runs on the device inside the fused constructor; the arithmetic and "
                            "the throughput are a distorted camera's, the match counts towards the image edges are lower than a real lens would give), "
                            "640x480, 1000 ORB feat/frame, 20-KF local BA of monocular edges"),
}


CONFIGS["C3"] = dict(CONFIGS["C2"], server=dict(tick_every=2), expect_gpus=2,
                     label="C3 (configs[2]): stereo clients 640x480 as C2, one per GPU, PLUS the server's place recognition in the same job -- every "
                           "2 keyframes all agents all-gather their new KeyFrame wire blocks (RCCL over xGMI), the server (rank 0's GPU, a thread of "
                           "its own) rebuilds each KeyFrame, computes its bag of words, queries DetectNBestCandidates and runs SearchByBoW(KF, KF) + "
                           "SearchByProjection(KF, Scw, map points)")
CONFIGS["C5"] = dict(per_rank=["C2", "mono"], server=dict(tick_every=2), expect_gpus=8,
                     label="C5 (configs[4]): mixed clients, one per GPU -- even ranks stereo 640x480 (as C2), odd ranks monocular 640x480 (as mono) -- "
                           "PLUS the server tick of C3 (RCCL all-gather of every agent's new KeyFrame blocks every 2 keyframes, place recognition "
                           "and KeyFrame matching on rank 0's GPU)")


def build_workload(scene, cfg, n_frames, api, views, synth, device):
    """Run the pipeline once per distinct frame to cache the host-side views a Tracking thread would hold (last-frame view,
    pose guesses) and to build the map a LocalMapping thread would have built: every FRAMES_PER_KF-th frame is a keyframe
    that adds map points for its features with depth that did NOT match a point of the local map of the keyframes before it
    (SearchLocalPoints with the true pose -- what Tracking + LocalMapping::CreateNewMapPoints leave behind).  Not timed."""
    import torch
    cam = scene.cam
    W, H = scene.W, scene.H
    p = scene.frame_view_params()
    ex = api.ORBextractor(cfg["n_features"], 1.2, 8, 20, 7, W, H, n_cams=2 if cfg["stereo"] else 1, device=device)
    rng = np.random.RandomState(1234)
    F = api.Frame(cfg["frame_cap"], device)
    LM = api.LocalMap(cfg["map_cap"], device)
    m_dedupe = api.ORBmatcher(0.8, True, device)
    dist = cfg.get("dist")
    bounds = p["bounds"] if dist is None else api.image_bounds(W, H, p["cam"][:4], dist, device)      # Frame::ComputeImageBounds
    fv0, _keep0 = views.frame_view(np.zeros(1, capi_dtype()), np.zeros((1, 32), np.uint8), None, None, bounds, p["cam"], 8, 1.2)
    frames, imgs, host_imgs = [], [], []
    kf_chunks = {}
    for k in range(n_frames):
        L, R, Tcw = scene.stereo_pair(k)
        host_imgs.append((np.ascontiguousarray(L), np.ascontiguousarray(R) if cfg["stereo"] else None))
        if cfg["stereo"]:
            dL = torch.from_numpy(L).to("cuda:%d" % device)
            dR = torch.from_numpy(R).to("cuda:%d" % device)
            imgs.append((dL, dR))
            nl, nr, kl, dl, ur, dp = ex.frame_stereo_dev(F, fv0, dL.data_ptr(), dR.data_ptr(), W, H, W, float(cam["bf"]), float(cam["b"]),
                                                         download=True)
            kl, dl, ur, dp = kl.copy(), dl.copy(), ur.copy(), dp.copy()
        else:
            dL = torch.from_numpy(L).to("cuda:%d" % device)
            imgs.append((dL, None))
            nl, kraw, kl, dl = ex.frame_mono(F, fv0, L, dist)       # Frame::Frame(mono); kl = mvKeysUn (= mvKeys without distortion)
            dp = scene.depth_at(kraw, Tcw)           # a mono agent's map comes from triangulation; here: scene geometry
            ur = np.full(nl, -1.0, np.float32)
        Pw, valid = synth.unproject_to_world(kl, dp, Tcw, cam)
        lv, keep = views.lastframe_view(valid.astype(np.uint8), np.zeros(nl, np.uint8), Pw, dl, kl["octave"], kl["angle"],
                                        np.full(nl, 3, np.int32), Tcw.astype(np.float32))
        if k % FRAMES_PER_KF == 0:
            dp_new = dp.copy()
            prev = [kf_chunks[j] for j in sorted(kf_chunks) if j >= k - FRAMES_PER_KF * (cfg["local_kfs"] - 1)]
            if prev:
                mp = {key: np.concatenate([c[key] for c in prev]) for key in prev[0]}
                wv, keep_w = views.worldpoints_view(mp["pos"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], mp["n_obs"], mp["bad"])
                LM.upload(wv)
                amp = np.full(nl, -1, np.int32); aob = np.zeros(nl, np.int32)
                amp, aob, nmatch = m_dedupe.SearchLocalPoints(F, LM, Tcw.astype(np.float32), 1.0, False, 0.0, amp, aob, None)
                dp_new[amp >= 0] = -1.0                  # the feature already observes a map point
            kf_chunks[k] = synth.map_from_frame(kl, dl, dp_new, Tcw, cam)
        frames.append(dict(Tcw=Tcw, guess=synth.perturb_pose(Tcw, rng).astype(np.float32), last_view=(lv, keep),
                           n=nl, stereo=int((ur > 0).sum()), kf_feats=(kl.copy(), dl.copy()) if k % FRAMES_PER_KF == 0 else None))
    return ex, imgs, host_imgs, frames, kf_chunks, fv0, bounds


def capi_dtype():
    from multi_orbslam3_amd import _capi
    return _capi.KEYPOINT_DTYPE


class LocalMaps:
    """The local map of the Tracking thread: all map points created by the last `n_kf` DISTINCT keyframes visited (SURVEY.md
    8d: 20 keyframes at C2, 50 at C4).  The flattened views are cached per keyframe set (a SLAM system keeps them as the
    point list of its local map; flattening 5-20 k points in Python inside the timed loop would measure numpy)."""

    def __init__(self, kf_chunks, n_kf, views):
        self.chunks, self.n_kf, self.views = kf_chunks, n_kf, views
        self.recent = []                  # keyframe ids, most recent last, unique
        self.cache = {}
        self.sizes = []

    def prefill(self, seq, first_step):
        """The keyframes the sequence passed before step `first_step` (oldest first)."""
        back, j = [], first_step - 1
        while len(back) < self.n_kf and j > first_step - 1 - 2 * len(seq):
            k = seq[j % len(seq)]
            if k % FRAMES_PER_KF == 0 and k not in back:
                back.append(k)
            j -= 1
        for k in reversed(back):
            self.visit(k)

    def visit(self, kf_id):
        if kf_id in self.recent:
            self.recent.remove(kf_id)
        self.recent.append(kf_id)
        self.recent = self.recent[-self.n_kf:]

    def view(self):
        key = tuple(sorted(self.recent))
        got = self.cache.get(key)
        if got is None:
            parts = [self.chunks[j] for j in key]
            mp = {k2: np.concatenate([c[k2] for c in parts]) for k2 in parts[0]}
            got = self.views.worldpoints_view(mp["pos"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], mp["n_obs"], mp["bad"])
            self.cache[key] = got
            self.sizes.append(int(got[0].m))
        return got[0]


def host_cpu():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return dict(nproc=os.cpu_count(), model=model, affinity_cpus=len(os.sched_getaffinity(0)))


def cpu_baseline(scene, cfg, synth, views, n_frames, host_imgs, frames, kf_chunks, seq, cpus=None, build_native=True):
    """The CPU oracle (a restatement of the reference path), rebuilt -O3 -march=native for this host, on a bounded sample of
    the same workload (same images, same sequence, same local maps) with the reference's threading: left / right extraction
    on two threads (S/Frame.cc:92-95), the tracking steps on the calling thread, local BA on its own thread next to tracking
    (S/ClientSystem.cc:105-106) -- at most 3 busy cores.  cpus: restrict the sample's threads to these CPUs (per-agent
    baselines of a multi-GPU run use disjoint core triples)."""
    import queue
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding as ob
    lib_path = ob.use_native(build=build_native)
    if cpus:
        os.sched_setaffinity(0, cpus)                     # threads created below inherit it
    cam = scene.cam
    nf = cfg["n_features"]
    exL = ob.Extractor(n_features=nf, max_width=scene.W, max_height=scene.H)
    exR = ob.Extractor(n_features=nf, max_width=scene.W, max_height=scene.H)
    p = scene.frame_view_params()
    base_bounds = p["bounds"] if cfg.get("dist") is None else ob.image_bounds(scene.W, scene.H, p["cam"][:4], cfg["dist"])
    rng = np.random.RandomState(1234)
    nfree, nfix, npts = cfg["lba"]
    prob = synth.make_lba_problem(n_free=nfree, n_fixed=nfix, n_points=npts, width=scene.W, height=scene.H, mono_frac=cfg["mono_frac"])
    lp, keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"])
    pool = ThreadPoolExecutor(2)
    q = queue.Queue()
    lba_times = []

    def lba_worker():
        while True:
            job = q.get()
            if job is None:
                return
            t0 = time.perf_counter()
            ob.lba_solve(lp)                              # ctypes releases the GIL: runs next to the frame loop
            lba_times.append(time.perf_counter() - t0)

    worker = threading.Thread(target=lba_worker, daemon=True)
    worker.start()
    maps = LocalMaps(kf_chunks, cfg["local_kfs"], views)
    n_fill = 2 * FRAMES_PER_KF                            # the first frames only fill the last-frame view
    maps.prefill(seq, 0)                                  # the map of the keyframes before the sample's first frame
    wv = maps.view()
    last = None
    t_start = None
    done = 0
    n_map = []
    th_frame, mono = (7.0, False) if cfg["stereo"] else (15.0, True)
    for i in range(n_frames + n_fill):
        k = seq[i % len(seq)]
        L, R = host_imgs[k]
        Tcw = frames[k]["Tcw"]
        if i == n_fill:
            t_start = time.perf_counter()
        if cfg["stereo"]:
            fl, fr = pool.submit(exL.extract, L), pool.submit(exR.extract, R)
            rc, kl, dl, _ = fl.result()
            rc, kr, dr, _ = fr.result()
            ur, dp = ob.stereo_match(exL, exR, kl, dl, kr, dr, float(cam["bf"]), float(cam["b"]))
        else:
            rc, kl, dl, _ = exL.extract(L, (0, 1000))
            ur, dp = None, scene.depth_at(kl, Tcw)
            kl = ob.undistort_keypoints(kl, p["cam"][:4], cfg.get("dist"))           # Frame::UndistortKeyPoints (a copy without distortion)
        fv, keep1 = views.frame_view(kl, dl, ur, dp if cfg["stereo"] else None, base_bounds, p["cam"], 8, 1.2)
        n = len(kl)
        amp = np.full(n, -1, np.int32); aob = np.zeros(n, np.int32)
        guess = synth.perturb_pose(Tcw, rng).astype(np.float32)
        if last is not None:
            amp, aob, nm1 = ob.search_by_projection_frame(fv, guess, last[0], th_frame, mono, True, amp, aob)
            amp, aob, nm2 = ob.search_local_points(fv, wv, guess, 1.0, False, 0.0, 0.8, amp, aob)
        if k % FRAMES_PER_KF == 0:
            maps.visit(k)
        if i % FRAMES_PER_KF == 0:
            wv = maps.view()
            n_map.append(int(wv.m))
            if i >= n_fill:
                q.put(1)
        if i >= n_fill:
            done += 1
        Pw, valid = synth.unproject_to_world(kl, dp, Tcw, cam)
        last = views.lastframe_view(valid.astype(np.uint8), np.zeros(n, np.uint8), Pw, dl, kl["octave"], kl["angle"],
                                    np.full(n, 3, np.int32), Tcw.astype(np.float32))
    q.put(None)
    worker.join()                                         # every LBA triggered by the sample has finished
    wall = time.perf_counter() - t_start
    pool.shutdown()
    fps = done / wall
    native = lib_path.endswith(os.path.join("_native", "liboracle.so"))
    return dict(value=round(fps, 3), unit="frames/s", cores=3 if cfg["stereo"] else 2, kind="port",
                build="g++ -O3 -march=native -ffp-contract=off, built on this host" if native else "g++ -O3 -msse4.2 (portable build)",
                sample="%d frames (%s) + %d local BAs of %.1f ms on their own thread, local map of %d points on average, %.1f s wall; "
                       "threads as in the reference (S/Frame.cc:92-95, S/ClientSystem.cc:105-106)"
                       % (done, "L/R extraction on 2 threads" if cfg["stereo"] else "mono extraction on the tracking thread",
                          len(lba_times), 1e3 * float(np.mean(lba_times)) if lba_times else 0.0, int(np.mean(n_map)) if n_map else 0, wall))


def _bits_equal(a, b):
    """Bit-for-bit equality of two arrays (structured or plain; -0.0 != 0.0, NaN == NaN of the same payload)."""
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def parity_gate(n_gate, first_step, run_step, finish, scene, cfg, views, frames, host_imgs, seq, map_view_of_step, lp, lba_out, th_frame, mono,
                nn_map=0.8, bounds=None):
    """SURVEY.md 8(d) "parity gates run in the same job": the frames the product loop has just been TIMED on, once more through
    the same loop (`run_step(i)` = one step of libagentloop.so -- or of the Python loop -- configured as the main timed region, returning
    what it left on the host: mvKeys / mDescriptors / mvuRight / mvDepth as delivered by the constructor, F.mvpMapPoints after
    SearchByProjection(Current, Last) and after SearchLocalPoints), and the CPU oracle on the same images, pose guesses, last-frame
    views and local maps.  Bit-exact: keypoints (24 B records), descriptors, uRight / depth (f32 bits), both match arrays and match
    counts.  Local BA: the result the loop's last solve delivered against the oracle's solve of the same problem: status, LM
    iterations of both rounds, outlier flags, |pose| / |point| difference <= 1e-4 after the float32 write-back, (lambda, chi2, trials)
    trace within 1e-9 relative.  Outside every timed region.  Returns the `parity` object of the line; ok = every field true / <= 1e-4."""
    from oracle import binding as ob
    W, H = scene.W, scene.H
    cam = scene.cam
    p = scene.frame_view_params()
    stereo = cfg["stereo"]
    dist = cfg.get("dist")
    if dist is not None:                                   # Frame::ComputeImageBounds: the product's bounds (device) against the oracle's
        ob_bounds = ob.image_bounds(W, H, p["cam"][:4], dist)
        if tuple(np.float32(bounds)) != tuple(np.float32(ob_bounds)):
            return dict(ok=False, frames=0, violations=["image bounds %s differ from the oracle's %s" % (tuple(bounds), ob_bounds)])
    bounds = p["bounds"] if bounds is None else bounds
    exL = ob.Extractor(n_features=cfg["n_features"], max_width=W, max_height=H)
    exR = ob.Extractor(n_features=cfg["n_features"], max_width=W, max_height=H) if stereo else None
    res = dict(frames=0, extract_bit_exact=True, stereo_bit_exact=True if stereo else None, match_frame_equal=True, match_map_equal=True,
               keypoints_checked=0, matches_frame_checked=0, matches_map_checked=0, first_step=int(first_step))
    bad = []
    import zlib
    digest = 0
    for i in range(first_step, first_step + n_gate):
        k, k_last = seq[i % len(seq)], seq[(i - 1) % len(seq)]
        got = run_step(i)
        # digest of everything this agent's loop delivered for the frame (multi-agent runs: equal to the 1-agent run of the same seed)
        for key in ("kps", "kps_un", "desc", "uright", "depth", "amp_frame", "amp", "aob"):
            if got.get(key) is not None:
                a_ = np.ascontiguousarray(got[key][: got["nl"]])
                digest = zlib.crc32(a_.tobytes(), digest)
        L, R = host_imgs[k]
        if stereo:
            rc, okl, odl, _ = exL.extract(L)
            rc, okr, odr, _ = exR.extract(R)
            our, odp = ob.stereo_match(exL, exR, okl, odl, okr, odr, float(cam["bf"]), float(cam["b"]))
        else:
            rc, okl, odl, _ = exL.extract(L, (0, 1000))
            our = odp = None
        n = len(okl)
        e_ok = got["nl"] == n and _bits_equal(got["kps"][:n], okl) and _bits_equal(got["desc"][:n], odl)
        if not stereo:
            okl = ob.undistort_keypoints(okl, p["cam"][:4], dist)         # mvKeysUn: what the grid and the searches read
            e_ok = e_ok and _bits_equal(got["kps_un"][:n], okl)
        if stereo:
            e_ok = e_ok and got["nr"] == len(okr)
            s_ok = got["nl"] == n and _bits_equal(got["uright"][:n], our) and _bits_equal(got["depth"][:n], odp)
            res["stereo_bit_exact"] = bool(res["stereo_bit_exact"] and s_ok)
            if not s_ok:
                bad.append("step %d (frame %d): uRight / depth differ from the oracle" % (i, k))
        res["extract_bit_exact"] = bool(res["extract_bit_exact"] and e_ok)
        if not e_ok:
            bad.append("step %d (frame %d): keypoints / descriptors differ from the oracle (%d vs %d left)" % (i, k, got["nl"], n))
            continue                                      # the searches of a frame with other features cannot agree
        fv, keep1 = views.frame_view(okl, odl, our, odp, bounds, p["cam"], 8, 1.2)
        guess = frames[k]["guess"]
        amp = np.full(n, -1, np.int32); aob = np.zeros(n, np.int32)
        amp1, aob1, n1 = ob.search_by_projection_frame(fv, guess, frames[k_last]["last_view"][0], th_frame, mono, True, amp, aob)
        f_ok = got["n1"] == n1 and np.array_equal(got["amp_frame"][:n], amp1)
        amp2, aob2, n2 = ob.search_local_points(fv, map_view_of_step(i), guess, 1.0, False, 0.0, nn_map, amp1, aob1)
        m_ok = got["n2"] == n2 and np.array_equal(got["amp"][:n], amp2) and np.array_equal(got["aob"][:n], aob2)
        res["match_frame_equal"] = bool(res["match_frame_equal"] and f_ok)
        res["match_map_equal"] = bool(res["match_map_equal"] and m_ok)
        if not f_ok:
            bad.append("step %d (frame %d): SearchByProjection(Current, Last) %d matches vs oracle %d" % (i, k, got["n1"], n1))
        if not m_ok:
            bad.append("step %d (frame %d): SearchLocalPoints %d matches vs oracle %d" % (i, k, got["n2"], n2))
        res["frames"] += 1; res["keypoints_checked"] += n + (len(okr) if stereo else 0)
        res["matches_frame_checked"] += int(n1); res["matches_map_checked"] += int(n2)
    # ---- the local BA the loop ran last (every keyframe step solves the same problem) against the oracle's solve of it
    finish()                                              # every local BA the gate's steps submitted has delivered its result
    g = lba_out
    digest = zlib.crc32(np.ascontiguousarray(g.points).tobytes(), zlib.crc32(np.ascontiguousarray(g.poses).tobytes(), digest))
    res["agent_digest"] = int(digest)
    o = ob.lba_solve(lp)
    d_pose = float(np.abs(g.poses - o.poses).max()); d_pt = float(np.abs(g.points - o.points).max())
    tg, to = g.trace_rows(), o.trace_rows()
    tr_ok = tg.shape == to.shape and bool(np.array_equal(tg[:, 2], to[:, 2]))
    tr_rel = float(np.max(np.abs(tg[:, :2] - to[:, :2]) / np.maximum(np.abs(to[:, :2]), 1e-300))) if tr_ok and len(to) else (0.0 if tr_ok else float("inf"))
    res.update(lba_max_abs=max(d_pose, d_pt), lba_pose_max_abs=d_pose, lba_point_max_abs=d_pt,
               lba_iters_equal=bool(g.status == o.status and tuple(g.iters) == tuple(o.iters)), lba_iters=[int(x) for x in g.iters],
               lba_outliers_equal=bool(np.array_equal(g.edge_outlier, o.edge_outlier) and g.n_outliers == o.n_outliers),
               lba_trace_trials_equal=tr_ok, lba_trace_max_rel=tr_rel, lba_edges=int(lp.n_edges))
    if not res["lba_iters_equal"]:
        bad.append("local BA: status / iterations %s %s vs oracle %s %s" % (g.status, tuple(g.iters), o.status, tuple(o.iters)))
    if not (res["lba_max_abs"] <= 1e-4):
        bad.append("local BA: |pose| %.3g |point| %.3g above 1e-4" % (d_pose, d_pt))
    if not res["lba_outliers_equal"]:
        bad.append("local BA: outlier sets differ")
    if not (tr_ok and tr_rel <= 1e-9):
        bad.append("local BA: (lambda, chi2, trials) trace differs (max rel %.3g)" % tr_rel)
    res["ok"] = not bad
    res["violations"] = bad[:8]
    res["oracle"] = "oracle/ (CPU restatement of the reference path; parity unpinned by the reference, see DESIGN.md section 5)"
    return res


def server_tick_parity(server, views):
    """The in-job gate, server side (rank 0): the LAST tick's blocks through the oracle -- wire unpack, bag of words,
    DetectNBestCandidates (from the score state the tick started with), SearchByBoW(KF, KF), SearchByProjection(KF, Scw, points) --
    against what the device produced inside the timed region."""
    from oracle import binding as ob
    if server is None or not server.is_server or not server.last:
        return None
    fv = server.fv
    ok = dict(blocks=0, wire_equal=True, bow_equal=True, candidates_equal=True, bow_matches_equal=True, projection_matches_equal=True)
    for r in server.last:
        n = r["n"]
        wire = r["wire"] if r["wire"] is not None else r["blk"].cpu().numpy()
        kps, desc = ob.wire_unpack(np.ascontiguousarray(wire), n)
        server.KF.from_wire(fv, wire=np.ascontiguousarray(wire), n=n)
        gk, gd = server.KF.download()
        ok["wire_equal"] &= bool(gk.tobytes() == kps.tobytes() and gd.tobytes() == desc.tobytes())
        (bw, bv), (fn, fs, ff) = ob.vocab_bow(server.vocab_view, desc, server.LEVELS_UP)
        ok["bow_equal"] &= bool(np.array_equal(bw, r["bow"][0]) and np.array_equal(bv, r["bow"][1]) and np.array_equal(fn, r["fv"][0])
                                and np.array_equal(fs, r["fv"][1]) and np.array_equal(ff, r["fv"][2]))
        score = r["before"].copy()
        ol, om = ob.detect_n_best_candidates(server.db_view, bw, bv, r["con"], r["agent"], 3, score)
        ok["candidates_equal"] &= bool(np.array_equal(ol, r["loop"]) and np.array_equal(om, r["merge"]))
        c = server.db_kfs[r["cand"]]
        fvk, keepk = views.frame_view(kps, desc, None, None, (fv.min_x, fv.max_x, fv.min_y, fv.max_y), (fv.fx, fv.fy, fv.cx, fv.cy, fv.bf, fv.b),
                                      fv.n_levels, fv.scale_factor)
        fvK, k1 = views.featvec_view(fn, fs, ff)
        fv1, k2 = views.featvec_view(*c["fv"])
        om12, onb = ob.search_by_bow_kf(fvk, fvK, np.ones(n, np.uint8), c["desc"], c["valid"], c["angle"], fv1, 0.9, True)
        ok["bow_matches_equal"] &= bool(onb == r["nb"] and np.array_equal(om12, r["m12"]))
        omt, onp = ob.search_by_projection_sim3(fvk, server.map_view, r["T"], server.free[:n], 8, 1.5)
        ok["projection_matches_equal"] &= bool(onp == r["nproj"] and np.array_equal(omt, r["matched"]))
        ok["blocks"] += 1
    ok["ok"] = all(v for k, v in ok.items() if k != "blocks") and ok["blocks"] > 0
    return ok


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this same command (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, as torch.distributed.run sets them), rank 0 inheriting stdout -- it prints the ONE line.  The parent
    never loads the HIP runtime: it counts the GPUs in the KFD topology under /sys (harness.count_gpus_without_runtime; only where
    sysfs does not show it does it fall back to torch.cuda.device_count()).  Returns the exit code: 0 when every rank returned 0;
    2 when the node has fewer GPUs than ranks (ORBG_BENCH_SHARE_GPU=1, a test aid, lets ranks share the GPUs that exist)."""
    import socket
    import subprocess
    from multi_orbslam3_amd import harness as _h
    n_dev = _h.count_gpus_without_runtime()
    if n_dev is None:
        import torch
        n_dev = torch.cuda.device_count()
    if n_dev < n and os.environ.get("ORBG_BENCH_SHARE_GPU") != "1":
        sys.stderr.write("bench.py: --gpus %d but this node has %d GPU%s (one agent per GPU; ORBG_BENCH_SHARE_GPU=1 shares them for a dry run)\n"
                         % (n, n_dev, "" if n_dev == 1 else "s"))
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = list(range(n))
    while pending:
        for r in list(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.remove(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, code))
                for q in pending:
                    procs[q].terminate()                   # (exactly the processes started above)
        time.sleep(0.05)
    return rc


def copy_bandwidth_gbs(torch, device, mib=512, reps=10):
    """Device-to-device copy of `mib` MiB (read + write counted): the second denominator SURVEY.md 8(d) asks for."""
    a = torch.empty(mib << 20, dtype=torch.uint8, device="cuda:%d" % device)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * (mib << 20) * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C2")
    ap.add_argument("--frames", type=int, default=100,
                    help="distinct synthetic frames; the sequence is their ping-pong, 2 * frames - 2 steps long (SURVEY.md 8d: 200 frames)")
    ap.add_argument("--prewarm-steps", type=int, default=200,
                    help="untimed steps before --warmup (at least this many; the pre-warm runs until the step rate is stationary, "
                         "0.3 - 3 s): first-use allocations, clocks, hardware queues -- so that a short timed region is at steady state")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--first-agent", type=int, default=0,
                    help="agent id of rank 0 (rank r runs agent first-agent + r: its own scene / local-BA seeds); a 1-rank run with --first-agent a "
                         "reproduces agent a of a multi-rank run (tests compare parity.agent_digests)")
    ap.add_argument("--parity-frames", type=int, default=20,
                    help="in-job parity gate (SURVEY.md 8d): this many frames of the timed sequence go through the product loop once more, "
                         "outside the timed region, and every output is compared with the CPU oracle; the process exits with code 3 on a "
                         "violation (0: no gate, the line then carries parity: null)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the value_host_images / value_with_pose_opt regions")
    ap.add_argument("--secondary-steps", type=int, default=300)
    ap.add_argument("--separate-calls", action="store_true",
                    help="call extract / ComputeStereoMatches / grid as three entry points instead of the fused "
                         "Frame-constructor entry point orbx_frame_stereo_dev")
    ap.add_argument("--pose-opt", action="store_true",
                    help="also run Optimizer::PoseOptimization (SURVEY row f-2) after each of the two searches in the MAIN "
                         "timed region; by default that accounting is reported as value_with_pose_opt")
    ap.add_argument("--device-images", action="store_true",
                    help="MAIN timed region with the images already resident in HBM (no host staging / PCIe copy); by default that "
                         "accounting is reported as value_device_images")
    ap.add_argument("--loop", choices=["cxx", "python"], default="cxx",
                    help="cxx: the Tracking-thread loop is libagentloop.so (multi_orbslam3_amd/csrc/agent_loop.cpp: C++ above the C-ABI, as the "
                         "reference's Tracking.cc is); python: the same steps from this script through the ctypes wrappers (mono / "
                         "--separate-calls / --profile-stages always use it)")
    ap.add_argument("--submit-order", choices=["before-wait", "after-wait"], default="before-wait",
                    help="pipelined constructor: hand frame t+1 over before or after frame t's constructor is collected")
    ap.add_argument("--ctor-ahead", type=int, default=1, choices=[1, 2, 3],
                    help="pipelined constructor: frames handed over ahead of the one being tracked (ring of N + 1 extractor handles / frame "
                         "objects; the python loop supports 1)")
    ap.add_argument("--no-host-features", action="store_true",
                    help="pipelined constructor: leave the features in HBM (rounds 1-3); by default every frame's keypoints, descriptors, "
                         "uRight and depth are delivered into host arrays when the constructor is collected (orbx_set_frame_outputs)")
    ap.add_argument("--ingest", choices=["thread", "inline"], default="thread",
                    help="host images: thread = the library's ingest thread packs the rows into pinned staging and enqueues the "
                         "constructor (orbx_frame_stereo_submit, ORBX_SUBMIT_ASYNC); inline = the tracking thread does")
    ap.add_argument("--last-frame-view", choices=["resident", "inplace"], default="resident",
                    help="SearchByProjection(Current, Last): resident = mLastFrame's view is uploaded when the tracking of that frame ends "
                         "(orbm_lastview_upload, inside the step) and the next frame's search reads it from HBM; inplace = the search kernel "
                         "reads the view from pinned host memory over PCIe (the form of rounds 1-3)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the main K-step region is run this many times back to back; value = the FIRST one, value_min / _median / _max "
                         "over all of them are reported next to it (1: no repeats)")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip tests/cpp/dropin_bench (the per-frame path and the local BA timed through the reference-signature glue "
                         "over mock Frame / KeyFrame / MapPoint objects; reported as value_dropin)")
    ap.add_argument("--closed-loop-frames", type=int, default=200,
                    help="in-job closed-loop parity (tests/cpp/closed_loop, rank 0, config C2): this many frames of a stereo agent through the "
                         "reference-signature glue, product and oracle each feeding on their own outputs, every product call shadowed on "
                         "the oracle; reported as parity.closed_loop, a violation fails the job like the open-loop gate (0: skip)")
    ap.add_argument("--profile-stages", action="store_true", help="bracket every extractor stage with HIP events")
    ap.add_argument("--no-numa-pin", action="store_true",
                    help="do not restrict the process to the CPUs of the GPU's NUMA node (default: like numactl --cpunodebind)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="construct every frame synchronously before it is tracked; by default frame t+1's constructor "
                         "(orbx_frame_stereo_dev_submit on a second extractor handle) runs while frame t is tracked")
    ap.add_argument("--lba-mode", choices=["async", "inline"], default="async",
                    help="async: LBA runs on the library's worker thread + its own HIP stream concurrently with tracking, as the "
                         "reference's LocalMapping thread does (S/ClientSystem.cc:105-106); inline: LBA blocks the frame loop")
    ap.add_argument("--server-tick", action="store_true",
                    help="also time the server-side exchange (RCCL all-gather of KF wire blocks -> rebuilt KeyFrame -> "
                         "SearchByProjection(KF, Scw, map)) for 2 and 8 KF blocks; reported under config.server_tick")
    args = ap.parse_args()
    # ---- N > 1 launched the way the driver launches N = 1 (`python bench.py --gpus N`, no RANK in the environment): this process
    # starts the N rank processes itself -- fresh children, decided BEFORE anything here touches the GPU -- and relays their exit code
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank%s (WORLD_SIZE): refusing to print a line whose n_gpus is not "
                         "what was asked for\n" % (args.gpus, world_env, "" if world_env == 1 else "s"))
        sys.exit(2)
    cfg = CONFIGS[args.config]
    server_cfg = cfg.get("server")
    if "per_rank" in cfg:
        # mixed clients: the configuration of THIS rank's agent
        sub = cfg["per_rank"][int(os.environ.get("RANK", "0")) % len(cfg["per_rank"])]
        cfg = dict(CONFIGS[sub], server=server_cfg, label=cfg["label"] + " -- this rank: " + sub, agent_kind=sub)
    stereo = cfg["stereo"]

    # The agent keeps six HIP streams busy (two extractor handles, two frames, the local map, the local BA).  The ROCm
    # runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with 4, the search kernels sometimes
    # share a queue with the local BA's chain; with 6 every stream has its own; 8 and 12 are slower again.  Must be set
    # before the runtime initialises.  (Without the pipelined constructor the agent has three busy streams and the default
    # of 4 is the good setting -- more hardware queues than busy streams cost dispatch latency on every one of them.)
    pipeline = not (args.no_pipeline or args.separate_calls)
    host_images = not args.device_images and not args.separate_calls
    # Measured this round (3 / 4 / 5 / 6 / 8 queues: 5550 / 8290 / 8270 / 8240 / 4380 frames/s): the runtime's default of 4 is as good
    # as 5 or 6 now, and the server tick (--server-tick) needs it (with 6 its streams share queues erratically: 8 blocks 354 us vs 1470)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
    # Every agent keeps up to three threads spinning on completion words (tracking thread, local-BA worker, image-ingest thread).  If
    # the container's CPU quota (cgroup cpu.max) cannot feed that for all ranks of this node, the policy is per thread ROLE
    # (orbg_set_wait_policy / ORBG_NO_POLL, include/orbgpu.h "Host threads"): the tracking thread keeps spinning -- it is on the frame's
    # critical path -- and the local-BA worker and the ingest thread block (runtime waits / condition variables: 6-10 us more latency
    # per wait, a fraction of a CPU each); only below one CPU per rank does the tracking thread block too.
    wait_policy = {"caller": "spin", "lba": "spin", "ingest": "spin", "reason": "no cpu quota (or quota >= 3 cpus per rank)"}
    ranks_here = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if os.environ.get("ORBG_NO_POLL") is None:
        try:
            with open("/sys/fs/cgroup/cpu.max") as f:
                q, per = f.read().split()
            if q != "max":
                cpus = float(q) / float(per)
                if cpus < 1.0 * ranks_here:
                    os.environ["ORBG_NO_POLL"] = "all"
                elif cpus < 3.0 * ranks_here:
                    os.environ["ORBG_NO_POLL"] = "lba,ingest"
                wait_policy["reason"] = "cpu quota %.1f for %d rank%s" % (cpus, ranks_here, "" if ranks_here == 1 else "s")
        except (OSError, ValueError):
            pass
    else:
        wait_policy["reason"] = "ORBG_NO_POLL=%s in the environment" % os.environ["ORBG_NO_POLL"]
    np_env = os.environ.get("ORBG_NO_POLL")
    if np_env is not None:
        every = np_env in ("", "1", "all") or not any(t in np_env for t in ("caller", "lba", "ingest"))
        for role in ("caller", "lba", "ingest"):
            if every or role in np_env:
                wait_policy[role] = "block"
    wait_mode = "polling" if all(wait_policy[r] == "spin" for r in ("caller", "lba", "ingest")) else \
        "per role: tracking thread %ss, local-BA worker %ss, ingest thread %ss (%s)" % (wait_policy["caller"], wait_policy["lba"], wait_policy["ingest"],
                                                                                       wait_policy["reason"])
    import torch
    from multi_orbslam3_amd import _capi as capi
    from multi_orbslam3_amd import api, harness, synth, views

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    # --server-tick on one GPU still goes through RCCL: a single-rank process group
    # control plane on gloo, the RCCL group of the server tick is created by its first collective (harness.AgentGroup)
    grp = harness.AgentGroup("nccl", force_group=args.server_tick or server_cfg is not None, device_index=(
        int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1) if os.environ.get("ORBG_BENCH_SHARE_GPU") == "1" else None))
    rank, local_rank, world = grp.rank, grp.local_rank, grp.world
    device = local_rank
    if os.environ.get("ORBG_BENCH_SHARE_GPU") == "1":
        # test aid: several ranks on the GPUs that exist (a 2-rank launch on a 1-GPU box exercises the multi-rank code path of this
        # file -- barriers, MAX over ranks, per-agent statistics; the rate it prints says nothing)
        device = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device)
    affinity_at_start = os.sched_getaffinity(0)
    cpu_affinity = None if args.no_numa_pin else harness.pin_to_gpu_numa_node(device)
    core_pair = None if args.no_numa_pin else harness.cores_for_agent(device, local_rank, per_agent=3)

    W, H = cfg["W"], cfg["H"]
    agent_id = args.first_agent + rank
    scene = synth.Scene(W, H, seed=synth.SEED_IMAGES + agent_id)      # one agent per GPU, distinct seeds
    cam = scene.cam
    ex, imgs, host_imgs, frames, kf_chunks, fv, frame_bounds = build_workload(scene, cfg, args.frames, api, views, synth, device)
    p = scene.frame_view_params()
    dist = cfg.get("dist")
    dist_c = None if dist is None else capi.OrbxDistortion(*[float(v) for v in dist])
    F = api.Frame(cfg["frame_cap"], device)
    # frame t+1 is constructed (second extractor handle, second frame object) while frame t is tracked
    ctor_ahead = args.ctor_ahead if (args.loop == "cxx" and not args.separate_calls and not args.profile_stages) else 1
    n_ring = 2 if ctor_ahead == 1 else 4        # an even ring: consecutive frames alternate between the two extractor streams
    exs = [ex] + [api.ORBextractor(cfg["n_features"], 1.2, 8, 20, 7, W, H, n_cams=2 if stereo else 1, device=device) for _ in range(n_ring - 1)] if pipeline else [ex]
    Fs = [F] + [api.Frame(cfg["frame_cap"], device) for _ in range(n_ring - 1)] if pipeline else [F]
    in_flight = [False] * 4
    # the pipelined constructor delivers mvKeys / mDescriptors / mvuRight / mvDepth into host arrays at every _wait (orbx_set_frame_outputs):
    # what an unchanged Tracking / KeyFrame / Communicator reads of a frame is on the host inside the timed step
    feature_outputs = [e.set_frame_outputs(cfg["frame_cap"]) for e in exs] if (pipeline and not args.no_host_features) else None
    LM = api.LocalMap(cfg["map_cap"], device)
    m_frame = api.ORBmatcher(0.9, True, device)
    m_map = api.ORBmatcher(0.8, True, device)
    opt = api.Optimizer(device)
    nfree, nfix, npts = cfg["lba"]
    prob = synth.make_lba_problem(n_free=nfree, n_fixed=nfix, n_points=npts, seed=synth.SEED_LBA + agent_id, width=W, height=H,
                                  mono_frac=cfg["mono_frac"])
    lp, lp_keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"], device=device)
    bf, bb = float(cam["bf"]), float(cam["b"])
    if core_pair is not None and args.lba_mode == "async":
        # the library's local-BA worker inherits the affinity of the thread that makes the first asynchronous call
        os.sched_setaffinity(0, core_pair[1])
        opt.LocalBundleAdjustmentAsync(lp, views.LbaOutput(lp.n_poses, lp.n_points, lp.n_edges))
        opt.wait()
        os.sched_setaffinity(0, core_pair[0])
        cpu_affinity = "%s; tracking thread on cpus %s, local-BA worker on cpus %s" % (cpu_affinity, sorted(core_pair[0]), sorted(core_pair[1]))
    if pipeline and args.ingest == "thread":
        # the library's ingest thread is created by the first asynchronous submission and inherits that caller's affinity
        if core_pair is not None and len(core_pair) >= 3:
            os.sched_setaffinity(0, core_pair[2])
        if stereo:
            exs[1].frame_stereo_submit(Fs[1], fv, host_imgs[0][0], host_imgs[0][1], bf, bb, async_ingest=True)
        else:
            exs[1].frame_mono_submit(Fs[1], fv, host_imgs[0][0], dist_c, async_ingest=True)
        exs[1].frame_stereo_dev_wait()
        if core_pair is not None:
            os.sched_setaffinity(0, core_pair[0])
            if len(core_pair) >= 3:
                cpu_affinity = "%s, image-ingest thread on cpus %s" % (cpu_affinity, sorted(core_pair[2]))
    nF = len(frames)
    seq = list(range(nF)) + list(range(nF - 2, 0, -1))             # ping-pong: consecutive frames stay adjacent
    po_prob = synth.make_pose_opt_problem(n=450, seed=77 + agent_id)
    po1, po1_keep = views.pose_opt_problem(po_prob["Xw"], po_prob["u"], po_prob["v"], po_prob["ur"], po_prob["inv_sigma2"],
                                          po_prob["cam"], po_prob["Tcw"], device=device)
    po_prob2 = synth.make_pose_opt_problem(n=650, seed=78 + agent_id)
    po2, po2_keep = views.pose_opt_problem(po_prob2["Xw"], po_prob2["u"], po_prob2["v"], po_prob2["ur"], po_prob2["inv_sigma2"],
                                          po_prob2["cam"], po_prob2["Tcw"], device=device)
    for e in exs:
        e.set_profiling(2 if args.profile_stages else 1)
    ev_overhead_ms = ex.event_overhead_ms(100)
    lba_ev_overhead_ms = opt.event_overhead_ms(100)
    opt.set_profiling(True, reset=True)
    FAST_BRACKET_EVERY = 1 if args.profile_stages else 4         # the event pair costs ~5 us of stream time: sample every 4th frame
    th_frame, mono_flag = (7.0, False) if stereo else (15.0, True)

    lba_out = views.LbaOutput(lp.n_poses, lp.n_points, lp.n_edges)      # result arrays are allocated once, like a SLAM system would
    amp_buf = np.full(2 * cfg["frame_cap"], -1, np.int32); aob_buf = np.zeros(2 * cfg["frame_cap"], np.int32)

    class Region:
        """Accumulators of one timed region."""
        def __init__(self, n):
            self.stage = dict(extract=0.0, stereo=0.0, grid=0.0, match_frame=0.0, match_map=0.0, lba=0.0, map_upload=0.0, pose_opt=0.0)
            self.kern = dict(fast_kernel_ms=0.0, octree_host_ms=0.0)
            if args.profile_stages:
                self.kern.update(pyramid_ms=0.0, fast_ms=0.0, desc_ms=0.0, stereo_ms=0.0)
            self.stats = dict(kp=0, stereo=0, m_frame=0, m_map=0, lba_iters=0, lba_calls=0, lba_s=0.0)
            self.step_s = np.zeros(max(n, 1))
            self.async_t0 = None
            self.async_timed = False
            self.sync_tail_s = 0.0                            # time spent inside the closing torch.cuda.synchronize()
            self.tail = (0.0, 0.0, 0.0)
            self.worst_step = None                           # (ms, [extract, match_frame, match_map, pose_opt, map_upload, lba, last_view] us, index in the region) of the slowest step
            self.timeline = np.zeros((0, 5), np.float32)     # per constructor: queue / pack / enqueue / wait / latency [us] (orbx_get_ctor_timeline)

    def collect_async(reg):
        if reg.async_t0 is None:
            return
        out = opt.wait()
        if reg.async_timed:
            reg.stats["lba_iters"] += sum(out.iters); reg.stats["lba_calls"] += 1
            reg.stats["lba_s"] += opt.last_solve_ms * 1e-3
        reg.async_t0 = None

    maps = LocalMaps(kf_chunks, cfg["local_kfs"], views)
    maps.prefill(seq, 0)
    ingest_async = args.ingest == "thread"
    submit_first = args.submit_order == "before-wait"

    def submit_ctor(c, k_img, host_images):
        if not stereo:
            if host_images:
                exs[c].frame_mono_submit(Fs[c], fv, host_imgs[k_img][0], dist_c, async_ingest=ingest_async)
            else:
                exs[c].frame_mono_submit(Fs[c], fv, None, dist_c, device_ptr=imgs[k_img][0].data_ptr(), size=(W, H, W))
        elif host_images:
            exs[c].frame_stereo_submit(Fs[c], fv, host_imgs[k_img][0], host_imgs[k_img][1], bf, bb, async_ingest=ingest_async)
        else:
            exs[c].frame_stereo_dev_submit(Fs[c], fv, imgs[k_img][0].data_ptr(), imgs[k_img][1].data_ptr(), W, H, W, bf, bb)

    current_map_view = [None]                     # python loop: the view the last LM.upload() made resident

    def step(i, reg, timed, pose_opt, host_images, pipelined, slot=None, last=False, capture=None):
        k, k_last = seq[i % len(seq)], seq[(i - 1) % len(seq)]
        fr = frames[k]
        dL, dR = imgs[k]
        t0 = time.perf_counter()
        Fc, exc = F, ex
        kl = kun = dl = None
        if not stereo and not pipelined:
            # mono agent: Frame::Frame(mono) = ExtractORB(0, im, 0, 1000) (S/Frame.cc:289) + UndistortKeyPoints + grid, one fused submission
            if host_images:
                nl, kl, kun, dl = ex.frame_mono(F, fv, host_imgs[k][0], dist_c)
            else:
                nl, kl, kun, dl = ex.frame_mono(F, fv, None, dist_c, device_ptr=imgs[k][0].data_ptr(), size=(W, H, W))
            nr = 0
            t1 = t2 = time.perf_counter()
        elif pipelined:
            c = i & 1
            Fc, exc = Fs[c], exs[c]
            if not in_flight[c]:                          # first step only: nothing was submitted ahead
                submit_ctor(c, k, host_images)
            if submit_first and not last:
                # Frame::Frame(t+1) is handed over (other handle, other frame object) BEFORE frame t is collected: its staging
                # copy and launches overlap the tail of frame t's constructor; the image pair t+1 is needed at the same
                # moment either way (the start of step t)
                submit_ctor(c ^ 1, seq[(i + 1) % len(seq)], host_images)
                in_flight[c ^ 1] = True
            nl, nr = exc.frame_stereo_dev_wait()
            in_flight[c] = False
            if not submit_first and not last:
                submit_ctor(c ^ 1, seq[(i + 1) % len(seq)], host_images)   # Frame::Frame(t+1) runs during the tracking of frame t
                in_flight[c ^ 1] = True
            t1 = t2 = time.perf_counter()
        elif host_images:
            # the reference's constructor takes host images (cv::Mat): the two H2D copies are inside the step
            nl, nr = ex.frame_stereo(F, fv, host_imgs[k][0], host_imgs[k][1], bf, bb, download=False)
            t1 = t2 = time.perf_counter()
        elif args.separate_calls:
            nl, nr = ex.extract_stereo_dev(dL.data_ptr(), dR.data_ptr(), W, H, W)[:2]
            t1 = time.perf_counter()
            ex.ComputeStereoMatches(bf, bb, download=False)
            t2 = time.perf_counter()
            F.from_extractor(ex, fv, nl)
        else:
            # Frame::Frame(stereo): extraction + ComputeStereoMatches + grid in ONE submission / ONE final sync
            nl, nr = ex.frame_stereo_dev(F, fv, dL.data_ptr(), dR.data_ptr(), W, H, W, bf, bb)
            t1 = t2 = time.perf_counter()
        t3 = time.perf_counter()
        amp = amp_buf[:nl]; aob = aob_buf[:nl]
        amp.fill(-1); aob.fill(0)                         # F.mvpMapPoints starts empty (S/Frame.cc:113)
        amp, aob, n1 = m_frame.SearchByProjectionFrame(Fc, fr["guess"], frames[k_last]["last_view"][0], th_frame, mono_flag, amp, aob, inplace=True)
        t4 = time.perf_counter()
        if capture is not None:
            capture["amp_frame"] = amp.copy()
        amp, aob, n2 = m_map.SearchLocalPoints(Fc, LM, fr["guess"], 1.0, False, 0.0, amp, aob, None, inplace=True)
        t5 = time.perf_counter()
        if capture is not None:
            capture.update(nl=nl, nr=nr, n1=n1, n2=n2, amp=amp.copy(), aob=aob.copy(), map_view=current_map_view[0])
            if kl is not None:
                capture.update(kps=kl, kps_un=kun, desc=dl)
            elif feature_outputs is not None and pipelined:
                o_ = feature_outputs[i & 1]
                capture.update(kps=o_["kps"], kps_un=o_["kps_un"], desc=o_["desc"], uright=o_["uright"], depth=o_["depth"])
        if pose_opt:
            # TrackWithMotionModel / TrackLocalMap call PoseOptimization after each search (S/Tracking.cc:2649,2712);
            # ~450 and ~650 correspondences as the two searches produce here
            opt.PoseOptimization(po1)
            opt.PoseOptimization(po2)
            tpo = time.perf_counter()
        else:
            tpo = t5
        t6 = t7 = tpo
        if k % FRAMES_PER_KF == 0:
            maps.visit(k)                              # this frame's map points joined the map when it was a keyframe
        if i % FRAMES_PER_KF == 0:
            current_map_view[0] = maps.view()
            LM.upload(current_map_view[0])             # Tracking::UpdateLocalMap: points of the last 20 / 50 keyframes
            t6 = time.perf_counter()
            if args.lba_mode == "async":
                collect_async(reg)                     # the previous keyframe's LBA (long finished in steady state)
                reg.async_t0, reg.async_timed = time.perf_counter(), timed
                opt.LocalBundleAdjustmentAsync(lp, lba_out)
                t7 = time.perf_counter()
            else:
                out = opt.LocalBundleAdjustment(lp, out=lba_out)
                t7 = time.perf_counter()
                if timed:
                    reg.stats["lba_iters"] += sum(out.iters); reg.stats["lba_calls"] += 1; reg.stats["lba_s"] += t7 - t6
        if timed:
            st = reg.stage
            for key, dt in (("extract", t1 - t0), ("stereo", t2 - t1), ("grid", t3 - t2), ("match_frame", t4 - t3),
                            ("match_map", t5 - t4), ("pose_opt", tpo - t5), ("map_upload", t6 - tpo), ("lba", t7 - t6)):
                st[key] += dt
            if args.profile_stages:
                tm = exc.timings()
                for key in reg.kern:
                    reg.kern[key] += tm[key]
            reg.stats["kp"] += nl + nr; reg.stats["m_frame"] += n1; reg.stats["m_map"] += n2
            if slot is not None:
                reg.step_s[slot] = time.perf_counter() - t0

    # local map must exist before the first frame
    current_map_view[0] = maps.view()
    LM.upload(current_map_view[0])

    # ---- the same loop in C++ (libagentloop.so): the per-frame host work of a client is C++ in the reference (Tracking.cc)
    use_cxx = args.loop == "cxx" and not args.separate_calls and not args.profile_stages
    loop = None
    if use_cxx:
        from multi_orbslam3_amd import agent as agent_mod
        # the local map uploaded at keyframe step i is periodic in i (sequence length x frames per keyframe): one view per keyframe
        # step of the period, taken from the second pass (the first one starts from the prefill)
        period = int(np.lcm(len(seq), FRAMES_PER_KF))
        sim = LocalMaps(kf_chunks, cfg["local_kfs"], views)
        sim.prefill(seq, 0)
        kf_views = []
        for i in range(2 * period):
            kk = seq[i % len(seq)]
            if kk % FRAMES_PER_KF == 0:
                sim.visit(kk)
            if i % FRAMES_PER_KF == 0 and i >= period:
                kf_views.append(sim.view())
        maps.sizes = list(sim.sizes)
        frames_in = [dict(host=host_imgs[k], dev=(imgs[k][0].data_ptr(), imgs[k][1].data_ptr() if stereo else None),
                          guess=np.ascontiguousarray(frames[k]["guess"], np.float32).reshape(16), last_view=frames[k]["last_view"][0])
                     for k in range(nF)]
        last_dev = api.LastFrameOnDevice(2 * cfg["frame_cap"], device) if args.last_frame_view == "resident" else None
        loop = agent_mod.AgentLoop(exs, Fs, LM, opt, fv, W, H, W, bf, bb, frames_in, seq, kf_views, lp, lba_out, [po1, po2], FRAMES_PER_KF,
                                   2 * cfg["frame_cap"], th_frame, mono_flag, last_view=last_dev, mono_agent=not stereo, dist=dist_c)

    # ---- C3 / C5: the server's place recognition runs in this job (harness.ServerTick; a collective constructor)
    server = None
    if server_cfg is not None:
        if not use_cxx:
            raise SystemExit("bench.py: --config %s needs the default C++ loop" % args.config)
        agent_kfs = [(k, frames[k]["kf_feats"][0], frames[k]["kf_feats"][1], frames[k]["Tcw"]) for k in range(nF) if frames[k]["kf_feats"] is not None]
        allp = {key: np.concatenate([kf_chunks[j][key] for j in sorted(kf_chunks)]) for key in kf_chunks[sorted(kf_chunks)[0]]}
        server_map_view, server_map_keep = views.worldpoints_view(allp["pos"], allp["normal"], allp["min_dist"], allp["max_dist"], allp["desc"],
                                                                  allp["n_obs"], allp["bad"])
        server = harness.ServerTick(grp, api, views, synth, device, fv, server_map_view, agent_kfs, FRAMES_PER_KF,
                                    tick_every=server_cfg["tick_every"], max_features=cfg["frame_cap"],
                                    shared_gpu=os.environ.get("ORBG_BENCH_SHARE_GPU") == "1" and world > 1)

    def ctxt_switches():
        """Involuntary context switches of every thread of this process so far: a spinning thread that loses its core to a
        neighbour's job shows up here (the bench boxes are shared hosts)."""
        tot = 0
        try:
            for t in os.listdir("/proc/self/task"):
                with open("/proc/self/task/%s/status" % t) as f:
                    for line in f:
                        if line.startswith("nonvoluntary_ctxt_switches"):
                            tot += int(line.split()[1])
        except OSError:
            return None
        return tot

    def run_region_cxx(n_steps, n_warm, pose_opt, host_images, pipelined, first_index):
        reg = Region(n_steps)
        loop.configure(pipelined, host_images, ingest_async, submit_first, args.lba_mode == "async", pose_opt, ahead=ctor_ahead)
        st = agent_mod.Stats()

        def sync():
            t_a = time.perf_counter()
            loop.drain(st, True)
            t_b = time.perf_counter()
            capi.check(capi.load().orbg_quiesce(device), "orbg_quiesce")     # spin until the library's streams are empty ...
            t_s = time.perf_counter()
            torch.cuda.synchronize()                                          # ... so that the contract's synchronisation has nothing to block on
            reg.sync_tail_s = time.perf_counter() - t_s
            reg.tail = (t_b - t_a, t_s - t_b, reg.sync_tail_s)               # drain (last local BA), quiesce, runtime synchronisation

        loop.run(first_index, n_warm)
        loop.drain()
        base = first_index + n_warm
        step_s = np.zeros(max(n_steps, 1))
        capi.check(capi.load().orbg_quiesce(device), "orbg_quiesce")      # (the same bracket on both sides; also: nothing of the closing
        grp.barrier(); torch.cuda.synchronize()                          # bracket is a first use inside the timed region)
        for e in exs:
            e.ctor_timeline(reset=True)
        cs0 = ctxt_switches()
        th0 = harness.cgroup_throttled()
        t0 = time.perf_counter()
        if server is None:
            loop.run(base, n_steps, last_is_final=True, timed=True, step_s=step_s, stats=st)
        else:
            # C3 / C5: the same K steps, handed to libagentloop in chunks that end with a keyframe step; after each the agent packs the
            # keyframe into a wire block, and every tick_every-th keyframe ALL ranks meet in the server tick's all-gather
            s_, ring_ = 0, loop.c.ring
            while s_ < n_steps:
                i_ = base + s_
                n_ = min(((-i_) % FRAMES_PER_KF) + 1, n_steps - s_)
                loop.run(i_, n_, last_is_final=(s_ + n_ == n_steps), timed=True, step_s=step_s[s_:], stats=st)
                s_ += n_
                last_ = base + s_ - 1
                if last_ % FRAMES_PER_KF == 0:
                    k_ = seq[last_ % len(seq)]
                    server.on_keyframe(Fs[(last_ % ring_) if pipelined else 0], st.last_nl, k_, frames[k_]["Tcw"])
            server.drain()                        # (rank 0) every tick exchanged in the region has been processed by the server thread
        sync()
        dt = time.perf_counter() - t0             # this rank's K steps, everything it launched complete; the job's time is the
        grp.barrier()                             # MAX over ranks (they started together) -- the barrier's own latency (gloo:
        elapsed = grp.max_over_ranks(dt)          # TCP, a few hundred microseconds against a 2.7 ms region) is not step time
        cs1 = ctxt_switches()
        reg.stats["nonvoluntary_ctxt_switches"] = None if cs0 is None or cs1 is None else cs1 - cs0
        th1 = harness.cgroup_throttled()
        reg.stats["cgroup_throttled_us"] = None if th0 is None or th1 is None else th1[1] - th0[1]
        for key, j in (("extract", 0), ("match_frame", 1), ("match_map", 2), ("pose_opt", 3), ("map_upload", 4), ("lba", 5), ("last_view_upload", 6)):
            reg.stage[key] = st.stage_s[j]
        reg.stats.update(kp=st.kp, m_frame=st.m_frame, m_map=st.m_map, lba_iters=st.lba_iters, lba_calls=st.lba_calls, lba_s=st.lba_s)
        reg.step_s = step_s
        reg.worst_step = (round(1e3 * st.worst_step_s, 3), [round(1e6 * st.worst_stage_s[q], 1) for q in range(7)], int(st.worst_step_index - base))
        reg.timeline = np.concatenate([e.ctor_timeline() for e in exs])
        return reg, elapsed

    def run_region(n_steps, n_warm, pose_opt, host_images, pipelined, first_index):
        """W untimed steps, then exactly n_steps timed ones between barrier + synchronize on both sides; MAX over ranks."""
        if use_cxx:
            return run_region_cxx(n_steps, n_warm, pose_opt, host_images, pipelined, first_index)
        reg = Region(n_steps)

        def sync():
            collect_async(reg)                         # every LBA triggered inside the timed region has finished
            for c in range(len(exs)):                  # ... and so has the frame constructor submitted by the last step
                if in_flight[c]:
                    exs[c].frame_stereo_dev_wait()
                    in_flight[c] = False
            capi.check(capi.load().orbg_quiesce(device), "orbg_quiesce")
            t_s = time.perf_counter()
            torch.cuda.synchronize()
            reg.sync_tail_s = time.perf_counter() - t_s

        for i in range(n_warm):
            step(first_index + i, reg, False, pose_opt, host_images, pipelined)
        collect_async(reg)
        base = first_index + n_warm
        # (the last timed step does not hand a further frame over: the region holds exactly n_steps constructors)
        for e in exs:
            e.ctor_timeline(reset=True)
        cs0 = ctxt_switches()
        elapsed = grp.timed(lambda i: step(base + i, reg, True, pose_opt, host_images, pipelined, slot=i, last=(i == n_steps - 1)), n_steps, sync)
        cs1 = ctxt_switches()
        reg.stats["nonvoluntary_ctxt_switches"] = None if cs0 is None or cs1 is None else cs1 - cs0
        reg.timeline = np.concatenate([e.ctor_timeline() for e in exs])
        return reg, elapsed

    # internal pre-warm, independent of --warmup: at least --prewarm-steps steps AND at least 50 ms of the main configuration
    prewarm_done = 0
    t_pw = time.perf_counter()
    scratch = Region(1)
    # (... and such that the first TIMED step is a keyframe step: the region then holds exactly K / FRAMES_PER_KF local BAs, the
    # last of them submitted FRAMES_PER_KF steps before the clock stops, whatever K and --warmup are)
    if use_cxx:
        loop.configure(pipeline, host_images, ingest_async, submit_first, args.lba_mode == "async", args.pose_opt, ahead=ctor_ahead)
    # The pre-warm runs until the step rate is STATIONARY: chunks of 100 steps, until at least --prewarm-steps steps and 0.3 s have
    # passed and the last three chunks agree within 1.5 % (cap: 3 s).  Measured on fresh boxes: after 400 steps / 50 ms the first
    # K = 20 region was 4-7 % below the four that followed it (and 20 % on the driver's box in round 3); after 4000 steps it is
    # within 1 % -- clocks and power states of a GPU that has just been leased settle over a few hundred milliseconds.
    chunk_rates = []
    while True:
        t_c = time.perf_counter()
        n_c = 100
        if use_cxx:
            loop.run(prewarm_done, n_c)
        else:
            for j in range(n_c):
                step(prewarm_done + j, scratch, False, args.pose_opt, host_images, pipeline)
        prewarm_done += n_c
        chunk_rates.append(n_c / (time.perf_counter() - t_c))
        t_all = time.perf_counter() - t_pw
        settled = len(chunk_rates) >= 3 and (max(chunk_rates[-3:]) - min(chunk_rates[-3:])) <= 0.015 * max(chunk_rates[-3:])
        if prewarm_done >= args.prewarm_steps and ((t_all >= 0.3 and settled) or t_all >= 3.0):
            break
    while (prewarm_done + args.warmup) % FRAMES_PER_KF != 0:
        if use_cxx:
            loop.run(prewarm_done, 1)
        else:
            step(prewarm_done, scratch, False, args.pose_opt, host_images, pipeline)
        prewarm_done += 1
    prewarm_s = time.perf_counter() - t_pw
    if use_cxx:
        loop.drain()
    collect_async(scratch)
    for e in exs:
        e.set_profile_interval(FAST_BRACKET_EVERY, reset=True)       # every handle brackets every 4th of ITS frames: one frame in four overall,
                                                                      # whatever the size of the ring (an event pair holds the stream for ~2 x 10 us)
    opt.set_profiling(True, reset=True)
    reg, elapsed = run_region(args.steps, args.warmup, args.pose_opt, host_images, pipeline, prewarm_done)
    # what the shared host did to the region: involuntary context switches of this process's threads inside it, and how busy
    # OTHER tenants keep the hardware threads of the agent's cores right after it (this process sleeps during the sample)
    host_noise = {"nonvoluntary_ctxt_switches_in_region": reg.stats.get("nonvoluntary_ctxt_switches"),
                  "cgroup_throttled_us_in_region": reg.stats.get("cgroup_throttled_us"), "cgroup_cpu_quota": harness.cgroup_cpu_quota()}
    if core_pair is not None:
        try:
            busy = harness._cpu_busy(0.1)
            host_noise["agent_cores_busy_after_region"] = {str(t): round(busy.get(t, 0.0), 2) for c in core_pair for t in sorted(c)}
        except OSError:
            pass
    # the same K-step region four more times, back to back (same configuration, same clock brackets): `value` stays the FIRST
    # region, literally; min / median / max over the five say how much of it is the draw of one 3 ms window on a shared host
    repeat_values = [world * args.steps / elapsed]
    repeat_detail = [(round(1e3 * elapsed, 4), round(1e3 * float(reg.step_s.sum()), 4), [round(1e6 * x, 1) for x in reg.tail],
                      round(1e3 * reg.stats["lba_s"] / max(reg.stats["lba_calls"], 1), 3), reg.worst_step)]
    if not args.no_secondary and args.repeats > 1:
        for rep in range(1, args.repeats):
            first = prewarm_done + rep * (args.steps + args.warmup + FRAMES_PER_KF)
            first += (-(first + args.warmup)) % FRAMES_PER_KF                # the first timed step is a keyframe step again
            rr, er = run_region(args.steps, args.warmup, args.pose_opt, host_images, pipeline, first)
            repeat_values.append(world * args.steps / er)
            repeat_detail.append((round(1e3 * er, 4), round(1e3 * float(rr.step_s.sum()), 4), [round(1e6 * x, 1) for x in rr.tail],
                                  round(1e3 * rr.stats["lba_s"] / max(rr.stats["lba_calls"], 1), 3), rr.worst_step))
    solver_sum_ms, solver_n, solver_unknowns, solver_mfma = opt.solver_stats()
    fast_sum, fast_n = 0.0, 0                              # bracket times accumulated inside the library over the timed region
    for e in exs:
        fs_, fn_ = e.fast_kernel_stats()
        fast_sum += fs_; fast_n += fn_
    # census of the other kernels of the Frame-constructor chain (event pair moved to each in turn, every frame, a few
    # dozen untimed steps): the roofline object below has to describe whichever kernel holds the most device time per step
    chain_ms = {}
    if not args.profile_stages:
        for kname in ("octree_kernel", "orient_desc_gpu_kernel", "pyr_tower_kernel"):
            for e in exs:
                e.set_profile_kernel(kname)
                e.set_profile_interval(1, reset=True)
            run_region(max(min(40, args.steps), 4), 4, False, host_images, pipeline, 30000)
            cs, cn = 0.0, 0
            for e in exs:
                s_, n_ = e.fast_kernel_stats()
                cs += s_; cn += n_
            if cn:
                chain_ms[kname] = (max(cs / cn - ev_overhead_ms, 1e-6), cn)
        for e in exs:
            e.set_profile_kernel("fast_cells_kernel")
    opt.set_profiling(False, reset=False)
    for e in exs:
        e.set_profiling(0)

    secondary = {}
    if not args.no_secondary:
        ns = max(min(args.secondary_steps, args.steps), 1)
        nw = max(min(args.warmup, 40), 1)
        if host_images and pipeline:
            r2, e2 = run_region(ns, nw, False, False, True, 10000)
            secondary["value_device_images"] = round(world * ns / e2, 3)
            secondary["value_device_images_note"] = ("the image(s) already resident in HBM (orbx_frame_stereo_dev_submit / orbx_frame_mono_dev_submit): no host "
                                                     "staging, no PCIe copy; pipelined constructor, %d timed steps" % ns)
        if pipeline:
            r4, e4 = run_region(ns, nw, False, True, False, 15000)
            secondary["value_sync_ctor_host_images"] = round(world * ns / e4, 3)
            secondary["value_sync_ctor_host_images_note"] = ("orbx_frame_stereo / orbx_frame_mono: the constructor an UNCHANGED Tracking thread calls -- host "
                                                             "images in, synchronous, nothing overlaps the tracking of the previous frame; "
                                                             "%d timed steps" % ns)
        if not args.pose_opt:
            r3, e3 = run_region(ns, nw, True, host_images, pipeline, 20000)
            secondary["value_with_pose_opt"] = round(world * ns / e3, 3)
            secondary["value_with_pose_opt_note"] = ("the two PoseOptimization calls of Tracking per frame inside the step "
                                                     "(S/Tracking.cc:2649,2712; 450 / 650 correspondences), %d timed steps" % ns)

    # informational: one PoseOptimization call
    for _ in range(5):
        opt.PoseOptimization(po1)
    t0 = time.perf_counter()
    for _ in range(20):
        opt.PoseOptimization(po1)
    pose_opt_ms = 1e3 * (time.perf_counter() - t0) / 20

    server_tick = None
    if args.server_tick:
        server_tick = run_server_tick(grp, api, views, torch, device, frames, kf_chunks, fv, LM, scene)

    copy_gbs = copy_bandwidth_gbs(torch, device) if rank == 0 else None
    # The drop-in child is a separate dynamically linked program: under a profiler (its preloaded library has initialised the GPU in
    # this process, and the child would inherit the preload) that fork + exec is the hop this pool forbids -- never from a profiled run.
    under_profiler = any(k.startswith(("ROCP", "ROCPROF", "ROCTRACER")) for k in os.environ) or \
        any(t in os.environ.get("LD_PRELOAD", "") for t in ("rocprof", "roctracer", "rocprofiler"))
    dropin = run_dropin_bench() if (rank == 0 and not args.no_dropin and not under_profiler and args.config == "C2") else None

    if rank == 0:
        K = args.steps
        stage, kern, stats = reg.stage, reg.kern, reg.stats
        ms_per_step = 1e3 * elapsed / K
        # fast_cells_kernel (the frame path's HBM-streaming kernel) is bracketed by a HIP event pair on the extractor's stream;
        # an EMPTY pair on that stream already measures ev_overhead_ms, so the launch duration is the bracket minus that
        fast_ms_raw = fast_sum / max(fast_n, 1)
        fast_ms = max(fast_ms_raw - ev_overhead_ms, 1e-6)
        pyr_px = sum(int(round(W / 1.2 ** l)) * int(round(H / 1.2 ** l)) for l in range(8))   # ~ the reference's float32 level sizes
        fast_bytes = (2 if stereo else 1) * pyr_px + (stats["kp"] / K) * 4.0
        fast_gbs = fast_bytes / (fast_ms * 1e-3) / 1e9 if fast_n else 0.0
        # the DOMINANT kernel of the step (top row of profiles/r2_*_kernel_stats: 8 launches per local BA): the LDL^T + solve of
        # the reduced camera system on the FP64 matrix cores.  Algorithmic FLOPs n^3/3 + 2 n^2 (dense LDL^T),
        # duration from a HIP event pair on the local BA's stream (one bracketed launch per solve) minus the empty-pair cost
        n_unk = solver_unknowns
        ldlt_flops = n_unk ** 3 / 3.0 + 2.0 * n_unk ** 2
        ldlt_ms_raw = solver_sum_ms / max(solver_n, 1)
        ldlt_ms = max(ldlt_ms_raw - lba_ev_overhead_ms, 1e-6)
        ldlt_tflops = ldlt_flops / (ldlt_ms * 1e-3) / 1e12 if solver_n else 0.0
        # ---- which kernel dominates the step?  device time per step = average launch x launches per step
        lm_per_ba = stats["lba_iters"] / max(stats["lba_calls"], 1)
        # the variant ldltm::pick() chooses for this many unknowns (tile rows T of the bordered matrix)
        ldlt_T = ((((n_unk + 3) & ~3) + 1) + 15) // 16
        xcd_on = os.environ.get("ORBG_LDLT_XCD", "1") != "0"          # (lba.hip: the eight-workgroup kernel from 9 tile rows on)
        ldlt_name = (("ldltm::k_ldlt_cols" if ldlt_T <= 8 else "ldltx::k_ldlt_xcd" if xcd_on and ldlt_T <= 19 else "ldltm::k_ldlt_mfma" if ldlt_T <= 9 else
                      "ldltm::k_ldlt_big" if ldlt_T <= 15 else "ldltm::k_ldlt_big48")
                     if solver_mfma else "k_wide_panel / k_wide_update")
        # HBM-side traffic of THAT kernel on THIS configuration: rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs of this
        # same command, KB per launch), committed under profiles/ (newest round; C4 has a file of its own: a 300 x 300 triangle is
        # 361 KB, the C2 kernel's 58 KB says nothing about it).  No matching kernel in the file: null, not somebody else's figure.
        traffic, traffic_src = None, None
        try:
            import glob
            suffix = "" if args.config == "C2" else "_" + args.config
            pj = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_write_per_kernel%s.json" % suffix)))[-1]
            pm = json.load(open(pj))
            want = ldlt_name.split("::")[-1].split(" ")[0]
            key = [k2 for k2 in pm if want in k2][0]
            traffic = int(1024 * (pm[key]["FETCH_SIZE_KB_avg"] + pm[key]["WRITE_SIZE_KB_avg"]))
            traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; kernel %s)" % (os.path.basename(pj), key)
        except Exception:
            pass
        per_step = {ldlt_name: ldlt_ms * lm_per_ba / FRAMES_PER_KF if solver_n else 0.0, "fast_cells_kernel": fast_ms if fast_n else 0.0}
        for kname, (kms, kn) in chain_ms.items():
            per_step[kname] = kms
        dominant = max(per_step, key=per_step.get)
        kp_per_frame = stats["kp"] / K
        chain_bytes = {   # algorithmic bytes per launch ("Kernels")
            "octree_kernel": 4.0 * 4.5 * kp_per_frame + 8.0 * kp_per_frame,            # ~4.5 candidates per selected keypoint in, one selection record out
            "orient_desc_gpu_kernel": kp_per_frame * (37 * 37 + 32 + 16),              # the keypoint's 37 x 37 patch once, descriptor + keypoint out
            "pyr_tower_kernel": (2 if stereo else 1) * (W * H + sum(int(round(W / 1.2 ** l) + 38) * int(round(H / 1.2 ** l) + 38) for l in range(8))),
            "fast_cells_kernel": fast_bytes,
        }
        step_ms = 1e3 * reg.step_s
        total_bytes_per_step = fast_bytes * 6.8 + 3.9e6 * (stats["lba_iters"] / max(stats["lba_calls"], 1)) / FRAMES_PER_KF
        line = {
            "metric": "tracking+localBA frames/sec (aggregate over agents; 1 agent per GPU)",
            "value": round(world * K / elapsed, 3),
            "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32 (ORB front-end, Hamming), f64 (local BA)",
            "data": "synthetic",
            "timed_region_s": round(elapsed, 4), "prewarm_steps": int(prewarm_done), "prewarm_s": round(prewarm_s, 3),
            "prewarm_last_chunk_fps": round(chunk_rates[-1], 1),
            "fps_formula": round(1.0 / (sum(v for k2, v in stage.items() if k2 != "lba") / K +
                                        stats["lba_s"] / max(stats["lba_calls"], 1) / FRAMES_PER_KF), 3),
            "fps_formula_note": "BASELINE.md protocol: 1 / (t_frontend + t_LBA / K), K = %d frames per keyframe; t_frontend = host wall "
                                "time of the frame path per step, t_LBA = wall time of one local BA (both from the main timed region)" % FRAMES_PER_KF,
            "step_ms_p50": round(float(np.percentile(step_ms, 50)), 4), "step_ms_p95": round(float(np.percentile(step_ms, 95)), 4),
            "step_ms_max": round(float(step_ms.max()), 4),
            "config": {"workload": cfg["label"] + ", 1 LBA per %d frames; local map = map points of the last %d keyframes (%d on average), "
                                   "%d-frame ping-pong sequence of %d distinct frames"
                                   % (FRAMES_PER_KF, cfg["local_kfs"], int(np.mean(maps.sizes)) if maps.sizes else 0, len(seq), nF), "name": args.config,
                       "per_agent_fps": round(K / elapsed, 3), "frames_per_keyframe": FRAMES_PER_KF,
                       "stage_ms_per_frame": {k2: round(1e3 * v / K, 4) for k2, v in stage.items()},
                       "device_ms_per_frame": dict({k2: round(v / K, 4) for k2, v in kern.items()}, fast_kernel_ms=round(fast_ms_raw, 4)),
                       "avg_keypoints_per_frame": round(stats["kp"] / K, 1),
                       "avg_matches_frame": round(stats["m_frame"] / K, 1), "avg_matches_map": round(stats["m_map"] / K, 1),
                       "lba_mode": args.lba_mode, "pose_opt_in_step": bool(args.pose_opt),
                       "tracking_loop": ("libagentloop.so (multi_orbslam3_amd/csrc/agent_loop.cpp): the per-frame loop is C++ above the C-ABI, timed as one call "
                                         "of K steps" if use_cxx else "python (ctypes wrappers, one step per call)"), "host_images_in_step": bool(host_images),
                       "image_ingest": ("host images -> pinned staging slot (%s) -> copy kernel on the extractor's stream -> HBM"
                                        % ("library ingest thread" if pipeline and ingest_async else "calling thread")) if host_images else "images resident in HBM",
                       "last_frame_view": ("resident: uploaded at the end of the frame's tracking (inside the step), read from HBM by the next frame's search"
                                           if (use_cxx and args.last_frame_view == "resident") else "read in place from pinned host memory"),
                       "local_map_points_avg": int(np.mean(maps.sizes)) if maps.sizes else 0, "local_map_keyframes": cfg["local_kfs"],
                       "sequence_frames": len(seq),
                       "cpu_affinity": cpu_affinity, "host_noise": host_noise, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "completion_wait": wait_mode, "wait_policy": wait_policy,
                       "frame_ctor": ("pipelined, %d frame(s) ahead: Frame(t+1 .. t+%d) are submitted (host images: orbx_frame_stereo_submit) on the other extractor "
                                      "handles of a ring of %d before frame t is tracked, and collected at the start of their own steps; the timed region holds "
                                      "exactly K constructors (the first step submits its own and the ones ahead, the last ones hand no further frame over)"
                                      % (ctor_ahead, ctor_ahead, n_ring)) if pipeline else "synchronous",
                       "frame_ctor_ahead": ctor_ahead if pipeline else 0,
                       "features_on_host": ("every frame: mvKeys / mDescriptors / mvuRight / mvDepth of the left image are copied into host arrays when the "
                                            "constructor is collected (orbx_set_frame_outputs), inside the step") if feature_outputs else
                                           "no: the features stay in HBM (counts only)",
                       "pose_opt_ms_per_call_450_correspondences": round(pose_opt_ms, 4),
                       "lba_ms_per_call": round(1e3 * stats["lba_s"] / max(stats["lba_calls"], 1), 3),
                       "sequential_fps_formula": round(1.0 / (sum(v for k2, v in stage.items() if k2 != "lba") / K +
                                                             stats["lba_s"] / max(stats["lba_calls"], 1) / FRAMES_PER_KF), 3),
                       "lba_lm_iterations_per_call": round(stats["lba_iters"] / max(stats["lba_calls"], 1), 2),
                       "whole_step_hbm": {"algorithmic_bytes_per_step": int(total_bytes_per_step),
                                          "achieved_GBps": round(total_bytes_per_step / (ms_per_step * 1e-3) / 1e9, 2),
                                          "frac_of_hbm_peak": round(total_bytes_per_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                          "note": "SURVEY.md 8(d): 6.5 MB per image + 3.9 MB per LM step; the step is launch / latency bound"},
                       "frame_path_kernel": {"kernel": "fast_cells_kernel", "bound": "hbm", "achieved_GBps": round(fast_gbs, 2),
                                             "frac_of_hbm_peak": round(fast_gbs / HBM_PEAK_GBS, 5), "avg_launch_ms": round(fast_ms, 5),
                                             "event_pair_overhead_ms": round(ev_overhead_ms, 5), "bracketed_launches": int(fast_n)},
                       "device_copy_GBps_measured": round(copy_gbs, 1), "host_cpu": host_cpu()},
            "roofline": {"kernel": ldlt_name,
                         "bound": "mfma", "achieved": round(ldlt_tflops, 6), "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ldlt_tflops / FP64_MATRIX_PEAK_TFLOPS, 8), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_flops_per_launch": int(ldlt_flops),
                         "unknowns": int(n_unk), "launches_per_local_ba": round(stats["lba_iters"] / max(stats["lba_calls"], 1), 2),
                         "avg_launch_ms": round(ldlt_ms, 5), "avg_launch_ms_event_bracket_raw": round(ldlt_ms_raw, 5),
                         "event_pair_overhead_ms": round(lba_ev_overhead_ms, 5), "bracketed_launches": int(solver_n),
                         "peak_source": "AMD MI355X spec sheet, FP64 matrix 78.6 TF (MI355X_MICROARCH.md has no FP64 row); measured issue rate "
                                        "of v_mfma_f64_16x16x4_f64: 1 per 66 cycles per SIMD = 76 TF at 2.4 GHz (tools/micro/mfma_f64_latency.hip)",
                         "note": "a 120 x 120 LDL^T + solve is 0.6 MFLOP on a dependent chain of 120 pivots (one workgroup): "
                                 "time-to-solution is the figure of merit, the FLOP fraction is reported as the contract asks"},
        }
        # ---- the diagnosis of THIS run as top-level scalars (nested objects do not survive into the driver's record)
        line["value_min"] = round(min(repeat_values), 3); line["value_median"] = round(float(np.median(repeat_values)), 3)
        line["value_max"] = round(max(repeat_values), 3); line["value_regions"] = len(repeat_values)
        line["host_noise_ctxt_switches"] = host_noise.get("nonvoluntary_ctxt_switches_in_region")
        line["host_cgroup_cpu_quota"] = host_noise.get("cgroup_cpu_quota")                    # CPUs this container may use per 100 ms period (None: no quota)
        line["host_noise_cgroup_throttled_us"] = host_noise.get("cgroup_throttled_us_in_region")   # > 0: the container spent its quota inside the region
        busy_after = host_noise.get("agent_cores_busy_after_region") or {}
        line["host_noise_max_core_busy"] = max(busy_after.values()) if busy_after else None
        tl = reg.timeline
        if len(tl):
            for j, nm in enumerate(("ingest_queue_us", "ingest_pack_us", "ctor_enqueue_us", "ctor_wait_us", "ctor_latency_us")):
                line[nm + "_p50"] = round(float(np.percentile(tl[:, j], 50)), 1)
                line[nm + "_max"] = round(float(tl[:, j].max()), 1)
        for k2, v in stage.items():
            if k2 in ("extract", "match_frame", "match_map", "map_upload", "lba", "last_view_upload"):
                line["stage_%s_us" % k2] = round(1e6 * v / K, 1)
        line["lba_ms_per_call"] = round(1e3 * stats["lba_s"] / max(stats["lba_calls"], 1), 3)
        line["sync_tail_us"] = round(1e6 * reg.sync_tail_s, 1)
        line["config"]["regions_ms_elapsed__ms_in_steps__tail_us_drain_quiesce_sync__lba_ms__worst_step"] = repeat_detail
        line["config"]["device_ms_per_step_by_kernel"] = {k2: round(v, 5) for k2, v in sorted(per_step.items(), key=lambda kv: -kv[1])}
        if dominant != ldlt_name:
            # a kernel of the constructor chain holds more device time per step than the LDL^T: it is the roofline's subject
            # (the LDL^T's figures stay in the line as config.lba_solver_kernel)
            line["config"]["lba_solver_kernel"] = line["roofline"]
            kms = per_step[dominant]
            kbytes = chain_bytes[dominant]
            gbs = kbytes / (kms * 1e-3) / 1e9
            traffic2 = None
            try:
                key2 = [k2 for k2 in pm if dominant in k2][0]
                traffic2 = int(1024 * (pm[key2]["FETCH_SIZE_KB_avg"] + pm[key2]["WRITE_SIZE_KB_avg"]))
            except Exception:
                pass
            line["roofline"] = {"kernel": dominant, "bound": "hbm", "achieved": round(gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(gbs / HBM_PEAK_GBS, 8), "traffic": traffic2, "traffic_source": traffic_src,
                                "algorithmic_bytes_per_launch": int(kbytes), "avg_launch_ms": round(kms, 5),
                                "event_pair_overhead_ms": round(ev_overhead_ms, 5),
                                "bracketed_launches": int(fast_n if dominant == "fast_cells_kernel" else chain_ms[dominant][1]),
                                "launches_per_step": 1,
                                "note": "one launch per Frame constructor (both cameras); the kernel works out of LDS on a few tens of "
                                        "KB per workgroup and is bound by dependent LDS round trips, not by HBM: the byte fraction is "
                                        "reported as the contract asks" if dominant == "octree_kernel" else
                                        "one launch per Frame constructor (both cameras)"}
        line.update(secondary)
        if dropin is not None:
            line["value_dropin"] = dropin.get("frames_per_s_frame_path")
            line["value_dropin_note"] = ("tests/cpp/dropin_bench: Frame::Frame(stereo, host images) + SearchByProjection(Cur, Last) + SearchLocalPoints "
                                         "called through the reference-signature glue (include/orbgpu_dropin.hpp) on mock Frame / MapPoint "
                                         "objects, synchronous constructor, one thread: frames/s of that frame path; per-call glue / upload / "
                                         "C-ABI microseconds under config.dropin")
            line["value_dropin_with_pose_opt"] = dropin.get("frames_per_s_frame_path_with_pose_opt")
            line["value_dropin_with_pose_opt_note"] = ("the same + both Optimizer::PoseOptimization calls of a frame (S/Tracking.cc:2649, :2712) through the glue: "
                                                       "what an UNCHANGED Tracking thread (synchronous constructor, one thread) gets per frame")
            line["config"]["dropin"] = dropin
        if server_tick is not None:
            line["config"]["server_tick"] = server_tick
    # CPU baseline: every rank (= agent) runs its own sample on its own three physical cores ("for A agents run A
    # independent baseline processes pinned to disjoint cores"); rank 0 reports its sample and min / median over the ranks
    if not args.no_cpu_baseline and not under_profiler:      # (the baseline builds the oracle -march=native: g++ children)
        n_base = 300 if args.config != "C4" else 60
        own = None
        if core_pair is not None and world > 1:
            own = set().union(*core_pair)
        else:
            os.sched_setaffinity(0, affinity_at_start)     # one agent: the baseline's three threads get the whole machine again
        # one rank of the node builds the -march=native oracle, the others wait for the file (they must not all write it at once)
        if local_rank == 0:
            from oracle import binding as ob_
            ob_.use_native()
        grp.barrier()
        base = cpu_baseline(scene, cfg, synth, views, n_base, host_imgs, frames, kf_chunks, seq, cpus=own, build_native=False)
        per_rank = grp.gather_floats(base["value"])
        if rank == 0:
            base["host_cpu"] = host_cpu()
            base["per_agent"] = {"values": [round(v, 3) for v in per_rank], "min": round(min(per_rank), 3),
                                 "median": round(float(np.median(per_rank)), 3),
                                 "pinning": "each agent's sample on its own physical cores %s" % (sorted(own) if own else "(unpinned: one agent)")}
            line["cpu_baseline"] = base
    # ---- in-job parity gate (SURVEY.md 8d last row): frames of the TIMED sequence once more through the same loop, every output
    # against the CPU oracle.  After the cpu_baseline leg (which selects the oracle build of this process), outside every timed region.
    parity = None
    if args.parity_frames > 0 and under_profiler:
        parity = {"skipped": "under a profiler (the oracle library may have to be rebuilt: a child process)"}
    elif args.parity_frames > 0:
        K_ = FRAMES_PER_KF
        gate_first = 40000                                   # (a multiple of FRAMES_PER_KF: the gate's first step is a keyframe step)
        if use_cxx:
            # the main region's configuration (pipelined or not, host or device images); the features come to the host through the
            # constructor's own delivery (orbx_set_frame_outputs for the two-halves form, its output arguments for the synchronous one)
            loop.drain()
            sync_out = None
            if pipeline and feature_outputs is None:
                feature_outputs = [e.set_frame_outputs(cfg["frame_cap"]) for e in exs]
            if not pipeline:
                sync_out = loop.set_sync_outputs(cfg["frame_cap"])
            loop.configure(pipeline, host_images, ingest_async, submit_first, args.lba_mode == "async", False, ahead=ctor_ahead)
            loop.capture_first_search(True)
            loop.run(gate_first - K_, K_)                    # the keyframe step before the gate makes its local map resident

            def run_step(i):
                st = loop.run(i, 1)
                out = sync_out if sync_out is not None else feature_outputs[i % loop.c.ring]
                return dict(nl=st.last_nl, nr=st.last_nr, n1=st.last_n1, n2=st.last_n2, kps=out["kps"], kps_un=out["kps_un"], desc=out["desc"],
                            uright=out["uright"], depth=out["depth"], amp_frame=loop.amp_after_frame, amp=loop.amp, aob=loop.aob)

            def map_view_of_step(i):
                return kf_views[((i // K_) - (1 if i % K_ == 0 else 0)) % len(kf_views)]
        else:
            scratch2 = Region(1)
            for j in range(gate_first - K_, gate_first):
                step(j, scratch2, False, False, host_images, pipeline)
            gate_maps = {}

            def run_step(i):
                cap_ = {}
                step(i, scratch2, False, False, host_images, pipeline, capture=cap_)
                gate_maps[i] = cap_["map_view"]
                return cap_

            def map_view_of_step(i):
                return gate_maps[i]
        if use_cxx or (pipeline and feature_outputs is not None) or (not stereo and not pipeline):
            parity = parity_gate(args.parity_frames, gate_first, run_step, (loop.drain if use_cxx else (lambda: collect_async(scratch2))), scene, cfg,
                                 views, frames, host_imgs, seq, map_view_of_step, lp, lba_out, th_frame, mono_flag, bounds=frame_bounds)
            if use_cxx:
                loop.capture_first_search(False)
                loop.set_sync_outputs(0)
            parity["loop"] = "libagentloop.so, configured as the main timed region" if use_cxx else "python loop (ctypes wrappers)"
        else:
            parity = {"skipped": "this python-loop configuration leaves the features in HBM (use the default --loop cxx)"}
        if server is not None and "skipped" not in parity:
            sp = server_tick_parity(server, views)
            if sp is not None:
                parity["server_tick"] = sp
                if not sp["ok"]:
                    parity["ok"] = False
                    parity.setdefault("violations", []).append("server tick: %s" % {k2: v for k2, v in sp.items() if v is False})
        # ---- the optimisers on frames of a two-camera rig (no BASELINE configuration has one: a parity-only leg, rank 0)
        if rank == 0 and "skipped" not in parity:
            rp = rig_parity(api, views)
            parity["camera_rig"] = rp
            if rp.get("ok") is not True:
                parity["ok"] = False
                parity.setdefault("violations", []).append("camera rig: %s" % {k2: v for k2, v in rp.items() if k2 != "what"})
        # ---- closed loop (rank 0): state carried from call to call, 200 frames, product vs oracle (tests/cpp/closed_loop.cpp)
        if rank == 0 and args.closed_loop_frames > 0 and args.config == "C2" and "skipped" not in parity:
            cl = run_closed_loop(args.closed_loop_frames)
            parity["closed_loop"] = cl
            if cl.get("ok") is not True:
                parity["ok"] = False
                parity.setdefault("violations", []).append("closed loop: %s" % {k2: v for k2, v in cl.items() if k2 in ("error", "shadow_mismatches", "first_divergent_frame", "first_divergent_frame_no_caches")})
            # ... and of a two-fisheye agent (tests/cpp/rig_loop.cpp: the constructor's stereo matcher, the two-camera matchers, both optimisers)
            rl = run_rig_loop(120)
            parity["rig_closed_loop"] = rl
            if rl.get("ok") is not True:
                parity["ok"] = False
                parity.setdefault("violations", []).append("rig closed loop: %s" % {k2: v for k2, v in rl.items() if k2 != "what"})
        oks = grp.gather_floats(1.0 if parity.get("ok", True) else 0.0)
        parity["agents_ok"] = [bool(v) for v in oks]
        parity["agent_digests"] = [int(v) for v in grp.gather_floats(float(parity.get("agent_digest", 0)))]
        parity["agent_ids"] = [args.first_agent + r_ for r_ in range(world)]
        if not all(parity["agents_ok"]):
            parity["ok"] = False
    # rank -> GPU -> NUMA node -> CPU sets: where every agent of the job ran (what a multi-GPU run must show next to its scaling figure)
    place = harness.gpu_placement(device)
    place.update(rank=rank, agent=args.first_agent + rank, cpu_affinity=cpu_affinity,
                 cores=None if core_pair is None else {"tracking": sorted(core_pair[0]), "local_ba_worker": sorted(core_pair[1]) if len(core_pair) > 1 else None,
                                                       "image_ingest": sorted(core_pair[2]) if len(core_pair) > 2 else None})
    placement = grp.gather_objects(place)
    kinds = grp.gather_floats(0.0 if stereo else 1.0)
    pol = grp.gather_floats(float(sum(1 << j for j, r in enumerate(("caller", "lba", "ingest")) if wait_policy[r] == "block")))
    server_report = server.report() if server is not None else None
    if rank == 0:
        line["parity"] = parity
        line["config"]["agents"] = ["stereo" if v == 0.0 else "mono" for v in kinds]
        line["config"]["placement_per_rank"] = placement
        line["config"]["wait_policy_per_rank"] = ["".join(("b" if (int(v) >> j) & 1 else "s") for j in range(3)) + " (tracking / local-BA worker / ingest: s = spins, b = blocks)"
                                                  for v in pol]
        if server_report is not None:
            line["config"]["server_tick_in_job"] = server_report
        # (RCCL writes a version banner through C stdio, which is flushed at exit when stdout is a file: flush it now so that the
        # JSON line is the LAST line of the output)
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
    grp.close()
    if parity is not None and parity.get("ok") is False:
        sys.stderr.write("bench.py: PARITY VIOLATION against the oracle: %s\n" % "; ".join(parity.get("violations", []) or ["(another agent)"]))
        sys.exit(3)


def run_dropin_bench():
    """tests/cpp/dropin_bench (built by __graft_entry__.build(); rebuilt here if missing): the path through the reference-side glue."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "dropin_bench")
    try:
        if not os.path.exists(exe):
            import __graft_entry__ as ge
            ge.build_dropin_bench()
        r = subprocess.run([exe, "40"], capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            return {"error": "dropin_bench exit code %d: %s" % (r.returncode, (r.stdout + r.stderr)[-300:])}
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as e:                                # the headline does not depend on it
        return {"error": repr(e)[:300]}


def rig_parity(api, views):
    """LocalBundleAdjustment and PoseOptimization on keyframes / a Frame of a two-fisheye rig (KannalaBrandt8 models, the right camera's
    EdgeSE3ProjectXYZToBody edges, S/Optimizer.cc:1085-1151, 2021-2120) and the tracking matchers on a two-camera Frame against the oracle.  The oracle is the checker here, outside
    every timed region."""
    try:
        from multi_orbslam3_amd import synth
        from oracle import binding as ob
        pr = synth.make_lba_rig_problem(n_free=8, n_fixed=4, n_points=600)
        p, keep = views.lba_problem(pr["poses"], pr["pose_fixed"], pr["points"], pr["edges"], pr["cam"], rig=views.camera_rig(*pr["rig"]))
        opt = api.Optimizer()
        g = opt.LocalBundleAdjustment(p)
        o = ob.lba_solve(p)
        d = {"lba_edges": int(p.n_edges), "lba_right_camera_edges": int((pr["edges"]["ur"] <= -1.5).sum()),
             "lba_iterations_equal": bool(g.iters == o.iters and g.status == o.status),
             "lba_max_pose_diff": float(np.abs(g.poses - o.poses).max()), "lba_max_point_diff": float(np.abs(g.points - o.points).max()),
             "lba_outlier_flags_differing": int((g.edge_outlier != o.edge_outlier).sum())}
        po = synth.make_pose_opt_rig_problem(n_left=300, n_right=200)
        q, keep2 = views.pose_opt_problem(po["Xw"], po["u"], po["v"], po["ur"], po["inv_sigma2"], po["cam"], po["Tcw"],
                                          rig=views.camera_rig(*po["rig"]))
        g2 = opt.PoseOptimization(q)
        o2 = ob.pose_optimize(q)
        d.update({"pose_opt_max_pose_diff": float(np.abs(g2.Tcw.astype(np.float64) - o2.Tcw.astype(np.float64)).max()),
                  "pose_opt_outlier_flags_differing": int((g2.outliers != o2.outliers).sum()),
                  "pose_opt_inliers": [int(g2.n_inliers), int(o2.n_inliers)]})
        # the tracking matchers on a two-camera Frame (Nleft != -1): isInFrustum through either camera, SearchByProjection(Frame, MapPoints)
        # with the right camera's block, SearchByProjection(CurrentFrame, LastFrame) -- S/Frame.cc:545-554, S/ORBmatcher.cc:44-214, 1970-2186
        sc = synth.make_rig_track_scene()
        bounds = (0, sc["size"], 0, sc["size"])
        cam = (sc["left"][1], sc["left"][2], sc["left"][3], sc["left"][4], 0.0, 0.1)
        fl, k1 = views.frame_view(sc["kps_left"], sc["desc_left"], None, None, bounds, cam)
        fr, k2 = views.frame_view(sc["kps_right"], sc["desc_right"], None, None, bounds, cam)
        wv, k3 = views.worldpoints_view(sc["pos"], sc["normal"], sc["min_dist"], sc["max_dist"], sc["desc"], sc["n_obs"], sc["bad"])
        rig = views.camera_rig(sc["left"], sc["right"], sc["Trl"])
        FL, FR = api.Frame().upload(fl, k1), api.Frame().upload(fr, k2)
        gt, ot = FL.isInFrustumRig(sc["Tcw"], rig, sc["Tlr"], wv), ob.is_in_frustum_rig(fl, sc["Tcw"], rig, sc["Tlr"], wv)
        mk = lambda t: views.mappoints_view(t["track_in_view"], sc["bad"], t["proj_x"], t["proj_y"], t["proj_x"], t["track_depth"], t["scale_level"],
                                            t["view_cos"], sc["desc"], sc["n_obs"])
        (mv, k4), (mvr, k5) = mk(ot[0]), mk(ot[1])
        m = api.ORBmatcher(0.8, True)
        gs = m.SearchByProjectionRig(FL, FR, mv, mvr, sc["left_to_right"], sc["right_to_left"], 3.0, True, 6.0, sc["assigned_mp"], sc["assigned_obs"])
        os_ = ob.search_by_projection_mps_rig(fl, fr, mv, mvr, sc["left_to_right"], sc["right_to_left"], 3.0, True, 6.0, 0.8, sc["assigned_mp"], sc["assigned_obs"])
        last = synth.rig_last_frame(sc)
        lv, k6 = views.lastframe_view(last["mp_valid"], last["outlier"], last["world_pos"], last["desc"], last["octave"], last["angle"], last["n_obs"], last["Tcw"])
        gf = m.SearchByProjectionFrameRig(FL, FR, sc["Tcw"], rig, lv, 7.0, False, sc["assigned_mp"], sc["assigned_obs"])
        of = ob.search_by_projection_frame_rig(fl, fr, sc["Tcw"], rig, lv, 7.0, 0, 1, sc["assigned_mp"], sc["assigned_obs"])
        d.update({"matcher_features": [len(sc["kps_left"]), len(sc["kps_right"])], "matcher_map_points": int(wv.m),
                  "frustum_flags_and_levels_equal": bool(all(np.array_equal(gt[s_][k], ot[s_][k]) for s_ in (0, 1) for k in ("track_in_view", "scale_level"))),
                  "frustum_max_projection_diff_px": float(max(np.abs(gt[s_][k].astype(np.float64) - ot[s_][k]).max() for s_ in (0, 1) for k in ("proj_x", "proj_y"))),
                  "search_map_points_matches": [int(gs[2]), int(os_[2])], "search_map_points_features_differing": int((gs[0] != os_[0]).sum()),
                  "search_last_frame_matches": [int(gf[2]), int(of[2])], "search_last_frame_features_differing": int((gf[0] != of[0]).sum())})
        # ... and the left-right matcher of the two-fisheye Frame constructor (Frame::ComputeStereoFishEyeMatches, S/Frame.cc:1093-1150)
        fs = synth.make_fisheye_stereo_scene()
        fsv, k7 = views.fisheye_stereo_view(fs["kps_left"], fs["desc_left"], fs["mono_left"], fs["kps_right"], fs["desc_right"], fs["mono_right"], fs["left"],
                                            fs["right"], fs["Tlr"], fs["level_sigma2"])
        gc, oc = api.ComputeStereoFishEyeMatches(fsv), ob.fisheye_stereo_matches(fsv)
        hit = oc[0] >= 0
        d.update({"fisheye_stereo_matches": [int(gc[4]), int(oc[4])],
                  "fisheye_partner_arrays_equal": bool(np.array_equal(gc[0], oc[0]) and np.array_equal(gc[1], oc[1])),
                  "fisheye_max_relative_depth_diff": float(np.abs(gc[2][hit] - oc[2][hit]).max() / np.abs(oc[2][hit]).max()) if hit.any() else 0.0})
        d["ok"] = bool(d["lba_iterations_equal"] and d["lba_max_pose_diff"] <= 1e-4 and d["lba_max_point_diff"] <= 1e-4 and
                       d["lba_outlier_flags_differing"] <= 1 and d["pose_opt_max_pose_diff"] <= 1e-5 and
                       d["fisheye_partner_arrays_equal"] and gc[4] == oc[4] > 150 and d["fisheye_max_relative_depth_diff"] <= 1e-5 and
                       d["pose_opt_outlier_flags_differing"] <= 1 and d["frustum_flags_and_levels_equal"] and
                       d["frustum_max_projection_diff_px"] <= 3e-4 and gs[2] == os_[2] > 300 and d["search_map_points_features_differing"] == 0 and
                       gf[2] == of[2] > 200 and d["search_last_frame_features_differing"] == 0)
        d["what"] = ("8 + 4 keyframes / 600 points of a two-fisheye rig through lba_solve_h, 300 + 200 features through pose_optimize, "
                     "isInFrustum / SearchByProjection(F, MPs) / SearchByProjection(Cur, Last) on a two-camera frame, ComputeStereoFishEyeMatches; "
                     "product vs oracle; tolerances of tests/test_gpu_parity.py (-k 'rig or two_camera')")
        return d
    except Exception as e:
        return {"ok": False, "error": repr(e)[:300]}


def run_closed_loop(n_frames):
    """tests/cpp/closed_loop (built by __graft_entry__.build(); rebuilt here if missing): the closed-loop parity run.  The executable
    links the oracle -- it is the checker, outside every timed region -- and liborbgpu; its last JSON line is the report."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "closed_loop")
    try:
        if not os.path.exists(exe):
            import __graft_entry__ as ge
            ge.build_closed_loop()
        t0 = time.time()
        r = subprocess.run([exe, str(int(n_frames)), "5"], capture_output=True, text=True, timeout=900)
        rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if not rows:
            return {"ok": False, "error": "closed_loop exit code %d: %s" % (r.returncode, (r.stdout + r.stderr)[-300:])}
        d = json.loads(rows[-1])["closed_loop"]
        d["ok"] = bool(d.get("ok")) and r.returncode == 0
        d["wall_s"] = round(time.time() - t0, 1)
        d["what"] = ("stereo agent through include/orbgpu_dropin.hpp over mock Frame / KeyFrame / MapPoint / Map objects: motion model -> "
                     "SearchByProjection(Cur, Last) -> PoseOptimization -> SearchLocalPoints -> PoseOptimization every frame, keyframe + "
                     "LocalBundleAdjustment every 5th; the product run and the oracle run each carry their own state; shadow_* = every "
                     "product call repeated on the oracle with identical inputs; first_divergent_frame = -1: every discrete digest equal "
                     "on every frame")
        return d
    except Exception as e:
        return {"ok": False, "error": repr(e)[:300]}


def run_rig_loop(n_frames):
    """tests/cpp/rig_loop (built with closed_loop): a two-fisheye agent through the glue, product vs oracle, each on its own outputs."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "rig_loop")
    try:
        if not os.path.exists(exe):
            import __graft_entry__ as ge
            ge.build_closed_loop()
        r = subprocess.run([exe, str(int(n_frames))], capture_output=True, text=True, timeout=600)
        rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if not rows:
            return {"ok": False, "error": "rig_loop exit code %d: %s" % (r.returncode, (r.stdout + r.stderr)[-300:])}
        d = json.loads(rows[-1])["rig_loop"]
        d["ok"] = bool(d.get("ok")) and r.returncode == 0
        r2 = subprocess.run([exe, str(int(n_frames)), "--shadow"], capture_output=True, text=True, timeout=600)      # every call repeated on the oracle with identical inputs
        rows2 = [ln for ln in r2.stdout.splitlines() if ln.startswith("{")]
        sh = json.loads(rows2[-1])["rig_loop_shadow"] if rows2 else {"ok": False, "error": (r2.stdout + r2.stderr)[-300:]}
        d["shadow"] = sh
        d["ok"] = d["ok"] and bool(sh.get("ok")) and r2.returncode == 0
        d["what"] = ("two-fisheye agent through include/orbgpu_dropin.hpp over mock objects: ComputeStereoFishEyeMatches -> motion model -> "
                     "SearchByProjection(Cur, Last) -> PoseOptimization -> SearchLocalPoints -> PoseOptimization every frame; the product run and "
                     "the oracle run each carry their own state (a fisheye agent's runs may part by a match or two where a projection sits within 1e-4 px of a "
                     "window edge: ok = they stay together); shadow = every product call repeated on the oracle with identical inputs, exact")
        return d
    except Exception as e:
        return {"ok": False, "error": repr(e)[:300]}


def run_server_tick(grp, api, views, torch, device, frames, kf_chunks, fv, LM, scene, reps=50):
    """Server tick of configs[2]/[4] (S/Communicator.cc:124-146 hand-over, S/LoopClosing.cc:657,769,795 matching): every
    agent contributes ALL its new KeyFrame wire blocks (R/msg/KF.msg:29-31) in ONE RCCL all-gather per tick
    (AgentGroup.all_gather_keyframe_blocks: fixed-size buffers, feature counts in the header), the server rebuilds each
    KeyFrame on the device (orbk_frame_from_wire) and runs SearchByProjection(KF, Scw, map points) on it.  2 and 8 blocks per
    tick in total: with fewer ranks than blocks every rank contributes several keyframes to the same collective."""
    from multi_orbslam3_amd import _capi as capi
    p = scene.frame_view_params()
    kf_ids = sorted(kf_chunks)[:8]
    blocks = []
    Ftmp = api.Frame(8192, device)
    for k in kf_ids:
        ch = kf_chunks[k]
        n = min(len(ch["src_idx"]), 2000)
        kps = np.zeros(n, capi.KEYPOINT_DTYPE)
        kps["x"] = 100.0 + (np.arange(n) % 400); kps["y"] = 80.0 + (np.arange(n) // 400) * 7.0
        kps["size"] = 31.0; kps["angle"] = 0.0; kps["response"] = 20.0; kps["octave"] = 0
        fvk, keepk = views.frame_view(kps, ch["desc"][:n], None, None, p["bounds"], p["cam"], 8, 1.2)
        Ftmp.upload(fvk, keepk)
        w = torch.zeros(47 * n, dtype=torch.uint8, device="cuda:%d" % device)
        Ftmp.pack_wire(device_ptr=w.data_ptr())
        blocks.append((n, w, frames[k]["Tcw"].astype(np.float32)))
    torch.cuda.synchronize()
    m = api.ORBmatcher(0.75, True, device)
    K = api.Frame(8192, device)
    grp.open_data_plane()                                                  # collective: the RCCL group of the exchange, outside every timed rep
    bufs = grp.tick_buffers(max_features=2048, device="cuda:%d" % device, max_blocks=8)
    out = {}
    for n_blocks in (2, 8):
        per_rank = max((n_blocks + grp.world - 1) // grp.world, 1)
        mine = [(n, w) for (n, w, T) in blocks[:per_rank]]
        free = np.full(8192, -1, np.int32)

        def tick():
            done = 0
            got = grp.all_gather_keyframe_blocks(bufs, mine)          # ONE collective + one header read-back per tick
            for lst in got:
                for j, (nr, blk) in enumerate(lst):
                    if done >= n_blocks:
                        break
                    K.from_wire(fv, n=nr, device_ptr=blk.data_ptr())
                    m.SearchByProjectionSim3(K, blocks[j][2], LM, free[:nr], 4, 1.5)
                    done += 1
            return done
        for _ in range(5):
            tick()
        torch.cuda.synchronize(); grp.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            done = tick()
        torch.cuda.synchronize(); grp.barrier()
        dt = grp.max_over_ranks(time.perf_counter() - t0)
        out["%d_kf_blocks_us" % n_blocks] = round(1e6 * dt / reps, 1)
    # the collective alone (8 blocks)
    mine = [(n, w) for (n, w, T) in blocks[:max((8 + grp.world - 1) // grp.world, 1)]]
    torch.cuda.synchronize(); grp.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        grp.all_gather_keyframe_blocks(bufs, mine)
    torch.cuda.synchronize(); grp.barrier()
    out["exchange_only_8_blocks_us"] = round(1e6 * grp.max_over_ranks(time.perf_counter() - t0) / reps, 1)
    out["note"] = ("per tick: ONE RCCL all-gather of every agent's new KeyFrame wire blocks (47 B/feature, fixed-size buffers of %d bytes, "
                   "backend nccl, %d rank%s) + one read-back of the headers, then orbk_frame_from_wire + SearchByProjection(KF, Scw, map "
                   "points) per block" % (bufs["cap"], grp.world, "" if grp.world == 1 else "s"))
    return out


if __name__ == "__main__":
    main()
