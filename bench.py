#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric):
"tracking+localBA frames/sec per agent, 640x480 stereo, 1/2/4/8 agents".

One STEP = one stereo frame of one agent through the hot path, inputs already resident in HBM:
  [Frame ctor: extract L+R (pyramid, FAST, quad-tree, angle, rBRIEF) -> ComputeStereoMatches -> feature grid]
  ->  SearchByProjection(cur, last)  ->  SearchLocalPoints (isInFrustum + SearchByProjection over the local map)
and, every FRAMES_PER_KF-th step (a keyframe), one Local Bundle Adjustment (20 free + 10 fixed KFs, 2000 points)
plus the upload of the refreshed local map.  By default (--lba-mode async) the LBA runs on the library's own worker
thread and HIP stream (lba_solve_async / lba_wait), concurrently with the frame loop, exactly as the reference runs
LocalMapping next to Tracking (S/ClientSystem.cc:105-106); every LBA triggered in the timed region is waited for
before the clock stops.  `--lba-mode thread` does the same from a Python thread (pays for GIL hand-overs).
`--lba-mode inline` gives the serial accounting fps = 1 / (t_frontend + t_LBA / FRAMES_PER_KF), which is also
reported as config.sequential_fps_formula.

Agents shard one per GPU with no data-path collective -> weak scaling; `value` is the
aggregate over all ranks.  Launch for N>1:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
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
HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy peak is ~6300 GB/s
# algorithmic bytes of the FAST kernel per launch: every pyramid pixel of both cameras read once
PYR_PIXELS_640x480 = 950532


def build_workload(scene, n_frames, api, views, synth, device):
    """Run the pipeline once per distinct frame to cache the host-side views a Tracking thread would hold
    (last-frame view, local-map chunks, pose guesses).  Not timed."""
    import torch
    from multi_orbslam3_amd import _capi as capi
    cam = scene.cam
    W, H = scene.W, scene.H
    ex = api.ORBextractor(1000, 1.2, 8, 20, 7, W, H, n_cams=2, device=device)
    rng = np.random.RandomState(1234)
    frames = []
    imgs = []
    for k in range(n_frames):
        L, R, Tcw = scene.stereo_pair(k)
        dL = torch.from_numpy(L).to("cuda:%d" % device)
        dR = torch.from_numpy(R).to("cuda:%d" % device)
        imgs.append((dL, dR))
        nl, nr, kl, dl = ex.extract_stereo_dev(dL.data_ptr(), dR.data_ptr(), W, H, W, download_left=True)
        ur, dp = ex.ComputeStereoMatches(float(cam["bf"]), float(cam["b"]), n_left=nl)
        kl, dl, ur, dp = kl.copy(), dl.copy(), ur.copy(), dp.copy()
        Pw, valid = synth.unproject_to_world(kl, dp, Tcw, cam)
        lv, keep = views.lastframe_view(valid.astype(np.uint8), np.zeros(nl, np.uint8), Pw, dl, kl["octave"], kl["angle"],
                                        np.full(nl, 3, np.int32), Tcw.astype(np.float32))
        chunk = synth.map_from_frame(kl, dl, dp, Tcw, cam)
        frames.append(dict(Tcw=Tcw, guess=synth.perturb_pose(Tcw, rng).astype(np.float32), last_view=(lv, keep), chunk=chunk,
                           n=nl, stereo=int((ur > 0).sum())))
    return ex, imgs, frames


def local_map_for(frames, k, n_kf=6):
    """Union of the map points created by the last n_kf keyframes before frame k (keyframe = every FRAMES_PER_KF-th)."""
    ids = [((k // FRAMES_PER_KF) - j) * FRAMES_PER_KF for j in range(1, n_kf + 1)]
    ids = [i % len(frames) for i in ids]
    parts = [frames[i]["chunk"] for i in ids]
    return {key: np.concatenate([p[key] for p in parts]) for key in parts[0]}


def cpu_baseline(scene, synth, views, n_frames=300, n_distinct=16):
    """The CPU oracle (a restatement of the reference path) on a bounded sample of the same workload, with the reference's
    threading: left / right extraction on two threads (S/Frame.cc:92-95), the tracking steps on the calling thread, local
    BA on its own thread next to tracking (S/ClientSystem.cc:105-106) -- at most 3 busy cores."""
    import queue
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding as ob
    cam = scene.cam
    exL = ob.Extractor(n_features=1000, max_width=scene.W, max_height=scene.H)
    exR = ob.Extractor(n_features=1000, max_width=scene.W, max_height=scene.H)
    p = scene.frame_view_params()
    rng = np.random.RandomState(1234)
    imgs = [scene.stereo_pair(k) for k in range(n_distinct)]
    seq = list(range(n_distinct)) + list(range(n_distinct - 2, 0, -1))
    prob = synth.make_lba_problem(n_free=20, n_fixed=10, n_points=2000)
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
    last = None
    chunks = []
    t_start = None
    done = 0
    for i in range(n_frames + 8):
        L, R, Tcw = imgs[seq[i % len(seq)]]
        if i == 8:                                        # the first frames only fill the last-frame view / local map
            t_start = time.perf_counter()
        fl, fr = pool.submit(exL.extract, L), pool.submit(exR.extract, R)
        rc, kl, dl, _ = fl.result()
        rc, kr, dr, _ = fr.result()
        ur, dp = ob.stereo_match(exL, exR, kl, dl, kr, dr, float(cam["bf"]), float(cam["b"]))
        fv, keep1 = views.frame_view(kl, dl, ur, dp, p["bounds"], p["cam"], 8, 1.2)
        n = len(kl)
        amp = np.full(n, -1, np.int32); aob = np.zeros(n, np.int32)
        guess = synth.perturb_pose(Tcw, rng).astype(np.float32)
        if last is not None and len(chunks) >= 2:
            amp, aob, nm1 = ob.search_by_projection_frame(fv, guess, last[0], 7.0, False, True, amp, aob)
            mp = {key: np.concatenate([c[key] for c in chunks[-6:]]) for key in chunks[0]}
            wv, keep2 = views.worldpoints_view(mp["pos"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], mp["n_obs"], mp["bad"])
            amp, aob, nm2 = ob.search_local_points(fv, wv, guess, 1.0, False, 0.0, 0.8, amp, aob)
        if i >= 8:
            done += 1
            if done % FRAMES_PER_KF == 0:
                q.put(1)
        Pw, valid = synth.unproject_to_world(kl, dp, Tcw, cam)
        last = views.lastframe_view(valid.astype(np.uint8), np.zeros(n, np.uint8), Pw, dl, kl["octave"], kl["angle"],
                                    np.full(n, 3, np.int32), Tcw.astype(np.float32))
        if i < 8 or i % FRAMES_PER_KF == 0:
            chunks.append(synth.map_from_frame(kl, dl, dp, Tcw, cam))
            chunks = chunks[-6:]
    q.put(None)
    worker.join()                                         # every LBA triggered by the sample has finished
    wall = time.perf_counter() - t_start
    pool.shutdown()
    fps = done / wall
    return dict(value=round(fps, 3), unit="frames/s", cores=3, kind="port",
                sample="%d stereo frames (L/R extraction on 2 threads) + %d local BAs of %.1f ms on their own thread, %.1f s wall; "
                       "threads as in the reference (S/Frame.cc:92-95, S/ClientSystem.cc:105-106)"
                       % (done, len(lba_times), 1e3 * float(np.mean(lba_times)) if lba_times else 0.0, wall))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--frames", type=int, default=16, help="distinct synthetic stereo frames (ping-pong sequence)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--separate-calls", action="store_true",
                    help="call extract / ComputeStereoMatches / grid as three entry points instead of the fused "
                         "Frame-constructor entry point orbx_frame_stereo_dev")
    ap.add_argument("--pose-opt", action="store_true",
                    help="also run Optimizer::PoseOptimization (SURVEY row f-2) after each of the two searches, as "
                         "Tracking does; off by default so that the metric stays the one SURVEY.md 8(d) defines")
    ap.add_argument("--profile-stages", action="store_true",
                    help="bracket every extractor stage with HIP events (more API calls per frame); by default only "
                         "fast_cells_kernel (the roofline kernel) is bracketed")
    ap.add_argument("--no-numa-pin", action="store_true",
                    help="do not restrict the process to the CPUs of the GPU's NUMA node (default: like numactl --cpunodebind)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="construct every frame synchronously before it is tracked; by default frame t+1's constructor "
                         "(orbx_frame_stereo_dev_submit on a second extractor handle) runs while frame t is tracked")
    ap.add_argument("--lba-mode", choices=["async", "thread", "inline"], default="async",
                    help="async: LBA runs on the library's worker thread + its own HIP stream concurrently with tracking, as the "
                         "reference's LocalMapping thread does (S/ClientSystem.cc:105-106); thread: the same from a Python "
                         "thread; inline: LBA blocks the frame loop")
    args = ap.parse_args()

    # The agent keeps six HIP streams busy (two extractor handles, two frames, the local map, the local BA).  The ROCm
    # runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with 4, the search kernels sometimes
    # share a queue with the local BA's chain (bimodal 4200 / 4950 frames/s run to run); with 6 every stream has its own
    # (stable 4930-5040); 8 and 12 are slower again (4500).  Must be set before the runtime initialises.
    # (Without the pipelined constructor the agent has three busy streams and the default of 4 is the good setting: 4600 vs
    # 3000 frames/s with 6 or 8 -- more hardware queues than busy streams cost dispatch latency on every one of them.)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4" if (args.no_pipeline or args.separate_calls) else "6")
    # Every agent keeps two threads spinning on completion words (tracking thread, local-BA worker).  If the container's CPU
    # quota cannot feed that for all ranks of this node (cgroup cpu.max), fall back to the runtime's blocking waits
    # (ORBG_NO_POLL=1: ~6-10 us more latency per wait, a fraction of a CPU per rank) instead of being throttled.
    wait_mode = "polling"
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        if q != "max":
            ranks_here = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
            if float(q) / float(per) < 3.0 * ranks_here:
                os.environ.setdefault("ORBG_NO_POLL", "1")
                wait_mode = "runtime waits (cpu quota %.1f for %d ranks)" % (float(q) / float(per), ranks_here)
    except (OSError, ValueError):
        pass
    if os.environ.get("ORBG_NO_POLL"):
        wait_mode = "runtime waits" if wait_mode == "polling" else wait_mode
    import torch
    from multi_orbslam3_amd import _capi as capi
    from multi_orbslam3_amd import api, harness, synth, views

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    grp = harness.AgentGroup("nccl")
    rank, local_rank, world = grp.rank, grp.local_rank, grp.world
    device = local_rank
    torch.cuda.set_device(device)
    affinity_at_start = os.sched_getaffinity(0)
    cpu_affinity = None if args.no_numa_pin else harness.pin_to_gpu_numa_node(device)
    core_pair = None if args.no_numa_pin else harness.core_pair_for_agent(device, local_rank)

    W, H = 640, 480
    scene = synth.Scene(W, H, seed=synth.SEED_IMAGES + rank)      # one agent per GPU, distinct seeds
    cam = scene.cam
    ex, imgs, frames = build_workload(scene, args.frames, api, views, synth, device)
    p = scene.frame_view_params()
    fv, fv_keep = views.frame_view(np.zeros(1, capi.KEYPOINT_DTYPE), np.zeros((1, 32), np.uint8), None, None, p["bounds"], p["cam"], 8, 1.2)
    F = api.Frame(4096, device)
    pipeline = not (args.no_pipeline or args.separate_calls)
    # frame t+1 is constructed (second extractor handle, second frame object) while frame t is tracked
    exs = [ex, api.ORBextractor(1000, 1.2, 8, 20, 7, W, H, n_cams=2, device=device)] if pipeline else [ex]
    Fs = [F, api.Frame(4096, device)] if pipeline else [F]
    in_flight = [False, False]
    LM = api.LocalMap(16384, device)
    m_frame = api.ORBmatcher(0.9, True, device)
    m_map = api.ORBmatcher(0.8, True, device)
    opt = api.Optimizer(device)
    prob = synth.make_lba_problem(n_free=20, n_fixed=10, n_points=2000, seed=synth.SEED_LBA + rank)
    lp, lp_keep = views.lba_problem(prob["poses"], prob["pose_fixed"], prob["points"], prob["edges"], prob["cam"], device=device)
    bf, bb = float(cam["bf"]), float(cam["b"])
    if core_pair is not None and args.lba_mode == "async":
        # the library's local-BA worker inherits the affinity of the thread that makes the first asynchronous call
        os.sched_setaffinity(0, core_pair[1])
        opt.LocalBundleAdjustmentAsync(lp, views.LbaOutput(lp.n_poses, lp.n_points, lp.n_edges))
        opt.wait()
        os.sched_setaffinity(0, core_pair[0])
        cpu_affinity = "%s; tracking thread on cpus %s, local-BA worker on cpus %s" % (cpu_affinity, sorted(core_pair[0]), sorted(core_pair[1]))
    nF = len(frames)
    seq = list(range(nF)) + list(range(nF - 2, 0, -1))             # ping-pong: consecutive frames stay adjacent
    stage = dict(extract=0.0, stereo=0.0, grid=0.0, match_frame=0.0, match_map=0.0, lba=0.0, map_upload=0.0, pose_opt=0.0)
    po_prob = synth.make_pose_opt_problem(n=450, seed=77 + rank)
    po1, po1_keep = views.pose_opt_problem(po_prob["Xw"], po_prob["u"], po_prob["v"], po_prob["ur"], po_prob["inv_sigma2"],
                                          po_prob["cam"], po_prob["Tcw"], device=device)
    po_prob2 = synth.make_pose_opt_problem(n=650, seed=78 + rank)
    po2, po2_keep = views.pose_opt_problem(po_prob2["Xw"], po_prob2["u"], po_prob2["v"], po_prob2["ur"], po_prob2["inv_sigma2"],
                                          po_prob2["cam"], po_prob2["Tcw"], device=device)
    for e in exs:
        e.set_profiling(2 if args.profile_stages else 1)
    ev_overhead_ms = ex.event_overhead_ms(100)
    FAST_BRACKET_EVERY = 1 if args.profile_stages else 4         # the event pair costs ~5 us of stream time: sample every 4th frame
    kern = dict(fast_kernel_ms=0.0, octree_host_ms=0.0)
    if args.profile_stages:
        kern.update(pyramid_ms=0.0, fast_ms=0.0, desc_ms=0.0, stereo_ms=0.0)
    stats = dict(kp=0, stereo=0, m_frame=0, m_map=0, lba_iters=0, lba_calls=0, lba_s=0.0)

    import queue
    import threading
    lba_q = queue.Queue()

    lba_out = views.LbaOutput(lp.n_poses, lp.n_points, lp.n_edges)      # result arrays are allocated once, like a SLAM system would

    def lba_worker():
        while True:
            job = lba_q.get()
            if job is None:
                lba_q.task_done()
                return
            t0 = time.perf_counter()
            out = opt.LocalBundleAdjustment(lp, out=lba_out)
            dt = time.perf_counter() - t0
            if job:
                stats["lba_iters"] += sum(out.iters); stats["lba_calls"] += 1; stats["lba_s"] += dt
            lba_q.task_done()

    async_state = dict(t0=None, timed=False)

    def collect_async():
        if async_state["t0"] is None:
            return
        out = opt.wait()
        if async_state["timed"]:
            stats["lba_iters"] += sum(out.iters); stats["lba_calls"] += 1
            stats["lba_s"] += opt.last_solve_ms * 1e-3
        async_state["t0"] = None

    worker = None
    if args.lba_mode == "thread":
        worker = threading.Thread(target=lba_worker, daemon=True)
        worker.start()

    amp_buf = np.full(8192, -1, np.int32); aob_buf = np.zeros(8192, np.int32)

    def step(i, timed):
        k, k_last = seq[i % len(seq)], seq[(i - 1) % len(seq)]
        fr = frames[k]
        dL, dR = imgs[k]
        t0 = time.perf_counter()
        Fc, exc = F, ex
        if pipeline:
            c = i & 1
            Fc, exc = Fs[c], exs[c]
            if not in_flight[c]:                          # first step only: nothing was submitted ahead
                exc.frame_stereo_dev_submit(Fc, fv, dL.data_ptr(), dR.data_ptr(), W, H, W, bf, bb)
            nl, nr = exc.frame_stereo_dev_wait()
            in_flight[c] = False
            nxt = imgs[seq[(i + 1) % len(seq)]]           # Frame::Frame(t+1) runs during the tracking of frame t
            exs[c ^ 1].frame_stereo_dev_submit(Fs[c ^ 1], fv, nxt[0].data_ptr(), nxt[1].data_ptr(), W, H, W, bf, bb)
            in_flight[c ^ 1] = True
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
        amp, aob, n1 = m_frame.SearchByProjectionFrame(Fc, fr["guess"], frames[k_last]["last_view"][0], 7.0, False, amp, aob, inplace=True)
        t4 = time.perf_counter()
        amp, aob, n2 = m_map.SearchLocalPoints(Fc, LM, fr["guess"], 1.0, False, 0.0, amp, aob, None, inplace=True)
        t5 = time.perf_counter()
        if args.pose_opt:
            # TrackWithMotionModel / TrackLocalMap call PoseOptimization after each search (S/Tracking.cc:2649,2712);
            # ~450 and ~650 correspondences as the two searches produce here
            opt.PoseOptimization(po1)
            opt.PoseOptimization(po2)
            tpo = time.perf_counter()
            if timed:
                stage["pose_opt"] += tpo - t5
        else:
            tpo = t5
        t6 = t7 = tpo
        if i % FRAMES_PER_KF == 0:
            mp = local_map_for(frames, k)
            wv, keep = views.worldpoints_view(mp["pos"], mp["normal"], mp["min_dist"], mp["max_dist"], mp["desc"], mp["n_obs"], mp["bad"])
            LM.upload(wv)
            t6 = time.perf_counter()
            if args.lba_mode == "async":
                collect_async()                        # the previous keyframe's LBA (long finished in steady state)
                async_state["t0"], async_state["timed"] = time.perf_counter(), timed
                opt.LocalBundleAdjustmentAsync(lp, lba_out)
                t7 = time.perf_counter()
            elif worker is not None:
                lba_q.put(timed)                       # LocalMapping thread picks the keyframe up
                t7 = t6
            else:
                out = opt.LocalBundleAdjustment(lp, out=lba_out)
                t7 = time.perf_counter()
                if timed:
                    stats["lba_iters"] += sum(out.iters); stats["lba_calls"] += 1; stats["lba_s"] += t7 - t6
        if timed:
            for key, dt in (("extract", t1 - t0), ("stereo", t2 - t1), ("grid", t3 - t2), ("match_frame", t4 - t3),
                            ("match_map", t5 - t4), ("map_upload", t6 - tpo), ("lba", t7 - t6)):
                stage[key] += dt
            if args.profile_stages:
                tm = exc.timings()
                for key in kern:
                    kern[key] += tm[key]
            stats["kp"] += nl + nr; stats["m_frame"] += n1; stats["m_map"] += n2

    # local map must exist before the first frame
    mp0 = local_map_for(frames, 0)
    wv0, keep0 = views.worldpoints_view(mp0["pos"], mp0["normal"], mp0["min_dist"], mp0["max_dist"], mp0["desc"], mp0["n_obs"], mp0["bad"])
    LM.upload(wv0)
    for i in range(args.warmup):
        step(i, False)
    lba_q.join()
    collect_async()

    def sync():
        lba_q.join()                                   # every LBA triggered inside the timed region has finished
        collect_async()
        for c in range(len(exs)):                      # ... and so has the frame constructor submitted by the last step
            if pipeline and in_flight[c]:
                exs[c].frame_stereo_dev_wait()
                in_flight[c] = False
        torch.cuda.synchronize()

    for e in exs:
        e.set_profile_interval(max(FAST_BRACKET_EVERY // len(exs), 1), reset=True)
    # barrier + synchronize, exactly K steps, synchronize + barrier, MAX over ranks (harness.AgentGroup.timed)
    elapsed = grp.timed(lambda i: step(args.warmup + i, True), args.steps, sync)
    # informational: one PoseOptimization call (not part of `value` unless --pose-opt)
    for _ in range(5):
        opt.PoseOptimization(po1)
    t0 = time.perf_counter()
    for _ in range(20):
        opt.PoseOptimization(po1)
    pose_opt_ms = 1e3 * (time.perf_counter() - t0) / 20
    if rank == 0:
        K = args.steps
        ms_per_step = 1e3 * elapsed / K
        # fast_cells_kernel is bracketed by a HIP event pair on the extractor's stream in every timed step; an EMPTY pair on
        # that stream already measures ev_overhead_ms (event-record commands are not free), so the kernel's launch
        # duration is the bracket minus that constant -- this is the figure that agrees with rocprofv3's kernel trace
        fast_sum, fast_n = 0.0, 0                              # bracket times accumulated inside the library over the timed region
        for e in exs:
            fs_, fn_ = e.fast_kernel_stats()
            fast_sum += fs_; fast_n += fn_
        fast_ms_raw = fast_sum / max(fast_n, 1)
        fast_ms = max(fast_ms_raw - ev_overhead_ms, 1e-6)
        fast_bytes = 2 * PYR_PIXELS_640x480 + (stats["kp"] / K) * 4.0     # both cameras' pyramid pixels + packed candidates
        achieved = fast_bytes / (fast_ms * 1e-3) / 1e9 if fast_ms > 0 else 0.0
        # HBM-side traffic of the roofline kernel: rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs of this
        # same command, KB per launch), committed under profiles/; byte-wide loads are not the "wide coalesced" case for
        # which gfx950 halves FETCH_SIZE, so no x2 correction is applied (calibration: profiles/README.md)
        traffic, traffic_src = None, None
        try:
            import glob
            pj = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_write_per_kernel.json")))[-1]   # newest round
            pm = json.load(open(pj))["fast_cells_kernel"]
            traffic = int(1024 * (pm["FETCH_SIZE_KB_avg"] + pm["WRITE_SIZE_KB_avg"]))
            traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)" % os.path.basename(pj)
        except Exception:
            pass
        line = {
            "metric": "tracking+localBA frames/sec (aggregate over agents; 1 agent per GPU)",
            "value": round(world * K / elapsed, 3),
            "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32 (ORB front-end, Hamming), f64 (local BA)",
            "data": "synthetic",
            "config": {"workload": "C2: 1 client stereo 640x480 synthetic, 1000 ORB feat/frame, 20-KF local BA window "
                                   "(20 free + 10 fixed KFs, 2000 points), 1 LBA per %d frames" % FRAMES_PER_KF,
                       "per_agent_fps": round(K / elapsed, 3), "frames_per_keyframe": FRAMES_PER_KF,
                       "stage_ms_per_frame": {k2: round(1e3 * v / K, 4) for k2, v in stage.items()},
                       "device_ms_per_frame": dict({k2: round(v / K, 4) for k2, v in kern.items()}, fast_kernel_ms=round(fast_ms_raw, 4)),
                       "avg_keypoints_per_stereo_frame": round(stats["kp"] / K, 1),
                       "avg_matches_frame": round(stats["m_frame"] / K, 1), "avg_matches_map": round(stats["m_map"] / K, 1),
                       "lba_mode": args.lba_mode, "pose_opt_in_step": bool(args.pose_opt), "cpu_affinity": cpu_affinity,
                       "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "completion_wait": wait_mode,
                       "frame_ctor": ("pipelined: Frame(t+1) is submitted on a second extractor handle before frame t is tracked and "
                                      "collected at the start of step t+1; the constructor left in flight by the last timed step is "
                                      "waited for inside the timed region") if pipeline else "synchronous",
                       "pose_opt_ms_per_call_450_correspondences": round(pose_opt_ms, 4),
                       "lba_ms_per_call": round(1e3 * stats["lba_s"] / max(stats["lba_calls"], 1), 3),
                       "sequential_fps_formula": round(1.0 / (sum(v for k2, v in stage.items() if k2 != "lba") / K +
                                                             stats["lba_s"] / max(stats["lba_calls"], 1) / FRAMES_PER_KF), 3),
                       "lba_lm_iterations_per_call": round(stats["lba_iters"] / max(stats["lba_calls"], 1), 2)},
            "roofline": {"kernel": "fast_cells_kernel", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(fast_bytes), "avg_launch_ms": round(fast_ms, 5),
                         "avg_launch_ms_event_bracket_raw": round(fast_ms_raw, 5), "event_pair_overhead_ms": round(ev_overhead_ms, 5),
                         "bracketed_launches": int(fast_n), "bracket_every_nth_frame": FAST_BRACKET_EVERY,
                         "note": "per-frame work is a few MB: the path is launch/latency bound, not bandwidth bound (SURVEY.md 0-10)"},
        }
        if not args.no_cpu_baseline:
            os.sched_setaffinity(0, affinity_at_start)     # the CPU baseline's three threads get the whole machine again
            line["cpu_baseline"] = cpu_baseline(scene, synth, views)
        print(json.dumps(line))
    grp.close()


if __name__ == "__main__":
    main()
