"""The C++ host side above the C-ABI (include/orbgpu_adapters.hpp: classes with the reference's names; include/orbgpu_dropin.hpp:
the INTEGRATION.md bodies with the reference's signatures) compiles with plain g++, links against liborbgpu.so, fails loudly
without a GPU, and -- through header-only mocks of Frame / KeyFrame / MapPoint / Map -- agrees with the CPU oracle on a GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _build(name, with_oracle=False):
    lib_dir = os.path.join(ROOT, "multi_orbslam3_amd")
    exe = os.path.join(CPP, name)
    # (opencv_branches: the HAVE_OPENCV lines of the adapters against the signature-only stub of tests/cpp/opencv_stub)
    stub = ["-I", os.path.join(CPP, "opencv_stub")] if name == "opencv_branches" else []
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CPP] + stub + [
           os.path.join(CPP, name + ".cpp"), "-o", exe, "-pthread", "-L", lib_dir, "-lorbgpu", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib",
           "-L/opt/rocm/lib"]
    if with_oracle:
        from oracle import binding as ob
        ob.build()
        odir = os.path.join(ROOT, "oracle")
        cmd += ["-L", odir, "-loracle", "-Wl,-rpath," + odir]
    subprocess.check_call(cmd)
    return exe


def _no_gpu():
    from multi_orbslam3_amd import _capi
    return _capi.load().orbg_device_count() <= 0


@pytest.mark.parametrize("name,with_oracle", [("adapter_smoke", False), ("dropin_parity", True), ("dropin_bench", False), ("opencv_branches", False), ("rig_loop", True)])
def test_host_side_compiles_links_and_fails_loudly_without_gpu(name, with_oracle):
    exe = _build(name, with_oracle)
    if not _no_gpu():
        pytest.skip("a GPU is present; see the gpu-marked tests")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "no usable HIP device" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_lba_window_cache_flattens_the_same_problems_as_the_uncached_glue():
    """tests/cpp/glue_cache_check (host only: the entry-point set records the flattened problem and returns a synthetic solve): six
    consecutive local-BA windows of one map with observations added and erased, points moved and made bad and a new keyframe between
    them -- the glue with its window cache (MapPoint::mnChangeStamp) and the glue reading every point produce
    byte-identical problems, and the cached run copies a third of the observation maps."""
    exe = _build("glue_cache_check")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALL OK" in r.stdout and r.stdout.count("identical") == 6, (r.stdout, r.stderr)


def test_ldlt_xcd_plan_and_schedule_for_every_size():
    """tests/cpp/ldlt_xcd_plan_check.hip (host only, built with hipcc): for all 176 system sizes the eight-workgroup LDL^T takes, every
    tile has exactly one wavefront slot, chain wavefronts hold the last four tiles of a column, and the kernel's row-by-row schedule
    -- replayed as WAIT / PUBLISH programs over the same flags -- runs to the end (no wait cycle), publishing every G and every panel
    tile exactly once."""
    exe = os.path.join(CPP, "ldlt_xcd_plan_check")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-o", exe, os.path.join(CPP, "ldlt_xcd_plan_check.hip")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "176 system sizes checked: ALL OK" in r.stdout, (r.stdout[-2000:], r.stderr[-500:])


def test_lba_glue_alone_is_timed_on_the_host():
    """tests/cpp/glue_cpu_bench: the LocalBundleAdjustment glue over the mocks with an entry-point set that returns at once."""
    import json
    exe = _build("glue_cpu_bench")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert 0 < d["lba_glue_cached_us"] < 0.7 * d["lba_glue_uncached_us"], d


@pytest.mark.gpu
def test_opencv_signature_branches_run_on_gpu():
    """ORBextractor::operator()(cv::InputArray, cv::InputArray, vector<cv::KeyPoint>&, cv::OutputArray, vector<int>&) -- the reference's exact
    signature, I/ORBextractor.h:61-63 -- and the cv::Mat helpers of the glue, compiled against tests/cpp/opencv_stub (no OpenCV in this
    image; the stub pins nothing about OpenCV): same features as the pointer overload, -1 on an empty image."""
    exe = _build("opencv_branches")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "identical to the pointer overload: 1" in r.stdout, (r.stdout, r.stderr)


@pytest.mark.gpu
def test_adapters_run_on_gpu():
    exe = _build("adapter_smoke")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)


@pytest.mark.gpu
def test_dropin_glue_matches_the_oracle_through_the_same_mocks():
    """Extractor / stereo Frame constructor, isInFrustum, SearchByProjection x2, SearchByBoW, PoseOptimization and
    LocalBundleAdjustment (graph collection, vToErase, 50 %-outlier early return, pbStopFlag raised before AND -- by a second
    thread, through the reference's own bool -- during the solve, SetPose / SetWorldPos lock flags, Map change index) with the
    reference's signatures; and both optimisers on keyframes / a Frame of the two-fisheye rig (mpCamera2, NLeft: KannalaBrandt8
    models, the right camera's ToBody edges), with and without the glue's window cache; Tracking::SearchLocalPoints,
    SearchByProjection(CurrentFrame, LastFrame) and SearchByBoW(KeyFrame, Frame) on a two-camera Frame of that rig (Nleft != -1: isInFrustum through either camera, the
    right camera's search blocks, stereo partners) -- assignments, visible counts and the track fields left in the map points."""
    exe = _build("dropin_parity", with_oracle=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "dropin parity ok" in r.stdout, (r.returncode, r.stdout[-3000:], r.stderr[-1000:])
    assert "LocalBundleAdjustment [two-fisheye rig]: status 0" in r.stdout and "PoseOptimization [two-fisheye rig]:" in r.stdout, r.stdout[-3000:]
    assert r.stdout.count("two-camera Frame (last frame") == 3, r.stdout[-3000:]      # the matcher's two-camera forms through the glue
    assert "ComputeStereoFishEyeMatches [two-fisheye rig]:" in r.stdout, r.stdout[-3000:]
    assert "monocular fisheye Frame:" in r.stdout, r.stdout[-3000:]


@pytest.mark.gpu
def test_rig_closed_loop_120_frames_product_and_oracle_each_feeding_on_their_own_outputs():
    """tests/cpp/rig_loop: 120 frames of a two-fisheye agent through the glue -- ComputeStereoFishEyeMatches, the motion model (frames 10,
    30, .. : TrackReferenceKeyFrame's SearchByBoW instead), SearchByProjection(Cur, Last), PoseOptimization, outliers dropped,
    SearchLocalPoints, PoseOptimization, mLastFrame, as Tracking::Track does, every eighth frame a keyframe and LocalBundleAdjustment over
    the last keyframes (write-back, erasures) -- each run carrying its OWN poses, matches, outlier decisions, keyframes and map points.
    (1) --shadow: every entry-point call of the product run (~730) repeated on the oracle with identical inputs: every match array, stereo
    partner array, frustum flag and level, outlier set, local-BA status and iteration count equal; poses 1e-4, local-BA state 5e-2.
    (2) Independent runs: KannalaBrandt8::project goes through the device's resp. the host libm's float32 atan2 / cos / sin, which differ
    in the last place (projections: 1e-4 px), so once in a few hundred frames a feature that close to a window's edge makes the runs
    differ by a match or two (here: frame 118, 2 entries) -- asked for: the runs stay together (at most 8 match entries in any frame,
    poses and keyframe poses 1e-3, points 5e-2), all local BAs applied, the agent keeps track of the truth."""
    import json
    exe = _build("rig_loop", with_oracle=True)
    r = subprocess.run([exe, "120", "--shadow"], capture_output=True, text=True, timeout=600)
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and rows, (r.returncode, r.stdout[-2000:], r.stderr[-500:])
    sh = json.loads(rows[-1])["rig_loop_shadow"]
    assert sh["ok"] and sh["mismatches"] == 0 and sh["calls"] > 700 and sh["max_projection_diff_px"] <= 3e-4, sh
    r = subprocess.run([exe, "120"], capture_output=True, text=True, timeout=600)
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and rows, (r.returncode, r.stdout[-2000:], r.stderr[-500:])
    d = json.loads(rows[-1])["rig_loop"]
    assert d["ok"] and d["frames"] == 120 and d["first_divergent_frame"] in (-1,) + tuple(range(40, 120)) and d["max_pose_diff"] <= 1e-3 and d["mean_inliers"] > 500, d
    assert d["max_pose_error_vs_truth"] < 0.05 and d["local_bas"] >= 14 and d["local_bas_applied"] >= 12 and d["max_keyframe_pose_diff"] <= 1e-3 and d["max_point_diff"] <= 5e-2, d
    assert d["max_match_entries_differing_in_a_frame"] <= 8, d


def test_closed_loop_scenario_tracks_on_the_oracle_alone():
    """tests/cpp/closed_loop --oracle-only (no GPU): the synthetic stereo agent of the closed-loop parity run -- every frame through the
    reference-signature glue over the CPU oracle, feeding on its own poses, matches, keyframes and local-BA results -- keeps track
    for 60 frames and 12 keyframes.  Pins the scenario itself (the scaffolding between the hot-path calls) where no GPU is needed."""
    exe = _build("closed_loop", with_oracle=True)
    r = subprocess.run([exe, "60", "5", "--oracle-only"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "oracle-only run: 12 keyframes, 10 applied local BAs" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-500:])


@pytest.mark.gpu
def test_closed_loop_200_frames_product_and_oracle_each_feeding_on_their_own_outputs():
    """tests/cpp/closed_loop: 200 frames / 40 keyframes of a stereo agent through the glue, once over liborbgpu (frames from the device
    constructor, the change-counter caches of the local map and of the local-BA window in use), once over liborbgpu as it compiles
    against an unmodified MapPoint (no caches), once over the CPU oracle; every run carries its OWN state from call to call as
    Tracking::Track / LocalMapping do (S/Tracking.cc:2572-2811, S/LocalMapping.cc:140-379): the motion model from its own poses,
    the last frame's points after its own outlier decisions, the local map from its own covisibility graph, new points, culling,
    fusions, and the local BA's write-back and erasures.
    (1) Shadow: every entry-point call of the product runs (~1900: searches, PoseOptimization, local BA, distinctive descriptors) is
    repeated on the oracle with identical inputs: match arrays / flags / counts / iteration counts exact, poses <= 1e-5 (measured:
    one float32 ulp), local-BA state <= 1e-4 (measured 1.5e-8).
    (2) Independent runs: every discrete digest equal on every one of the 200 frames (features, both match arrays, outlier flags,
    local-map make-up, new / culled / fused points, local-BA status / iterations / fixed keyframes, bad points, observation counts);
    poses and keyframe poses <= 1e-4, point state <= 5e-2 (an ulp fed back through a motion model and a local BA that stops short of
    convergence: tests/cpp/closed_loop.cpp's header has the measured spread)."""
    import json
    exe = _build("closed_loop", with_oracle=True)
    r = subprocess.run([exe, "200", "5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "closed loop ok" in r.stdout, (r.returncode, r.stdout[-3000:], r.stderr[-1000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])["closed_loop"]
    assert d["ok"] and d["frames"] == 200 and d["keyframes"] == 40 and d["local_bas_applied"] >= 30
    assert d["first_divergent_frame"] == -1 and d["first_divergent_frame_no_caches"] == -1
    assert d["shadow_calls"] >= 1800 and d["shadow_mismatches"] == 0 and d["shadow_max_pose_abs_diff"] <= 1e-6 and d["shadow_max_lba_abs_diff"] <= 1e-4
    assert d["bit_identical_leading_frames"] >= 10 and d["max_pose_abs_diff"] <= 1e-4


@pytest.mark.gpu
def test_dropin_bench_times_the_path_through_the_glue():
    """tests/cpp/dropin_bench: every call of the per-frame path and the local BA through the reference-signature glue over the
    mocks at C2 sizes; glue / upload / C-ABI time per call as one JSON line (bench.py surfaces it as value_dropin)."""
    import json
    exe = _build("dropin_bench")
    r = subprocess.run([exe, "12"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    rows = d["dropin_bench"]
    assert len(rows) == 9 and d["frames_per_s_frame_path"] > 0 and d["local_map_points"] > 2000
    assert 0 < d["frames_per_s_frame_path_with_pose_opt"] < d["frames_per_s_frame_path"]
    # the local BA's window cache: consecutive windows of one map cost the glue less than half of a first window
    lba_next = [r for n, r in rows.items() if "consecutive windows" in n][0]
    lba_first = [r for n, r in rows.items() if "first window of a map" in n][0]
    assert lba_next["glue_us"] < 0.5 * lba_first["glue_us"], (lba_next, lba_first)
    # the glue's cache of the flattened local map: on a frame whose local map is the previous frame's nothing is cloned or uploaded
    cached = [r for n, r in rows.items() if "unchanged since the last frame" in n][0]
    fresh = [r for n, r in rows.items() if "after a keyframe" in n][0]      # (the local BA moved four points in five: those are re-read)
    assert cached["glue_us"] < 0.5 * fresh["glue_us"], (cached, fresh)
    for name, row in rows.items():
        assert row["total_us"] > 0 and row["c_abi_us"] > 0 and 0 <= row["glue_frac"] < 1.0, (name, row)
