"""bench.py prints ONE JSON line with the fields the driver reads (metric / value / unit / ..., roofline, cpu_baseline),
the roofline describes the kernel that dominates a fresh kernel trace, and the other configurations produce a line too."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(extra), capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    d = _bench("--steps", "40", "--warmup", "10", "--secondary-steps", "20")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "timed_region_s", "step_ms_p50", "step_ms_p95",
                "value_device_images", "value_sync_ctor_host_images", "value_with_pose_opt", "fps_formula", "prewarm_steps"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 10
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) < 0.01 * d["value"]
    assert abs(d["timed_region_s"] - 40 * d["ms_per_step"] * 1e-3) < 0.02 * d["timed_region_s"]
    assert 0 < d["step_ms_p50"] <= d["step_ms_p95"]
    assert 0 < d["value_sync_ctor_host_images"] and 0 < d["value_device_images"] and 0 < d["value_with_pose_opt"] < d["value"] * 1.2
    assert d["prewarm_steps"] >= 200 and d["fps_formula"] > 0
    cfgd = d["config"]
    # `value` is the host-image accounting: the images enter through pinned staging + a PCIe copy inside the step
    assert cfgd["host_images_in_step"] is True and "pinned staging" in cfgd["image_ingest"]
    assert cfgd["local_map_keyframes"] == 20 and 2000 <= cfgd["local_map_points_avg"] <= 12000 and cfgd["sequence_frames"] >= 190
    assert d["data"] == "synthetic" and "workload" in cfgd and "model" not in cfgd and cfgd["name"] == "C2"
    assert cfgd["device_copy_GBps_measured"] > 500 and cfgd["host_cpu"]["nproc"] >= 1 and cfgd["host_cpu"]["model"]
    assert cfgd["whole_step_hbm"]["achieved_GBps"] > 0
    rf = d["roofline"]
    for key in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "bracketed_launches"):
        assert key in rf, key
    assert rf["bound"] in ("hbm", "mfma") and rf["peak"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6
    assert rf["bracketed_launches"] >= 2 and 0.002 < rf["avg_launch_ms"] < 1.0      # (one solve in four carries the event pair)
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "build", "host_cpu"):
        assert key in cb, key
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1


@pytest.mark.gpu
def test_roofline_kernel_is_the_top_row_of_a_fresh_kernel_trace():
    """The kernel the roofline describes must be the one with the largest total time in a rocprofv3 kernel trace of the same
    command (serial accounting, so that every kernel of the step is in one trace)."""
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not on PATH")
    out = tempfile.mkdtemp(prefix="orbg_trace_")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "run", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "10", "--no-cpu-baseline", "--no-secondary",
           "--lba-mode", "inline", "--no-pipeline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd="/tmp", env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    stats = glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True)
    assert stats, os.listdir(out)
    rows = list(csv.DictReader(open(stats[0])))
    # PoseOptimization runs only in the informational tail of bench.py (not in a step): leave it out of the ranking
    rows = [x for x in rows if "pose_opt_kernel" not in x["Name"]]
    # copies are not kernels of the library
    rows = [x for x in rows if "__amd_rocclr" not in x["Name"]]
    rows.sort(key=lambda x: -float(x["TotalDurationNs"]))
    # bench.py picks the subject from its own event brackets; two kernels within 15 % of each other in total time (at the
    # moment octree_kernel and the LDL^T) may swap places between two runs, so the subject has to be the top row or tied with it
    name = d["roofline"]["kernel"].split("::")[-1]
    mine = [x for x in rows if name in x["Name"]]
    assert mine, (name, [x["Name"][:60] for x in rows[:4]])
    assert float(mine[0]["TotalDurationNs"]) >= 0.85 * float(rows[0]["TotalDurationNs"]), (name, [(x["Name"][:50], x["TotalDurationNs"]) for x in rows[:4]])
    # and the live event-bracket duration agrees with the trace's average for that kernel
    avg_us = float(mine[0]["AverageNs"]) / 1e3
    assert abs(1e3 * d["roofline"]["avg_launch_ms"] - avg_us) < 0.25 * avg_us, (d["roofline"]["avg_launch_ms"], avg_us)
    per = d["config"]["device_ms_per_step_by_kernel"]
    assert list(per)[0].split("::")[-1] == name and len(per) >= 5
    shutil.rmtree(out, ignore_errors=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["C4", "mono"])
def test_other_configurations_produce_a_line(name):
    d = _bench("--config", name, "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--secondary-steps", "10")
    assert d["config"]["name"] == name and d["value"] > 0 and d["config"]["lba_ms_per_call"] > 0
    assert d["roofline"]["bracketed_launches"] >= 1
    if name == "C4":
        assert d["roofline"]["unknowns"] == 300 and "k_ldlt_big48" in d["roofline"]["kernel"]
    else:
        assert d["config"]["host_images_in_step"] is True


@pytest.mark.gpu
def test_server_tick_goes_through_rccl_on_one_rank():
    d = _bench("--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-secondary", "--server-tick")
    st = d["config"]["server_tick"]
    assert st["2_kf_blocks_us"] > 0 and st["8_kf_blocks_us"] > st["2_kf_blocks_us"] and "nccl" in st["note"]
    assert 0 < st["exchange_only_8_blocks_us"] < st["8_kf_blocks_us"] and "ONE RCCL all-gather" in st["note"]
