"""N>1 path of the harness on CPU: 2 gloo ranks (world_size 2).  Agents are independent units -- one process per
agent, no data-path collective; torch.distributed only provides the barriers and the MAX over
ranks of the elapsed time.  The per-agent work here is the CPU oracle on a tiny image (the HIP path needs a GPU)."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import json, os, sys, time, hashlib
    sys.path.insert(0, %r)
    import numpy as np
    from multi_orbslam3_amd import harness, synth
    from oracle import binding as ob
    grp = harness.AgentGroup("gloo")
    seed = grp.agent_seed(synth.SEED_IMAGES)
    sc = synth.Scene(160, 120, seed=seed, tex_size=(400, 300), px_per_m=50.0)
    ex = ob.Extractor(n_features=200, n_levels=4, max_width=160, max_height=120)
    imgs = [sc.stereo_pair(k)[0] for k in range(3)]
    digest = hashlib.sha256()
    def step(i):
        rc, k, d, _ = ex.extract(imgs[i %% 3])
        digest.update(k.tobytes()); digest.update(d.tobytes())
        if grp.rank == 1:
            time.sleep(0.02)          # a slower agent: the job time must be the MAX over ranks
    elapsed = grp.timed(step, 6)
    out = dict(rank=grp.rank, world=grp.world, seed=seed, elapsed=elapsed, rate=grp.aggregate_rate(6, elapsed),
               digest=digest.hexdigest())
    print("RESULT " + json.dumps(out), flush=True)
    grp.close()
''') % ROOT


def _run_ranks(world, port):
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=240)
        assert p.returncode == 0, se[-2000:]
        line = [l for l in so.splitlines() if l.startswith("RESULT ")][-1]
        outs.append(json.loads(line[7:]))
    return sorted(outs, key=lambda o: o["rank"])


def test_two_agents_gloo_no_cross_talk_and_max_time():
    two = _run_ranks(2, 29611)
    assert [o["world"] for o in two] == [2, 2]
    assert two[0]["seed"] != two[1]["seed"] and two[0]["digest"] != two[1]["digest"]      # distinct agents
    # both ranks report the same job time (MAX over ranks) and it includes the slow agent's sleeps
    assert abs(two[0]["elapsed"] - two[1]["elapsed"]) < 1e-9
    assert two[0]["elapsed"] >= 6 * 0.02
    assert abs(two[0]["rate"] - 2 * 6 / two[0]["elapsed"]) < 1e-9                         # whole-job aggregate
    # an agent's outputs do not depend on how many other agents run: rank 0 alone == rank 0 of the pair
    one = _run_ranks(1, 29612)
    assert one[0]["world"] == 1 and one[0]["digest"] == two[0]["digest"]


EXCHANGE_WORKER = textwrap.dedent('''
    import json, os, sys, hashlib
    sys.path.insert(0, %r)
    import numpy as np, torch
    from multi_orbslam3_amd import harness, synth
    from oracle import binding as ob
    grp = harness.AgentGroup("gloo")
    sc = synth.Scene(160, 120, seed=grp.agent_seed(synth.SEED_IMAGES), tex_size=(400, 300), px_per_m=50.0)
    ex = ob.Extractor(n_features=150 + 60 * grp.rank, n_levels=4, max_width=160, max_height=120)      # ragged block sizes
    rc, k, d, _ = ex.extract(sc.stereo_pair(1)[0])
    wire = ob.wire_pack(k, d)
    got = grp.all_gather_keyframes(torch.from_numpy(wire.copy()), len(k))
    out = dict(rank=grp.rank, n=len(k), own=hashlib.sha256(wire.tobytes()).hexdigest(), blocks=[])
    for r, (n_r, blk) in enumerate(got):
        blk = blk.numpy()
        kk, dd = ob.wire_unpack(blk, n_r)
        out["blocks"].append(dict(n=n_r, sha=hashlib.sha256(blk.tobytes()).hexdigest(), octaves=int(kk["octave"].sum()),
                                  desc=hashlib.sha256(dd.tobytes()).hexdigest()))
    if grp.rank == 0:
        out["roundtrip_ok"] = bool(np.array_equal(got[0][1].numpy(), wire))
    # one server tick with SEVERAL new keyframes per agent (rank r contributes r + 2 blocks of different sizes) in ONE collective
    mine = []
    for j in range(grp.rank + 2):
        rc, kj, dj, _ = ex.extract(sc.stereo_pair(2 + j)[0])
        mine.append((len(kj), torch.from_numpy(ob.wire_pack(kj, dj).copy())))
    calls = []
    real = grp.dist.all_gather_into_tensor
    grp.dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    assert grp.open_data_plane() is None                  # gloo data plane: the default group (and a barrier on every rank)
    bufs = grp.tick_buffers(max_features=400, device="cpu", max_blocks=4)
    tick = grp.all_gather_keyframe_blocks(bufs, mine)
    tick2 = grp.all_gather_keyframe_blocks(bufs, mine[:1])          # the buffers are reused tick after tick
    grp.dist.all_gather_into_tensor = real
    out["tick_collectives"] = len(calls)
    out["tick_own"] = [[n, hashlib.sha256(w.numpy().tobytes()).hexdigest()] for n, w in mine]
    out["tick"] = [[[n, hashlib.sha256(b.numpy().tobytes()).hexdigest()] for n, b in lst] for lst in tick]
    out["tick2_counts"] = [len(lst) for lst in tick2]
    print("RESULT " + json.dumps(out), flush=True)
    grp.close()
''') % ROOT


def test_keyframe_blocks_all_gather_gloo():
    """The one exchange of the system (row f-4 / 8e): every agent receives every agent's keyframe wire block, ragged sizes."""
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29613")
        procs.append(subprocess.Popen([sys.executable, "-c", EXCHANGE_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=240)
        assert p.returncode == 0, se[-2000:]
        outs.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][-1][7:]))
    outs.sort(key=lambda o: o["rank"])
    assert outs[0]["n"] != outs[1]["n"] and min(o["n"] for o in outs) > 30
    for o in outs:                                   # both ranks hold both blocks, byte-identical to what their owners packed
        assert [b["n"] for b in o["blocks"]] == [outs[0]["n"], outs[1]["n"]]
        assert [b["sha"] for b in o["blocks"]] == [outs[0]["own"], outs[1]["own"]]
    assert outs[0]["blocks"] == outs[1]["blocks"] and outs[0]["roundtrip_ok"]
    # the batched tick: 2 + 3 blocks, ONE all-gather per tick, every rank holds every block byte for byte
    for o in outs:
        assert o["tick_collectives"] == 2                                  # two ticks, one collective each
        assert o["tick"] == [outs[0]["tick_own"], outs[1]["tick_own"]]
        assert o["tick2_counts"] == [1, 1]
    assert [len(o["tick_own"]) for o in outs] == [2, 3]


def test_cpu_list_parsing_and_pinning_helpers_degrade_without_a_gpu():
    from multi_orbslam3_amd import harness
    assert harness._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert harness._parse_cpulist("") == set()
    # no GPU / no sysfs topology here: both helpers report "not available" instead of raising
    assert harness.pin_to_gpu_numa_node(0) is None
    assert harness.core_pair_for_agent(0, 0) is None


def test_agents_of_one_node_never_share_a_core_pair():
    from multi_orbslam3_amd import harness
    for ncores in (2, 3, 8, 16, 17, 64, 96):
        cores = [{c, c + 128} for c in range(ncores)]
        for n_local in (1, 2, 4, 8):
            got = [harness._pick_core_pair(cores, slot, n_local) for slot in range(n_local)]
            if ncores // 2 < n_local:
                assert all(g is None for g in got)
                continue
            flat = [frozenset(c) for g in got for c in g]
            assert len(set(flat)) == 2 * n_local                     # all distinct physical cores
            if (ncores - ncores // 2) // 2 >= n_local:                  # room in the upper half: housekeeping cores stay free
                assert min(min(c) for c in flat) >= ncores // 2
    # a single agent on a big node gets the first pair of the upper half
    assert harness._pick_core_pair([{c} for c in range(64)], 0, 1) == ({32}, {33})


def test_a_lone_agent_takes_the_idle_cores_of_one_l3_slice():
    from multi_orbslam3_amd import harness
    cores = [{c, c + 64} for c in range(32)]
    # idle machine: the last slice, its first three cores
    got = harness._pick_idle_cores(cores, {}, 3)
    assert got == ({24, 88}, {25, 89}, {26, 90})
    # a neighbour spins on the sibling thread of core 25 and on core 26: the slice is still the best one with three idle cores
    busy = {89: 0.95, 26: 0.9}
    got = harness._pick_idle_cores(cores, busy, 3)
    assert got == ({24, 88}, {27, 91}, {28, 92})
    # the whole upper slice busy: the next idle slice is taken; cores stay inside ONE slice
    busy = {c: 0.8 for c in range(24, 32)}
    got = harness._pick_idle_cores(cores, busy, 3)
    assert got == ({16, 80}, {17, 81}, {18, 82})
    assert harness._pick_idle_cores(cores[:2], {}, 3) is None
    b = harness._cpu_busy(0.02)
    assert b and all(0.0 <= v <= 1.0 for v in b.values())


def test_cgroup_quota_and_throttle_readers(monkeypatch, tmp_path):
    """harness.cgroup_cpu_quota / cgroup_throttled: cgroup v2 `cpu.max` (`max 100000` = no quota, `1600000 100000` = 16 CPUs) and
    `cpu.stat`; v1 files as the fall-back; unreadable files give None (bench.py then reports null, never fails)."""
    import builtins
    from multi_orbslam3_amd import harness
    files = {}
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if isinstance(path, str) and path.startswith("/sys/fs/cgroup"):
            if path in files:
                f = tmp_path / ("f%d" % (abs(hash(path)) % 100000))
                f.write_text(files[path])
                return real_open(f, *a, **k)
            raise OSError("no such cgroup file")
        return real_open(path, *a, **k)

    monkeypatch.setattr(builtins, "open", fake_open)
    assert harness.cgroup_cpu_quota() is None and harness.cgroup_throttled() is None
    files["/sys/fs/cgroup/cpu.max"] = "max 100000\n"
    assert harness.cgroup_cpu_quota() is None
    files["/sys/fs/cgroup/cpu.max"] = "1600000 100000\n"
    assert harness.cgroup_cpu_quota() == 16.0
    files["/sys/fs/cgroup/cpu.stat"] = "usage_usec 5\nnr_periods 449\nnr_throttled 289\nthrottled_usec 1442068891\n"
    assert harness.cgroup_throttled() == (289, 1442068891)
    del files["/sys/fs/cgroup/cpu.max"], files["/sys/fs/cgroup/cpu.stat"]
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "250000\n"; files["/sys/fs/cgroup/cpu/cpu.cfs_period_us"] = "100000\n"
    files["/sys/fs/cgroup/cpu/cpu.stat"] = "nr_periods 10\nnr_throttled 3\nthrottled_time 7000000\n"
    assert harness.cgroup_cpu_quota() == 2.5 and harness.cgroup_throttled() == (3, 7000)
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "-1\n"
    assert harness.cgroup_cpu_quota() is None


def test_a_failing_server_thread_releases_its_buffers_and_the_error_reaches_the_tick():
    """harness.ServerTick: rank 0 hands every tick's gathered blocks to a server thread and, two ticks later, waits for that job's
    event before it reuses the buffer set -- in front of an all-gather the other ranks are already in.  A job that raises (or that is
    skipped because an earlier one raised) must still release its event, and the next tick must re-raise the error after its
    collective instead of blocking (no GPU: the thread's machinery alone, with a server whose work fails on the second job)."""
    import queue
    import threading
    import time

    from multi_orbslam3_amd import harness

    st = object.__new__(harness.ServerTick)
    st.np = np
    st.q, st.error, st.is_server, st.shared_gpu = queue.Queue(), None, True, False
    st.stats = dict(ticks=0, exchange_s=0.0)
    st.pending, st.busy, st.bufs2 = [], [None, None], [None, None]
    calls = []

    def process(got, done):
        calls.append(got)
        if got == "job1":
            raise RuntimeError("server work failed")
    st._process = process

    class Grp:
        def all_gather_keyframe_blocks(self, bufs, blocks, with_poses=True):
            return "job%d" % st.stats["ticks"]
    st.grp = Grp()
    st.thread = threading.Thread(target=st._serve, daemon=True)
    st.thread.start()
    st.tick()                                  # job0: fine
    st.tick()                                  # job1: raises in the server thread
    t0 = time.time()
    raised = []

    def third():
        try:
            st.tick()                          # waits for job0's event (set), exchanges, then sees the error
        except RuntimeError as e:
            raised.append(str(e))
    th = threading.Thread(target=third, daemon=True)
    th.start()
    th.join(10.0)
    if not raised:                             # the error may land just after the third tick's check: the fourth one sees it for sure
        st.q.join()
        th = threading.Thread(target=third, daemon=True)
        th.start()
        th.join(10.0)
    assert not th.is_alive() and raised == ["server work failed"], (raised, time.time() - t0)
    assert st.busy[0].is_set() or st.busy[1].is_set()
    st.q.join()
    assert all(ev is None or ev.is_set() for ev in st.busy)      # every job handed over -- processed, failed or skipped -- released its buffers
    assert calls[:2] == ["job0", "job1"] and "job2" not in calls  # jobs after the failure are skipped, not run on a broken server
    with __import__("pytest").raises(RuntimeError):
        st.drain()
    st.q.put(None)
