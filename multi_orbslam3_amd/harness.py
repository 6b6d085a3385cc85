"""Multi-agent harness: one process per GPU / agent, no data-path collective on the per-frame hot path.

`torch.distributed` is the control plane: a barrier on both sides of the timed region and the MAX over ranks of the elapsed
time.  The ONE exchange the system has -- keyframe wire blocks going to the server-side matcher (row f-4, configs 3/5) -- is
`all_gather_keyframes`: an all-gather of the agents' new KF blocks (47 bytes per feature, a few blocks per server tick, well
under 1 MB: latency bound, so a single padded all-gather and no ring tuning).  backend "nccl" is RCCL over xGMI on ROCm; the CPU
tests use "gloo".
"""
import os
import time


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def count_gpus_without_runtime():
    """GPUs of this node as the KFD topology lists them (/sys/class/kfd/kfd/topology/nodes/*/properties: a node with simd_count > 0 is
    a GPU), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set -- WITHOUT loading the HIP runtime: a launcher process
    that starts its ranks as children must not have touched the GPU (on this pool a process that has may not replace itself, and
    the count must not depend on what torch.cuda.device_count() does or does not initialise).  None when sysfs does not show it."""
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        n = 0
        for node in sorted(os.listdir(base)):
            props = {}
            with open(os.path.join(base, node, "properties")) as f:
                for ln in f:
                    kv = ln.split()
                    if len(kv) == 2:
                        props[kv[0]] = kv[1]
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except Exception:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [x for x in v.split(",") if x.strip() != ""]
            n = min(n, len(ids))
    return n


def gpu_placement(device_index):
    """Where an agent runs: its GPU's PCI address and NUMA node, and the CPUs the process is allowed on right now (a dict for the
    bench line's per-rank table; fields are None where the topology is not visible)."""
    out = {"device": int(device_index), "pci": None, "numa_node": None, "cpus": sorted(os.sched_getaffinity(0))}
    try:
        import torch
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        out["pci"] = bdf
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            out["numa_node"] = int(f.read().strip())
    except Exception:
        pass
    return out


def pin_to_gpu_numa_node(device_index):
    """Restrict this process (and the threads it creates afterwards: the library's LBA worker) to the CPUs of the NUMA node
    the GPU hangs off -- what `numactl --cpunodebind` does for a deployed agent.  The per-frame path spins on words in pinned
    host memory and rings PCIe doorbells; from the other socket every one of those is a cross-socket round trip (measured
    on the 2-socket bench box: searches 47 -> 65-74 us per call when the scheduler places the thread there).
    Returns a short description, or None when the topology is not visible (containers without sysfs, single-node hosts)."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            cpus = _parse_cpulist(f.read())
        allowed = os.sched_getaffinity(0) & cpus
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return "numa node %d of GPU %s (%d cpus)" % (node, bdf, len(allowed))
    except Exception:                                     # topology not visible: run unpinned
        return None


def _pick_cores(cores, slot, n_local, per_agent):
    """cores: the physical cores of the node (sets of hardware threads), in order.  The first cores of a node serve
    interrupts and housekeeping, so agents take consecutive runs of `per_agent` cores from the upper half.  Two agents must
    never share a core (their spinning threads would halve each other): when the node has fewer runs than the launch has
    local ranks (all of which may sit on this node) every core is used, and when that is not enough either nothing is pinned."""
    n_local = max(n_local, slot + 1)
    half = len(cores) // 2
    if (len(cores) - half) // per_agent < n_local:
        half = 0
    if (len(cores) - half) // per_agent < n_local:
        return None
    base = half + per_agent * slot
    return tuple(cores[base + i] for i in range(per_agent))


def _pick_core_pair(cores, slot, n_local):
    return _pick_cores(cores, slot, n_local, 2)


def _physical_cores_of_gpu_node(device_index):
    import torch
    pr = torch.cuda.get_device_properties(device_index)
    bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
        node = int(f.read().strip())
    if node < 0:
        node = 0
    with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
        cpus = _parse_cpulist(f.read()) & os.sched_getaffinity(0)
    cores, seen = [], set()
    for c in sorted(cpus):
        if c in seen:
            continue
        with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
            sib = _parse_cpulist(f.read()) & cpus
        seen |= sib
        cores.append(sib)
    return cores


def _cpu_busy(interval=0.12):
    """Busy fraction of every hardware thread over `interval` seconds (two readings of /proc/stat)."""
    def read():
        out = {}
        with open("/proc/stat") as f:
            for line in f:
                if line.startswith("cpu") and line[3].isdigit():
                    p = line.split()
                    v = [int(x) for x in p[1:9]]
                    out[int(p[0][3:])] = (sum(v), v[3] + v[4])
        return out
    a = read()
    time.sleep(interval)
    b = read()
    busy = {}
    for c, (tot, idle) in a.items():
        dt = b[c][0] - tot
        busy[c] = 0.0 if dt <= 0 else 1.0 - (b[c][1] - idle) / dt
    return busy


def _pick_idle_cores(cores, busy, per_agent, group=8):
    """A lone agent on a SHARED host (the bench boxes run four tenants on one 2-socket machine) takes the `per_agent` least
    busy physical cores of the least busy group of `group` consecutive cores (one L3 slice: the three threads hand words to
    each other) instead of a fixed run: a neighbour's job sitting on the fixed run costs a third of the frame rate
    (8270-8340 frames/s on six runs of one box, 5650 on a seventh with the local BA worker sharing a core).  `cores`: sets of
    hardware threads per physical core, in order; `busy`: hardware thread -> busy fraction.  Returns None when there are
    fewer than `per_agent` cores."""
    if len(cores) < per_agent:
        return None
    load = [max(busy.get(t, 0.0) for t in c) for c in cores]
    best = None
    for g0 in range(0, len(cores), group):
        idx = sorted(range(g0, min(g0 + group, len(cores))), key=lambda i: (load[i], i))[:per_agent]
        if len(idx) < per_agent:
            continue
        key = (round(sum(load[i] for i in idx), 2), -g0)             # ties: the upper groups (the first cores serve interrupts)
        if best is None or key < best[0]:
            best = (key, idx)
    if best is None:
        return None
    return tuple(cores[i] for i in sorted(best[1]))


def cores_for_agent(device_index, slot, per_agent=3):
    """`per_agent` distinct physical cores (each returned with its SMT siblings) on the GPU's NUMA node for agent number
    `slot` on that node: tracking thread, the library's local-BA worker, the library's image-ingest thread.  All three spin
    on completion / hand-over words; when the scheduler happens to put two of them on the hardware threads of one core each
    runs at about half speed (searches 47 -> 65-74 us per call, observed in roughly one run out of five).  Falls back to a
    pair (the ingest thread then floats on the node) and returns None when the topology is not visible."""
    try:
        cores = _physical_cores_of_gpu_node(device_index)
        n_local = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
        if n_local == 1 and slot == 0:
            got = _pick_idle_cores(cores, _cpu_busy(), per_agent)
            if got is not None:
                return got
        got = _pick_cores(cores, slot, n_local, per_agent)
        if got is None and per_agent > 2:
            got = _pick_cores(cores, slot, n_local, 2)
        return got
    except Exception:
        return None


def core_pair_for_agent(device_index, slot):
    """Two distinct physical cores for agent `slot` (tracking thread, local-BA worker); see cores_for_agent."""
    got = cores_for_agent(device_index, slot, per_agent=2)
    return None if got is None else (got[0], got[1])


class AgentGroup:
    def __init__(self, backend=None, device_index=None, force_group=False):
        """force_group: create the process group even for a single rank (exercises the RCCL path of
        all_gather_keyframes on one GPU)."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.data_backend = backend          # backend of the DATA plane (keyframe exchange); the control plane is gloo, always
        self._data_group = None
        if self.world > 1 or force_group:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            self.data_backend = backend
            # Control plane (barriers around the timed region, MAX of the elapsed times, per-agent statistics) on gloo, always:
            # agents have no data-path collective, and an RCCL communicator in the process -- its streams and hardware queues --
            # costs an agent a third of its frame rate from the first barrier on (8400 -> 5450 frames/s with a 1-rank nccl group
            # that only ever ran barriers; csrc/common.hpp on hardware queues).  The RCCL group is the DATA plane of the server
            # tick and is created by open_data_plane() -- a collective of its own, called by every rank before the first exchange.
            self._device_index = self.local_rank if device_index is None else device_index
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.dist = dist
            self._data_group = None

    @property
    def backend(self):                       # (kept for callers of the earlier name)
        return self.data_backend

    def open_data_plane(self):
        """COLLECTIVE: creates the group of the data-path exchange (keyframe wire blocks) -- RCCL over xGMI when the data backend
        is "nccl"; with "gloo" (CPU tests) the exchange uses the default group and this is a barrier.  Every rank must call it,
        before its first exchange and outside any timed region (communicator set-up takes hundreds of milliseconds); the
        exchanges themselves never create a group, so a rank that skips one exchange cannot hang the others in new_group."""
        if self.dist is None:
            return None
        if self.data_backend == "nccl" and self._data_group is None:
            import torch
            self._data_group = self.dist.new_group(backend="nccl", device_id=torch.device("cuda", self._device_index))
        self.dist.barrier()
        return self._data_group

    def data_group(self, tensor=None):
        """The group an exchange of `tensor` goes through: the RCCL group for GPU tensors (open_data_plane() must have been
        called), the default gloo group for CPU tensors.  GPU tensors without an RCCL data plane are an error, not a silent
        fall-through to gloo."""
        if self.dist is None:
            return None
        on_gpu = tensor is not None and tensor.device.type == "cuda"
        if not on_gpu:
            return None
        if self.data_backend != "nccl":
            raise RuntimeError("GPU tensors need the nccl (RCCL) data plane; this group was created with data backend %r" % (self.data_backend,))
        if self._data_group is None:
            raise RuntimeError("call AgentGroup.open_data_plane() on every rank before the first keyframe exchange")
        return self._data_group

    def agent_seed(self, base):
        """Agents are independent clients: distinct seeds, no shared state."""
        return base + self.rank

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        import torch
        t = torch.tensor([seconds], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, value):
        """One float per rank -> list over ranks (on every rank)."""
        if self.dist is None:
            return [float(value)]
        import torch
        mine = torch.tensor([float(value)], dtype=torch.float64)
        out = [torch.zeros(1, dtype=torch.float64) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [float(v[0]) for v in out]

    def gather_objects(self, obj):
        """One small picklable object per rank -> list over ranks (on every rank; control plane)."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def timed(self, fn, steps, sync=None):
        """barrier + sync, exactly `steps` calls of fn(i), sync + barrier; returns MAX-over-ranks seconds."""
        self.barrier()
        if sync:
            sync()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(i)
        if sync:
            sync()
        dt = time.perf_counter() - t0          # (the closing barrier's own latency is not step time: MAX over ranks of the ranks' times)
        self.barrier()
        return self.max_over_ranks(dt)

    def aggregate_rate(self, steps, seconds):
        """Whole-job throughput: every rank did `steps` units in the (max) time."""
        return self.world * steps / seconds

    def all_gather_keyframes(self, wire, n_features):
        """Every agent contributes one keyframe wire block (torch uint8 tensor of 47 * n_features bytes, on the GPU with the
        nccl backend, on the CPU with gloo); returns [(n_features_r, block_r)] for all ranks r, block_r a view into the
        gathered buffer on the same device.  Two collectives: the feature counts (8 bytes per rank), then the blocks padded to
        the largest one."""
        import torch
        dev = wire.device
        if self.dist is None:
            return [(int(n_features), wire[: 47 * int(n_features)])]
        counts = torch.zeros(self.world, dtype=torch.int64, device=dev)
        mine = torch.tensor([int(n_features)], dtype=torch.int64, device=dev)
        self.dist.all_gather_into_tensor(counts, mine, group=self.data_group(mine))
        counts = [int(c) for c in counts.cpu()]
        pad = 47 * max(max(counts), 1)
        send = torch.zeros(pad, dtype=torch.uint8, device=dev)
        send[: 47 * int(n_features)] = wire[: 47 * int(n_features)]
        recv = torch.empty(self.world * pad, dtype=torch.uint8, device=dev)
        self.dist.all_gather_into_tensor(recv, send, group=self.data_group(send))
        return [(counts[r], recv[r * pad: r * pad + 47 * counts[r]]) for r in range(self.world)]

    # ---- server tick: ONE collective per tick, whatever the number of ranks and of new keyframes
    TICK_MAX_BLOCKS = 15                     # new keyframes one agent can contribute per tick
    TICK_HEADER_BYTES = 64                   # int32[16]: number of blocks, then the feature count of each block
    TICK_POSE_BYTES = 64                     # float32[16] per block: the keyframe's Tcw (KF.msg carries the pose next to the features)

    def tick_buffers(self, max_features, device, max_blocks=8):
        """Persistent send / receive buffers of a server tick for up to `max_blocks` keyframes of up to `max_features`
        features per agent: [64-byte header | the agent's wire blocks back to back].  The size is fixed for the group's life,
        so a tick needs no size exchange."""
        import torch
        assert 1 <= max_blocks <= self.TICK_MAX_BLOCKS
        cap = self.TICK_HEADER_BYTES + self.TICK_POSE_BYTES * int(max_blocks) + 47 * int(max_features) * int(max_blocks)
        cap = (cap + 255) & ~255
        send = torch.zeros(cap, dtype=torch.uint8, device=device)
        recv = torch.zeros(self.world * cap, dtype=torch.uint8, device=device)
        return dict(cap=cap, send=send, recv=recv, max_blocks=int(max_blocks), max_features=int(max_features))

    def all_gather_keyframe_blocks(self, bufs, blocks, with_poses=False):
        """One server tick: every agent contributes ALL its new keyframe wire blocks -- `blocks` = [(n_features, uint8 tensor
        of 47 * n_features bytes[, Tcw as 16 float32])], possibly empty -- in ONE all-gather of the fixed-size tick buffers (RCCL
        over xGMI on GPU tensors, gloo on CPU tensors; < 1 MB per agent: latency bound).  The per-block feature counts and poses
        travel in the buffer's header, so there is no second collective and no host round trip between collectives; the headers of
        all agents are read back with one small copy.  Returns [[(n_features, block view[, Tcw 4 x 4]), ...] for every rank]."""
        import numpy as np
        import torch
        cap, send, recv = bufs["cap"], bufs["send"], bufs["recv"]
        mb = bufs["max_blocks"]
        assert len(blocks) <= mb
        hb = self.TICK_HEADER_BYTES + self.TICK_POSE_BYTES * mb
        hdr = np.zeros(hb // 4, np.int32)
        hdr[0] = len(blocks)
        off = hb
        for b, blk in enumerate(blocks):
            n, w = int(blk[0]), blk[1]
            assert n <= bufs["max_features"]
            hdr[1 + b] = n
            if len(blk) > 2 and blk[2] is not None:
                hdr[16 + 16 * b: 32 + 16 * b] = np.ascontiguousarray(blk[2], np.float32).reshape(16).view(np.int32)
            send[off: off + 47 * n] = w[: 47 * n]
            off += 47 * n
        send[:hb] = torch.from_numpy(hdr.view(np.uint8)).to(send.device, non_blocking=True)
        if self.dist is None:
            recv[:cap] = send
        else:
            self.dist.all_gather_into_tensor(recv, send, group=self.data_group(send))
        heads = recv.view(self.world, cap)[:, :hb].cpu().numpy().view(np.int32)      # the tick's one read-back
        out = []
        for r in range(self.world):
            nb = int(heads[r, 0])
            off = r * cap + hb
            lst = []
            for b in range(nb):
                n = int(heads[r, 1 + b])
                if with_poses:
                    lst.append((n, recv[off: off + 47 * n], heads[r, 16 + 16 * b: 32 + 16 * b].view(np.float32).reshape(4, 4).copy()))
                else:
                    lst.append((n, recv[off: off + 47 * n]))
                off += 47 * n
            out.append(lst)
        return out

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def cgroup_cpu_quota():
    """CPU quota of this process's cgroup in CPUs (cgroup v2 cpu.max / v1 cpu.cfs_quota_us), or None when unlimited / unreadable.
    A container with a quota throttles ALL its threads for the rest of a 100 ms period once the quota is spent: three spinning
    agent threads next to other busy processes of the same container show up as ~80 ms pauses (round 4: the bench boxes run the
    command in a 16-CPU cgroup; tools/neighbour_load.py's `everywhere` mode exceeded it)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except (OSError, ValueError):
        return None


def cgroup_throttled():
    """(periods throttled, microseconds throttled) of this process's cgroup so far, or None."""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            d = dict(line.split()[:2] for line in open(path) if line.strip())
            us = int(d["throttled_usec"]) if "throttled_usec" in d else int(d.get("throttled_time", 0)) // 1000
            return int(d.get("nr_throttled", 0)), us
        except (OSError, ValueError, KeyError):
            continue
    return None


class ServerTick:
    """The compute side of the server of BASELINE configs[2] / configs[4], running in the SAME job next to the agents
    (S/ClientHandler.cc:100-129: one LoopClosing / place-recognition thread per client on the server; S/LoopClosing.cc:580-840
    NewDetectCommonRegions; S/Communicator.cc:951-1149 keyframes arriving from the clients).

    Every agent (= rank) packs each new keyframe into a wire block on its GPU (orbk_pack_frame, R/msg/KF.msg:29-31 layout); every
    `tick_every` keyframes ALL ranks meet in ONE all-gather of the blocks (RCCL over xGMI on device tensors; gloo over host copies
    when several ranks share one GPU, which RCCL refuses -- a test aid, flagged in the report).  The server lives on rank 0's GPU:
    for every block received it rebuilds the KeyFrame on the device (orbk_frame_from_wire), computes its bag of words
    (KeyFrame::ComputeBoW -> orbv_transform_frame + orbv_bow_assemble), asks the place-recognition database for loop / merge
    candidates (KeyFrameDatabase::DetectNBestCandidates -> orbd_detect_n_best_candidates), matches the keyframe against the best
    candidate by bag of words (ORBmatcher::SearchByBoW(KF, KF) -> orbm_search_by_bow_kf) and projects the server's map points into it
    (ORBmatcher::SearchByProjection(KF, Scw, points) -> orbm_search_by_projection_sim3).  The Sim3 solver between the two matchers
    is out of scope: the keyframe's own pose plays Scw.

    The database is built before the timed region from every agent's keyframes of the sequence (what the server has received so
    far): vocabulary = a synthetic DBoW2 tree over rank 0's descriptors (ORBvoc.txt is not in the reference tree), map id = agent."""

    LEVELS_UP = 1

    def __init__(self, grp, api, views, synth, device, fv, server_map_view, agent_kfs, frames_per_kf, tick_every=2, max_features=4096,
                 shared_gpu=False):
        """agent_kfs: this agent's keyframes of the sequence, [(frame id, keypoints, descriptors, Tcw)], host arrays; server_map_view:
        the map points the server projects into incoming keyframes (rank 0: a worldpoints view, uploaded into a map handle of the
        server's own).  COLLECTIVE: every rank constructs it at the same point (the keyframes travel to the server through the
        tick's own exchange).  The server's work runs on a thread of its own on rank 0 (the reference's server is a separate process
        with one place-recognition thread per client): the tracking thread only takes part in the exchange."""
        import queue
        import threading
        import numpy as np
        import torch
        self.grp, self.api, self.views, self.fv = grp, api, views, fv
        self.map_view = server_map_view
        self.device, self.tick_every, self.K = device, int(tick_every), int(frames_per_kf)
        self.shared_gpu = bool(shared_gpu)
        self.is_server = grp.rank == 0
        self.np, self.torch = np, torch
        self.dev = "cpu" if self.shared_gpu else "cuda:%d" % device
        self.max_features = int(max_features)
        # two sets of tick buffers, used alternately: the server thread may still read tick t's blocks while tick t + 1 is exchanged
        self.bufs2 = [grp.tick_buffers(max_features=self.max_features, device=self.dev, max_blocks=max(self.tick_every, 1)) for _ in range(2)]
        self.bufs = self.bufs2[0]
        self.busy = [None, None]                             # threading.Event per buffer set: set when the server thread is done with it
        self.wire = [torch.zeros(47 * self.max_features, dtype=torch.uint8, device="cuda:%d" % device) for _ in range(max(self.tick_every, 1))]
        self.pending = []                                    # [(n, wire tensor, Tcw)] packed since the last tick
        self.stats = dict(ticks=0, blocks=0, loop_candidates=0, merge_candidates=0, bow_matches=0, projection_matches=0, exchange_s=0.0,
                          server_s=0.0, pack_s=0.0)
        self.last = None                                     # inputs / outputs of the last tick (rank 0) for the parity gate
        self.n_agent_kfs = len(agent_kfs)
        self.kf_slot = {int(k): j for j, (k, _a, _b, _c) in enumerate(agent_kfs)}
        # ---- every agent's keyframes -> the server (host blocks over the control plane: set-up, untimed)
        F = api.Frame(self.max_features, device)
        self._F = F
        setup_bufs = grp.tick_buffers(max_features=self.max_features, device="cpu", max_blocks=8)
        mine = []
        for (k, kps, desc, Tcw) in agent_kfs:
            fvk, keepk = views.frame_view(kps, desc, None, None, (fv.min_x, fv.max_x, fv.min_y, fv.max_y),
                                          (fv.fx, fv.fy, fv.cx, fv.cy, fv.bf, fv.b), fv.n_levels, fv.scale_factor)
            F.upload(fvk, keepk)
            mine.append((len(kps), torch.from_numpy(F.pack_wire().copy()), np.asarray(Tcw, np.float32)))
        rounds = grp.max_over_ranks(float((len(mine) + 7) // 8))
        received = [[] for _ in range(grp.world)]
        for r in range(int(rounds)):
            got = grp.all_gather_keyframe_blocks(setup_bufs, mine[8 * r: 8 * r + 8], with_poses=True)
            for a in range(grp.world):
                received[a] += [(n, blk.numpy().copy(), T) for (n, blk, T) in got[a]]
        if grp.dist is not None and not self.shared_gpu:
            grp.open_data_plane()                            # the RCCL group of the ticks (collective, outside every timed region)
        elif grp.dist is not None:
            grp.barrier()
        if not self.is_server:
            return
        # ---- the server's state: vocabulary, database, the flattened keyframes the KeyFrame matcher reads
        all_desc = np.concatenate([d for (_k, _p, d, _T) in agent_kfs]) if agent_kfs else np.zeros((1, 32), np.uint8)
        voc = synth.make_vocabulary(k=10, L=3, seed=0xB0C, descriptors=all_desc)
        self.voc_arrays = voc
        vv, self._vkeep = views.vocab_view(voc["child_start"], voc["child_ids"], voc["desc"], voc["weight"], voc["word_id"], voc["L"])
        self.vocab_view = vv
        self.voc = api.ORBVocabulary(vv, self._vkeep, device)
        self.n_words = int(voc["word_id"].max()) + 1
        self.KF = api.Frame(self.max_features, device)       # the keyframe being processed
        self.matcher_bow = api.ORBmatcher(0.9, True, device)     # S/LoopClosing.cc:605-606: matcherBoW(0.9, true), matcher(0.75, true)
        self.matcher_proj = api.ORBmatcher(0.75, True, device)
        bows, inv, covis, map_id, self.db_kfs = [], {}, [], [], []
        for a in range(grp.world):
            base = len(bows)
            for j, (n, wire, T) in enumerate(received[a]):
                self.KF.from_wire(fv, wire=wire, n=n)
                (bw, bv), (fn, fs, ff) = self.voc.transform(frame=self.KF, levelsup=self.LEVELS_UP)
                kps, desc = self.KF.download()
                idx = len(bows)
                bows.append((bw, bv))
                for w in bw:
                    inv.setdefault(int(w), []).append(idx)
                map_id.append(a)
                self.db_kfs.append(dict(desc=desc.copy(), angle=np.ascontiguousarray(kps["angle"]).copy(), fv=(fn, fs, ff), Tcw=T, agent=a, slot=j,
                                        valid=np.ones(n, np.uint8)))
            n_a = len(bows) - base
            for j in range(n_a):
                covis.append([base + q for q in (j - 1, j + 1, j - 2, j + 2, j - 3, j + 3) if 0 <= q < n_a])
        self.agent_base = np.cumsum([0] + [len(received[a]) for a in range(grp.world)])
        nk = len(bows)
        self.db_arrays = dict(inv=inv, bows=bows, covis=covis, map_id=np.asarray(map_id, np.int32), bad=np.zeros(nk, np.uint8),
                              map_bad=np.zeros(nk, np.uint8), n_words=self.n_words)
        dv, self._dkeep = views.database_view(inv, bows, covis, self.db_arrays["map_id"], self.db_arrays["bad"], self.db_arrays["map_bad"], self.n_words)
        self.db_view = dv
        self.db = api.KeyFrameDatabase(dv, self._dkeep, device)
        self.place_score = np.zeros(nk, np.float32)
        self.free = np.full(self.max_features, -1, np.int32)
        self.LM = api.LocalMap(max(int(server_map_view.m), 1) + 64, device)      # the server's own map handle (never the agent's)
        self.LM.upload(server_map_view)
        self.q = queue.Queue()
        self.error = None
        self.thread = threading.Thread(target=self._serve, daemon=True)
        self.thread.start()

    def _serve(self):
        while True:
            job = self.q.get()
            try:
                if job is None:
                    return
                if self.error is None:
                    self._process(*job)
            except Exception as e:                           # reported by tick() / drain(): a failing server must not hang the agents
                self.error = e
            finally:
                # the buffer set of this job is free again WHATEVER happened to the job (failed, or skipped because an earlier one
                # failed): rank 0 waits on this event two ticks later, in front of a collective every other rank is already in
                if job is not None:
                    job[1].set()
                self.q.task_done()

    # ---- agent side
    def on_keyframe(self, frame, n_features, frame_id, Tcw):
        """The agent has just tracked a keyframe (device frame `frame`, still resident): pack it; every tick_every-th call is a tick."""
        t0 = time.perf_counter()
        w = self.wire[len(self.pending)]
        frame.n = int(n_features)
        frame.pack_wire(device_ptr=w.data_ptr())
        self.pending.append((int(n_features), w, self.np.asarray(Tcw, self.np.float32), int(frame_id)))
        self.stats["pack_s"] += time.perf_counter() - t0
        if len(self.pending) >= self.tick_every:
            self.tick()

    def tick(self):
        """COLLECTIVE: one all-gather of every agent's pending blocks, then the server's work on rank 0."""
        np = self.np
        t0 = time.perf_counter()
        blocks = [((n, w.cpu(), T) if self.shared_gpu else (n, w, T)) for (n, w, T, _k) in self.pending]
        self.pending = []
        b = self.stats["ticks"] & 1
        if self.busy[b] is not None:
            self.busy[b].wait()                              # the server thread has finished with this buffer set (two ticks ago)
        got = self.grp.all_gather_keyframe_blocks(self.bufs2[b], blocks, with_poses=True)
        if self.is_server and self.error is not None:        # (after the collective: the other ranks are not left waiting in it)
            raise self.error
        t1 = time.perf_counter()
        self.stats["exchange_s"] += t1 - t0
        self.stats["ticks"] += 1
        if not self.is_server:
            return
        import threading
        self.busy[b] = threading.Event()
        self.q.put((got, self.busy[b]))

    def _process(self, got, done):
        np = self.np
        t1 = time.perf_counter()
        record = []
        for a, lst in enumerate(got):
            for (n, blk, T) in lst:
                before = self.place_score.copy()
                if self.shared_gpu:
                    wire_host = blk.numpy().copy()
                    self.KF.from_wire(self.fv, wire=wire_host, n=n)
                else:
                    wire_host = None
                    self.KF.from_wire(self.fv, n=n, device_ptr=blk.data_ptr())
                (bw, bv), (fn, fs, ff) = self.voc.transform(frame=self.KF, levelsup=self.LEVELS_UP)
                # spConnectedKF: the database keyframes of the same agent that the new keyframe is covisible with (its neighbours in time)
                con = np.zeros(len(self.place_score), np.uint8)
                lo, hi = int(self.agent_base[a]), int(self.agent_base[a + 1])
                j = self._nearest_db_slot(a, T)
                con[max(lo, lo + j - 2): min(hi, lo + j + 3)] = 1
                loop_c, merge_c = self.db.DetectNBestCandidates(bw, bv, con, a, 3, self.place_score)
                cand = int(merge_c[0]) if len(merge_c) else int(loop_c[0]) if len(loop_c) else (lo + j + 5) % max(len(self.place_score), 1)
                c = self.db_kfs[cand]
                fvK, keepK = self.views.featvec_view(fn, fs, ff)
                fv1, keep1 = self.views.featvec_view(*c["fv"])
                m12, nb = self.matcher_bow.SearchByBoWKF(self.KF, fvK, np.ones(n, np.uint8), c["desc"], c["valid"], c["angle"], fv1)
                matched, nproj = self.matcher_proj.SearchByProjectionSim3(self.KF, T, self.LM, self.free[:n], 8, 1.5)
                self.stats["blocks"] += 1
                self.stats["loop_candidates"] += len(loop_c); self.stats["merge_candidates"] += len(merge_c)
                self.stats["bow_matches"] += int(nb); self.stats["projection_matches"] += int(nproj)
                record.append(dict(agent=a, n=n, wire=wire_host, blk=blk, T=T, bow=(bw, bv), fv=(fn, fs, ff), con=con, before=before,
                                   loop=loop_c, merge=merge_c, cand=cand, m12=m12, nb=nb, matched=matched, nproj=nproj))
        self.last = record
        self.stats["server_s"] += time.perf_counter() - t1        # (_serve releases `done`, on every path)

    def drain(self):
        """Every tick exchanged so far has been processed by the server thread (rank 0; a no-op elsewhere)."""
        if self.is_server:
            self.q.join()
            if self.error is not None:
                raise self.error

    def _nearest_db_slot(self, a, T):
        """Index (within agent a's database keyframes) of the keyframe whose pose is nearest to T (the sequence revisits its frames)."""
        np = self.np
        lo, hi = int(self.agent_base[a]), int(self.agent_base[a + 1])
        if hi <= lo:
            return 0
        d = [float(np.abs(self.db_kfs[q]["Tcw"] - T).sum()) for q in range(lo, hi)]
        return int(np.argmin(d))

    def flush(self):
        """COLLECTIVE: a tick for whatever is pending (end of a region), so that every rank leaves with empty hands."""
        n = self.grp.max_over_ranks(float(len(self.pending)))
        if n > 0:
            self.tick()

    def report(self, wall_s=None):
        s = dict(self.stats)
        t = max(s["ticks"], 1)
        out = dict(ticks=s["ticks"], keyframes_per_agent_per_tick=self.tick_every, exchange_us_per_tick=round(1e6 * s["exchange_s"] / t, 1),
                   pack_us_per_keyframe=round(1e6 * s["pack_s"] / max(s["ticks"] * self.tick_every, 1), 1),
                   exchange=("gloo over host copies (ranks share a GPU: RCCL refuses two ranks on one device; test aid)" if self.shared_gpu else
                             ("RCCL all-gather on device memory, %d rank%s" % (self.grp.world, "" if self.grp.world == 1 else "s")) if self.grp.dist is not None
                             else "single rank: device-to-device copy (no process group)"),
                   bytes_per_agent_per_tick=int(self.bufs["cap"]),
                   server_thread="rank 0, a thread of its own next to the agent's tracking thread (handles of its own: KeyFrame, vocabulary, "
                                 "database, map)")
        if self.is_server:
            out.update(server_rank=0, server_us_per_tick=round(1e6 * s["server_s"] / t, 1), blocks=s["blocks"],
                       database_keyframes=len(self.place_score), vocabulary_words=self.n_words,
                       loop_candidates=s["loop_candidates"], merge_candidates=s["merge_candidates"], bow_matches=s["bow_matches"],
                       projection_matches=s["projection_matches"])
        return out
