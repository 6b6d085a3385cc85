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
        if n_local == 1 and slot == 0 and os.environ.get("ORBG_FIXED_CORES", "0") != "1":
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

    def tick_buffers(self, max_features, device, max_blocks=8):
        """Persistent send / receive buffers of a server tick for up to `max_blocks` keyframes of up to `max_features`
        features per agent: [64-byte header | the agent's wire blocks back to back].  The size is fixed for the group's life,
        so a tick needs no size exchange."""
        import torch
        assert 1 <= max_blocks <= self.TICK_MAX_BLOCKS
        cap = self.TICK_HEADER_BYTES + 47 * int(max_features) * int(max_blocks)
        cap = (cap + 255) & ~255
        send = torch.zeros(cap, dtype=torch.uint8, device=device)
        recv = torch.zeros(self.world * cap, dtype=torch.uint8, device=device)
        return dict(cap=cap, send=send, recv=recv, max_blocks=int(max_blocks), max_features=int(max_features))

    def all_gather_keyframe_blocks(self, bufs, blocks):
        """One server tick: every agent contributes ALL its new keyframe wire blocks -- `blocks` = [(n_features, uint8 tensor
        of 47 * n_features bytes)], possibly empty -- in ONE all-gather of the fixed-size tick buffers (RCCL over xGMI on GPU
        tensors, gloo on CPU tensors; < 1 MB per agent: latency bound).  The per-block feature counts travel in the buffer's
        header, so there is no second collective and no host round trip between collectives; the headers of all agents are read
        back with one small copy.  Returns [[(n_features, block view), ...] for every rank]."""
        import numpy as np
        import torch
        cap, send, recv = bufs["cap"], bufs["send"], bufs["recv"]
        assert len(blocks) <= bufs["max_blocks"]
        hdr = np.zeros(self.TICK_HEADER_BYTES // 4, np.int32)
        hdr[0] = len(blocks)
        off = self.TICK_HEADER_BYTES
        for b, (n, w) in enumerate(blocks):
            n = int(n)
            assert n <= bufs["max_features"]
            hdr[1 + b] = n
            send[off: off + 47 * n] = w[: 47 * n]
            off += 47 * n
        send[: self.TICK_HEADER_BYTES] = torch.from_numpy(hdr.view(np.uint8)).to(send.device, non_blocking=True)
        if self.dist is None:
            recv[:cap] = send
        else:
            self.dist.all_gather_into_tensor(recv, send, group=self.data_group(send))
        heads = recv.view(self.world, cap)[:, : self.TICK_HEADER_BYTES].cpu().numpy().view(np.int32)      # the tick's one read-back
        out = []
        for r in range(self.world):
            nb = int(heads[r, 0])
            off = r * cap + self.TICK_HEADER_BYTES
            lst = []
            for b in range(nb):
                n = int(heads[r, 1 + b])
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
