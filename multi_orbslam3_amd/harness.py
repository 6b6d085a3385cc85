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


def _pick_core_pair(cores, slot, n_local):
    """cores: the physical cores of the node (sets of hardware threads), in order.  The first cores of a node serve
    interrupts and housekeeping, so agents take consecutive pairs from the upper half.  Two agents must never get the same
    pair (four spinning threads on two cores): when the node has fewer pairs than the launch has local ranks (all of which
    may sit on this node) every core is used, and when that is not enough either nothing is pinned."""
    n_local = max(n_local, slot + 1)
    half = len(cores) // 2
    if (len(cores) - half) // 2 < n_local:
        half = 0
    if (len(cores) - half) // 2 < n_local:
        return None
    base = half + 2 * slot
    return cores[base], cores[base + 1]


def core_pair_for_agent(device_index, slot):
    """Two distinct physical cores (each returned with its SMT siblings) on the GPU's NUMA node for agent number `slot` on
    that node: one for the tracking thread, one for the library's local-BA worker.  Both threads spin on completion words;
    when the scheduler happens to put them on the two hardware threads of one core each runs at about half speed
    (searches 47 -> 65-74 us per call, observed in roughly one run out of five).  Returns (main_cpus, worker_cpus) or None."""
    try:
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
        return _pick_core_pair(cores, slot, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except Exception:
        return None


class AgentGroup:
    def __init__(self, backend=None, device_index=None, force_group=False):
        """force_group: create the process group even for a single rank (exercises the RCCL path of
        all_gather_keyframes on one GPU; must then be constructed before any other GPU call of the process)."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.backend = backend
        if self.world > 1 or force_group:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            self.backend = backend
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", self.local_rank if device_index is None else device_index)
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist

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
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

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
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

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
        self.dist.all_gather_into_tensor(counts, mine)
        counts = [int(c) for c in counts.cpu()]
        pad = 47 * max(max(counts), 1)
        send = torch.zeros(pad, dtype=torch.uint8, device=dev)
        send[: 47 * int(n_features)] = wire[: 47 * int(n_features)]
        recv = torch.empty(self.world * pad, dtype=torch.uint8, device=dev)
        self.dist.all_gather_into_tensor(recv, send)
        return [(counts[r], recv[r * pad: r * pad + 47 * counts[r]]) for r in range(self.world)]

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
