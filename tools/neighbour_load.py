#!/usr/bin/env python3
"""Induced neighbour load for bench.py: reproduces a SHARED host (the bench boxes run several tenants on one 2-socket machine).

    python3 tools/neighbour_load.py <mode>[,<mode>...] -- python bench.py --gpus 1 --steps 20 --warmup 5

starts load processes pinned next to the agent's three threads, runs the command with ORBG_FIXED_CORES=1 (the harness then takes the
fixed run of cores this script computed the neighbours of, instead of looking for idle ones), prints the command's output and stops
the load processes it started (by PID).

modes (combine with commas):
  siblings   one spinning process on the SECOND hardware thread of each of the agent's three cores (SMT sharing: the agent's thread
             keeps a hardware thread of its own and shares the core's execution units)
  timeslice  one spinning process on EACH hardware thread of the agent's three cores (the agent's threads have no hardware thread
             of their own left: the scheduler time-slices them against the load)
  l3         memory-streaming processes on both hardware threads of the other cores of the agent's L3 slice (8 cores)
  membw      N memory-streaming processes spread over the rest of the GPU's NUMA node (default N = 16; membw:32 for 32)
  everywhere one spinning process on one hardware thread of every other core of the NUMA node, nearest cores first -- as many as the
             container's CPU quota leaves room for (see `budget` below)
  none       no load (the quiet reference run with the same fixed cores)
"""
import multiprocessing as mp
import os
import subprocess
import sys
import time


def spin(cpus):
    os.sched_setaffinity(0, cpus)
    x = 1
    while True:
        x = (x * 1103515245 + 12345) & 0x7FFFFFFF


def stream(cpus, mbytes=96):
    import numpy as np
    os.sched_setaffinity(0, cpus)
    a = np.ones(mbytes << 17, np.float64)          # mbytes MB
    b = np.empty_like(a)
    while True:
        np.copyto(b, a)
        np.copyto(a, b)


def physical_cores_of_first_gpu_node(harness):
    """As harness._physical_cores_of_gpu_node(0), from sysfs alone: this process must not initialise the GPU (it starts the
    benchmark as a child)."""
    node = None
    base = "/sys/bus/pci/devices"
    for d in sorted(os.listdir(base)):
        try:
            vendor = open(os.path.join(base, d, "vendor")).read().strip()
            cls = open(os.path.join(base, d, "class")).read().strip()
            if vendor == "0x1002" and (cls.startswith("0x12") or cls.startswith("0x03")):
                node = int(open(os.path.join(base, d, "numa_node")).read().strip())
                # (the boxes expose one GPU to the container; with several visible the first display/accelerator function wins)
                break
        except OSError:
            continue
    if node is None or node < 0:
        node = 0
    with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
        cpus = harness._parse_cpulist(f.read()) & os.sched_getaffinity(0)
    cores, seen = [], set()
    for c in sorted(cpus):
        if c in seen:
            continue
        with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
            sib = harness._parse_cpulist(f.read()) & cpus
        seen |= sib
        cores.append(sib)
    return cores


def main():
    if "--" not in sys.argv:
        raise SystemExit(__doc__)
    k = sys.argv.index("--")
    modes = [m for m in ",".join(sys.argv[1:k]).split(",") if m]
    cmd = sys.argv[k + 1:]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from multi_orbslam3_amd import harness
    cores = physical_cores_of_first_gpu_node(harness)          # sets of hardware threads per physical core, in order
    agent = harness._pick_cores(cores, 0, 1, 3)                # what cores_for_agent() returns with ORBG_FIXED_CORES=1
    if agent is None:
        raise SystemExit("topology not visible: cannot place the load")
    agent_idx = [cores.index(c) for c in agent]
    procs = []
    # The load must stay INSIDE the container's CPU quota (cgroup cpu.max; 16 CPUs on the bench boxes): a cgroup that spends its quota
    # has ALL its threads stopped for the rest of the 100 ms period -- the benchmark's included.  Round 4 measured exactly that as
    # "one 75-80 ms pause per region" under `everywhere` (63 spinners) before the quota was read here.  Six CPUs are left to the
    # benchmark (three spinning agent threads, the interpreter, the runtime's helpers); ORBG_LOAD_IGNORE_QUOTA=1 starts everything.
    quota = harness.cgroup_cpu_quota()
    budget = None if (quota is None or os.environ.get("ORBG_LOAD_IGNORE_QUOTA")) else max(int(quota) - 6, 0)
    skipped = [0]

    def start(fn, cpus, *a):
        if budget is not None and len(procs) >= budget:
            skipped[0] += 1
            return
        p = mp.Process(target=fn, args=(set(cpus),) + a, daemon=True)
        p.start()
        procs.append(p)

    for m in modes:
        name, _, arg = m.partition(":")
        if name == "none":
            pass
        elif name == "siblings":
            for c in agent:
                if len(c) > 1:
                    start(spin, [sorted(c)[-1]])
        elif name == "timeslice":
            for c in agent:
                for t in sorted(c):
                    start(spin, [t])
        elif name == "l3":
            g0 = (agent_idx[0] // 8) * 8
            for i in range(g0, min(g0 + 8, len(cores))):
                if i not in agent_idx:
                    for t in sorted(cores[i]):
                        start(stream, [t])
        elif name == "membw":
            n = int(arg or 16)
            others = [i for i in range(len(cores)) if i // 8 != agent_idx[0] // 8]
            for j in range(n):
                start(stream, [sorted(cores[others[(j * 7) % len(others)]])[0]])
        elif name == "everywhere":
            others = [i for i in range(len(cores)) if i not in agent_idx]
            # (nearest neighbours of the agent first: with a quota only the first few start)
            others.sort(key=lambda i: min(abs(i - a2) for a2 in agent_idx))
            for i in others:
                start(spin, [sorted(cores[i])[0]])
        else:
            raise SystemExit("unknown mode %r" % m)
    time.sleep(1.0)
    print("[neighbour_load] modes=%s agent cores=%s load processes=%d (cgroup cpu quota %s: %d not started)" %
          (modes, [sorted(c) for c in agent], len(procs), quota, skipped[0]), flush=True)
    env = dict(os.environ, ORBG_FIXED_CORES="1")
    try:
        rc = subprocess.call(cmd, env=env)
    finally:
        for p in procs:
            p.terminate()
        for p in procs:
            p.join(5)
    sys.exit(rc)


if __name__ == "__main__":
    main()
