#!/usr/bin/env python3
"""Host-side noise report of a GPU box: affinity, cgroup CPU quota, load, per-core busy fraction over a second.
Used to explain run-to-run spread of bench.py (the tracking / local-BA / ingest threads spin on host cores)."""
import os, time, json


def read_stat():
    out = {}
    with open("/proc/stat") as f:
        for line in f:
            if line.startswith("cpu") and line[3].isdigit():
                p = line.split()
                v = list(map(int, p[1:9]))
                out[int(p[0][3:])] = (sum(v), v[3] + v[4])
    return out


def busy(dt=1.0):
    a = read_stat(); time.sleep(dt); b = read_stat()
    r = {}
    for c in a:
        tot = b[c][0] - a[c][0]; idle = b[c][1] - a[c][1]
        r[c] = 0.0 if tot <= 0 else 1.0 - idle / tot
    return r


if __name__ == "__main__":
    aff = sorted(os.sched_getaffinity(0))
    info = {"nproc": os.cpu_count(), "affinity_n": len(aff), "affinity": "%d-%d" % (aff[0], aff[-1]) if aff else "", "loadavg": open("/proc/loadavg").read().split()[:3]}
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
        try:
            info[p] = open(p).read().strip()
        except Exception:
            pass
    b = busy(1.0)
    hot = {c: round(v, 2) for c, v in b.items() if v > 0.2}
    info["busy_cores_gt20pct"] = hot
    info["n_busy"] = len(hot)
    print(json.dumps(info))
