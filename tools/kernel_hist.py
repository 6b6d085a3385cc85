#!/usr/bin/env python3
"""Duration distribution of one kernel in a rocprofv3 --kernel-trace csv, and what else was on the GPU during its fastest and
slowest launches.  Usage: kernel_hist.py <kernel_trace.csv> <kernel name substring> [n_examples]"""
import csv, sys
from collections import Counter


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].split("<")[0]


def main():
    path, key = sys.argv[1], sys.argv[2]
    nex = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    mine = [(e - s, s, e) for s, e, n in rows if key in n]
    if not mine:
        print("no launches of", key); return
    d = sorted(x[0] for x in mine)
    q = lambda p: d[min(len(d) - 1, int(p * len(d)))] / 1e3
    print("%s: %d launches  min %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us" % (key, len(d), d[0] / 1e3, q(.1), q(.5), q(.9), d[-1] / 1e3))
    h = Counter(int(x / 1e3) for x in d)
    for k in sorted(h):
        print("  %3d us %5d %s" % (k, h[k], "#" * min(80, h[k])))

    def overlaps(s, e):
        c = Counter()
        for s2, e2, n in rows:
            if e2 <= s or s2 >= e or (s2 == s and e2 == e): continue
            c[n] += (min(e, e2) - max(s, s2)) / (e - s)
        return ", ".join("%s %.0f%%" % (n, 100 * v) for n, v in c.most_common(5)) or "(alone)"
    mine.sort()
    print("fastest:")
    for dur, s, e in mine[:nex]: print("  %.1f us  with: %s" % (dur / 1e3, overlaps(s, e)))
    print("slowest:")
    for dur, s, e in mine[-nex:]: print("  %.1f us  with: %s" % (dur / 1e3, overlaps(s, e)))
    alone = [dur for dur, s, e in mine if overlaps(s, e) == "(alone)"]
    if alone:
        alone.sort()
        print("alone on the GPU: %d launches, min %.1f p50 %.1f max %.1f us" % (len(alone), alone[0] / 1e3, alone[len(alone) // 2] / 1e3, alone[-1] / 1e3))


if __name__ == "__main__":
    main()
