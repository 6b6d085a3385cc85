#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 --pmc counter_collection.csv (several counters per run).
usage: pmc_counters.py <counter_collection.csv>"""
import csv, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    a = acc[name][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
counters = sorted({c for k in acc.values() for c in k})
print("%-34s %8s " % ("kernel", "launches") + " ".join("%26s" % c for c in counters))
for name in sorted(acc, key=lambda k: -sum(v[0] for v in acc[k].values())):
    n = max(v[1] for v in acc[name].values())
    print("%-34s %8d " % (name[:34], n) + " ".join("%26.1f" % (acc[name][c][0] / max(acc[name][c][1], 1)) for c in counters))
