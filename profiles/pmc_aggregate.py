#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files (one counter per run) into per-kernel averages.

usage: pmc_aggregate.py FETCH_SIZE=<csv> WRITE_SIZE=<csv> > out.json      (values are KB per launch, as rocprofv3 reports)
"""
import csv, json, sys
from collections import defaultdict

out = defaultdict(dict)
for arg in sys.argv[1:]:
    counter, path = arg.split("=", 1)
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        a = acc[name]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for name, (tot, n) in acc.items():
        out[name][counter + "_KB_avg"] = round(tot / n, 2)
        out[name]["launches"] = n
print(json.dumps(out, indent=1, sort_keys=True))
