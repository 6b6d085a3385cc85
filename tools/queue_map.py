#!/usr/bin/env python3
"""Which kernels of a rocprofv3 --kernel-trace csv went through which hardware queue (Queue_Id): HIP maps the streams of a
process onto GPU_MAX_HW_QUEUES hardware queues, and streams that share one are serialised.  Usage: queue_map.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].split("<")[0]


q = defaultdict(lambda: defaultdict(int))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        q[r["Queue_Id"]][short(r["Kernel_Name"])] += 1
for qid in sorted(q):
    ks = sorted(q[qid].items(), key=lambda kv: -kv[1])
    print("queue %s: %s" % (qid, ", ".join("%s x%d" % kv for kv in ks[:14])))
