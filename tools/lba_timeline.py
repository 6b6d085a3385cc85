"""Per-solve timeline of the local BA from a rocprofv3 kernel trace (csv): span from the first upload's start to k_errors_export's
end, kernel time, gaps, and the average duration of each kernel inside the solves.  python tools/lba_timeline.py run_kernel_trace.csv"""
import collections
import csv
import sys

import numpy as np

ks = []
for r in csv.DictReader(open(sys.argv[1])):
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    if nm.startswith('k_') or nm.startswith('ldltm::'):
        ks.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), nm.split('<')[0]))
ks.sort()
solves, cur = [], []
for k in ks:
    if k[2] == 'k_upload16' and cur and cur[-1][2] != 'k_upload16':
        cur = []
    cur.append(k)
    if k[2] in ('k_errors_export', 'k_export'):
        solves.append(cur)
        cur = []
solves = solves[len(solves) // 5:]                       # (warm-up)
span = np.array([(s[-1][1] - s[0][0]) / 1e3 for s in solves])
kern = np.array([sum(k[1] - k[0] for k in s) / 1e3 for s in solves])
print("solves %d: span median %.1f us (min %.1f, p90 %.1f), kernel time median %.1f, idle inside the span median %.1f" %
      (len(solves), np.median(span), span.min(), np.percentile(span, 90), np.median(kern), np.median(span - kern)))
per = collections.defaultdict(list)
for s in solves:
    for k in s:
        per[k[2]].append((k[1] - k[0]) / 1e3)
for nm, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print("  %-22s per solve %4.1f x %6.2f us = %6.1f us  (min %.2f, p90 %.2f)" % (nm, len(v) / len(solves), np.mean(v), sum(v) / len(solves), min(v), np.percentile(v, 90)))
gaps = collections.defaultdict(list)
for s in solves:
    for a, b in zip(s, s[1:]):
        gaps[(a[2], b[2])].append((b[0] - a[1]) / 1e3)
for (a, b), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / len(solves) > 0.5:
        print("  gap %-20s -> %-20s per solve %4.1f x %6.2f us = %6.1f us" % (a, b, len(v) / len(solves), np.mean(v), sum(v) / len(solves)))
