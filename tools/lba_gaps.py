"""Inter-kernel gaps on the LBA stream from a rocprofv3 kernel trace (csv): python tools/lba_gaps.py run_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
    ks.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), nm))
ks.sort()
lba = [k for k in ks if k[2].startswith('k_') or k[2].startswith('ldltm::')]
gaps = collections.defaultdict(list)
for a, b in zip(lba, lba[1:]):
    g = (b[0] - a[1]) / 1e3
    if g < 300:
        gaps[(a[2][:20], b[2][:20])].append(g)
tot = 0.0
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    if len(v) > 5:
        print("%-22s -> %-22s n=%4d avg gap %6.2f us  total %8.1f us" % (k[0], k[1], len(v), sum(v) / len(v), sum(v)))
        tot += sum(v)
n_solves = sum(1 for k in lba if k[2].startswith('k_export') or k[2].startswith('k_errors_export'))
print("solves %d, gap time per solve %.1f us, kernel time per solve %.1f us" % (n_solves, tot / max(n_solves, 1), sum(k[1] - k[0] for k in lba) / 1e3 / max(n_solves, 1)))
