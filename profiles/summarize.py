#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv as a short table (name, calls, avg us, total ms, %)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    name = r['Name'].replace('(anonymous namespace)::', '').split('(')[0]
    print("%-30s calls %5s avg %9.1f us  total %8.2f ms  %6.2f%%" % (name, r['Calls'], float(r['AverageNs']) / 1e3,
          float(r['TotalDurationNs']) / 1e6, float(r['Percentage'])))
