#!/bin/bash
# The synchronous-constructor region under the `everywhere` load (a spinner on every other core of the NUMA node), variants by environment:
#   tools/micro/everywhere_sync.sh "" ORBG_BENCH_NO_BRACKETS=1 ...
for sw in "$@"; do
  for rep in 1 2; do
    env $sw timeout 300 python3 tools/neighbour_load.py everywhere -- python bench.py --gpus 1 --steps 400 --warmup 40 --no-pipeline --no-cpu-baseline --no-secondary --no-dropin 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
k=[x for x in c if x.startswith('regions_ms')][0]
print('[$sw]', d['value'], d['value_min'], d['value_max'], 'p50/p95/max', d['step_ms_p50'], d['step_ms_p95'], d['step_ms_max'], 'worst', [r[-1] for r in c[k]][:5])"
  done
done
