#!/bin/bash
# N default bench runs in a row on one box: value, p50 / p95 of the step, local BA per call, host noise -- the run-to-run spread.
N=${1:-6}; shift || true
for i in $(seq $N); do
  timeout 300 python bench.py --no-cpu-baseline --no-dropin --secondary-steps 0 "$@" 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('run $i', d['value'], d['step_ms_p50'], d['step_ms_p95'], c['lba_ms_per_call'], c.get('host_noise'), c['cpu_affinity'].split(';')[1] if ';' in c['cpu_affinity'] else '')"
done
