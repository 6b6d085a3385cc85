#!/bin/bash
# N runs of the line exactly as the driver takes it (python bench.py --gpus 1 --steps 20 --warmup 5): value, step p50 / p95, local BA, host noise
N=${1:-6}
for i in $(seq $N); do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('run $i', d['value'], d['step_ms_p50'], d['step_ms_p95'], d['step_ms_max'], c['lba_ms_per_call'], c.get('host_noise'), d.get('value_device_images'))"
done
