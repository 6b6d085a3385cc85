#!/bin/bash
# N runs of the line exactly as the driver takes it (python bench.py --gpus 1 --steps 20 --warmup 5): value, step p50 / p95, local BA, host noise
N=${1:-6}
for i in $(seq $N); do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('run $i value', d['value'], 'min/median/max of 5 regions', d.get('value_min'), d.get('value_median'), d.get('value_max'), 'step p50/p95/max', d['step_ms_p50'], d['step_ms_p95'], d['step_ms_max'],
      'lba ms', c['lba_ms_per_call'], 'ctxt switches', d.get('host_noise_ctxt_switches'), 'core busy', d.get('host_noise_max_core_busy'), 'pack/wait/latency p50 us', d.get('ingest_pack_us_p50'), d.get('ctor_wait_us_p50'), d.get('ctor_latency_us_p50'),
      'device images', d.get('value_device_images'), 'sync ctor', d.get('value_sync_ctor_host_images'), 'with pose opt', d.get('value_with_pose_opt'), 'dropin', d.get('value_dropin'))"
done
