#!/bin/bash
# The driver's command under induced neighbour load (tools/neighbour_load.py): one line per run with the run's own diagnosis.
# Usage on the GPU box: [BENCH_ARGS='...'] bash tools/noise_matrix.sh [modes...]   (default: none siblings l3 membw everywhere)
MODES=${@:-none siblings l3 membw everywhere}
for m in $MODES; do
  timeout 600 python3 tools/neighbour_load.py $m -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dropin $BENCH_ARGS 2>/dev/null | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
keys=['value','value_min','value_median','value_max','step_ms_p50','step_ms_p95','value_device_images','value_sync_ctor_host_images','value_with_pose_opt','host_noise_ctxt_switches','host_noise_cgroup_throttled_us','host_cgroup_cpu_quota','host_noise_max_core_busy','ingest_queue_us_p50','ingest_queue_us_max','ingest_pack_us_p50','ingest_pack_us_max','ctor_enqueue_us_p50','ctor_wait_us_p50','ctor_wait_us_max','ctor_latency_us_p50','ctor_latency_us_max','stage_extract_us','stage_match_frame_us','stage_match_map_us','stage_map_upload_us','stage_lba_us','lba_ms_per_call']
print('$m', {k:d.get(k) for k in keys})"
done
