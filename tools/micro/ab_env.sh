#!/bin/bash
# A/B of environment switches on the bench line: tools/micro/ab_env.sh "<bench args>" VAR=1 "VAR2=x VAR3=y" ...   (first run: no switch)
ARGS=$1; shift
run() { tag=$1; shift; env "$@" python3 bench.py $ARGS > gpurun_out/ab_$tag.json 2>gpurun_out/ab_$tag.err; python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ab_$tag.json').read().strip().splitlines()[-1])
g=lambda k: d.get(k)
print('$tag', g('value'), g('value_min'), g('value_max'), 'ext', g('stage_extract_us'), 'mf', g('stage_match_frame_us'), 'mm', g('stage_match_map_us'), 'lat', g('ctor_latency_us_p50'), 'lba', g('lba_ms_per_call'), 'stage_lba', g('stage_lba_us'), 'sync', g('value_sync_ctor_host_images'), 'po', g('value_with_pose_opt'), 'dev', g('value_device_images'))
"; }
mkdir -p gpurun_out
run base ORBG_AB_NONE=1
i=0
for sw in "$@"; do i=$((i+1)); run "v$i[$sw]" $sw; done
