run() { tag=$1; shift; env "$@" python3 bench.py --ctor-ahead 2 --repeats 3 > gpurun_out/ab_$tag.json 2>gpurun_out/ab_$tag.err; python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ab_$tag.json').read().strip().splitlines()[-1])
print('$tag', d['value'], d['value_min'], d['value_max'], 'ext', d['stage_extract_us'], 'mf', d['stage_match_frame_us'], 'mm', d['stage_match_map_us'], 'lat', d['ctor_latency_us_p50'], 'lba', d['lba_ms_per_call'], 'stage_lba', d['stage_lba_us'])
"; }
run base X=1
run ldltprio ORBG_LDLT_PRIO=1
run poolL ORBG_POOL_PRIO=h---
run both ORBG_POOL_PRIO=h--- ORBG_LDLT_PRIO=1
run poolLlow ORBG_POOL_PRIO=hll-
