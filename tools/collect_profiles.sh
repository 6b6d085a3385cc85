#!/bin/bash
# Collects the evidence files of a build on the GPU box: rocprofv3 kernel stats (inline LBA so that every kernel is in one
# trace), the two PMC passes, and the bench lines.  Usage (on the box): bash tools/collect_profiles.sh <tag>   e.g. r1_o
# Results land in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
TAG=${1:-rX}
OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --lba-mode inline --no-pipeline > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_async -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/stats_async.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --lba-mode inline --no-pipeline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --lba-mode inline --no-pipeline > $OUT/pmc_write.log 2>&1
cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 400 --warmup 40 2>/dev/null | tail -1 > $OUT/bench_line_async_pipelined.json
python3 bench.py --steps 400 --warmup 40 --no-pipeline --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_line_async_sync_ctor.json
python3 bench.py --steps 400 --warmup 40 --lba-mode inline --no-pipeline --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_line_inline.json
python3 bench.py --steps 400 --warmup 40 --pose-opt --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_line_async_poseopt.json
python3 tools/lba_gaps.py $OUT/stats/run_kernel_trace.csv > $OUT/lba_gaps.txt 2>&1
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
ls -la $OUT $OUT/stats | head -30
for f in $OUT/bench_line_*.json; do python3 -c "import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['config']['lba_ms_per_call'], d['config']['stage_ms_per_frame'])"; done
