#!/bin/bash
# round-4 session 3: fused constructor tail / slots-direct quad-trees / polling upload -- parity, timeline, A/B
R="$GRAFT_REPO_ROOT"; OUT="$R/gpurun_out/s3"; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q -k "extract or frame or stereo or octree or quadtree or ctor or constructor or smoke or c4 or image" > "$OUT/pytest_extract.log" 2>&1; tail -5 "$OUT/pytest_extract.log"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_sync" -o run -- python3 "$R/bench.py" --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-dropin --lba-mode inline --no-pipeline > "$OUT/trace_sync.log" 2>&1
cd "$R"
python3 tools/ctor_gaps.py "$(find $OUT/trace_sync -name '*kernel_trace.csv' | head -1)" > "$OUT/ctor_gaps.txt" 2>&1
cat "$OUT/ctor_gaps.txt"
echo "--- default"; bash tools/noise_matrix.sh none
echo "--- split tail"; ORBG_CTOR_SPLIT_TAIL=1 bash tools/noise_matrix.sh none
echo "--- gather"; ORBG_OCT_GATHER=1 bash tools/noise_matrix.sh none
echo "--- two uploads"; ORBG_IMG_TWO_UPLOADS=1 bash tools/noise_matrix.sh none
echo "--- all old"; ORBG_IMG_TWO_UPLOADS=1 ORBG_OCT_GATHER=1 ORBG_CTOR_SPLIT_TAIL=1 bash tools/noise_matrix.sh none
