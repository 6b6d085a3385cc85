#!/bin/bash
# round-4 session 2: constructor timeline + noise experiments
R="$GRAFT_REPO_ROOT"; OUT="$R/gpurun_out/s2"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_sync" -o run -- python3 "$R/bench.py" --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-dropin --lba-mode inline --no-pipeline > "$OUT/trace_sync.log" 2>&1
cd "$R"
python3 tools/ctor_gaps.py "$(find $OUT/trace_sync -name '*kernel_trace.csv' | head -1)" > "$OUT/ctor_gaps.txt" 2>&1
cat "$OUT/ctor_gaps.txt"
echo "--- prewarm 4000"; BENCH_ARGS="--prewarm-steps 4000" bash tools/noise_matrix.sh none none
echo "--- siblings / l3"; bash tools/noise_matrix.sh siblings l3
echo "--- PACK_NT"; ORBG_PACK_NT=1 bash tools/noise_matrix.sh none l3
echo "--- NO_POLL timeslice"; bash tools/noise_matrix.sh timeslice; ORBG_NO_POLL=1 bash tools/noise_matrix.sh timeslice none
