#!/bin/bash
# Collects the evidence files of a build on the GPU box: rocprofv3 kernel stats (inline LBA so that every kernel is in one
# trace), the PMC passes (FETCH_SIZE / WRITE_SIZE for HBM traffic, the FP64-MFMA counters for the matrix-core LDL^T; one
# rocprofv3 run per counter group, no trace domains) and the bench lines.
# Usage (on the box): bash tools/collect_profiles.sh <tag>   e.g. r2_a      (run tools/micro/build_variants.sh in the container first:
# the phase tables come from the -DPO_PROFILE / -DOCT_PROFILE variants of the library, which must match the sources)
# Results land in gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
TAG=${1:-rX}
R="$GRAFT_REPO_ROOT"
OUT="$R/gpurun_out/$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
INL="--steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-dropin --lba-mode inline --no-pipeline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$R/bench.py" $INL > "$OUT/stats.log" 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_async" -o run -- python3 "$R/bench.py" --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --no-dropin > "$OUT/stats_async.log" 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c4" -o run -- python3 "$R/bench.py" --config C4 --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-dropin --lba-mode inline --no-pipeline > "$OUT/stats_c4.log" 2>&1 || true
PMC="--steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-dropin --lba-mode inline --no-pipeline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o run -- python3 "$R/bench.py" $PMC > "$OUT/pmc_fetch.log" 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o run -- python3 "$R/bench.py" $PMC > "$OUT/pmc_write.log" 2>&1 || true
C4PMC="--config C4 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-dropin --lba-mode inline --no-pipeline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_c4" -o run -- python3 "$R/bench.py" $C4PMC > "$OUT/pmc_fetch_c4.log" 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_c4" -o run -- python3 "$R/bench.py" $C4PMC > "$OUT/pmc_write_c4.log" 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d "$OUT/pmc_mfma" -o run -- python3 "$R/bench.py" $PMC > "$OUT/pmc_mfma.log" 2>&1 || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_sq" -o run -- python3 "$R/bench.py" $PMC > "$OUT/pmc_sq.log" 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d "$OUT/pmc_mfma_c4" -o run -- python3 "$R/bench.py" --config C4 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-dropin --lba-mode inline --no-pipeline > "$OUT/pmc_mfma_c4.log" 2>&1 || true
cd "$R"
python3 bench.py --steps 2000 --warmup 100 2>/dev/null | grep '^{' | tail -1 > "$OUT/bench_line_default.json" || true
python3 bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > "$OUT/bench_line_driver_20_steps.json" || true
python3 bench.py --steps 400 --warmup 40 --no-pipeline --no-cpu-baseline --no-secondary --no-dropin 2>/dev/null | grep '^{' | tail -1 > "$OUT/bench_line_sync_ctor.json" || true
python3 bench.py --steps 400 --warmup 40 --lba-mode inline --no-pipeline --no-cpu-baseline --no-secondary --no-dropin 2>/dev/null | grep '^{' | tail -1 > "$OUT/bench_line_inline.json" || true
python3 bench.py --config C4 --steps 200 --warmup 20 --no-secondary --no-dropin 2>/dev/null | grep '^{' | tail -1 > "$OUT/bench_line_c4.json" || true
python3 bench.py --config mono --steps 2000 --warmup 100 --no-dropin 2>/dev/null | grep '^{' | tail -1 > "$OUT/bench_line_mono.json" || true
python3 bench.py --config mono_dist --steps 2000 --warmup 100 --no-secondary --no-dropin 2>/dev/null | grep '^{' | tail -1 > "$OUT/bench_line_mono_dist.json" || true
python3 bench.py --config C3 --steps 2000 --warmup 100 --no-secondary --no-dropin 2>/dev/null | grep '^{' | tail -1 > "$OUT/bench_line_c3_1rank.json" || true
ORBG_BENCH_SHARE_GPU=1 python3 bench.py --config C5 --gpus 2 --steps 200 --warmup 20 --no-secondary --no-dropin --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > "$OUT/bench_line_c5_2ranks_shared_gpu.json" || true
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-secondary --no-dropin --server-tick 2>/dev/null | grep '^{' | tail -1 > "$OUT/bench_line_server_tick.json" || true
tests/cpp/dropin_bench 40 > "$OUT/dropin_bench.json" 2> "$OUT/dropin_bench.txt" || true
tools/micro/fp64_latency > "$OUT/micro_fp64_issue.txt" 2>&1 || true
[ -x tools/micro/ldlt_mfma_test_prof ] && timeout 120 tools/micro/ldlt_mfma_test_prof > "$OUT/ldlt_xcd_timeline.txt" 2>&1 || true
[ -x tools/micro/readlane_chain ] && timeout 60 tools/micro/readlane_chain > "$OUT/micro_readlane_chain.txt" 2>&1 || true
for v in 0 1; do ORBG_LDLT_XCD=$v python3 tools/lba_sizes.py 30; done > "$OUT/lba_sizes_xcd_off_on.txt" 2>&1 || true
python3 tools/search_large_map.py 1 8 32 128 > "$OUT/search_large_map.txt" 2>&1 || true
python3 tools/po_time.py > "$OUT/pose_opt_time.txt" 2>&1 || true
python3 tools/micro/po_prof.py > "$OUT/pose_opt_phases.txt" 2>&1 || true
python3 tools/lba_time.py C2 200 > "$OUT/lba_time.txt" 2>&1 || true
python3 tools/lba_time.py C4 40 >> "$OUT/lba_time.txt" 2>&1 || true
python3 tools/lba_gaps.py "$OUT/stats/run_kernel_trace.csv" > "$OUT/lba_gaps.txt" 2>&1 || true
python3 tools/ctor_gaps.py "$OUT/stats/run_kernel_trace.csv" > "$OUT/ctor_gaps.txt" 2>&1 || true
python3 tools/lba_timeline.py "$(find "$OUT/stats_async" -name "*kernel_trace.csv" | head -1)" > "$OUT/lba_timeline_agent.txt" 2>&1 || true
python3 tools/micro/oct_prof.py > "$OUT/octree_phases.txt" 2>&1 || true
SPREAD=0 python3 tools/search_large_map.py 1 8 32 128 > "$OUT/search_large_map_dense.txt" 2>&1 || true
python3 tools/po_sweep.py 500 > "$OUT/pose_opt_sweep.txt" 2>&1 || true
python3 tools/vocab_time.py 200 > "$OUT/vocab_full_size.txt" 2>&1 || true
python3 tools/rig_time.py > "$OUT/rig_time.txt" 2>&1 || true
python3 tools/rig_match_time.py > "$OUT/rig_match_time.txt" 2>&1 || true
tests/cpp/closed_loop 200 5 > "$OUT/closed_loop.txt" 2>&1 || true
{ tests/cpp/rig_loop 120 --shadow; tests/cpp/rig_loop 120; } > "$OUT/rig_loop.txt" 2>&1 || true
bash tools/bench_driver_repeat.sh 5 > "$OUT/bench_driver_repeat.txt" 2>&1 || true
f=$(find "$OUT/pmc_fetch" -name "*counter_collection.csv" | head -1); w=$(find "$OUT/pmc_write" -name "*counter_collection.csv" | head -1)
if [ -n "$f" ] && [ -n "$w" ]; then python3 profiles/pmc_aggregate.py FETCH_SIZE="$f" WRITE_SIZE="$w" > "$OUT/pmc_fetch_write_per_kernel.json" || true; fi
f=$(find "$OUT/pmc_fetch_c4" -name "*counter_collection.csv" | head -1); w=$(find "$OUT/pmc_write_c4" -name "*counter_collection.csv" | head -1)
if [ -n "$f" ] && [ -n "$w" ]; then python3 profiles/pmc_aggregate.py FETCH_SIZE="$f" WRITE_SIZE="$w" > "$OUT/pmc_fetch_write_per_kernel_C4.json" || true; fi
for grp in pmc_mfma pmc_sq pmc_mfma_c4; do
  c=$(find "$OUT/$grp" -name "*counter_collection.csv" | head -1)
  if [ -n "$c" ]; then python3 profiles/pmc_counters.py "$c" > "$OUT/${grp}_per_kernel.txt" || true; fi
done
find "$OUT" -name "*_kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
ls -la "$OUT" | head -40
