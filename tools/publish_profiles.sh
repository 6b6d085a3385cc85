#!/bin/bash
# Copies the judged summaries of gpurun_out/<tag>/ (written by tools/collect_profiles.sh on the GPU box) into profiles/<tag>_*.
set -euo pipefail
TAG=${1:?tag}; S=gpurun_out/$TAG; D=profiles
for f in bench_line_default bench_line_driver_20_steps bench_line_sync_ctor bench_line_inline bench_line_c4 bench_line_mono bench_line_mono_dist bench_line_c3_1rank bench_line_c5_2ranks_shared_gpu bench_line_server_tick dropin_bench pmc_fetch_write_per_kernel pmc_fetch_write_per_kernel_C4; do
  [ -s $S/$f.json ] && cp $S/$f.json $D/${TAG}_$f.json
done
for f in dropin_bench lba_time ldlt_xcd_timeline lba_sizes_xcd_off_on micro_readlane_chain micro_fp64_issue pmc_sq_per_kernel pose_opt_phases pose_opt_time search_large_map search_large_map_dense pose_opt_sweep ctor_gaps bench_driver_repeat lba_timeline_agent octree_phases lba_workgroup_timelines closed_loop rig_loop vocab_full_size rig_time rig_match_time; do
  [ -s $S/$f.txt ] && cp $S/$f.txt $D/${TAG}_$f.txt
done
[ -s $S/pmc_mfma_per_kernel.txt ] && cp $S/pmc_mfma_per_kernel.txt $D/${TAG}_pmc_mfma_f64_per_kernel.txt
[ -s $S/pmc_mfma_c4_per_kernel.txt ] && cp $S/pmc_mfma_c4_per_kernel.txt $D/${TAG}_pmc_mfma_f64_per_kernel_C4.txt
[ -s $S/lba_gaps.txt ] && cp $S/lba_gaps.txt $D/${TAG}_lba_stream_gaps.txt
for p in "stats:bench_steps100_inline" "stats_async:bench_steps100_async_pipelined" "stats_c4:bench_C4_steps40_inline"; do
  d=${p%%:*}; n=${p##*:}
  f=$(find $S/$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $D/${TAG}_kernel_stats_$n.csv
done
ls $D | grep "^${TAG}_"
