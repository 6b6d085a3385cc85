cd /tmp && export TMPDIR=/tmp
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_ldlt
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_ldlt -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-dropin --lba-mode inline > $GRAFT_REPO_ROOT/gpurun_out/prof_ldlt.log 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(find gpurun_out/prof_ldlt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    nm=r['Name'].replace('(anonymous namespace)::','').split('(')[0][:40]
    print("%-42s calls %5s avg %8.1f us total %8.1f us"%(nm,r['Calls'],float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/1e3))
PY
tail -1 gpurun_out/prof_ldlt.log | cut -c1-200
