#!/bin/bash
# PMC passes for the LBA kernels (one rocprofv3 run per counter group, no trace domains). Usage on the box: bash tools/pmc_ldlt.sh
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
cd /tmp && export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/pmc_ldlt"
rm -rf "$OUT"; mkdir -p "$OUT"
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin --lba-mode inline --no-pipeline > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if not (nm.startswith("k_ldlt") or nm.startswith("k_schur") or nm.startswith("octree")):
        continue
    a = acc[nm][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for nm, d in acc.items():
    print(nm, {k: round(v[0] / v[1], 1) for k, v in d.items()})
PY
  else
    tail -3 $OUT/$tag.log
  fi
done
