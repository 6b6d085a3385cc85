"""Timeline of the Frame-constructor chain from a rocprofv3 kernel trace (csv) of a run with the synchronous constructor
(bench.py --no-pipeline --lba-mode inline): per kernel of the chain the average start offset from the chain's first kernel, its
duration and the gap to its predecessor; and the same for the two search chains.   python3 tools/ctor_gaps.py run_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    nm = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').split('<')[0]
    ks.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), nm))
ks.sort()
CTOR = ("img_upload_kernel", "img_upload_pair_kernel", "pyr_tower_kernel", "fast_cells_kernel", "gather_cells_kernel", "fast_gather_kernel", "octree_kernel", "orient_desc_gpu_kernel",
        "stereo_match_kernel", "stereo_grid_kernel", "grid_build_finalize_kernel", "grid_build_kernel", "stereo_finalize_kernel")
chains = []
cur = None
for k in ks:
    if k[2] in CTOR:
        if cur is None or (k[2] in ("img_upload_kernel", "img_upload_pair_kernel", "pyr_tower_kernel") and cur and cur[-1][2] not in ("img_upload_kernel", "img_upload_pair_kernel")):
            if cur:
                chains.append(cur)
            cur = []
        cur.append(k)
    elif cur and cur[-1][2] in ("grid_build_finalize_kernel", "stereo_grid_kernel"):
        chains.append(cur)              # (the constructor posts its completion word itself: the chain ends with its last kernel)
        cur = None
if cur:
    chains.append(cur)
sig = collections.Counter(tuple(k[2] for k in c) for c in chains)
shape, cnt = sig.most_common(1)[0]
sel = [c for c in chains if tuple(k[2] for k in c) == shape][5:]
print("chains %d, most common shape x%d: %s" % (len(chains), cnt, " > ".join(shape)))
tot_k = tot_g = 0.0
for i, nm in enumerate(shape):
    off = sum(c[i][0] - c[0][0] for c in sel) / len(sel) / 1e3
    dur = sum(c[i][1] - c[i][0] for c in sel) / len(sel) / 1e3
    gap = sum(c[i][0] - c[i - 1][1] for c in sel) / len(sel) / 1e3 if i else 0.0
    tot_k += dur; tot_g += gap
    print("  %-28s start +%7.1f us  duration %6.1f us  gap before %5.1f us" % (nm, off, dur, gap))
span = sum(c[-1][1] - c[0][0] for c in sel) / len(sel) / 1e3
print("  first start -> last end: %.1f us (kernels %.1f + gaps %.1f), %d dispatches" % (span, tot_k, tot_g, len(shape)))
# the searches: kernels between two constructor chains
others = collections.defaultdict(list)
for k in ks:
    if k[2].startswith("search_") or k[2].startswith("cull_") or k[2] in ("frustum_kernel",):
        others[k[2]].append((k[1] - k[0]) / 1e3)
for nm, v in others.items():
    print("  %-28s n=%d avg %.1f us" % (nm, len(v), sum(v) / len(v)))
