#!/bin/bash
# Builds the in-kernel-clock variants of liborbgpu.so from the CURRENT sources (hipcc cross-compiles without a GPU):
#   variants/liborbgpu_poprof.so   -DPO_PROFILE   (pose_opt.hip: pose_opt_kernel phase stamps, tools/micro/po_prof.py)
#   variants/liborbgpu_octprof.so  -DOCT_PROFILE  (extractor.hip: octree_kernel phase stamps, tools/micro/oct_prof.py)
#   variants/liborbgpu_lbaprof.so  -DLBA_PROFILE  (lba.hip: k_errlin / k_schur / k_update workgroup timelines, tools/micro/lba_prof.py)
# The other objects come from multi_orbslam3_amd/csrc/obj (run csrc/build.sh first).  tools/collect_profiles.sh expects them fresh.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"; C="$HERE/../../multi_orbslam3_amd/csrc"; V="$HERE/variants"; T="$(mktemp -d)"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function"
mkdir -p "$V"
bash "$C/build.sh" > /dev/null
/opt/rocm/bin/hipcc $FLAGS -DPO_PROFILE -c "$C/pose_opt.hip" -o "$T/pose_opt.o" &
/opt/rocm/bin/hipcc $FLAGS -DOCT_PROFILE -c "$C/extractor.hip" -o "$T/extractor.o" &
/opt/rocm/bin/hipcc $FLAGS -DLBA_PROFILE -c "$C/lba.hip" -o "$T/lba.o" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$V/liborbgpu_poprof.so" "$C/obj/extractor.o" "$C/obj/matcher.o" "$C/obj/lba.o" "$T/pose_opt.o" "$C/obj/bow.o" "$C/obj/misc.o" "$C/obj/vocab_text.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$V/liborbgpu_octprof.so" "$T/extractor.o" "$C/obj/matcher.o" "$C/obj/lba.o" "$C/obj/pose_opt.o" "$C/obj/bow.o" "$C/obj/misc.o" "$C/obj/vocab_text.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$V/liborbgpu_lbaprof.so" "$C/obj/extractor.o" "$C/obj/matcher.o" "$T/lba.o" "$C/obj/pose_opt.o" "$C/obj/bow.o" "$C/obj/misc.o" "$C/obj/vocab_text.o"
rm -rf "$T"
echo "built $V/liborbgpu_poprof.so $V/liborbgpu_octprof.so"
