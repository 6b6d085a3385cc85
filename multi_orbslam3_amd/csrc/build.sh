#!/bin/bash
# Builds multi_orbslam3_amd/liborbgpu.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../liborbgpu.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function"
mkdir -p "$HERE/obj"
pids=()
for f in extractor matcher lba pose_opt bow; do
  if [ ! -f "$HERE/obj/$f.o" ] || [ "$HERE/$f.hip" -nt "$HERE/obj/$f.o" ] || [ "$HERE/common.hpp" -nt "$HERE/obj/$f.o" ] || [ "$HERE/wave.hpp" -nt "$HERE/obj/$f.o" ] || [ "$HERE/stereo_finalize.hpp" -nt "$HERE/obj/$f.o" ] || [ "$HERE/ldlt_mfma.hpp" -nt "$HERE/obj/$f.o" ] || [ "$HERE/ldlt_xcd.hpp" -nt "$HERE/obj/$f.o" ] || [ "$HERE/se3.hpp" -nt "$HERE/obj/$f.o" ] || [ "$HERE/grid_build.hpp" -nt "$HERE/obj/$f.o" ] || [ "$HERE/ldlt_jump_tables.inc" -nt "$HERE/obj/$f.o" ] || [ "$HERE/orb_pattern_data.inc" -nt "$HERE/obj/$f.o" ] || [ "$HERE/build.sh" -nt "$HERE/obj/$f.o" ] || [ "$HERE/../../include/orbgpu.h" -nt "$HERE/obj/$f.o" ]; then
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC $FLAGS -x hip -c "$HERE/misc.cpp" -o "$HERE/obj/misc.o"
g++ -O2 -std=c++17 -fPIC -Wall -c "$HERE/vocab_text.cpp" -o "$HERE/obj/vocab_text.o"      # host C++ only: the ORBvoc.txt parser
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$HERE/obj/extractor.o" "$HERE/obj/matcher.o" "$HERE/obj/lba.o" "$HERE/obj/pose_opt.o" "$HERE/obj/bow.o" "$HERE/obj/misc.o" "$HERE/obj/vocab_text.o"
# the Tracking-thread loop above the C-ABI (host C++ only: plain g++ against liborbgpu.so)
g++ -O2 -std=c++17 -fPIC -shared -Wall "$HERE/agent_loop.cpp" -o "$HERE/../libagentloop.so" -L"$HERE/.." -lorbgpu -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib
echo "built $OUT"
