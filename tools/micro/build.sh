#!/bin/bash
# Builds the stand-alone micro programs for gfx950 (hipcc cross-compiles without a GPU); intermediates stay in /tmp,
# the device assembly of program P is left in /tmp/P.s.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
TMP="$(mktemp -d)"
for p in ${@:-ldlt_mfma_test mfma_f64_latency}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -save-temps=obj -o "$TMP/$p" "$HERE/$p.hip"
  mv "$TMP/$p" "$HERE/$p"
  cp "$TMP/$p"-hip-amdgcn-amd-amdhsa-gfx950.s "/tmp/$p.s"
done
rm -rf "$TMP"
