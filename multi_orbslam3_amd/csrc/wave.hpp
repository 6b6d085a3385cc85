// Wavefront-wide (64 lanes) scans and reductions on the DPP data path of gfx950: six v_*_dpp instructions
// (row_shr 1/2/4/8 inside each row of 16 lanes, then row_bcast:15 and row_bcast:31 across rows) instead of six
// ds_bpermute round trips through the LDS crossbar.  All 64 lanes must be active at the call site.
#pragma once

#include <hip/hip_runtime.h>

namespace orbg {

#define ORBG_DPP(ident, v, ctrl, rmask) __builtin_amdgcn_update_dpp((int)(ident), (int)(v), ctrl, rmask, 0xF, false)

// inclusive prefix sum over the lanes
__device__ __forceinline__ int wave_incl_scan_add(int v) {
  v += ORBG_DPP(0, v, 0x111, 0xF);   // row_shr:1
  v += ORBG_DPP(0, v, 0x112, 0xF);   // row_shr:2
  v += ORBG_DPP(0, v, 0x114, 0xF);   // row_shr:4
  v += ORBG_DPP(0, v, 0x118, 0xF);   // row_shr:8
  v += ORBG_DPP(0, v, 0x142, 0xA);   // row_bcast:15 -> rows 1, 3
  v += ORBG_DPP(0, v, 0x143, 0xC);   // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ unsigned wave_incl_scan_add(unsigned v) { return (unsigned)wave_incl_scan_add((int)v); }

// sum / max / min over the lanes, returned wave-uniform (scalar register)
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_incl_scan_add(v), 63); }
__device__ __forceinline__ unsigned wave_sum(unsigned v) { return (unsigned)wave_sum((int)v); }

__device__ __forceinline__ int wave_max(int v) {
  constexpr int I = -2147483647 - 1;
  v = max(v, ORBG_DPP(I, v, 0x111, 0xF));
  v = max(v, ORBG_DPP(I, v, 0x112, 0xF));
  v = max(v, ORBG_DPP(I, v, 0x114, 0xF));
  v = max(v, ORBG_DPP(I, v, 0x118, 0xF));
  v = max(v, ORBG_DPP(I, v, 0x142, 0xA));
  v = max(v, ORBG_DPP(I, v, 0x143, 0xC));
  return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ unsigned wave_min(unsigned v) {
  constexpr unsigned I = 0xFFFFFFFFu;
  v = min(v, (unsigned)ORBG_DPP(I, v, 0x111, 0xF));
  v = min(v, (unsigned)ORBG_DPP(I, v, 0x112, 0xF));
  v = min(v, (unsigned)ORBG_DPP(I, v, 0x114, 0xF));
  v = min(v, (unsigned)ORBG_DPP(I, v, 0x118, 0xF));
  v = min(v, (unsigned)ORBG_DPP(I, v, 0x142, 0xA));
  v = min(v, (unsigned)ORBG_DPP(I, v, 0x143, 0xC));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

}  // namespace orbg
