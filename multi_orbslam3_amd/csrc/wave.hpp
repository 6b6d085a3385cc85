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

// FP64 sum over the lanes in a fixed (tree) order, returned in every lane: each DPP step moves the two 32-bit halves
__device__ __forceinline__ double wave_sum_f64(double v) {
#define ORBG_DPP64(ctrl, rmask)                                                                    \
  {                                                                                                \
    const int lo = ORBG_DPP(0, __double2loint(v), ctrl, rmask), hi = ORBG_DPP(0, __double2hiint(v), ctrl, rmask); \
    v += __hiloint2double(hi, lo);                                                                 \
  }
  // (row_shr steps with bound_ctrl: a lane without a source reads 0, so no "old" operand has to be zeroed first -- one v_mov less per
  // DPP move; the two row_bcast steps write only the rows of their mask and need the zero in the others)
#define ORBG_DPP64B(ctrl)                                                                          \
  {                                                                                                \
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xF, 0xF, true), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xF, 0xF, true); \
    v += __hiloint2double(hi, lo);                                                                 \
  }
  ORBG_DPP64B(0x111) ORBG_DPP64B(0x112) ORBG_DPP64B(0x114) ORBG_DPP64B(0x118) ORBG_DPP64(0x142, 0xA) ORBG_DPP64(0x143, 0xC)
#undef ORBG_DPP64B
#undef ORBG_DPP64
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

}  // namespace orbg
