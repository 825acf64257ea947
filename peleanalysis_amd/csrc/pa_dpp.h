// pa_dpp.h -- a double from the neighbouring lane of the wavefront (DPP wavefront shifts): the x-neighbour of a cell whose row of
// 64 cells is held one cell per lane.  Lane 0 (wave_shr) / lane 63 (wave_shl) keep their own value: the caller loads those.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ double lane_from_left(double v) {  // lane n <- lane n - 1
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xF, 0xF, false);  // wave_shr:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_from_right(double v) {  // lane n <- lane n + 1
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xF, 0xF, false);  // wave_shl:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
