// Wavefront-level sums without the LDS crossbar (shared by the operator apply and the solver kernels).
#pragma once
#include <hip/hip_runtime.h>

namespace storm {

// 64-lane sum with DPP moves (VALU only; __shfl_down compiles to ds_bpermute, which costs a
// trip through the LDS crossbar per step).  The total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
  v += dpp_mov<0xb1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4e, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x114, 0xf>(v);  // row_shr:4
  v += dpp_mov<0x118, 0xf>(v);  // row_shr:8   -> lanes 12..15 of each row hold the row sum
  v += dpp_mov<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_mov<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return v;
}

}  // namespace storm
