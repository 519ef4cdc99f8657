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


// The 64-lane sum in the order of the tree
//     for (off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);          (result in lane 0)
// -- the order every block / ticket fold of this library has had since round 1, so the bits of every reduction stay --
// without the LDS crossbar: the two long-distance steps are gfx950's permlane swaps (lanes 0..31 receive lanes 32..63;
// rows 0 and 2 receive rows 1 and 3), the four short ones DPP row shifts.  Lanes that take no part in lane 0's tree hold
// garbage afterwards.  (tools/wave_sum_check.hip compares it with the __shfl_down tree bit for bit.)
__device__ __forceinline__ double wave_sum_down(double v) {
  {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v += __hiloint2double((int)rh[1], (int)rl[1]);  // lanes 0..31: += lane + 32
  }
  {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v += __hiloint2double((int)rh[1], (int)rl[1]);  // lanes 0..15: += lane + 16
  }
  v += dpp_mov<0x108, 0xf>(v);  // row_shl:8
  v += dpp_mov<0x104, 0xf>(v);  // row_shl:4
  v += dpp_mov<0x102, 0xf>(v);  // row_shl:2
  v += dpp_mov<0x101, 0xf>(v);  // row_shl:1
  return v;
}
// ... and in every lane (the value of lane 0 of the tree above = of the xor-butterfly `v += __shfl_xor(v, off)`).
__device__ __forceinline__ double wave_sum_all(double v) {
  v = wave_sum_down(v);
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

}  // namespace storm
