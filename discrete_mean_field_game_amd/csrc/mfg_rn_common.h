// Building blocks shared by the reward-network kernels (mfg_reward_net.hip) and the reward network evaluated INSIDE the
// packed step kernel (mfg_rn_fused.h): DPP wave sums, lane shifts, scalar-cache pointers.
#pragma once
#include "mfg_device.h"

namespace mfg {

typedef float rn_v4f_t __attribute__((ext_vector_type(4)));
typedef float rn_v2f_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wave_sum_f32_dpp(float v) {
  v += dpp_mov_f32<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov_f32<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov_f32<0x141, 0xF>(v);  // row_half_mirror
  v += dpp_mov_f32<0x140, 0xF>(v);  // row_mirror
  v += dpp_mov_f32<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_mov_f32<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Four wave sums advancing together: lane 63 ends up with the four totals.  One block of 24 DPP adds -- every step's four
// instructions are independent and separate an instruction from the one that reads its result (the two wait states a DPP
// source needs); the last two steps add lane 15 / 31 of the previous rows into rows {1, 3} / {2, 3} in place (as separate
// move + add they are three instructions each).
__device__ __forceinline__ void wave_sum4_to_lane63(float (&v)[4]) {
  asm volatile(
      "s_nop 1\n"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
      "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
      "v_add_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
      "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
      "v_add_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
      "s_nop 1\n"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}


typedef const __attribute__((address_space(4))) float* RnConstF;  // read-only global memory: uniform reads are scalar loads
// max(x, 0) in ONE instruction (fmaxf first quiets a signalling NaN with a v_max_f32 x, x, x of its own)
__device__ __forceinline__ float relu_f32(float x) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}
// wave shifts by one lane (DPP wave_shr:1 / wave_shl:1): lane l takes the value of lane l - 1 / l + 1
__device__ __forceinline__ float lane_below(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_above(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, true));
}


}  // namespace mfg
