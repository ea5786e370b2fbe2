// gfx950 idioms shared by the kernels: branch-free predicated loads and 16-lane (DPP row) reductions.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- predicated loads without branches -------------------------------------------------------------------
// "load if valid, else zero" as a raw buffer load whose offset is pushed out of range for the invalid lanes (the
// hardware returns 0 and fetches nothing).  With `cond ? *p : 0` the compiler builds branches whose merge copies
// force an s_waitcnt vmcnt(0) in the middle of a burst of loads: one memory round trip each.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned OOB = 0xFFFFFFF0u;
__device__ __forceinline__ rsrc_t mk_rsrc(const void *p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x80000000, 0x00020000);
}
__device__ __forceinline__ f32x4 bl4(rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ float bl1(rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

// ---- reductions over the 16 lanes of a DPP row ------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// quad_perm[1,0,3,2] = 0xB1, quad_perm[2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}
