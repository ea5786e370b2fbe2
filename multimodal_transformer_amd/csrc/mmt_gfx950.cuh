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
// same with a wave-uniform part of the offset in an SGPR (not range-checked: push `byte_off` out of range to skip)
__device__ __forceinline__ f32x4 bl4s(rsrc_t r, unsigned byte_off, unsigned uniform_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, uniform_off, 0));
}
// cache-policy forms for data that workgroups hand to each other inside one launch: AUX_SC1 stores write through to
// memory, AUX_SC1 loads bypass this CU's L1 (which no other CU's store ever refreshes)
constexpr int AUX_PLAIN = 0, AUX_SC1 = 16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int AUX>
__device__ __forceinline__ f32x4 bl4x(rsrc_t r, unsigned byte_off, unsigned uniform_off = 0u) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, uniform_off, AUX));
}
template <int AUX>
__device__ __forceinline__ void bs4x(rsrc_t r, unsigned byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, AUX);
}
typedef __attribute__((address_space(1))) unsigned gu32;      // global (never flat) words that workgroups poll / raise
typedef __attribute__((address_space(1))) float gf32;
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

// ---- sums across the four DPP rows of a wave --------------------------------------------------------------
// v_permlane{16,32}_swap exchange half of the rows between two registers; fed the same value twice they leave the
// two registers holding lane l and lane l^16 (l^32) between them.  (Inline asm: the builtin pairs the results wrongly
// in this compiler.  s_nop 1 covers the VALU-write -> permlane-swap-read hazard the assembler does not see.)
__device__ __forceinline__ float xor16_sum(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float xor32_sum(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
// two values for the price of one: lanes 0..31 get x[l] + x[l + 32], lanes 32..63 get y[l - 32] + y[l]
// (s_nop 4: x and y may come straight from an MFMA, whose write -> VALU read distance the assembler does not see in here)
__device__ __forceinline__ float pair32_sum(float x, float y) {
  asm("s_nop 4\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  return x + y;
}
// even DPP rows get x[row] + x[row + 1], odd rows y[row - 1] + y[row]
__device__ __forceinline__ float pair16_sum(float x, float y) {
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
  return x + y;
}
__device__ __forceinline__ float wave64_sum(float v) { return xor32_sum(xor16_sum(row16_sum(v))); }
// the same total as ONE wave-uniform value (it lands in an SGPR): row sums, then row_bcast:15 into rows 1 and 3, row_bcast:31
// into rows 2 and 3, and lane 63 is read out - 7 instructions against the 18 of wave64_sum
__device__ __forceinline__ float wave64_total(float v) {
  v = row16_sum(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// 4x4 transpose inside every quad of lanes: lane j's register r  <->  lane r's register j
__device__ __forceinline__ void quad_transpose(f32x4 &a, int j) {
  float s, r;
  s = (j & 1) ? a[0] : a[1]; r = dpp_mov<0xB1>(s); if (j & 1) a[0] = r; else a[1] = r;
  s = (j & 1) ? a[2] : a[3]; r = dpp_mov<0xB1>(s); if (j & 1) a[2] = r; else a[3] = r;
  s = (j & 2) ? a[0] : a[2]; r = dpp_mov<0x4E>(s); if (j & 2) a[0] = r; else a[2] = r;
  s = (j & 2) ? a[1] : a[3]; r = dpp_mov<0x4E>(s); if (j & 2) a[1] = r; else a[3] = r;
}
