// Key encoding and wavefront-level select shared by the top-k kernels (topk.hip) and the lm_head
// weight-stream GEMM that emits stage-1 candidates from its epilogue (gemm_skinny.hip).
#pragma once
#include "common.h"

__device__ __forceinline__ uint32_t mono_bf16(bf16_t b) {  // 16-bit monotone key in the low half
  if ((b & 0x7fff) > 0x7f80) return 0xffffu;       // NaN: largest
  if (b == 0x8000) b = 0;                          // -0.0 == +0.0
  return (b & 0x8000) ? (uint32_t)(uint16_t)~b : (uint32_t)(b | 0x8000);
}
__device__ __forceinline__ uint32_t mono_f32(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
  if (u == 0x80000000u) u = 0;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// full-wave unsigned max, result uniform (read from lane 63)
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) {
  uint32_t t;
  t = __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xf, 0xf, false); v = v > t ? v : t;   // quad_perm [1,0,3,2]
  t = __builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xf, 0xf, false); v = v > t ? v : t;   // quad_perm [2,3,0,1]
  t = __builtin_amdgcn_update_dpp(0u, v, 0x141, 0xf, 0xf, false); v = v > t ? v : t;  // row_half_mirror
  t = __builtin_amdgcn_update_dpp(0u, v, 0x140, 0xf, 0xf, false); v = v > t ? v : t;  // row_mirror
  t = __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false); v = v > t ? v : t;  // row_bcast15 -> rows 1,3
  t = __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false); v = v > t ? v : t;  // row_bcast31 -> rows 2,3
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned long long wave_umax64(unsigned long long v) {
  const uint32_t hi = wave_umax((uint32_t)(v >> 32));
  const uint32_t lo = wave_umax(((uint32_t)(v >> 32) == hi) ? (uint32_t)v : 0u);
  return ((unsigned long long)hi << 32) | lo;
}

