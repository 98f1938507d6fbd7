// Row-wise top-k out of a 16-bit key image in LDS, for GEMM epilogues that never write their logits (gemm_rows_wide_kernel<MT, true>,
// gemm_skinny_keepx_kernel<..., TOPK>).  One 16-lane group selects the k best of one row: lane l holds columns 32 jj + 2l and + 1
// (one 4-byte read per pair, all reads issued before the first is used) as 32-bit keys = 16-bit monotone value << 16 | 0xffff - column
// (the lowest column wins a tie) and keeps its best FOUR (one max + three v_med3_u32 per key); a round takes the group's
// maximum (four DPP steps), lane 0 writes it as a candidate record in topk.hip's format, the lane that held it moves up its
// next key and looks at its columns again only after four wins.  Slots in [ncols, 32 * ceil(ncols / 32)) and columns the
// producer has no value for must hold key 0: their 32-bit keys lie below every real one (real keys are >= 0x007f0000), and a
// row must have at least k real columns... or the tail candidates come out as empty records (0), which the merge skips.
#pragma once
#include "common.h"

__device__ __forceinline__ void tk_insert4(uint32_t key, uint32_t& b1, uint32_t& b2, uint32_t& b3, uint32_t& b4) {
  uint32_t n2, n3, n4;                                     // b1 >= b2 >= b3 >= b4 stay the four largest keys seen
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(n4) : "v"(b3), "v"(b4), "v"(key));
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(n3) : "v"(b2), "v"(b3), "v"(key));
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(n2) : "v"(b1), "v"(b2), "v"(key));
  b1 = key > b1 ? key : b1; b2 = n2; b3 = n3; b4 = n4;
}
__device__ __forceinline__ uint32_t tk_group16_umax(uint32_t v) {   // the maximum of a 16-lane group, in every lane of the group
  uint32_t t;
  t = __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xf, 0xf, false); v = v > t ? v : t;   // quad_perm [1,0,3,2]
  t = __builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xf, 0xf, false); v = v > t ? v : t;   // quad_perm [2,3,0,1]
  t = __builtin_amdgcn_update_dpp(0u, v, 0x141, 0xf, 0xf, false); v = v > t ? v : t;  // row_half_mirror
  t = __builtin_amdgcn_update_dpp(0u, v, 0x140, 0xf, 0xf, false); v = v > t ? v : t;  // row_mirror
  return v;
}
template <int NPAIR, bool LIMITED>
__device__ __forceinline__ void tk_scan4(const unsigned char* krow, int ncols, int l16, uint32_t limit,
                                         uint32_t& b1, uint32_t& b2, uint32_t& b3, uint32_t& b4) {
  uint32_t w[NPAIR];
#pragma unroll
  for (int jj = 0; jj < NPAIR; ++jj) w[jj] = *reinterpret_cast<const uint32_t*>(krow + 2 * (32 * jj + 2 * l16));   // unguarded: the row holds 32 * NPAIR slots
  b1 = b2 = b3 = b4 = 0;
#pragma unroll
  for (int jj = 0; jj < NPAIR; ++jj) {
    if (32 * jj < ncols) {                                 // uniform
      const uint32_t c = 32 * jj + 2 * l16;
      uint32_t key0 = (w[jj] << 16) | (0xffffu - c);
      uint32_t key1 = (w[jj] & 0xffff0000u) | (0xfffeu - c);
      if (LIMITED) {
        key0 = key0 < limit ? key0 : 0u;
        key1 = key1 < limit ? key1 : 0u;
      }
      tk_insert4(key0, b1, b2, b3, b4);
      tk_insert4(key1, b1, b2, b3, b4);
    }
  }
}
// krow: this row's keys (16 bit each, 32 * NPAIR slots readable); base_idx: global index of column 0; out: k records (written
// by lane 0 of the group when `write`); every lane of the 16-lane group must call it
template <int NPAIR>
__device__ __forceinline__ void tk_select_group16(const unsigned char* krow, int ncols, uint32_t base_idx, int k,
                                                  unsigned long long* out, bool write, int l16) {
  uint32_t b1, b2, b3, b4;
  tk_scan4<NPAIR, false>(krow, ncols, l16, 0u, b1, b2, b3, b4);
  int wins = 0;
  for (int r = 0; r < k; ++r) {
    const uint32_t gm = tk_group16_umax(b1);
    if (l16 == 0 && write) {
      const uint32_t gidx = base_idx + (0xffffu - (gm & 0xffffu));
      out[r] = (gm >> 16) ? (((unsigned long long)(gm & 0xffff0000u) << 32) | (unsigned long long)(0xffffffffu - gidx)) : 0ull;
    }
    if (gm != 0 && b1 == gm) {
      b1 = b2; b2 = b3; b3 = b4; b4 = 0;
      if (++wins == 4) { tk_scan4<NPAIR, true>(krow, ncols, l16, gm, b1, b2, b3, b4); wins = 0; }   // its four known keys are spent
    }
  }
}
// topk.hip's monotone 16-bit key (NaN largest, -0 == +0)
__device__ __forceinline__ uint32_t tk_mono_bf16(bf16_t b) {
  if ((b & 0x7fff) > 0x7f80) return 0xffffu;
  if (b == 0x8000) b = 0;
  return (b & 0x8000) ? (uint32_t)(uint16_t)~b : (uint32_t)(b | 0x8000);
}
