// Per-row top-k over the vocabulary with a stated, deterministic tie-break (gfx950).
//
// Replaces torch.topk(last_headout, top_k) of the EAGLE-1 expansion loop (top_k = 10,
// eagle/llama2.py:11) and the first-index argmax of the agreement count
// (torch.max(..., 2), eagle/train.py:115-116) / greedy verification (row a22) with k = 1.
// Order: value descending, then index ascending (== a stable descending sort); -0.0 == +0.0;
// NaN sorts above +inf (torch.sort semantics).  Results are bit-exact against the oracle.
//
// Wavefront-level select: every element becomes an unsigned key whose max is "largest value,
// lowest index"; a wave finds its maximum with a 6-step DPP reduction (quad_perm, row mirrors,
// row_bcast15/31 -- no LDS, no ds_bpermute), the owning lane retires it, k rounds.
//   stage 1: one wave per 1024 logits (16 per lane, 32 contiguous bytes): k candidates per wave.
//            bf16 logits use a single 32-bit key (16-bit monotone value | 10-bit local index).
//   stage 2: one workgroup per row: each wave reduces its share of the candidates to k, wave 0
//            merges the 4 x k survivors and writes ids (int64) and values.
#include "common.h"

#define TK_THREADS 256
#define TK_WCHUNK 1024                 // logits per wave in stage 1
#define TK_PER_LANE (TK_WCHUNK / 64)   // 16
#define TK_MAXK 16
#define TK_CM 8                        // stage-2 candidates per thread (16 beyond 2048 candidates) -> wavechunks*k <= 4096

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

// ---- stage 1, bf16: 32-bit keys ----
// (m, s) <- log-sum-exp merge with (m2, s2): running maximum and sum of exp(x - m)
__device__ __forceinline__ void tk_lse_merge(float& m, float& s, float m2, float s2) {
  const float M = fmaxf(m, m2);
  if (M == -INFINITY) return;
  s = s * __expf(m - M) + s2 * __expf(m2 - M);
  m = M;
}

// lse_part (may be null): [rows][nwc] (max, sum of exp(x - max)) of every wave chunk, for a row log-sum-exp out of the
// same pass over the logits (the dynamic tree scores candidates by softmax probability: eagle_topk_lse_vocab).
template <bool LSE>
__global__ __launch_bounds__(TK_THREADS) void topk_stage1_bf16_kernel(const bf16_t* __restrict__ x, int64_t ld, int V, int k,
                                                                     int nwc, unsigned long long* __restrict__ cand,
                                                                     const int64_t* __restrict__ block_index, int64_t block_rows,
                                                                     float2* __restrict__ lse_part) {
  const int row = blockIdx.y;
  if (block_index) x += *block_index * block_rows * ld;
  const int lane = threadIdx.x & 63;
  const int wc = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wc >= nwc) return;
  const bf16_t* xr = x + (int64_t)row * ld;
  const int base = wc * TK_WCHUNK + lane * TK_PER_LANE;
  uint32_t key[TK_PER_LANE];
  const bool vec = ((ld & 7) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && (base + TK_PER_LANE <= V);
  if (vec) {
    const uint4 a = *reinterpret_cast<const uint4*>(xr + base);
    const uint4 b = *reinterpret_cast<const uint4*>(xr + base + 8);
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      key[2 * i] = (mono_bf16((bf16_t)(w[i] & 0xffff)) << 16) | (uint32_t)(TK_WCHUNK - 1 - (lane * TK_PER_LANE + 2 * i));
      key[2 * i + 1] = (mono_bf16((bf16_t)(w[i] >> 16)) << 16) | (uint32_t)(TK_WCHUNK - 1 - (lane * TK_PER_LANE + 2 * i + 1));
    }
  } else {
#pragma unroll
    for (int i = 0; i < TK_PER_LANE; ++i) {
      const int idx = base + i;
      // invalid slots get key 0
      key[i] = idx < V ? ((mono_bf16(xr[idx]) << 16) | (uint32_t)(TK_WCHUNK - 1 - (lane * TK_PER_LANE + i))) : 0u;
    }
  }
  if (LSE) {
    // the monotone key orders the values, so the lane's maximum comes from its best key; elements that are -inf,
    // NaN or beyond V contribute nothing (as in row_lse_kernel)
    float f[TK_PER_LANE];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < TK_PER_LANE; ++i) {
      const int idx = base + i;
      const float v = idx < V ? bf2f(xr[idx]) : -INFINITY;
      f[i] = (v > -INFINITY) ? v : -INFINITY;          // NaN compares false
      mx = fmaxf(mx, f[i]);
    }
    float sm = 0.f;
    if (mx > -INFINITY) {
#pragma unroll
      for (int i = 0; i < TK_PER_LANE; ++i) sm += __expf(f[i] - mx);   // exp(-inf) = 0
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float m2 = __shfl_xor(mx, o, 64), s2 = __shfl_xor(sm, o, 64);
      tk_lse_merge(mx, sm, m2, s2);
    }
    if (lane == 0) lse_part[(int64_t)row * nwc + wc] = make_float2(mx, sm);
  }
  // valid keys are > 0: the smallest value part is 0x007f (-inf); NaN patterns map to 0xffff
  // The selection is VALU-bound (19072 waves at 128 rows x 151936 logits), and most of it was the winner's rescan of its 16
  // keys after every round.  Each lane keeps its best AND its second-best key (keys are distinct: the slot is part of the
  // key); a lane that wins falls back on its second key at once and rescans only if it wins again before that -- 0.7
  // times per wave at k = 10 instead of 10 times.  Winners leave in decreasing order, so a lane's consumed keys are
  // exactly its keys >= the key it has just lost.
  uint32_t best = 0, second = 0;
#pragma unroll
  for (int i = 0; i < TK_PER_LANE; ++i) {
    const uint32_t lo = key[i] < best ? key[i] : best;
    second = lo > second ? lo : second;
    best = key[i] > best ? key[i] : best;
  }
  bool second_valid = true;
  unsigned long long* out = cand + ((int64_t)row * nwc + wc) * k;
  for (int r = 0; r < k; ++r) {
    const uint32_t bm = wave_umax(best);
    if (lane == 0) {
      unsigned long long o = 0;
      if (bm) {
        const uint32_t local = TK_WCHUNK - 1 - (bm & 0xffffu);
        const uint32_t gidx = (uint32_t)(wc * TK_WCHUNK) + local;
        o = ((unsigned long long)(bm & 0xffff0000u) << 32) | (unsigned long long)(0xffffffffu - gidx);
      }
      out[r] = o;
    }
    if (bm != 0 && best == bm) {
      if (second_valid) {
        best = second;
        second_valid = false;
      } else {
        best = 0; second = 0;
#pragma unroll
        for (int i = 0; i < TK_PER_LANE; ++i) {
          const uint32_t kk = key[i] < bm ? key[i] : 0u;
          const uint32_t lo = kk < best ? kk : best;
          second = lo > second ? lo : second;
          best = kk > best ? kk : best;
        }
        second_valid = true;
      }
    }
  }
}

// ---- stage 1, fp32: 64-bit keys ----
__global__ __launch_bounds__(TK_THREADS) void topk_stage1_f32_kernel(const float* __restrict__ x, int64_t ld, int V, int k,
                                                                    int nwc, unsigned long long* __restrict__ cand,
                                                                    const int64_t* __restrict__ block_index, int64_t block_rows) {
  const int row = blockIdx.y;
  if (block_index) x += *block_index * block_rows * ld;
  const int lane = threadIdx.x & 63;
  const int wc = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wc >= nwc) return;
  const float* xr = x + (int64_t)row * ld;
  const int base = wc * TK_WCHUNK + lane * TK_PER_LANE;
  unsigned long long key[TK_PER_LANE];
#pragma unroll
  for (int i = 0; i < TK_PER_LANE; ++i) {
    const int idx = base + i;
    key[i] = idx < V ? (((unsigned long long)mono_f32(xr[idx]) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)idx)) : 0ull;
  }
  unsigned long long best = 0;
#pragma unroll
  for (int i = 0; i < TK_PER_LANE; ++i) best = key[i] > best ? key[i] : best;
  unsigned long long* out = cand + ((int64_t)row * nwc + wc) * k;
  for (int r = 0; r < k; ++r) {
    const unsigned long long bm = wave_umax64(best);
    if (lane == 0) out[r] = bm;
    if (bm != 0 && best == bm) {
      best = 0;
#pragma unroll
      for (int i = 0; i < TK_PER_LANE; ++i) {
        if (key[i] == bm) key[i] = 0;
        best = key[i] > best ? key[i] : best;
      }
    }
  }
}

// K32 (every candidate index < 65536): the merge runs on 32-bit keys (16-bit monotone value | 0xffff - index), one DPP
// reduction per selection round instead of the two of a 64-bit key.
// CM = candidates per thread (CM * 256 >= ncand); x == nullptr: the values are rebuilt from the records' 16-bit keys (bf16 only;
// -0 comes back as +0): the records then need not come from a logits array at all (eagle_lm_head_topk_rows_bf16)
template <typename T, bool LSE, bool K32 = false, int CM = TK_CM>
__global__ __launch_bounds__(TK_THREADS) void topk_stage2_kernel(const unsigned long long* __restrict__ cand, int ncand, int k,
                                                                const T* __restrict__ x, int64_t ld,
                                                                int64_t* __restrict__ out_idx, T* __restrict__ out_val,
                                                                const int64_t* __restrict__ block_index, int64_t block_rows,
                                                                const float2* __restrict__ lse_part, int nwc,
                                                                float* __restrict__ lse_out, int out_group, int64_t out_gstride) {
  __shared__ unsigned long long surv[4 * TK_MAXK];
  if (block_index) x += *block_index * block_rows * ld;
  const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (K32) {
    // candidate c = value16 << 48 | (0xffffffff - index): with index < 65536 the low half is 0xffff0000 | (0xffff - index)
    uint32_t key[CM];
    uint32_t best = 0;
#pragma unroll
    for (int i = 0; i < CM; ++i) {
      const int c = t + i * TK_THREADS;
      const unsigned long long v = c < ncand ? cand[(int64_t)row * ncand + c] : 0ull;
      key[i] = v ? ((uint32_t)(v >> 32) | ((uint32_t)v & 0xffffu)) : 0u;
      best = key[i] > best ? key[i] : best;
    }
    for (int r = 0; r < k; ++r) {
      const uint32_t bm = wave_umax(best);
      if (lane == 0) surv[wave * TK_MAXK + r] = bm ? (((unsigned long long)(bm & 0xffff0000u) << 32) | 0xffff0000ull | (bm & 0xffffu)) : 0ull;
      if (bm != 0 && best == bm) {
        best = 0;
#pragma unroll
        for (int i = 0; i < CM; ++i) {
          if (key[i] == bm) key[i] = 0;
          best = key[i] > best ? key[i] : best;
        }
      }
    }
  } else {
  unsigned long long key[CM];
#pragma unroll
  for (int i = 0; i < CM; ++i) {
    const int c = t + i * TK_THREADS;
    key[i] = c < ncand ? cand[(int64_t)row * ncand + c] : 0ull;
  }
  unsigned long long best = 0;
#pragma unroll
  for (int i = 0; i < CM; ++i) best = key[i] > best ? key[i] : best;
  for (int r = 0; r < k; ++r) {
    const unsigned long long bm = wave_umax64(best);
    if (lane == 0) surv[wave * TK_MAXK + r] = bm;
    if (bm != 0 && best == bm) {
      best = 0;
#pragma unroll
      for (int i = 0; i < CM; ++i) {
        if (key[i] == bm) key[i] = 0;
        best = key[i] > best ? key[i] : best;
      }
    }
  }
  }
  __syncthreads();
  if (LSE && wave == 1) {                  // row log-sum-exp from the stage-1 partials, beside wave 0's final merge
    float m = -INFINITY, sm = 0.f;
    for (int c = lane; c < nwc; c += 64) {
      const float2 q = lse_part[(int64_t)row * nwc + c];
      tk_lse_merge(m, sm, q.x, q.y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(sm, o, 64);
      tk_lse_merge(m, sm, m2, s2);
    }
    if (lane == 0) lse_out[row] = m + __logf(sm);
  }
  if (wave != 0) return;
  unsigned long long mine = 0;
  if (lane < 4 * k) mine = surv[(lane / k) * TK_MAXK + (lane % k)];
  unsigned long long won = 0;                  // lane r keeps the r-th winner: the k value loads and
  if (K32) {                                   // stores below then go out together, not one per round
    uint32_t m32 = mine ? ((uint32_t)(mine >> 32) | ((uint32_t)mine & 0xffffu)) : 0u;
    for (int r = 0; r < k; ++r) {
      const uint32_t bm = wave_umax(m32);
      if (lane == r) won = bm ? (0xffff0000ull | (bm & 0xffffu)) : 0ull;      // only the low (index) half is read below
      if (bm != 0 && m32 == bm) m32 = 0;
    }
  } else {
    for (int r = 0; r < k; ++r) {
      const unsigned long long bm = wave_umax64(mine);
      if (lane == r) won = bm;
      if (bm != 0 && mine == bm) mine = 0;
    }
  }
  if (lane < k) {
    // fewer than k entries can only happen when V < k; emit index -1 then
    const int64_t idx = won ? (int64_t)(0xffffffffu - (uint32_t)(won & 0xffffffffu)) : -1;
    // output row: groups of out_group rows, out_gstride elements apart (out_group = 0: plain [rows, k])
    const int64_t opos = out_group ? (int64_t)(row / out_group) * out_gstride + (int64_t)(row % out_group) * k : (int64_t)row * k;
    out_idx[opos + lane] = idx;
    if (out_val) {
      if (x) out_val[opos + lane] = idx >= 0 ? x[(int64_t)row * ld + idx] : (T)0;
      else if (sizeof(T) == 2) {
        const uint32_t k16 = (uint32_t)(won >> 48);
        const uint32_t b = k16 == 0xffffu ? 0x7fc0u : (k16 & 0x8000u) ? (k16 & 0x7fffu) : (~k16 & 0xffffu);
        out_val[opos + lane] = idx >= 0 ? (T)b : (T)0;
      }
    }
  }
}

// ---- one launch for small vocabularies (bf16, <= 64 wave chunks: V <= 65536) ----
// One 16-wave workgroup per row: wave w selects the k best of its CPW wave chunks {w, w+16, ...} in ONE selection (keys =
// 16-bit monotone value | 12-bit slot among the wave's CPW*1024 logits, ordered so that the lowest global index wins a
// tie: the wave's chunks are taken in ascending order), the 16 x k survivors go through LDS and wave 0 merges them -- the
// same total order of (value, index) as the two-launch form, so the same result, in two selection passes instead of
// three and without the second launch (a dependent small kernel costs ~5 us on the draft cycle: six top-k calls a cycle).
#define TK_FUSED_WAVES 16
#define TK_FUSED_MAXWC 64
template <int CPW>
__global__ __launch_bounds__(TK_FUSED_WAVES * 64) void topk_fused_bf16_kernel(const bf16_t* __restrict__ x, int64_t ld, int V, int k, int nwc,
                                                                            int64_t* __restrict__ out_idx, bf16_t* __restrict__ out_val,
                                                                            const int64_t* __restrict__ block_index, int64_t block_rows,
                                                                            int out_group, int64_t out_gstride) {
  __shared__ unsigned long long surv[TK_FUSED_WAVES * TK_MAXK];
  const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (block_index) x += *block_index * block_rows * ld;
  const bf16_t* xr = x + (int64_t)row * ld;
  const bool al = ((ld & 7) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  constexpr int NKEY = CPW * TK_PER_LANE;
  constexpr uint32_t SLOTS = CPW * TK_WCHUNK;                        // <= 4096: 12 bits
  uint32_t key[NKEY];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int wc = wave + c * TK_FUSED_WAVES;
    const int base = wc * TK_WCHUNK + lane * TK_PER_LANE;
    const uint32_t slot0 = (uint32_t)(c * TK_WCHUNK + lane * TK_PER_LANE);   // ascending with the global index inside a wave
    if (wc < nwc && al && base + TK_PER_LANE <= V) {
      const uint4 a = *reinterpret_cast<const uint4*>(xr + base);
      const uint4 b = *reinterpret_cast<const uint4*>(xr + base + 8);
      const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        key[c * TK_PER_LANE + 2 * i] = (mono_bf16((bf16_t)(w[i] & 0xffff)) << 16) | (SLOTS - 1 - (slot0 + 2 * i));
        key[c * TK_PER_LANE + 2 * i + 1] = (mono_bf16((bf16_t)(w[i] >> 16)) << 16) | (SLOTS - 1 - (slot0 + 2 * i + 1));
      }
    } else {
#pragma unroll
      for (int i = 0; i < TK_PER_LANE; ++i) {
        const int idx = base + i;
        key[c * TK_PER_LANE + i] = (wc < nwc && idx < V) ? ((mono_bf16(xr[idx]) << 16) | (SLOTS - 1 - (slot0 + i))) : 0u;
      }
    }
  }
  uint32_t best = 0;
#pragma unroll
  for (int i = 0; i < NKEY; ++i) best = key[i] > best ? key[i] : best;
  for (int r = 0; r < k; ++r) {
    const uint32_t bm = wave_umax(best);
    if (lane == 0) {
      unsigned long long o = 0;
      if (bm) {
        const uint32_t slot = SLOTS - 1 - (bm & 0xffffu);
        const uint32_t gidx = (uint32_t)((wave + (slot >> 10) * TK_FUSED_WAVES) * TK_WCHUNK) + (slot & (TK_WCHUNK - 1));
        o = ((unsigned long long)(bm & 0xffff0000u) << 32) | (unsigned long long)(0xffffffffu - gidx);
      }
      surv[wave * TK_MAXK + r] = o;
    }
    if (bm != 0 && best == bm) {
      best = 0;
#pragma unroll
      for (int i = 0; i < NKEY; ++i) {
        if (key[i] == bm) key[i] = 0;
        best = key[i] > best ? key[i] : best;
      }
    }
  }
  __syncthreads();
  if (wave != 0) return;
  unsigned long long key2[4];                                        // 16 x k <= 256 survivors: four per lane
  unsigned long long best2 = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    key2[i] = c < TK_FUSED_WAVES * k ? surv[(c / k) * TK_MAXK + (c % k)] : 0ull;
    best2 = key2[i] > best2 ? key2[i] : best2;
  }
  unsigned long long won = 0;
  for (int r = 0; r < k; ++r) {
    const unsigned long long bm = wave_umax64(best2);
    if (lane == r) won = bm;
    if (bm != 0 && best2 == bm) {
      best2 = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (key2[i] == bm) key2[i] = 0;
        best2 = key2[i] > best2 ? key2[i] : best2;
      }
    }
  }
  if (lane < k) {
    const int64_t idx = won ? (int64_t)(0xffffffffu - (uint32_t)(won & 0xffffffffu)) : -1;
    const int64_t opos = out_group ? (int64_t)(row / out_group) * out_gstride + (int64_t)(row % out_group) * k : (int64_t)row * k;
    out_idx[opos + lane] = idx;
    if (out_val) out_val[opos + lane] = idx >= 0 ? xr[idx] : (bf16_t)0;
  }
}

// ---- the merge alone on 16 waves: rows x ncand candidate records (<= 6144) -> k best per row ----
// For producers that leave many short candidate lists (the lm_head epilogue of the one-request tree steps: 250 workgroups x k
// records per row).  topk_stage2_kernel holds CM = 16-24 records per thread on 4 waves and rescans them every round it wins;
// here a thread holds at most CW (2500 records: 3), every wave selects its k best, wave 0 merges the 16 x k survivors --
// the structure of topk_fused_bf16_kernel's second half.  Same total order (record value = key << 48 | ~index).
// K32 (every candidate index < 65536): both selections run on 32-bit keys (16-bit value | 0xffff - index), one DPP reduction per round.
template <int CW, bool K32>
__global__ __launch_bounds__(TK_FUSED_WAVES * 64) void topk_merge_wide_kernel(const unsigned long long* __restrict__ cand, int ncand, int k,
                                                                            int64_t* __restrict__ out_idx, bf16_t* __restrict__ out_val,
                                                                            int out_group, int64_t out_gstride) {
  __shared__ unsigned long long surv[TK_FUSED_WAVES * TK_MAXK];
  const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  unsigned long long won = 0;
  if (K32) {
    uint32_t* surv32 = reinterpret_cast<uint32_t*>(surv);
    uint32_t key[CW], best = 0;
#pragma unroll
    for (int i = 0; i < CW; ++i) {
      const int c = t + i * (TK_FUSED_WAVES * 64);
      const unsigned long long v = c < ncand ? cand[(int64_t)row * ncand + c] : 0ull;
      key[i] = v ? ((uint32_t)(v >> 32) | ((uint32_t)v & 0xffffu)) : 0u;
      best = key[i] > best ? key[i] : best;
    }
    for (int r = 0; r < k; ++r) {
      const uint32_t bm = wave_umax(best);
      if (lane == 0) surv32[wave * TK_MAXK + r] = bm;
      if (bm != 0 && best == bm) {
        best = 0;
#pragma unroll
        for (int i = 0; i < CW; ++i) {
          if (key[i] == bm) key[i] = 0;
          best = key[i] > best ? key[i] : best;
        }
      }
    }
    __syncthreads();
    if (wave != 0) return;
    uint32_t key2[4], best2 = 0;                                     // 16 x k <= 256 survivors: four per lane
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      key2[i] = c < TK_FUSED_WAVES * k ? surv32[(c / k) * TK_MAXK + (c % k)] : 0u;
      best2 = key2[i] > best2 ? key2[i] : best2;
    }
    for (int r = 0; r < k; ++r) {
      const uint32_t bm = wave_umax(best2);
      if (lane == r) won = bm ? (((unsigned long long)(bm & 0xffff0000u) << 32) | 0xffff0000ull | (bm & 0xffffu)) : 0ull;
      if (bm != 0 && best2 == bm) {
        best2 = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (key2[i] == bm) key2[i] = 0;
          best2 = key2[i] > best2 ? key2[i] : best2;
        }
      }
    }
  } else {
  unsigned long long key[CW];
  unsigned long long best = 0;
#pragma unroll
  for (int i = 0; i < CW; ++i) {
    const int c = t + i * (TK_FUSED_WAVES * 64);
    key[i] = c < ncand ? cand[(int64_t)row * ncand + c] : 0ull;
    best = key[i] > best ? key[i] : best;
  }
  for (int r = 0; r < k; ++r) {
    const unsigned long long bm = wave_umax64(best);
    if (lane == 0) surv[wave * TK_MAXK + r] = bm;
    if (bm != 0 && best == bm) {
      best = 0;
#pragma unroll
      for (int i = 0; i < CW; ++i) {
        if (key[i] == bm) key[i] = 0;
        best = key[i] > best ? key[i] : best;
      }
    }
  }
  __syncthreads();
  if (wave != 0) return;
  unsigned long long key2[4];                                        // 16 x k <= 256 survivors: four per lane
  unsigned long long best2 = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    key2[i] = c < TK_FUSED_WAVES * k ? surv[(c / k) * TK_MAXK + (c % k)] : 0ull;
    best2 = key2[i] > best2 ? key2[i] : best2;
  }
  for (int r = 0; r < k; ++r) {
    const unsigned long long bm = wave_umax64(best2);
    if (lane == r) won = bm;
    if (bm != 0 && best2 == bm) {
      best2 = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (key2[i] == bm) key2[i] = 0;
        best2 = key2[i] > best2 ? key2[i] : best2;
      }
    }
  }
  }
  if (lane < k) {
    const int64_t idx = won ? (int64_t)(0xffffffffu - (uint32_t)(won & 0xffffffffu)) : -1;
    const int64_t opos = out_group ? (int64_t)(row / out_group) * out_gstride + (int64_t)(row % out_group) * k : (int64_t)row * k;
    out_idx[opos + lane] = idx;
    if (out_val) {                                                   // the value back from the record's 16-bit key (-0 comes back as +0)
      const uint32_t k16 = (uint32_t)(won >> 48);
      const uint32_t b = k16 == 0xffffu ? 0x7fc0u : (k16 & 0x8000u) ? (k16 & 0x7fffu) : (~k16 & 0xffffu);
      out_val[opos + lane] = idx >= 0 ? (bf16_t)b : (bf16_t)0;
    }
  }
}

// k = 1 over many long rows (the verifier's argmax at every tree node of a request batch: 832 rows x 151936 logits): one
// workgroup per row reads the row once and writes the index -- no candidate buffer and no second launch (the two-launch
// form spends 126k one-kilobyte waves on it: 62 + 4.7 us at 4.1 TB/s).  Same order as everywhere: the monotone 16-bit key,
// NaN largest, -0 == +0, lowest index among equals (a thread scans ascending indices and takes strict improvements only).
#define TK_AM_THREADS 256
typedef unsigned short tk_u16x2 __attribute__((ext_vector_type(2)));
typedef short tk_i16x2 __attribute__((ext_vector_type(2)));
// two monotone keys from one word of two bf16 (packed 16-bit ops): exact for everything but NaN and -0, which the caller
// detects (raw halves > 0xff80, keys > 0xff80, a row maximum of +-0) and sends down the exact path
__device__ __forceinline__ tk_u16x2 tk_fast_keys(uint32_t w) {
  union { uint32_t u; tk_u16x2 v; tk_i16x2 s; } c; c.u = w;
  const tk_i16x2 sign = c.s >> 15;                                    // 0xffff for negative halves
  union { tk_i16x2 s; tk_u16x2 v; } m; m.s = sign;
  return c.v ^ (m.v | (tk_u16x2){0x8000, 0x8000});
}
__global__ __launch_bounds__(TK_AM_THREADS) void argmax_rows_bf16_kernel(const bf16_t* __restrict__ x, int64_t ld, int V,
                                                                        int64_t* __restrict__ out_idx, bf16_t* __restrict__ out_val,
                                                                        const int64_t* __restrict__ block_index, int64_t block_rows,
                                                                        int out_group, int64_t out_gstride) {
  __shared__ unsigned long long wbest[TK_AM_THREADS / 64];
  __shared__ int wflag[TK_AM_THREADS / 64];
  const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (block_index) x += *block_index * block_rows * ld;
  const bf16_t* xr = x + (int64_t)row * ld;
  const bool al = ((ld & 7) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  const int nvec = al ? V >> 3 : 0;
  auto reduce = [&](unsigned long long key, int flag, unsigned long long& won) -> int {   // block-wide: best key, OR of flags
    key = wave_umax64(key);
    flag = __any(flag) ? 1 : 0;
    __syncthreads();                                                  // (second use: the first round's reads are done)
    if (lane == 0) { wbest[wave] = key; wflag[wave] = flag; }
    __syncthreads();
    won = 0; int f = 0;
#pragma unroll
    for (int w = 0; w < TK_AM_THREADS / 64; ++w) { won = wbest[w] > won ? wbest[w] : won; f |= wflag[w]; }
    return f;
  };
  // ---- fast pass: packed keys, one compare per 16-byte vector, the winning vector re-read at the end ----
  uint32_t bk = 0, bv = 0, rawmax = 0;                                // best key, its vector; largest raw half (negative NaN: > 0xff80)
  auto take = [&](const uint4& a, int v) {
    const tk_u16x2 k0 = tk_fast_keys(a.x), k1 = tk_fast_keys(a.y), k2 = tk_fast_keys(a.z), k3 = tk_fast_keys(a.w);
    const tk_u16x2 km = __builtin_elementwise_max(__builtin_elementwise_max(k0, k1), __builtin_elementwise_max(k2, k3));
    union { uint32_t u; tk_u16x2 v; } r0, r1, r2, r3; r0.u = a.x; r1.u = a.y; r2.u = a.z; r3.u = a.w;
    const tk_u16x2 rm = __builtin_elementwise_max(__builtin_elementwise_max(r0.v, r1.v), __builtin_elementwise_max(r2.v, r3.v));
    const uint32_t vm = km.x > km.y ? km.x : km.y;
    const uint32_t rr = rm.x > rm.y ? rm.x : rm.y;
    rawmax = rr > rawmax ? rr : rawmax;
    if (vm > bk) { bk = vm; bv = (uint32_t)v; }
  };
  int v = t;
  for (; v + 3 * TK_AM_THREADS < nvec; v += 4 * TK_AM_THREADS) {      // four 16-byte loads in flight per thread
    const uint4 a0 = *reinterpret_cast<const uint4*>(xr + (int64_t)v * 8);
    const uint4 a1 = *reinterpret_cast<const uint4*>(xr + (int64_t)(v + TK_AM_THREADS) * 8);
    const uint4 a2 = *reinterpret_cast<const uint4*>(xr + (int64_t)(v + 2 * TK_AM_THREADS) * 8);
    const uint4 a3 = *reinterpret_cast<const uint4*>(xr + (int64_t)(v + 3 * TK_AM_THREADS) * 8);
    take(a0, v); take(a1, v + TK_AM_THREADS); take(a2, v + 2 * TK_AM_THREADS); take(a3, v + 3 * TK_AM_THREADS);
  }
  for (; v < nvec; v += TK_AM_THREADS) take(*reinterpret_cast<const uint4*>(xr + (int64_t)v * 8), v);
  uint32_t bi = 0;
  if (bk) {                                                           // first element of the winning vector that holds the key
    const uint4 a = *reinterpret_cast<const uint4*>(xr + (int64_t)bv * 8);
    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
    bi = 8;
#pragma unroll
    for (int i = 3; i >= 0; --i) {
      const tk_u16x2 kk = tk_fast_keys(w[i]);
      if (kk.y == bk) bi = 2 * i + 1;
      if (kk.x == bk) bi = 2 * i;
    }
    bi += bv * 8;
  }
  for (int i = nvec * 8 + t; i < V; i += TK_AM_THREADS) {              // the unaligned row / the last V % 8 logits
    union { uint32_t u; tk_u16x2 v; } c; c.u = xr[i];
    const uint32_t kk = tk_fast_keys(c.u).x;
    rawmax = c.u > rawmax ? c.u : rawmax;
    if (kk > bk) { bk = kk; bi = (uint32_t)i; }
  }
  // value first, then the lowest index: the same 64-bit order as the candidate records of the two-launch form
  unsigned long long key = bk ? (((unsigned long long)bk << 48) | (unsigned long long)(0xffffffffu - bi)) : 0ull;
  unsigned long long won;
  int special = reduce(key, (rawmax > 0xff80u || bk > 0xff80u) ? 1 : 0, won);
  const uint32_t wk = (uint32_t)(won >> 48);
  special |= (wk == 0x8000u || wk == 0x7fffu) ? 1 : 0;                // a row maximum of +-0: -0 == +0 must tie by index
  if (special) {
    // ---- exact pass (rows with NaN, or whose maximum is a zero): the monotone key of mono_bf16 on every element ----
    bk = 0; bi = 0;
    for (int i = t; i < V; i += TK_AM_THREADS) {
      const uint32_t kk = mono_bf16(xr[i]);
      if (kk > bk) { bk = kk; bi = (uint32_t)i; }
    }
    key = bk ? (((unsigned long long)bk << 48) | (unsigned long long)(0xffffffffu - bi)) : 0ull;
    reduce(key, 0, won);
  }
  if (t == 0) {
    const int64_t idx = won ? (int64_t)(0xffffffffu - (uint32_t)(won & 0xffffffffu)) : -1;
    const int64_t opos = out_group ? (int64_t)(row / out_group) * out_gstride + (int64_t)(row % out_group) : (int64_t)row;
    out_idx[opos] = idx;
    if (out_val) out_val[opos] = idx >= 0 ? xr[idx] : (bf16_t)0;
  }
}

static int g_topk_two_stage = 0;     // diagnostics (tests): 1 = always the two-launch form, 2 = the one-launch form wherever legal
static int g_tk_merge_narrow = 0;    // bit 2 of the same word: eagle_topk_merge_candidates always on the four-wave kernel
extern "C" void eagle_debug_set_topk(int two_stage) { g_topk_two_stage = two_stage & 3; g_tk_merge_narrow = (two_stage >> 2) & 1; }

extern "C" int64_t eagle_topk_workspace_bytes(int rows, int V, int k) {
  const int64_t nwc = (V + TK_WCHUNK - 1) / TK_WCHUNK;
  return (int64_t)rows * nwc * (k + 1) * 8;     // k candidates + one (max, sum-exp) partial per wave chunk
}

static int topk_launch(const void* logits, int64_t ld, int dtype, int rows, int V, int k, int64_t* out_idx, void* out_val,
                       void* workspace, const int64_t* block_index, int64_t block_rows, float* lse_out, hipStream_t stream,
                       int out_group = 0, int64_t out_gstride = 0) {
  if (!logits || !out_idx || !workspace) return EAGLE_ERR_ARG;
  if (block_index && block_rows < rows) return EAGLE_ERR_SHAPE;
  if (rows < 1 || V < 1 || k < 1 || k > TK_MAXK || ld < V) return EAGLE_ERR_SHAPE;
  const int nwc = (V + TK_WCHUNK - 1) / TK_WCHUNK;
  if ((int64_t)nwc * k > TK_THREADS * 16) return EAGLE_ERR_SHAPE;     // stage 2 holds 8 (<= 2048 candidates) or 16 per thread
  const bool big = (int64_t)nwc * k > TK_THREADS * TK_CM;             // > 2048 candidates: V > 131072 at k > 13 (k = 16 over 151936)
  unsigned long long* cand = (unsigned long long*)workspace;
  float2* part = lse_out ? reinterpret_cast<float2*>(cand + (int64_t)rows * nwc * k) : nullptr;
  dim3 g1((nwc + 3) / 4, rows);
  // one launch where it is the faster form (measured in the batch-1 draft cycle, V = 32000: k = 1 7.6 us against 4.5 + 4.5;
  // k = 10 14.3 us against 5.8 + 6.3 -- ten serial selection rounds on ONE workgroup cost more than the launch they save)
  if (dtype == 0 && !lse_out && k == 1 && nwc > TK_FUSED_MAXWC && rows >= 256 && g_topk_two_stage != 1) {
    // many long rows: a workgroup per row fills the chip by itself (below ~256 rows the two-launch form spreads a row wider)
    hipLaunchKernelGGL(argmax_rows_bf16_kernel, dim3(rows), dim3(TK_AM_THREADS), 0, stream, (const bf16_t*)logits, ld, V, out_idx,
                       (bf16_t*)out_val, block_index, block_rows, out_group, out_gstride);
  } else if (dtype == 0 && !lse_out && nwc <= TK_FUSED_MAXWC && (k <= 2 || g_topk_two_stage == 2) && g_topk_two_stage != 1) {
#define TK_FUSED_LAUNCH(CPW) hipLaunchKernelGGL(topk_fused_bf16_kernel<CPW>, dim3(rows), dim3(TK_FUSED_WAVES * 64), 0, stream, \
    (const bf16_t*)logits, ld, V, k, nwc, out_idx, (bf16_t*)out_val, block_index, block_rows, out_group, out_gstride)
    if (nwc <= TK_FUSED_WAVES) TK_FUSED_LAUNCH(1);
    else if (nwc <= 2 * TK_FUSED_WAVES) TK_FUSED_LAUNCH(2);
    else TK_FUSED_LAUNCH(4);
#undef TK_FUSED_LAUNCH
  } else if (dtype == 0 && lse_out) {
    hipLaunchKernelGGL(topk_stage1_bf16_kernel<true>, g1, dim3(TK_THREADS), 0, stream, (const bf16_t*)logits, ld, V, k, nwc, cand,
                       block_index, block_rows, part);
    if (big)
      hipLaunchKernelGGL((topk_stage2_kernel<bf16_t, true, false, 16>), dim3(rows), dim3(TK_THREADS), 0, stream, cand, nwc * k, k,
                         (const bf16_t*)logits, ld, out_idx, (bf16_t*)out_val, block_index, block_rows, part, nwc, lse_out, out_group, out_gstride);
    else
    hipLaunchKernelGGL((topk_stage2_kernel<bf16_t, true>), dim3(rows), dim3(TK_THREADS), 0, stream, cand, nwc * k, k,
                       (const bf16_t*)logits, ld, out_idx, (bf16_t*)out_val, block_index, block_rows, part, nwc, lse_out, out_group, out_gstride);
  } else if (dtype == 0) {
    hipLaunchKernelGGL(topk_stage1_bf16_kernel<false>, g1, dim3(TK_THREADS), 0, stream, (const bf16_t*)logits, ld, V, k, nwc, cand,
                       block_index, block_rows, part);
    if (V <= 65536)
      hipLaunchKernelGGL((topk_stage2_kernel<bf16_t, false, true>), dim3(rows), dim3(TK_THREADS), 0, stream, cand, nwc * k, k,
                         (const bf16_t*)logits, ld, out_idx, (bf16_t*)out_val, block_index, block_rows, part, nwc, lse_out, out_group, out_gstride);
    else if (big)
      hipLaunchKernelGGL((topk_stage2_kernel<bf16_t, false, false, 16>), dim3(rows), dim3(TK_THREADS), 0, stream, cand, nwc * k, k,
                         (const bf16_t*)logits, ld, out_idx, (bf16_t*)out_val, block_index, block_rows, part, nwc, lse_out, out_group, out_gstride);
    else
      hipLaunchKernelGGL((topk_stage2_kernel<bf16_t, false>), dim3(rows), dim3(TK_THREADS), 0, stream, cand, nwc * k, k,
                         (const bf16_t*)logits, ld, out_idx, (bf16_t*)out_val, block_index, block_rows, part, nwc, lse_out, out_group, out_gstride);
  } else if (dtype == 1 && !lse_out) {
    hipLaunchKernelGGL(topk_stage1_f32_kernel, g1, dim3(TK_THREADS), 0, stream, (const float*)logits, ld, V, k, nwc, cand, block_index, block_rows);
    if (big)
      hipLaunchKernelGGL((topk_stage2_kernel<float, false, false, 16>), dim3(rows), dim3(TK_THREADS), 0, stream, cand, nwc * k, k,
                         (const float*)logits, ld, out_idx, (float*)out_val, block_index, block_rows, (const float2*)nullptr, nwc,
                         (float*)nullptr, out_group, out_gstride);
    else
    hipLaunchKernelGGL((topk_stage2_kernel<float, false>), dim3(rows), dim3(TK_THREADS), 0, stream, cand, nwc * k, k,
                       (const float*)logits, ld, out_idx, (float*)out_val, block_index, block_rows, (const float2*)nullptr, nwc,
                       (float*)nullptr, out_group, out_gstride);
  } else {
    return EAGLE_ERR_ARG;
  }
  return eagle_check_launch();
}

// Stage 2 alone: rows x ncand candidate records (value key << 48 | 0xffffffff - index; 0 = empty) -> the k best of every row,
// ids and bf16 values (rebuilt from the keys).  ncand <= 6144.  The records come from eagle_lm_head_topk_rows_bf16's GEMM
// epilogue (or from any producer that keeps the record order: value first, then the lowest index).
// index_bound: every candidate index is below it (0 = unknown); below 65536 the 16-wave form merges 32-bit keys
int eagle_topk_merge_candidates_bounded(const void* cand, int ncand, int rows, int k, int64_t* out_idx, void* out_val,
                                        int group, int64_t group_stride, int index_bound, hipStream_t stream) {
  if (!cand || !out_idx) return EAGLE_ERR_ARG;
  if (rows < 1 || ncand < 1 || ncand > TK_THREADS * 24 || k < 1 || k > TK_MAXK) return EAGLE_ERR_SHAPE;
  if (group < 0 || (group > 0 && (rows % group || group_stride < (int64_t)group * k))) return EAGLE_ERR_SHAPE;
#define TK_MERGE(CMV) hipLaunchKernelGGL((topk_stage2_kernel<bf16_t, false, false, CMV>), dim3(rows), dim3(TK_THREADS), 0, stream,      \
    (const unsigned long long*)cand, ncand, k, (const bf16_t*)nullptr, (int64_t)0, out_idx, (bf16_t*)out_val, (const int64_t*)nullptr, \
    (int64_t)0, (const float2*)nullptr, 0, (float*)nullptr, group, group_stride)
  // few rows of many records (one request's tree step): the 16-wave form; many rows (a request batch) keep four waves per row
#define TK_MERGE_WIDE(CWV) do {                                                                                                  \
    if (index_bound > 0 && index_bound <= 65536)                                                                                 \
      hipLaunchKernelGGL((topk_merge_wide_kernel<CWV, true>), dim3(rows), dim3(TK_FUSED_WAVES * 64), 0, stream,                  \
                         (const unsigned long long*)cand, ncand, k, out_idx, (bf16_t*)out_val, group, group_stride);             \
    else                                                                                                                         \
      hipLaunchKernelGGL((topk_merge_wide_kernel<CWV, false>), dim3(rows), dim3(TK_FUSED_WAVES * 64), 0, stream,                 \
                         (const unsigned long long*)cand, ncand, k, out_idx, (bf16_t*)out_val, group, group_stride);             \
  } while (0)
  if (rows <= 16 && ncand > TK_THREADS * 4 && !g_tk_merge_narrow) {
    if (ncand <= TK_FUSED_WAVES * 64 * 3) TK_MERGE_WIDE(3);
    else TK_MERGE_WIDE(6);
  }
  else if (ncand <= TK_THREADS * 8) TK_MERGE(8);
  else if (ncand <= TK_THREADS * 16) TK_MERGE(16);
  else TK_MERGE(24);
#undef TK_MERGE
#undef TK_MERGE_WIDE
  return eagle_check_launch();
}

extern "C" int eagle_topk_merge_candidates(const void* cand, int ncand, int rows, int k, int64_t* out_idx, void* out_val,
                                           int group, int64_t group_stride, hipStream_t stream) {
  return eagle_topk_merge_candidates_bounded(cand, ncand, rows, k, out_idx, out_val, group, group_stride, 0, stream);
}

// dtype: 0 = bf16 logits, 1 = fp32 logits
extern "C" int eagle_topk_vocab(const void* logits, int64_t ld, int dtype, int rows, int V, int k,
                                int64_t* out_idx, void* out_val, void* workspace, const int64_t* block_index,
                                int64_t block_rows, hipStream_t stream) {
  return topk_launch(logits, ld, dtype, rows, V, k, out_idx, out_val, workspace, block_index, block_rows, nullptr, stream);
}

// eagle_topk_vocab writing into a batched table: logits row r = b*group + j goes to out + b*group_stride + j*k (elements)
extern "C" int eagle_topk_vocab_grouped(const void* logits, int64_t ld, int dtype, int rows, int V, int k, int64_t* out_idx,
                                        void* out_val, void* workspace, int group, int64_t group_stride, hipStream_t stream) {
  if (group < 1 || rows % group || group_stride < (int64_t)group * k) return EAGLE_ERR_SHAPE;
  return topk_launch(logits, ld, dtype, rows, V, k, out_idx, out_val, workspace, nullptr, 0, nullptr, stream, group, group_stride);
}

// bf16 logits: the top-k of eagle_topk_vocab plus lse_out[rows] = fp32 log-sum-exp of every row, from the same pass
extern "C" int eagle_topk_lse_vocab(const void* logits, int64_t ld, int rows, int V, int k, int64_t* out_idx, void* out_val,
                                    float* lse_out, void* workspace, hipStream_t stream) {
  if (!lse_out) return EAGLE_ERR_ARG;
  return topk_launch(logits, ld, 0, rows, V, k, out_idx, out_val, workspace, nullptr, 0, lse_out, stream);
}

// ---- the entry point that allocates nothing (no counterpart in the reference) ----
// For callers that cannot call hipMalloc or hand in a workspace (code under HIP graph capture, tight serving loops,
// harnesses that forbid allocation): the stage-1 candidate buffer is a file-scope device array.  Every GPU the library
// runs on holds its own 2 MiB copy until the process ends.  The array's NAME is a host-side stand-in, not a device
// address: host code only ever uses the per-device address hipGetSymbolAddress returned.
#define TK_SCRATCH_BYTES (2 << 20)
#define TK_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[TK_SCRATCH_BYTES];
static void* g_tk_scratch_ptr[TK_MAX_DEVICES];          // per device: each GPU has the array at its own address

// Same contract as eagle_topk_vocab (bf16 / fp32 logits, optional ring cursor) without the workspace argument.
// Limit: eagle_topk_workspace_bytes(rows, V, k) <= 2 MiB (e.g. 159 rows of a 151936-entry vocabulary at k = 10),
// larger problems return EAGLE_ERR_SHAPE.  All calls on one GPU share that GPU's array: calls that overlap on one
// GPU (two streams, two threads) race.  Nothing in the array has to start at zero: stage 1 writes every candidate slot
// stage 2 reads, so a call never sees the previous call's contents.  The first call on a device resolves the array's
// device address (hipGetSymbolAddress: not capturable -- make one call per device before capturing a graph).
extern "C" int eagle_topk_vocab_noalloc(const void* logits, int64_t ld, int dtype, int rows, int V, int k, int64_t* out_idx,
                                        void* out_val, const int64_t* block_index, int64_t block_rows, hipStream_t stream) {
  if (rows < 1 || V < 1 || k < 1 || k > TK_MAXK) return EAGLE_ERR_SHAPE;
  if (eagle_topk_workspace_bytes(rows, V, k) > (int64_t)TK_SCRATCH_BYTES) return EAGLE_ERR_SHAPE;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TK_MAX_DEVICES) return EAGLE_ERR_ARG;
  void* p = __atomic_load_n(&g_tk_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (!p) {
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return EAGLE_ERR_LAUNCH;
    __atomic_store_n(&g_tk_scratch_ptr[dev], p, __ATOMIC_RELEASE);   // same value whichever thread stores it
  }
  return topk_launch(logits, ld, dtype, rows, V, k, out_idx, out_val, p, block_index, block_rows, nullptr, stream);
}
