// Per-row top-k over the vocabulary with a stated, deterministic tie-break (gfx950).
//
// Replaces torch.topk(last_headout, top_k) of the EAGLE-1 expansion loop (top_k = 10,
// eagle/llama2.py:11) and the first-index argmax of the agreement count
// (torch.max(..., 2), eagle/train.py:115-116) / greedy verification (row a22) with k = 1.
// Order: value descending, then index ascending (== a stable descending sort); -0.0 == +0.0;
// NaN sorts above +inf (torch.sort semantics).  Results are bit-exact against the oracle.
//
// Two stages, both wavefront-level selects over 64-bit keys
//     key = (monotone(value) << 32) | (0xFFFFFFFF - index)
// so that one unsigned max gives "largest value, lowest index":
//   stage 1: grid (chunks of 4096 logits) x rows; each lane holds 16 keys in registers, k
//            rounds of {wave max via DPP shuffles -> block max via LDS -> owner removes};
//   stage 2: one wave per row merges chunks x k candidates the same way.
#include "common.h"

#define TK_THREADS 256
#define TK_CHUNK 4096
#define TK_PER_THREAD (TK_CHUNK / TK_THREADS)  // 16
#define TK_MAXK 16
#define TK_CMAX 16  // stage-2 candidates per lane -> chunks*k <= 1024

__device__ __forceinline__ uint32_t mono_bf16(bf16_t b) {
  if ((b & 0x7fff) > 0x7f80) return 0xffff0000u;   // NaN: largest
  if (b == 0x8000) b = 0;                          // -0.0 == +0.0
  uint16_t m = (b & 0x8000) ? (uint16_t)~b : (uint16_t)(b | 0x8000);
  return ((uint32_t)m) << 16;
}
__device__ __forceinline__ uint32_t mono_f32(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
  if (u == 0x80000000u) u = 0;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long x = __shfl_xor(v, o, 64);
    v = x > v ? x : v;
  }
  return v;
}

template <typename T>
__global__ __launch_bounds__(TK_THREADS) void topk_stage1_kernel(const T* __restrict__ x, int64_t ld, int V, int k,
                                                                unsigned long long* __restrict__ cand) {
  __shared__ unsigned long long sh[4];
  const int row = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const T* xr = x + (int64_t)row * ld;
  unsigned long long key[TK_PER_THREAD];
  const int base = chunk * TK_CHUNK + t * TK_PER_THREAD;
#pragma unroll
  for (int i = 0; i < TK_PER_THREAD; ++i) {
    const int idx = base + i;
    unsigned long long kk = 0;
    if (idx < V) {
      uint32_t m;
      if (sizeof(T) == 2) m = mono_bf16(((const bf16_t*)xr)[idx]); else m = mono_f32(((const float*)xr)[idx]);
      kk = ((unsigned long long)m << 32) | (unsigned long long)(0xffffffffu - (uint32_t)idx);
    }
    key[i] = kk;
  }
  unsigned long long best = 0;
#pragma unroll
  for (int i = 0; i < TK_PER_THREAD; ++i) best = key[i] > best ? key[i] : best;
  for (int r = 0; r < k; ++r) {
    unsigned long long wm = wave_max_u64(best);
    if (lane == 0) sh[wave] = wm;
    __syncthreads();
    unsigned long long bm = sh[0];
    bm = sh[1] > bm ? sh[1] : bm;
    bm = sh[2] > bm ? sh[2] : bm;
    bm = sh[3] > bm ? sh[3] : bm;
    __syncthreads();
    if (t == 0) cand[((int64_t)row * nchunks + chunk) * k + r] = bm;
    if (bm != 0 && best == bm) {  // keys are unique: exactly one owner
      best = 0;
#pragma unroll
      for (int i = 0; i < TK_PER_THREAD; ++i) {
        if (key[i] == bm) key[i] = 0;
        best = key[i] > best ? key[i] : best;
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(64) void topk_stage2_kernel(const unsigned long long* __restrict__ cand, int ncand, int k,
                                                        const T* __restrict__ x, int64_t ld,
                                                        int64_t* __restrict__ out_idx, T* __restrict__ out_val) {
  const int row = blockIdx.x, lane = threadIdx.x;
  unsigned long long key[TK_CMAX];
#pragma unroll
  for (int i = 0; i < TK_CMAX; ++i) {
    const int c = lane + i * 64;
    key[i] = c < ncand ? cand[(int64_t)row * ncand + c] : 0ull;
  }
  unsigned long long best = 0;
#pragma unroll
  for (int i = 0; i < TK_CMAX; ++i) best = key[i] > best ? key[i] : best;
  for (int r = 0; r < k; ++r) {
    const unsigned long long bm = wave_max_u64(best);
    if (lane == 0) {
      // fewer than k finite entries can only happen when V < k; emit index -1 then
      const int64_t idx = bm ? (int64_t)(0xffffffffu - (uint32_t)(bm & 0xffffffffu)) : -1;
      out_idx[(int64_t)row * k + r] = idx;
      if (out_val) out_val[(int64_t)row * k + r] = idx >= 0 ? x[(int64_t)row * ld + idx] : (T)0;
    }
    if (bm != 0 && best == bm) {
      best = 0;
#pragma unroll
      for (int i = 0; i < TK_CMAX; ++i) {
        if (key[i] == bm) key[i] = 0;
        best = key[i] > best ? key[i] : best;
      }
    }
  }
}

extern "C" int64_t eagle_topk_workspace_bytes(int rows, int V, int k) {
  const int64_t nchunks = (V + TK_CHUNK - 1) / TK_CHUNK;
  return (int64_t)rows * nchunks * k * 8;
}

// dtype: 0 = bf16 logits, 1 = fp32 logits
extern "C" int eagle_topk_vocab(const void* logits, int64_t ld, int dtype, int rows, int V, int k,
                                int64_t* out_idx, void* out_val, void* workspace, hipStream_t stream) {
  if (!logits || !out_idx || !workspace) return EAGLE_ERR_ARG;
  if (rows < 1 || V < 1 || k < 1 || k > TK_MAXK || ld < V) return EAGLE_ERR_SHAPE;
  const int nchunks = (V + TK_CHUNK - 1) / TK_CHUNK;
  if (nchunks * k > 64 * TK_CMAX) return EAGLE_ERR_SHAPE;
  unsigned long long* cand = (unsigned long long*)workspace;
  dim3 g1(nchunks, rows);
  if (dtype == 0) {
    hipLaunchKernelGGL(topk_stage1_kernel<bf16_t>, g1, dim3(TK_THREADS), 0, stream, (const bf16_t*)logits, ld, V, k, cand);
    hipLaunchKernelGGL(topk_stage2_kernel<bf16_t>, dim3(rows), dim3(64), 0, stream, cand, nchunks * k, k,
                       (const bf16_t*)logits, ld, out_idx, (bf16_t*)out_val);
  } else if (dtype == 1) {
    hipLaunchKernelGGL(topk_stage1_kernel<float>, g1, dim3(TK_THREADS), 0, stream, (const float*)logits, ld, V, k, cand);
    hipLaunchKernelGGL(topk_stage2_kernel<float>, dim3(rows), dim3(64), 0, stream, cand, nchunks * k, k,
                       (const float*)logits, ld, out_idx, (float*)out_val);
  } else {
    return EAGLE_ERR_ARG;
  }
  return eagle_check_launch();
}
