// Weight-streaming "skinny" GEMM for the decode side of the draft head (gfx950).
//
//   out[M,N] = epilogue( x[M,K] . W[N,K]^T )        M <= 64 per pass, bf16 in, fp32 accumulate
//
// Replaces the nn.Linear calls of eagle/llama2.py at draft time (fc :542, q/k/v/o_proj
// :222-224/:280, gate/up/down_proj :317) and the frozen lm_head (eagle/train.py:215-228) when
// the number of rows is a handful (tree steps: 1..8 rows).  At that size the op is HBM-bound on
// the weight bytes, so the design is a pure weight stream:
//   * W is the MFMA A operand (16 output features per tile), x^T the B operand (16 tokens):
//     v_mfma_f32_16x16x32_bf16.  Weights go HBM -> VGPR directly (no LDS round trip), each lane
//     16 B per load, 64 B contiguous per row per wave-instruction, 4 instructions deep per
//     128-wide K super-step.
//   * a block = 8 waves that split K between them (in-block split-K, reduced through LDS), so
//     the epilogue (bias, residual add, SiLU*up) runs in the producing kernel.
//   * the activation row can be the concatenation of two gathered sources:
//     fc(cat(embed_tokens[ids], hidden[parent_row]))  (eagle/llama2.py:515, :542).
#include "common.h"

#define SK_WAVES 8
#define EPI_LINEAR 0     // (+bias) (+residual)
#define EPI_SILU_MUL 1   // out = silu(x.W^T) * (x.W2^T)     (LlamaMLP, eagle/llama2.py:317)

struct SkinnyArgs {
  const bf16_t* x0; const int64_t* idx0; int64_t ld0; int K0;
  const int64_t* ind0; int ind_n; int64_t idx0_bstride;  // idx0 row = idx0[(m/ind_n)*bstride + ind0[m%ind_n]]
  float norm_eps;                                        // > 0: scale row m by rsqrt(mean(x[m]^2) + eps)
  const bf16_t* x1; const int64_t* idx1; int64_t ld1;
  const bf16_t* W; const bf16_t* W2;
  const bf16_t* bias; const bf16_t* res; int64_t ldr;
  bf16_t* out; int64_t ldo;
  int M, N, K;
};

template <int RT, int MT, int EPI>
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_skinny_kernel(SkinnyArgs p) {
  constexpr int NT = RT * MT;
  __shared__ __attribute__((aligned(16))) float red[SK_WAVES][NT][64][4];
  __shared__ float red_ss[SK_WAVES][MT][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int rows_per_block = (EPI == EPI_SILU_MUL) ? 16 : RT * 16;
  const int n0 = blockIdx.x * rows_per_block;

  const bf16_t* wrow[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    int n = n0 + ((EPI == EPI_SILU_MUL) ? 0 : t * 16) + r16;
    n = n < p.N ? n : p.N - 1;  // clamp: rows past N are computed on valid memory and dropped
    const bf16_t* base = (EPI == EPI_SILU_MUL && t == 1) ? p.W2 : p.W;
    wrow[t] = base + (int64_t)n * p.K + 8 * g;
  }
  const bf16_t* xa[MT];
  const bf16_t* xb[MT];
  bool mvalid[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int m = mt * 16 + r16;
    mvalid[mt] = m < p.M;
    int64_t ra = 0, rb = 0;
    if (mvalid[mt]) {
      if (p.ind0) ra = p.idx0[(int64_t)(m / p.ind_n) * p.idx0_bstride + p.ind0[m % p.ind_n]];
      else ra = p.idx0 ? p.idx0[m] : m;
      rb = p.idx1 ? p.idx1[m] : m;
    }
    xa[mt] = p.x0 + ra * p.ld0 + 8 * g;
    xb[mt] = p.x1 ? (p.x1 + rb * p.ld1 + 8 * g - p.K0) : xa[mt];
  }

  f32x4 acc[RT][MT];
  float ssq[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) ssq[mt] = 0.f;
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_norm = p.norm_eps > 0.f;

  const int nfull = p.K >> 7;
  for (int ss = wave; ss < nfull; ss += SK_WAVES) {
    const int kb = ss << 7;
    bf16x8 a[4][RT], b[4][MT];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = kb + s * 32;
#pragma unroll
      for (int t = 0; t < RT; ++t) a[s][t] = *reinterpret_cast<const bf16x8*>(wrow[t] + k);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bf16_t* src = (k < p.K0) ? xa[mt] : xb[mt];
        bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        b[s][mt] = mvalid[mt] ? *reinterpret_cast<const bf16x8*>(src + k) : z;
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][t], b[s][mt], acc[t][mt], 0, 0, 0);
    if (do_norm) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float xv = (float)b[s][mt][j]; ssq[mt] = fmaf(xv, xv, ssq[mt]); }
    }
  }
  // tail: K % 128 in {32, 64, 96}, taken by the wave whose turn it would be
  const int tail_steps = (p.K & 127) >> 5;
  if (tail_steps && wave == (nfull % SK_WAVES)) {
    for (int s = 0; s < tail_steps; ++s) {
      const int k = (nfull << 7) + s * 32;
      bf16x8 a[RT], b[MT];
#pragma unroll
      for (int t = 0; t < RT; ++t) a[t] = *reinterpret_cast<const bf16x8*>(wrow[t] + k);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bf16_t* src = (k < p.K0) ? xa[mt] : xb[mt];
        bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        b[mt] = mvalid[mt] ? *reinterpret_cast<const bf16x8*>(src + k) : z;
      }
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], b[mt], acc[t][mt], 0, 0, 0);
      if (do_norm) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float xv = (float)b[mt][j]; ssq[mt] = fmaf(xv, xv, ssq[mt]); }
      }
    }
  }

  // in-block split-K reduction
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      *reinterpret_cast<f32x4*>(&red[wave][t * MT + mt][lane][0]) = acc[t][mt];
  if (do_norm) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float v = ssq[mt];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (g == 0) red_ss[wave][mt][r16] = v;
    }
  }
  __syncthreads();
  if (wave != 0) return;
  float rstd[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    rstd[mt] = 1.f;
    if (do_norm) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < SK_WAVES; ++w) v += red_ss[w][mt][r16];
      rstd[mt] = rsqrtf(v / (float)p.K + p.norm_eps);
    }
  }
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < SK_WAVES; ++w) s += *reinterpret_cast<const f32x4*>(&red[w][t * MT + mt][lane][0]);
      acc[t][mt] = s * rstd[mt];
    }

  // epilogue: D layout of mfma 16x16: col = lane&15 (token), row = 4*(lane>>4)+r (feature)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 16 + r16;
    if (m >= p.M) continue;
    if (EPI == EPI_SILU_MUL) {
      const int nb = n0 + 4 * g;
      float y[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float gt = rbf(acc[0][mt][r]), up = rbf(acc[1][mt][r]);
        float s = rbf(gt / (1.0f + __expf(-gt)));
        y[r] = s * up;
      }
      bf16_t* o = p.out + (int64_t)m * p.ldo + nb;
      if (nb + 3 < p.N && ((p.ldo | nb) & 3) == 0) {
        uint2 v = {pack2bf(y[0], y[1]), pack2bf(y[2], y[3])};
        *reinterpret_cast<uint2*>(o) = v;
      } else {
        for (int r = 0; r < 4; ++r) if (nb + r < p.N) o[r] = f2bf(y[r]);
      }
    } else {
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const int nb = n0 + t * 16 + 4 * g;
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = nb + r;
          float v = acc[t][mt][r];
          if (p.bias && n < p.N) v += bf2f(p.bias[n]);
          v = rbf(v);
          if (p.res && n < p.N) v = v + bf2f(p.res[(int64_t)m * p.ldr + n]);
          y[r] = v;
        }
        bf16_t* o = p.out + (int64_t)m * p.ldo + nb;
        if (nb + 3 < p.N && ((p.ldo | nb) & 3) == 0) {
          uint2 v = {pack2bf(y[0], y[1]), pack2bf(y[2], y[3])};
          *reinterpret_cast<uint2*>(o) = v;
        } else {
          for (int r = 0; r < 4; ++r) if (nb + r < p.N) o[r] = f2bf(y[r]);
        }
      }
    }
  }
}

template <int RT, int MT, int EPI>
static int launch_skinny(const SkinnyArgs& a, hipStream_t st) {
  const int rows_per_block = (EPI == EPI_SILU_MUL) ? 16 : RT * 16;
  dim3 grid((a.N + rows_per_block - 1) / rows_per_block), block(SK_WAVES * 64);
  hipLaunchKernelGGL((gemm_skinny_kernel<RT, MT, EPI>), grid, block, 0, st, a);
  return eagle_check_launch();
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// See include/eagle_hip.h for the contract.
extern "C" int eagle_gemm_skinny_bf16(const void* x0, const int64_t* idx0, int64_t ld0, int K0,
                                      const int64_t* ind0, int ind_n, int64_t idx0_bstride, float norm_eps,
                                      const void* x1, const int64_t* idx1, int64_t ld1,
                                      const void* W, const void* W2, const void* bias,
                                      const void* res, int64_t ldr, void* out, int64_t ldo,
                                      int M, int N, int K, int epi, hipStream_t stream) {
  if (!x0 || !W || !out) return EAGLE_ERR_ARG;
  if (epi != EPI_LINEAR && epi != EPI_SILU_MUL) return EAGLE_ERR_ARG;
  if (ind0 && (!idx0 || ind_n < 1 || M > (epi == EPI_SILU_MUL ? 32 : 64))) return EAGLE_ERR_ARG;
  if (norm_eps < 0.f || (norm_eps > 0.f && x1)) return EAGLE_ERR_ARG;
  if (epi == EPI_SILU_MUL && (!W2 || bias || res)) return EAGLE_ERR_ARG;
  if (M < 1 || N < 1 || K < 32 || (K & 31)) return EAGLE_ERR_SHAPE;
  if (x1) { if (K0 <= 0 || K0 >= K || (K0 & 31) || (ld1 & 7) || !al16(x1)) return EAGLE_ERR_SHAPE; }
  else K0 = K;
  if ((ld0 & 7) || !al16(x0) || !al16(W) || (W2 && !al16(W2))) return EAGLE_ERR_SHAPE;
  if (ld0 < K0 && !idx0 && M > 1) return EAGLE_ERR_SHAPE;
  if (ldo < N || (res && ldr < N)) return EAGLE_ERR_SHAPE;

  const int chunk = (epi == EPI_SILU_MUL) ? 32 : 64;  // rows per weight pass
  for (int m0 = 0; m0 < M; m0 += chunk) {
    SkinnyArgs a;
    const int mc = (M - m0) < chunk ? (M - m0) : chunk;
    a.x0 = (const bf16_t*)x0 + (idx0 ? 0 : (int64_t)m0 * ld0); a.idx0 = idx0 ? idx0 + m0 : nullptr;
    a.ld0 = ld0; a.K0 = K0;
    a.ind0 = ind0; a.ind_n = ind_n; a.idx0_bstride = idx0_bstride; a.norm_eps = norm_eps;
    a.x1 = x1 ? (const bf16_t*)x1 + (idx1 ? 0 : (int64_t)m0 * ld1) : nullptr; a.idx1 = idx1 ? idx1 + m0 : nullptr;
    a.ld1 = ld1;
    a.W = (const bf16_t*)W; a.W2 = (const bf16_t*)W2; a.bias = (const bf16_t*)bias;
    a.res = res ? (const bf16_t*)res + (int64_t)m0 * ldr : nullptr; a.ldr = ldr;
    a.out = (bf16_t*)out + (int64_t)m0 * ldo; a.ldo = ldo;
    a.M = mc; a.N = N; a.K = K;
    int rc;
    if (epi == EPI_SILU_MUL) {
      if (mc <= 16) rc = launch_skinny<2, 1, EPI_SILU_MUL>(a, stream);
      else rc = launch_skinny<2, 2, EPI_SILU_MUL>(a, stream);
    } else if (mc <= 16) {
      if (N >= 16384) rc = launch_skinny<2, 1, EPI_LINEAR>(a, stream);
      else rc = launch_skinny<1, 1, EPI_LINEAR>(a, stream);
    } else if (mc <= 32) {
      rc = launch_skinny<1, 2, EPI_LINEAR>(a, stream);
    } else {
      rc = launch_skinny<1, 4, EPI_LINEAR>(a, stream);
    }
    if (rc) return rc;
  }
  return EAGLE_OK;
}
