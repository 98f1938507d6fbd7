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
//   * PACKED weights (eagle_pack_weight_bf16, round 3): the fragment shape above reads 16 rows x 64 B per wave-instruction,
//     i.e. HALF cache lines whose other half follows with the next instruction.  Measured on MI355X (tools/stream_probe.hip,
//     262 MB from cold caches, non-temporal): that shape streams at 5.4 TB/s, wave-instructions that read 1 KiB contiguous
//     at 6.3 TB/s (+17 %).  A packed copy of W stores every 16-row x 32-k fragment as ONE contiguous 1 KiB block in lane order
//     ([N/16][K/32][64 lanes][8 bf16]), so the same registers are filled by whole-line reads; the arithmetic (and therefore
//     every output bit) is unchanged.
#include "common.h"
#include "topk_select.h"
#include <type_traits>

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
  bool nt;                                               // host-side only: stream W with non-temporal loads
  int rot;                                               // keepx: neighbouring blocks walk their K super-steps from staggered starts
  int stage;                                             // keepx: collect a block's outputs in LDS and write them when its stream is done
  int packed;                                            // host-side only: W / W2 are eagle_pack_weight_bf16 images
  unsigned long long* cand; int topk;                    // keepx<..., TKP>: [M][blocks][topk] candidate records instead of `out`
};

// Packed image: tile T = rows [16T, 16T+16), k-block f = columns [32f, 32f+32): 512 elements at (T * K/32 + f) * 512, lane
// l = r16 + 16 g holds W[16T + r16][32f + 8g .. +8].  Rows past N are zero.  Address of lane `lane`'s piece of (T, k):
__device__ __forceinline__ const bf16_t* sk_pk(const bf16_t* base, int T, int K, int k, int lane) {
  return base + ((int64_t)T * (K >> 5) + (k >> 5)) * 512 + lane * 8;
}

template <int RT, int MT, int EPI, bool PK = false>
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
    if (PK) {                                           // whole tiles: the image is zero-padded to 16 rows
      const int T = min((n0 + ((EPI == EPI_SILU_MUL) ? 0 : t * 16)) >> 4, (p.N - 1) >> 4);
      wrow[t] = sk_pk(base, T, p.K, 0, lane);
    } else {
      wrow[t] = base + (int64_t)n * p.K + 8 * g;
    }
  }
  constexpr int KS = PK ? 16 : 1;                       // element step per k: a packed k-block of 32 columns is 512 elements on
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
      for (int t = 0; t < RT; ++t) a[s][t] = *reinterpret_cast<const bf16x8*>(wrow[t] + k * KS);
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
      for (int t = 0; t < RT; ++t) a[t] = *reinterpret_cast<const bf16x8*>(wrow[t] + k * KS);
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


// ---------------------------------------------------------------------------------------------
// Streaming variant for M <= 16 (every tree step): same math, different schedule.
//   * balanced row partition: block b owns rows [b*R, b*R + R) with R = ceil(N / grid); tiles that
//     stick out of the range are masked per lane (no load is issued for a masked row), so every
//     block streams the same number of weight bytes whatever N is (11008 = 256 * 43);
//   * software pipeline: the loads of K super-step j+1 (and j+2) are in flight while super-step j
//     is multiplied, across tile-step boundaries and across the reduction barrier (raw s_barrier:
//     it drains LDS traffic only, the weight loads stay outstanding);
//   * the first weight loads are issued BEFORE the activation row indices are resolved (those are
//     dependent global loads: ids -> row -> x);
//   * weights are read once: non-temporal loads keep them from evicting x and the partial sums.
// RT = 16-row weight tiles per tile-step (EPI_SILU_MUL: RT/2 gate tiles + RT/2 up tiles).
// ---------------------------------------------------------------------------------------------
template <int RT> struct SkStage { bf16x8 a[4][RT]; bf16x8 b[4]; };

template <int RT, int EPI, bool NT, bool PK = false>
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_skinny_stream_kernel(SkinnyArgs p, int rows_per_block, int n_iter) {
  constexpr int RTW = (EPI == EPI_SILU_MUL) ? RT / 2 : RT;   // distinct 16-row groups per tile-step
  constexpr int RPS = RTW * 16;                               // output rows per tile-step
  __shared__ __attribute__((aligned(16))) float red[2][SK_WAVES][RT][64][4];
  __shared__ float red_ss[2][SK_WAVES][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int row_lo = blockIdx.x * rows_per_block;
  const int row_hi = min(p.N, row_lo + rows_per_block);
  if (row_lo >= row_hi) return;
  const int n_ts = (row_hi - row_lo + RPS - 1) / RPS;
  // K % 128 == 0 (launcher): wave w owns K super-steps w, w+8, ...; waves may differ by one step
  n_iter = ((p.K >> 7) - wave + SK_WAVES - 1) / SK_WAVES;
  const int J = n_ts * n_iter;
  const bool do_norm = p.norm_eps > 0.f;
  const bf16_t* xa = p.x0;   // resolved after the first weight loads are in flight
  const bf16_t* xb = p.x0;
  const bool mvalid = r16 < p.M;

  // No load below is predicated (a "load or zero" select makes hipcc branch around each load and
  // wait vmcnt(0) behind it, which serialises the stream): rows outside [row_lo,row_hi) re-read the
  // block's last valid row (same cache lines as the neighbouring lanes, so no extra HBM bytes) and
  // token columns >= M re-read token 0; both only feed accumulator entries that are never stored.
  auto load_w = [&](SkStage<RT>& st, int ts, int it) {
    const int kb = (wave + SK_WAVES * it) << 7;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int nb = row_lo + ts * RPS + ((EPI == EPI_SILU_MUL) ? (t >> 1) : t) * 16;
      int n = nb + r16;
      n = n < row_hi ? n : row_hi - 1;
      const bf16_t* base = (EPI == EPI_SILU_MUL && (t & 1)) ? p.W2 : p.W;
      // packed: row_lo is a multiple of 16 (launcher); tiles past the block's last one re-read it
      const bf16_t* wp = PK ? sk_pk(base, min(nb >> 4, (row_hi - 1) >> 4), p.K, kb, lane) : base + (int64_t)n * p.K + 8 * g + kb;
      constexpr int SS = PK ? 512 : 32;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        st.a[s][t] = NT ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + s * SS)) : *reinterpret_cast<const bf16x8*>(wp + s * SS);
    }
  };
  auto load_x = [&](SkStage<RT>& st, int it) {
    const int kb = (wave + SK_WAVES * it) << 7;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = kb + s * 32;
      const bf16_t* src = (k < p.K0) ? xa : xb;
      st.b[s] = *reinterpret_cast<const bf16x8*>(src + k);
    }
  };

  f32x4 acc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float ssq = 0.f;
  int parity = 0, done_ts = 0;

  auto finish_tile_step = [&](int ts) {
#pragma unroll
    for (int t = 0; t < RT; ++t) *reinterpret_cast<f32x4*>(&red[parity][wave][t][lane][0]) = acc[t];
    if (do_norm) {
      float v = ssq;
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (g == 0) red_ss[parity][wave][r16] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave == (done_ts & (SK_WAVES - 1))) {
      float rstd = 1.f;
      if (do_norm) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < SK_WAVES; ++w) v += red_ss[parity][w][r16];
        rstd = rsqrtf(v / (float)p.K + p.norm_eps);
      }
      f32x4 sum[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        f32x4 sv = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < SK_WAVES; ++w) sv += *reinterpret_cast<const f32x4*>(&red[parity][w][t][lane][0]);
        sum[t] = sv * rstd;
      }
      // D layout of mfma 16x16: col = lane&15 (token m), rows 4*(lane>>4)+r (features)
      const int m = r16;
      if (m < p.M) {
#pragma unroll
        for (int u = 0; u < RTW; ++u) {
          const int nb = row_lo + ts * RPS + u * 16 + 4 * g;
          float y[4];
          if (EPI == EPI_SILU_MUL) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float gt = rbf(sum[2 * u][r]), up = rbf(sum[2 * u + 1][r]);
              y[r] = rbf(gt / (1.0f + __expf(-gt))) * up;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int n = nb + r;
              float v = sum[u][r];
              if (p.bias && n < row_hi) v += bf2f(p.bias[n]);
              v = rbf(v);
              if (p.res && n < row_hi) v = v + bf2f(p.res[(int64_t)m * p.ldr + n]);
              y[r] = v;
            }
          }
          bf16_t* o = p.out + (int64_t)m * p.ldo + nb;
          if (nb + 3 < row_hi && ((p.ldo | nb) & 3) == 0) {
            uint2 v = {pack2bf(y[0], y[1]), pack2bf(y[2], y[3])};
            *reinterpret_cast<uint2*>(o) = v;
          } else {
            for (int r = 0; r < 4; ++r) if (nb + r < row_hi) o[r] = f2bf(y[r]);
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ssq = 0.f;
    parity ^= 1;
    ++done_ts;
  };

  auto compute = [&](const SkStage<RT>& st, int ts, int it) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st.a[s][t], st.b[s], acc[t], 0, 0, 0);
    if (do_norm) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float xv = (float)st.b[s][j]; ssq = fmaf(xv, xv, ssq); }
    }
    if (it == n_iter - 1) finish_tile_step(ts);
  };

  SkStage<RT> s0, s1;
  int lts = 0, lit = 0;                       // load cursor
  auto adv = [&](int& ts, int& it) { if (++it == n_iter) { it = 0; ++ts; } };
  // prologue.  Gathered activations (fc: embed_tokens[flat_ids[tree_indices]]) need two dependent
  // index loads before the first x load can be issued; the first two weight stages are requested
  // before that chain so that HBM is already streaming (they are re-requested below, from L2, in
  // the steady-state order: hipcc's vmcnt counting at the loop head follows the issue order of its
  // predecessor blocks, and only an identical order keeps the refill of one stage in flight while
  // the other stage is multiplied).
  if (p.idx0 || p.idx1) {
    load_w(s0, 0, 0);
    { int t1 = 0, i1 = 0; adv(t1, i1); load_w(s1, t1, i1); }
    int64_t ra = 0, rb = 0;
    if (mvalid) {
      const int m = r16;
      if (p.ind0) ra = p.idx0[(int64_t)(m / p.ind_n) * p.idx0_bstride + p.ind0[m % p.ind_n]];
      else ra = p.idx0 ? p.idx0[m] : m;
      rb = p.idx1 ? p.idx1[m] : m;
    }
    xa = p.x0 + ra * p.ld0 + 8 * g;
    xb = p.x1 ? (p.x1 + rb * p.ld1 + 8 * g - p.K0) : xa;
  } else {
    const int m = mvalid ? r16 : 0;
    xa = p.x0 + (int64_t)m * p.ld0 + 8 * g;
    xb = p.x1 ? (p.x1 + (int64_t)m * p.ld1 + 8 * g - p.K0) : xa;
  }
  __builtin_amdgcn_sched_barrier(0);
  load_w(s0, lts, lit); load_x(s0, lit); adv(lts, lit);
  __builtin_amdgcn_sched_barrier(0);
  // unconditional (a stage past the end re-reads clamped rows): a guarded load here would make the
  // loop head wait as if stage 1 had never been requested
  load_w(s1, lts, lit); load_x(s1, lit); adv(lts, lit);
  __builtin_amdgcn_sched_barrier(0);
  int cts = 0, cit = 0;                       // compute cursor
  // steady state: both stage refills are unconditional, so that hipcc can count vmcnt across the
  // loop (a refill under `if` makes it wait for everything at the loop head)
  int j = 0;
  for (; j + 3 < J; j += 2) {
    compute(s0, cts, cit); adv(cts, cit);
    load_w(s0, lts, lit); load_x(s0, lit); adv(lts, lit);
    compute(s1, cts, cit); adv(cts, cit);
    load_w(s1, lts, lit); load_x(s1, lit); adv(lts, lit);
  }
  const int rem = J - j;                      // 1, 2 or 3 super-steps left; s0 = j, s1 = j + 1
  compute(s0, cts, cit); adv(cts, cit);
  if (rem >= 2) {
    if (rem == 3) { load_w(s0, lts, lit); load_x(s0, lit); }
    compute(s1, cts, cit); adv(cts, cit);
    if (rem == 3) compute(s0, cts, cit);
  }
}


// K == 128 * SK_WAVES * NI exactly (NI = 4: K = 4096, the q|k|v, o, gate/up and lm_head projections):
// every wave owns NI fixed K super-steps, so its x fragments are loaded ONCE and stay in registers
// for all of the block's rows (no per-tile-step re-read of x through L2: at 8 tokens that re-read
// is half as many bytes again as the weight stream), the RMSNorm row statistics are computed once,
// and the super-step loop is fully unrolled: stage parity, x fragment and refill target are static.
// MT = 16-row activation tiles whose fragments stay in registers (1: <= 16 rows, every tree step of one request; 2: <= 32
// rows, a 32-request batch at one row each -- each weight fragment then feeds two MFMAs instead of one).
#define SK_OST_COLS 640
// TKP > 0 (lm_head + top-k of 1-16 rows, no logits leave the kernel): the block's outputs become 16-bit monotone keys in LDS
// ([row][32 * TKP column slots], the slots past the block's columns hold key 0); when its stream is done every 16-lane group
// selects the p.topk best of one row (topk_select.h) and writes them as candidate records [row][block][topk] in topk.hip's
// format -- eagle_topk_merge_candidates finishes the job.  Against logits + the two-stage top-k: one launch and one 4 us scan
// of the logits less per draft step, the same ids and values (same rounding, same order: value, then lowest index).
#define SK_KSTRIDE(TKP) (64 * (TKP) + 64)               // bytes per key row: 4 rows read side by side fall in 4 different bank quarters
template <int RT, int EPI, int NI, bool NT, int S = 2, int MT = 1, bool STAGE = false, bool PK = false, int TKP = 0>
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_skinny_keepx_kernel(SkinnyArgs p, int rows_per_block) {
  static_assert(NI % S == 0, "S register stages rotate through the NI super-steps of a tile-step");
  static_assert(TKP == 0 || (EPI == EPI_LINEAR && !STAGE), "the top-k form has the linear epilogue and its own LDS image");
  constexpr int RTW = (EPI == EPI_SILU_MUL) ? RT / 2 : RT;
  constexpr int RPS = RTW * 16;
  __shared__ __attribute__((aligned(16))) float red[2][SK_WAVES][MT * RT][64][4];
  __shared__ float red_ss[SK_WAVES][MT * 16];
  // Output staging.  A tile-step yields 16 * RTW columns of every row: 32-byte pieces of rows that lie ld * 2 bytes apart,
  // one batch every few microseconds, written back by L2 as partial lines in the middle of the weight stream -- measured
  // at 151936 x 4096: +0.9 us per output ROW (253 us at 32 rows against 234 at 17 with the same instructions; with
  // non-temporal stores, which go to memory at once, 292).  The block's outputs wait here instead (its columns are
  // consecutive: up to SK_OST_COLS of them) and leave as whole rows of the block's column range when its stream is done
  // (32 rows: 253 -> 230 us).  The copy is a serial tail of ~1.5 us, so the launcher asks for it from M x N >= 400000 only
  // (the batch-1 lm_head of 32000 columns at 1-8 rows is faster without: 52.7 vs 54.2 us).
  __shared__ __attribute__((aligned(16))) bf16_t ost[STAGE ? MT * 16 : 1][STAGE ? SK_OST_COLS + 8 : 8];
  __shared__ __attribute__((aligned(16))) unsigned char kimg[TKP ? MT * 16 * SK_KSTRIDE(TKP) : 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  if (TKP) {                                           // key 0 everywhere; the first tile-step's barrier orders it before the key stores
    for (int i = threadIdx.x; i < MT * 16 * SK_KSTRIDE(TKP) / 16; i += SK_WAVES * 64) reinterpret_cast<uint4*>(kimg)[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  const bool stage_out = STAGE && rows_per_block <= SK_OST_COLS && ((rows_per_block | (int)(p.ldo & 7)) & 7) == 0 &&
                         ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
  const int row_lo = blockIdx.x * rows_per_block;
  const int row_hi = min(p.N, row_lo + rows_per_block);
  if (row_lo >= row_hi) return;
  const int n_ts = (row_hi - row_lo + RPS - 1) / RPS;
  const bool do_norm = p.norm_eps > 0.f;

  // Neighbouring blocks take their NI super-steps in rotated order: side by side they then read different 2 KB column
  // groups of the weight rows instead of marching through the same HBM channels in step (registers stay statically
  // indexed by `it`; only the addresses rotate).
  static_assert((NI & (NI - 1)) == 0, "rotation masks with NI - 1");
  const int rot = p.rot ? (blockIdx.x & (NI - 1)) : 0;
  bf16x8 a[S][4][RT];
  auto load_w = [&](bf16x8 (&st)[4][RT], int ts, int it) {
    const int kb = (wave + SK_WAVES * ((it + rot) & (NI - 1))) << 7;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int nb = row_lo + ts * RPS + ((EPI == EPI_SILU_MUL) ? (t >> 1) : t) * 16;
      int n = nb + r16;
      n = n < row_hi ? n : row_hi - 1;             // see gemm_skinny_stream_kernel: no predicated loads
      const bf16_t* base = (EPI == EPI_SILU_MUL && (t & 1)) ? p.W2 : p.W;
      const bf16_t* wp = PK ? sk_pk(base, min(nb >> 4, (row_hi - 1) >> 4), p.K, kb, lane) : base + (int64_t)n * p.K + 8 * g + kb;
      constexpr int SS = PK ? 512 : 32;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        st[s][t] = NT ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + s * SS)) : *reinterpret_cast<const bf16x8*>(wp + s * SS);
    }
  };
#pragma unroll
  for (int st = 0; st < S; ++st) load_w(a[st], 0, st);
  // activation rows (possibly gathered: two dependent index loads) -> registers, once
  bf16x8 xk[MT][NI][4];
  float rstd[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const bf16_t* xa;
    const bf16_t* xb;
    {
      int64_t ra = 0, rb = 0;
      const int m = mt * 16 + r16;
      if (m < p.M) {
        if (p.ind0) ra = p.idx0[(int64_t)(m / p.ind_n) * p.idx0_bstride + p.ind0[m % p.ind_n]];
        else ra = p.idx0 ? p.idx0[m] : m;
        rb = p.idx1 ? p.idx1[m] : m;
      }
      xa = p.x0 + ra * p.ld0 + 8 * g;
      xb = p.x1 ? (p.x1 + rb * p.ld1 + 8 * g - p.K0) : xa;
    }
#pragma unroll
    for (int it = 0; it < NI; ++it)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int k = ((wave + SK_WAVES * ((it + rot) & (NI - 1))) << 7) + s * 32;
        xk[mt][it][s] = *reinterpret_cast<const bf16x8*>(((k < p.K0) ? xa : xb) + k);
      }
    rstd[mt] = 1.f;
  }
  if (do_norm) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float ssq = 0.f;
#pragma unroll
      for (int it = 0; it < NI; ++it)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float xv = (float)xk[mt][it][s][j]; ssq = fmaf(xv, xv, ssq); }
      ssq += __shfl_xor(ssq, 16, 64);
      ssq += __shfl_xor(ssq, 32, 64);
      if (g == 0) red_ss[wave][mt * 16 + r16] = ssq;
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < SK_WAVES; ++w) v += red_ss[w][mt * 16 + r16];
      rstd[mt] = rsqrtf(v / (float)p.K + p.norm_eps);
    }
  }

  f32x4 acc[MT][RT];
  // One tile-step.  LAST (the block's final one): no refill past the block's rows -- with packed images those refills would
  // re-read the last tile's 1 KiB pieces from HBM (non-temporal lines do not stay): 64 KB per block, 6 % of the lm_head's bytes and
  // 16 % of q|k|v's (PMC, profiles/r03_lm_head_pmc.json); a peeled copy instead of a guarded load keeps hipcc's vmcnt counts exact.
  auto tile_step = [&](const int ts, auto last_c) {
    constexpr bool LAST = decltype(last_c)::value;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NI; ++it) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[it % S][s][t], xk[mt][it][s], acc[mt][t], 0, 0, 0);
      // refill this stage with the super-step S ahead
      if (!(LAST && it + S >= NI)) load_w(a[it % S], ts + (it + S) / NI, (it + S) % NI);
    }
    const int parity = ts & 1;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < RT; ++t) *reinterpret_cast<f32x4*>(&red[parity][wave][mt * RT + t][lane][0]) = acc[mt][t];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave == (ts & (SK_WAVES - 1))) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x4 sum[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          f32x4 sv = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int w = 0; w < SK_WAVES; ++w) sv += *reinterpret_cast<const f32x4*>(&red[parity][w][mt * RT + t][lane][0]);
          sum[t] = sv * rstd[mt];
        }
        const int m = mt * 16 + r16;
        if (m < p.M) {
#pragma unroll
          for (int u = 0; u < RTW; ++u) {
            const int nb = row_lo + ts * RPS + u * 16 + 4 * g;
            float y[4];
            if (EPI == EPI_SILU_MUL) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float gt = rbf(sum[2 * u][r]), up = rbf(sum[2 * u + 1][r]);
                y[r] = rbf(gt / (1.0f + __expf(-gt))) * up;
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int n = nb + r;
                float v = sum[u][r];
                if (p.bias && n < row_hi) v += bf2f(p.bias[n]);
                v = rbf(v);
                if (p.res && n < row_hi) v = v + bf2f(p.res[(int64_t)m * p.ldr + n]);
                y[r] = v;
              }
            }
            if constexpr (TKP > 0) {               // keys instead of logits; columns past row_hi keep key 0
              uint32_t k16[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) k16[r] = (nb + r < row_hi) ? tk_mono_bf16(f2bf(y[r])) : 0u;
              *reinterpret_cast<uint2*>(kimg + m * SK_KSTRIDE(TKP) + 2 * (nb - row_lo)) = make_uint2(k16[0] | (k16[1] << 16), k16[2] | (k16[3] << 16));
              continue;
            }
            bf16_t* o = p.out + (int64_t)m * p.ldo + nb;
            if (stage_out) {                       // columns past row_hi are staged too and never copied out
              uint2 v = {pack2bf(y[0], y[1]), pack2bf(y[2], y[3])};
              *reinterpret_cast<uint2*>(&ost[m][nb - row_lo]) = v;
            } else if (nb + 3 < row_hi && ((p.ldo | nb) & 3) == 0) {
              uint2 v = {pack2bf(y[0], y[1]), pack2bf(y[2], y[3])};
              *reinterpret_cast<uint2*>(o) = v;
            } else {
              for (int r = 0; r < 4; ++r) if (nb + r < row_hi) o[r] = f2bf(y[r]);
            }
          }
        }
      }
    }
  };
  for (int ts = 0; ts + 1 < n_ts; ++ts) tile_step(ts, std::false_type{});
  tile_step(n_ts - 1, std::true_type{});
  if constexpr (TKP > 0) {                             // selection: one 16-lane group per row
    __syncthreads();
    const int l16 = threadIdx.x & 15;
    for (int R = threadIdx.x >> 4; R < min(p.M, MT * 16); R += SK_WAVES * 4)
      tk_select_group16<TKP>(kimg + R * SK_KSTRIDE(TKP), row_hi - row_lo, (uint32_t)row_lo, p.topk,
                             p.cand + ((int64_t)R * gridDim.x + blockIdx.x) * p.topk, true, l16);
    return;
  }
  if (!stage_out) return;
  __syncthreads();
  const int ncols = row_hi - row_lo, cpr = (ncols + 7) >> 3;        // 16-byte chunks per row
  const int mrows = min(p.M, MT * 16);
  for (int idx = threadIdx.x; idx < mrows * cpr; idx += SK_WAVES * 64) {
    const int row = idx / cpr, c8 = (idx - row * cpr) * 8;
    const bf16_t* src = &ost[row][c8];
    bf16_t* o = p.out + (int64_t)row * p.ldo + row_lo + c8;
    if (c8 + 8 <= ncols) *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(src);
    else for (int r = 0; c8 + r < ncols; ++r) o[r] = src[r];
  }
}

static int g_skinny_variant = 3;    // 3: per-shape choice (default); 0: tile kernel, 1: streaming kernel, 2: register-resident x where legal
static int g_sk_blocks_per_cu = 2;
static int g_sk_min_rows = 16;
static int g_sk_keepx_stages = 2;   // register stages of the keepx kernel (2 or 4)
static int g_sk_keepx_bpc = 1;      // keepx workgroups per CU
static int g_sk_keepx_rt = 0;       // 0: by shape; 1 / 2: forced 16-row weight tiles per tile-step (EPI_LINEAR)
static int g_sk_silu_tile = 0;      // 1: SiLU*up projection of <= 16 rows on the tile kernel
static int g_sk_rot = 1;            // keepx: staggered K walk of neighbouring blocks (bit 15 of the variant word turns it off)
extern "C" void eagle_debug_set_skinny_variant(int v, int blocks_per_cu, int min_rows) {
  g_skinny_variant = v & 15;                       // bits 4-7: keepx register stages, bits 8-11: keepx blocks per CU
  if ((v >> 4) & 15) g_sk_keepx_stages = (v >> 4) & 15;
  if ((v >> 8) & 15) g_sk_keepx_bpc = (v >> 8) & 15;
  g_sk_keepx_rt = (v >> 12) & 3;
  g_sk_silu_tile = (v >> 14) & 1;
  g_sk_rot = ((v >> 15) & 1) ? 0 : 1;
  if (blocks_per_cu > 0) g_sk_blocks_per_cu = blocks_per_cu;
  if (min_rows > 0) g_sk_min_rows = min_rows;
}
static int skinny_variant() { return g_skinny_variant; }
static int sk_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// Cache policy of the weight stream.  A draft forward streams more weight bytes (734 MB, Llama-2-7B dims)
// than the 256 MB Infinity Cache holds, five times per cycle: with one policy for all of them nothing
// survives from one forward to the next.  The caller marks, PER CALL (EAGLE_GEMM_KEEP_RESIDENT in `epi`), the
// matrices that should stay resident (the host that owns the model picks the smallest ones, up to ~3/4 of the
// cache); those are read with the default policy, every other weight with non-temporal loads so that it does not
// displace them.  No state is kept in the library.
#define EAGLE_GEMM_KEEP_RESIDENT 0x100
#define EAGLE_GEMM_W_PACKED 0x200
static bool sk_stream_nt(const SkinnyArgs& a, int epi) { (void)epi; return a.nt; }

template <int RT, int EPI, int MT, bool PK>
static int launch_skinny_keepx_(const SkinnyArgs& a, hipStream_t st) {
  int grid = sk_num_cus() * g_sk_keepx_bpc;
  int R = (a.N + grid - 1) / grid;
  if (R < 16) R = 16;
  R = PK ? ((R + 15) & ~15) : ((R + 7) & ~7);          // 16-byte runs when the staged outputs are copied out; packed: whole 16-row tiles
  grid = (a.N + R - 1) / R;
  const bool nt = sk_stream_nt(a, EPI);
  if constexpr (MT == 1 && !PK) {
    if (g_sk_keepx_stages == 4) {
      if (nt) hipLaunchKernelGGL((gemm_skinny_keepx_kernel<RT, EPI, 4, true, 4>), dim3(grid), dim3(SK_WAVES * 64), 0, st, a, R);
      else hipLaunchKernelGGL((gemm_skinny_keepx_kernel<RT, EPI, 4, false, 4>), dim3(grid), dim3(SK_WAVES * 64), 0, st, a, R);
      return eagle_check_launch();
    }
  }
  if (a.stage) {                                       // the instantiation that collects its outputs in LDS (see the kernel)
    if (nt) hipLaunchKernelGGL((gemm_skinny_keepx_kernel<RT, EPI, 4, true, 2, MT, true, PK>), dim3(grid), dim3(SK_WAVES * 64), 0, st, a, R);
    else hipLaunchKernelGGL((gemm_skinny_keepx_kernel<RT, EPI, 4, false, 2, MT, true, PK>), dim3(grid), dim3(SK_WAVES * 64), 0, st, a, R);
    return eagle_check_launch();
  }
  if (nt) hipLaunchKernelGGL((gemm_skinny_keepx_kernel<RT, EPI, 4, true, 2, MT, false, PK>), dim3(grid), dim3(SK_WAVES * 64), 0, st, a, R);
  else hipLaunchKernelGGL((gemm_skinny_keepx_kernel<RT, EPI, 4, false, 2, MT, false, PK>), dim3(grid), dim3(SK_WAVES * 64), 0, st, a, R);
  return eagle_check_launch();
}
template <int RT, int EPI, int MT = 1>
static int launch_skinny_keepx(const SkinnyArgs& a, hipStream_t st) {
  return a.packed ? launch_skinny_keepx_<RT, EPI, MT, true>(a, st) : launch_skinny_keepx_<RT, EPI, MT, false>(a, st);
}

// lm_head + top-k of 1..16 rows over a packed image (K = 4096): blocks / columns per block of the launch, 0 if not served
static int sk_topk_blocks(int M, int N, int K, int k, int* R_out) {
  if (M < 1 || M > 16 || K != 128 * SK_WAVES * 4 || N < 16 || k < 1 || k > 16) return 0;
  int grid = sk_num_cus() * g_sk_keepx_bpc;
  int R = (N + grid - 1) / grid;
  if (R < 16) R = 16;
  R = (R + 15) & ~15;
  grid = (N + R - 1) / R;
  if (R > 640 || (int64_t)grid * k > 256 * 24 || N < k) return 0;     // the merge takes <= 6144 candidates (a last block with fewer than k columns leaves empty records, which it skips)
  if (R_out) *R_out = R;
  return grid;
}
int eagle_skinny_topk_blocks(int M, int N, int K, int k) { return sk_topk_blocks(M, N, K, k, nullptr); }
// Called by eagle_lm_head_topk_rows_bf16 (gemm_rows.hip) for 1..16 rows after its argument checks; cand: [M][blocks][k] records.
int eagle_launch_skinny_topk(const void* x0, const int64_t* idx0, int64_t ld0, const int64_t* ind0, int ind_n, int64_t idx0_bstride,
                             const void* Wpacked, const void* bias, int M, int N, int K, int k, void* cand, bool nt, hipStream_t st) {
  int R = 0;
  const int grid = sk_topk_blocks(M, N, K, k, &R);
  if (!grid) return EAGLE_ERR_SHAPE;
  SkinnyArgs a;
  a.x0 = (const bf16_t*)x0; a.idx0 = idx0; a.ld0 = ld0; a.K0 = K;
  a.ind0 = ind0; a.ind_n = ind_n; a.idx0_bstride = idx0_bstride; a.norm_eps = 0.f;
  a.x1 = nullptr; a.idx1 = nullptr; a.ld1 = 0;
  a.W = (const bf16_t*)Wpacked; a.W2 = nullptr; a.bias = (const bf16_t*)bias; a.res = nullptr; a.ldr = 0;
  a.out = nullptr; a.ldo = 0; a.M = M; a.N = N; a.K = K;
  a.nt = nt; a.rot = g_sk_rot; a.stage = 0; a.packed = 1;
  a.cand = (unsigned long long*)cand; a.topk = k;
#define SK_TOPK(TKP_) do {                                                                                                          \
    if (nt) hipLaunchKernelGGL((gemm_skinny_keepx_kernel<1, EPI_LINEAR, 4, true, 2, 1, false, true, TKP_>), dim3(grid), dim3(SK_WAVES * 64), 0, st, a, R);  \
    else hipLaunchKernelGGL((gemm_skinny_keepx_kernel<1, EPI_LINEAR, 4, false, 2, 1, false, true, TKP_>), dim3(grid), dim3(SK_WAVES * 64), 0, st, a, R);   \
  } while (0)
  if (R <= 128) SK_TOPK(4);
  else if (R <= 256) SK_TOPK(8);
  else SK_TOPK(20);
#undef SK_TOPK
  return eagle_check_launch();
}

template <int RT, int EPI, bool PK>
static int launch_skinny_stream_(const SkinnyArgs& a, hipStream_t st) {
  static int occ = 0;                                  // resident 512-thread blocks per CU for this instantiation
  if (!occ) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gemm_skinny_stream_kernel<RT, EPI, true, PK>, SK_WAVES * 64, 0) != hipSuccess || occ < 1) occ = 1;
  }
  int grid = sk_num_cus() * (occ < g_sk_blocks_per_cu ? occ : g_sk_blocks_per_cu);
  int R = (a.N + grid - 1) / grid;
  if (R < g_sk_min_rows) R = g_sk_min_rows;            // few rows per block waste MFMA lanes, not bandwidth
  R = PK ? ((R + 15) & ~15) : ((R + 3) & ~3);          // keeps the 8-byte output stores aligned; packed: whole 16-row tiles
  grid = (a.N + R - 1) / R;
  const int nfull = (a.K + 127) >> 7;
  const int n_iter = (nfull + SK_WAVES - 1) / SK_WAVES;
  if (sk_stream_nt(a, EPI)) hipLaunchKernelGGL((gemm_skinny_stream_kernel<RT, EPI, true, PK>), dim3(grid), dim3(SK_WAVES * 64), 0, st, a, R, n_iter);
  else hipLaunchKernelGGL((gemm_skinny_stream_kernel<RT, EPI, false, PK>), dim3(grid), dim3(SK_WAVES * 64), 0, st, a, R, n_iter);
  return eagle_check_launch();
}
template <int RT, int EPI>
static int launch_skinny_stream(const SkinnyArgs& a, hipStream_t st) {
  return a.packed ? launch_skinny_stream_<RT, EPI, true>(a, st) : launch_skinny_stream_<RT, EPI, false>(a, st);
}

template <int RT, int MT, int EPI>
static int launch_skinny(const SkinnyArgs& a, hipStream_t st) {
  const int rows_per_block = (EPI == EPI_SILU_MUL) ? 16 : RT * 16;
  dim3 grid((a.N + rows_per_block - 1) / rows_per_block), block(SK_WAVES * 64);
  if (a.packed) hipLaunchKernelGGL((gemm_skinny_kernel<RT, MT, EPI, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((gemm_skinny_kernel<RT, MT, EPI, false>), grid, block, 0, st, a);
  return eagle_check_launch();
}

// ---------------------------------------------------------------------------------------------
// eagle_pack_weight_bf16: the packed image of W [N, K] (row stride ldw) -- see the head of this file.  One 16-byte piece per
// thread; run once per weight version (the host caches the result beside its other derived weights).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weight_kernel(const bf16_t* __restrict__ W, int64_t ldw, int N, int K, bf16_t* __restrict__ out,
                                                          int64_t pieces) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= pieces) return;
  const int KF = K >> 5;
  const int lane = (int)(c & 63);
  const int64_t blk = c >> 6;
  const int f = (int)(blk % KF);
  const int64_t n = (blk / KF) * 16 + (lane & 15);
  uint4 v = {0u, 0u, 0u, 0u};
  if (n < N) v = *reinterpret_cast<const uint4*>(W + n * ldw + 32 * f + 8 * (lane >> 4));
  *reinterpret_cast<uint4*>(out + c * 8) = v;
}

extern "C" int64_t eagle_packed_weight_bytes(int N, int K) {
  if (N < 1 || K < 32 || (K & 31)) return 0;
  return (int64_t)((N + 15) / 16) * 16 * K * (int64_t)sizeof(bf16_t);
}

extern "C" int eagle_pack_weight_bf16(const void* W, int64_t ldw, int N, int K, void* packed, hipStream_t stream) {
  if (!W || !packed) return EAGLE_ERR_ARG;
  if (N < 1 || K < 32 || (K & 31) || ldw < K || (ldw & 7)) return EAGLE_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(packed) & 15)) return EAGLE_ERR_SHAPE;
  const int64_t pieces = (int64_t)((N + 15) / 16) * 16 * K / 8;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)W, ldw, N, K,
                     (bf16_t*)packed, pieces);
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
  const bool keep_resident = (epi & EAGLE_GEMM_KEEP_RESIDENT) != 0;
  const bool packed = (epi & EAGLE_GEMM_W_PACKED) != 0;
  epi &= ~(EAGLE_GEMM_KEEP_RESIDENT | EAGLE_GEMM_W_PACKED);
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
    a.M = mc; a.N = N; a.K = K; a.nt = !keep_resident; a.rot = g_sk_rot; a.packed = packed ? 1 : 0;
    a.cand = nullptr; a.topk = 0;
    a.stage = (int64_t)mc * N >= 400000 ? 1 : 0;
    int rc;
    const int var = skinny_variant();   // 3 = per-shape choice (measured); 0/1/2 force one kernel family
    const bool big = (epi == EPI_SILU_MUL) || N >= 8192;
    if (mc <= 16 && K == 128 * SK_WAVES * 4 && (var == 2 || (var == 3 && big)) && !(g_sk_silu_tile && epi == EPI_SILU_MUL)) {
      if (epi == EPI_SILU_MUL) rc = launch_skinny_keepx<2, EPI_SILU_MUL>(a, stream);
      // one 16-row weight tile per tile-step: 32 KB in flight per CU stream as fast as 64 KB and the shorter tile-step
      // ends sooner (lm_head 32000 x 4096 from cold caches 52.2 -> 49.0 us, in-cycle 53.7 -> 52.4; tools/skinny_sweep.py)
      else if (g_sk_keepx_rt == 2) rc = launch_skinny_keepx<2, EPI_LINEAR>(a, stream);
      else rc = launch_skinny_keepx<1, EPI_LINEAR>(a, stream);
    } else if (mc > 16 && mc <= 32 && K == 128 * SK_WAVES * 4 && epi == EPI_LINEAR && N >= 8192 && (var == 2 || var == 3)) {
      // 17-32 rows against a tall matrix (the lm_head of a 32-request batch at one row each): both 16-row x tiles in registers
      rc = launch_skinny_keepx<1, EPI_LINEAR, 2>(a, stream);
    } else if (mc <= 16 && (K & 127) == 0 && K >= 128 * SK_WAVES && (var == 1 || var == 2 || (var == 3 && K >= 8192)) &&
               !(g_sk_silu_tile && epi == EPI_SILU_MUL)) {
      if (epi == EPI_SILU_MUL) rc = launch_skinny_stream<2, EPI_SILU_MUL>(a, stream);
      else if (N >= 8192) rc = launch_skinny_stream<2, EPI_LINEAR>(a, stream);
      else rc = launch_skinny_stream<1, EPI_LINEAR>(a, stream);
    } else if (epi == EPI_SILU_MUL) {
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
