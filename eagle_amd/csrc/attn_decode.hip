// Tree-masked decode attention for the draft head, fused with RoPE and the KV-cache append.
//
// Replaces, for q_len <= 16 rows per call, the body of LlamaAttention.forward after the q/k/v
// projections (eagle/llama2.py:226-273): RoPE at explicit position_ids (:61-69, :233-234), the
// cache concat (:236-239, here an in-place append into a preallocated cache), scores/sqrt(hd)
// (:247), the additive mask of _prepare_decoder_attention_mask (:460-491: causal when q_len>1,
// padding, tree overlay on the last T columns), fp32 softmax cast back to bf16 (:263), P.V (:264)
// and the transpose back to [B, q, H*hd] (:272-273).
//
// One workgroup per (batch, head).  K rows stream HBM/L2 -> VGPR as MFMA A fragments
// (v_mfma_f32_16x16x32_bf16, keys x queries), scores and probabilities live in LDS as bf16 (the
// reference rounds both to bf16), P.V runs on the VALU with 256-B coalesced V-row reads.
// Intermediate roundings follow the reference's bf16 eager op order.
#include "common.h"

#define AD_THREADS 256
#define AD_MAXM 16
#define AD_MAXHD 128
#define BF_NEG_INF ((bf16_t)0xFF80)

struct AttnDecArgs {
  const bf16_t* qkv; int64_t ld_qkv;
  const int64_t* pos;
  const bf16_t* cos_t; const bf16_t* sin_t; int n_pos;
  bf16_t* kcache; bf16_t* vcache; int64_t c_bstride; int64_t c_hstride;
  const int32_t* past_len; int past_off;
  const uint8_t* tree_mask; int T; int64_t tm_bstride;
  const uint8_t* pad_mask; int64_t pm_stride;
  bf16_t* out; int64_t ld_out;
  int* err_flag;
  int B, M, H, HK, hd, cap, max_kv, causal;
  int rows_per_req, row_off;  // row of request b, query m = b*rows_per_req + row_off + m
  float sqrt_hd;
};

__global__ __launch_bounds__(AD_THREADS) void attn_decode_kernel(AttnDecArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // carve: q_s[16][hd] bf16 | knew[16][hd] | vnew[16][hd] | S[16][max_kv_pad] bf16
  const int hd = p.hd;
  bf16_t* q_s = reinterpret_cast<bf16_t*>(smem);
  bf16_t* knew = q_s + AD_MAXM * AD_MAXHD;
  bf16_t* vnew = knew + AD_MAXM * AD_MAXHD;
  bf16_t* S = vnew + AD_MAXM * AD_MAXHD;
  const int ldS = (p.max_kv + 15) & ~15;

  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int rep = p.H / p.HK, hk = h / rep;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int M = p.M;
  const int past = p.past_len[b] + p.past_off;
  const int total = past + M;
  if (past < 0 || total > p.cap || total > p.max_kv) {
    if (t == 0 && p.err_flag) atomicExch(p.err_flag, 1);
    return;
  }
  bf16_t* kc = p.kcache + (int64_t)b * p.c_bstride + (int64_t)hk * p.c_hstride;
  bf16_t* vc = p.vcache + (int64_t)b * p.c_bstride + (int64_t)hk * p.c_hstride;
  const int half = hd >> 1;

  // ---- phase 0: RoPE on q,k ; stage new k,v rows in LDS ; append them to the cache ----
  for (int i = t; i < AD_MAXM * hd; i += AD_THREADS) {
    const int m = i / hd, d = i - m * hd;
    if (m < M) {
      const int64_t row = (int64_t)b * p.rows_per_req + p.row_off + m;
      int64_t ps = p.pos[row];
      ps = ps < 0 ? 0 : (ps >= p.n_pos ? p.n_pos - 1 : ps);
      const float c = bf2f(p.cos_t[ps * hd + d]), s = bf2f(p.sin_t[ps * hd + d]);
      const int dp = d < half ? d + half : d - half;
      const bf16_t* qrow = p.qkv + row * p.ld_qkv + h * hd;
      const bf16_t* krow = p.qkv + row * p.ld_qkv + (p.H + hk) * hd;
      const bf16_t* vrow = p.qkv + row * p.ld_qkv + (p.H + p.HK + hk) * hd;
      float qv = bf2f(qrow[d]), qr = bf2f(qrow[dp]);
      float kv = bf2f(krow[d]), kr = bf2f(krow[dp]);
      if (d < half) { qr = -qr; kr = -kr; }
      const bf16_t qe = f2bf(rbf(qv * c) + rbf(qr * s));
      const bf16_t ke = f2bf(rbf(kv * c) + rbf(kr * s));
      const bf16_t ve = vrow[d];
      q_s[m * AD_MAXHD + d] = qe;
      knew[m * AD_MAXHD + d] = ke;
      vnew[m * AD_MAXHD + d] = ve;
      if (h % rep == 0) {
        kc[(int64_t)(past + m) * hd + d] = ke;
        vc[(int64_t)(past + m) * hd + d] = ve;
      }
    } else {
      q_s[m * AD_MAXHD + d] = 0;
    }
  }
  __syncthreads();

  // ---- phase 1: scores = bf16(bf16(q.k) / sqrt(hd)) + mask, keys x queries on MFMA ----
  const int r16 = lane & 15, g = lane >> 4;
  const int ksteps = (hd + 31) >> 5;
  bf16x8 qf[AD_MAXHD / 32];
#pragma unroll
  for (int s = 0; s < AD_MAXHD / 32; ++s) {
    bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    const int d0 = 32 * s + 8 * g;
    qf[s] = (s < ksteps && d0 < hd) ? *reinterpret_cast<const bf16x8*>(q_s + r16 * AD_MAXHD + d0) : z;
  }
  const uint8_t* tm = p.tree_mask ? p.tree_mask + (int64_t)b * p.tm_bstride : nullptr;
  const uint8_t* pm = p.pad_mask ? p.pad_mask + (int64_t)b * p.pm_stride : nullptr;
  const int tree0 = total - p.T;
  for (int c0 = wave * 16; c0 < total; c0 += 4 * 16) {
    const int c = c0 + r16;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < AD_MAXHD / 32; ++s) {
      if (s < ksteps) {
        bf16x8 kf = {0, 0, 0, 0, 0, 0, 0, 0};
        const int d0 = 32 * s + 8 * g;
        if (d0 < hd && c < total) {
          kf = (c < past) ? *reinterpret_cast<const bf16x8*>(kc + (int64_t)c * hd + d0)
                          : *reinterpret_cast<const bf16x8*>(knew + (c - past) * AD_MAXHD + d0);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], acc, 0, 0, 0);
      }
    }
    // lane holds query m = r16, keys c0 + 4g + r
    const int m = r16;
    uint32_t w[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cc = c0 + 4 * g + r;
      bool vis = (m < M) && (cc < total);
      if (p.causal && M > 1) vis = vis && (cc <= past + m);
      if (pm) vis = vis && (cc < total) && (pm[cc < total ? cc : 0] != 0);
      if (tm && cc >= tree0 && cc < total) vis = vis && (tm[m < M ? m * p.T + (cc - tree0) : 0] != 0);
      const bf16_t sc = vis ? f2bf(rbf(acc[r]) / p.sqrt_hd) : BF_NEG_INF;
      if (r & 1) w[r >> 1] |= ((uint32_t)sc) << 16; else w[r >> 1] = sc;
    }
    *reinterpret_cast<uint2*>(S + m * ldS + c0 + 4 * g) = make_uint2(w[0], w[1]);
  }
  __syncthreads();

  // ---- phase 2: fp32 softmax per row, probabilities rounded to bf16 in place ----
  {
    const int m = t >> 4, j = t & 15;
    if (m < M) {
      bf16_t* row = S + m * ldS;
      float mx = -INFINITY;
      for (int c = j; c < total; c += 16) mx = fmaxf(mx, bf2f(row[c]));
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
      float sum = 0.f;
      for (int c = j; c < total; c += 16) sum += expf(bf2f(row[c]) - mx);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
      for (int c = j; c < total; c += 16) row[c] = f2bf(expf(bf2f(row[c]) - mx) / sum);
      // zero the padding up to a multiple of 8 so phase 3 can read 16-byte groups
      for (int c = total + j; c < ((total + 7) & ~7); c += 16) row[c] = 0;
    }
  }
  __syncthreads();

  // ---- phase 3: O = P.V on the VALU; thread = (2 adjacent d, row group) ----
  {
    const int dp = t % half, mg = t / half, nmg = AD_THREADS / half;
    if (mg < AD_MAXM) {
      constexpr int RPT = 4;  // rows per thread (nmg * RPT >= 16 when hd = 128)
      float a0[RPT], a1[RPT];
#pragma unroll
      for (int i = 0; i < RPT; ++i) a0[i] = a1[i] = 0.f;
      for (int mb = mg; mb < M; mb += nmg * RPT) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) a0[i] = a1[i] = 0.f;
        for (int c0 = 0; c0 < total; c0 += 8) {
          uint32_t vv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int c = c0 + u;
            vv[u] = 0;
            if (c < total)
              vv[u] = (c < past) ? *reinterpret_cast<const uint32_t*>(vc + (int64_t)c * hd + 2 * dp)
                                 : *reinterpret_cast<const uint32_t*>(vnew + (c - past) * AD_MAXHD + 2 * dp);
          }
#pragma unroll
          for (int i = 0; i < RPT; ++i) {
            const int m = mb + i * nmg;
            if (m < M) {
              const uint4 pv = *reinterpret_cast<const uint4*>(S + m * ldS + c0);
              const uint32_t* pu = reinterpret_cast<const uint32_t*>(&pv);
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const float pr = (u & 1) ? hi_bf(pu[u >> 1]) : lo_bf(pu[u >> 1]);
                a0[i] += pr * lo_bf(vv[u]);
                a1[i] += pr * hi_bf(vv[u]);
              }
            }
          }
        }
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          const int m = mb + i * nmg;
          if (m < M)
            *reinterpret_cast<uint32_t*>(p.out + ((int64_t)b * p.rows_per_req + p.row_off + m) * p.ld_out + h * hd + 2 * dp) = pack2bf(a0[i], a1[i]);
        }
      }
    }
  }
}

extern "C" int64_t eagle_attn_decode_lds_bytes(int max_kv) {
  return (int64_t)3 * AD_MAXM * AD_MAXHD * 2 + (int64_t)AD_MAXM * (((max_kv + 15) & ~15) + 16) * 2;
}

// See include/eagle_hip.h for the contract.
extern "C" int eagle_tree_attention_bf16(const void* qkv, int64_t ld_qkv, const int64_t* position_ids,
                                         const void* cos_table, const void* sin_table, int n_pos,
                                         void* kcache, void* vcache, int64_t cache_batch_stride, int cap,
                                         const int32_t* past_len, int past_off,
                                         const uint8_t* tree_mask, int T, int64_t tree_mask_batch_stride,
                                         const uint8_t* pad_mask, int64_t pad_mask_stride,
                                         void* out, int64_t ld_out, int* err_flag,
                                         int B, int M, int H, int HK, int hd, int max_kv, int causal,
                                         int rows_per_req, int row_off, hipStream_t stream) {
  if (!qkv || !position_ids || !cos_table || !sin_table || !kcache || !vcache || !past_len || !out) return EAGLE_ERR_ARG;
  if (rows_per_req < M || row_off < 0 || row_off + M > rows_per_req) return EAGLE_ERR_SHAPE;
  if (B < 1 || M < 1 || M > AD_MAXM || H < 1 || HK < 1 || (H % HK) || hd < 8 || hd > AD_MAXHD || (hd & 7)) return EAGLE_ERR_SHAPE;
  if (AD_THREADS % (hd / 2) || (AD_THREADS / (hd / 2)) * 4 < 1) return EAGLE_ERR_SHAPE;
  if (max_kv < M || max_kv > cap || T < 0 || T > max_kv || (tree_mask && T < 1)) return EAGLE_ERR_SHAPE;
  if ((ld_qkv & 7) || (ld_out & 1)) return EAGLE_ERR_SHAPE;
  const int64_t lds = eagle_attn_decode_lds_bytes(max_kv);
  if (lds > 160 * 1024) return EAGLE_ERR_SHAPE;
  static int64_t lds_attr = 0;
  if (lds > lds_attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) return EAGLE_ERR_LAUNCH;
    lds_attr = lds;
  }
  AttnDecArgs a;
  a.qkv = (const bf16_t*)qkv; a.ld_qkv = ld_qkv; a.pos = position_ids;
  a.cos_t = (const bf16_t*)cos_table; a.sin_t = (const bf16_t*)sin_table; a.n_pos = n_pos;
  a.kcache = (bf16_t*)kcache; a.vcache = (bf16_t*)vcache; a.c_bstride = cache_batch_stride; a.c_hstride = (int64_t)cap * hd;
  a.past_len = past_len; a.past_off = past_off;
  a.tree_mask = tree_mask; a.T = tree_mask ? T : 0; a.tm_bstride = tree_mask_batch_stride;
  a.pad_mask = pad_mask; a.pm_stride = pad_mask_stride;
  a.out = (bf16_t*)out; a.ld_out = ld_out; a.err_flag = err_flag;
  a.B = B; a.M = M; a.H = H; a.HK = HK; a.hd = hd; a.cap = cap; a.max_kv = max_kv; a.causal = causal;
  a.rows_per_req = rows_per_req; a.row_off = row_off;
  a.sqrt_hd = sqrtf((float)hd);
  hipLaunchKernelGGL(attn_decode_kernel, dim3(B * H), dim3(AD_THREADS), (size_t)lds, stream, a);
  return eagle_check_launch();
}
