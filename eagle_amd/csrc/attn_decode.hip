// Tree-masked decode attention for the draft head, fused with RoPE and the KV-cache append.
//
// Replaces, for q_len <= 16 rows per call, the body of LlamaAttention.forward after the q/k/v
// projections (eagle/llama2.py:226-273): RoPE at explicit position_ids (:61-69, :233-234), the
// cache concat (:236-239, here an in-place append into a preallocated cache), scores/sqrt(hd)
// (:247), the additive mask of _prepare_decoder_attention_mask (:460-491: causal when q_len>1,
// padding, tree overlay on the last T columns), fp32 softmax (:263), P.V (:264) and the
// transpose back to [B, q, H*hd] (:272-273).
//
// Split-KV ("flash-decoding") layout so that a batch-1 request still fills the chip:
//   kernel 1: one workgroup per (64-key chunk, head, request).  K rows stream HBM/L2 -> VGPR as
//             MFMA A fragments (v_mfma_f32_16x16x32_bf16: keys x queries), chunk-local softmax
//             in fp32, P.V on the VALU with every V load of the chunk in flight at once;
//             writes (m, l, O~) partials.
//   kernel 2: one workgroup per (head, request) merges the partials and writes bf16 rows.
// q/k RoPE roundings follow the reference's bf16 eager op order; scores are rounded to bf16
// before and after the 1/sqrt(hd) scaling like the reference; the probabilities stay fp32 (the
// reference rounds them to bf16) -- inside the stated bf16 tolerance.
#include "common.h"

#define AD_THREADS 256
#define AD_MAXM 16
#define AD_MAXHD 128
#define AD_CH 64  // keys per workgroup

struct AttnDecArgs {
  const bf16_t* qkv; int64_t ld_qkv;
  const int64_t* pos; int pos_off; int pos_stride;
  const bf16_t* cos_t; const bf16_t* sin_t; int n_pos;
  bf16_t* kcache; bf16_t* vcache; int64_t c_bstride; int64_t c_hstride;
  const int32_t* past_len; int past_off;
  const uint8_t* tree_mask; int T; int64_t tm_bstride;
  const uint8_t* pad_mask; int64_t pm_stride;
  bf16_t* out; int64_t ld_out;
  float* ws; int max_splits;
  int* err_flag;
  int B, M, H, HK, hd, cap, causal;
  int rows_per_req, row_off;  // row of request b, query m = b*rows_per_req + row_off + m
  int vec16;                  // hd % 16 == 0 and every base 16-byte aligned: 16-byte phase-0 accesses
  float sqrt_hd;
  int split_keys;             // keys per partial record (64: attn_split_kernel; 512: attn_wide_kernel)
  bf16_t* q_roped;            // wide path: [B][AD_MAXM][H*hd] roped queries (behind the partial records in the workspace)
  int gq;                     // wide path: query heads of one KV head packed into the 16 MFMA columns of a workgroup (1, 2 or 4)
};

// workspace record per (b, h, split): [AD_MAXM][AD_MAXHD] fp32 O~  then [AD_MAXM][2] fp32 (m, l)
#define AD_REC (AD_MAXM * AD_MAXHD + AD_MAXM * 2)

// MM = row capacity of this instantiation (4, 8 or 16 >= M): everything but q_s (all 16 MFMA columns, rows >= M zero) has
// MM rows, so a 4-row tree step takes 15 KB of LDS and 4 workgroups (116 registers per lane) share a CU where the 16-row
// layout (48 KB) fits 3
// -- a wide ragged batch runs tens of workgroups per CU, each a chain of memory latencies (PMC: waves wait 50 % of
// their cycles at 2 resident waves per SIMD).
// HD = compile-time head_dim (128, both shipped configs) or 0 = read it from the arguments: with the constant every
// index division, the key-group geometry and the small-head_dim paths fold away (the kernel is VALU- and
// latency-bound on wide batches: PMC 61 % VALU-busy, a third of it integer and predicate work).
template <int MM, int HD>
__global__ __launch_bounds__(AD_THREADS) void attn_split_kernel(AttnDecArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t q_s[AD_MAXM * AD_MAXHD];
  __shared__ __attribute__((aligned(16))) bf16_t knew[MM * AD_MAXHD];
  __shared__ __attribute__((aligned(16))) bf16_t vnew[MM * AD_MAXHD];
  __shared__ __attribute__((aligned(16))) float P[MM * AD_CH];
  __shared__ __attribute__((aligned(16))) float red[4][MM][AD_MAXHD];
  __shared__ float rowm[AD_MAXM], rowl[AD_MAXM];

  const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int hd = HD ? HD : p.hd, M = p.M;
  const int rep = p.H / p.HK, hk = h / rep;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int past = p.past_len[b] + p.past_off;
  const int total = past + M;
  if (past < 0 || total > p.cap || (total + AD_CH - 1) / AD_CH > p.max_splits) {
    if (t == 0 && split == 0 && p.err_flag) atomicExch(p.err_flag, 1);
    return;
  }
  const int c_begin = split * AD_CH;
  if (c_begin >= total) return;
  const int c_end = (c_begin + AD_CH < total) ? c_begin + AD_CH : total;
  bf16_t* kc = p.kcache + (int64_t)b * p.c_bstride + (int64_t)hk * p.c_hstride;
  bf16_t* vc = p.vcache + (int64_t)b * p.c_bstride + (int64_t)hk * p.c_hstride;
  const int half = hd >> 1;
  const bool need_new = c_end > past;           // this chunk contains rows appended by this call
  const bool writer = (split == 0) && (h % rep == 0);

  // ---- early loads: cached K fragments and V pairs do not depend on this call's RoPE, so they
  //      are issued first and their latency overlaps phases 0-2 ----
  const int r16 = lane & 15, g = lane >> 4;
  const int ksteps = (hd + 31) >> 5;
  const int kc0 = c_begin + wave * 16;           // first key of this wave's MFMA tile
  const int kcl = kc0 + r16;                     // this lane's key
  bf16x8 kf[AD_MAXHD / 32];
#pragma unroll
  for (int s = 0; s < AD_MAXHD / 32; ++s) {
    bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    const int d0 = 32 * s + 8 * g;
    kf[s] = (s < ksteps && d0 < hd && kcl < c_end && kcl < past)
                ? *reinterpret_cast<const bf16x8*>(kc + (int64_t)kcl * hd + d0) : z;
  }
  const int dpn = half;                          // d pairs
  const int dp = t % dpn, kg = t / dpn;          // hd=128: 64 pairs x 4 key groups
  const int ngroups = AD_THREADS / dpn;          // >= 4
  const int kpg = AD_CH / ngroups;               // keys per group: 16 for hd=128, fewer below
  uint32_t vv[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int c = c_begin + kg * kpg + u;
    vv[u] = (u < kpg && c < c_end && c < past) ? *reinterpret_cast<const uint32_t*>(vc + (int64_t)c * hd + 2 * dp) : 0u;
  }

  // ---- phase 0: RoPE on q (always) and on the new k rows (if needed); append to the cache ----
  if (p.vec16) {
    // 8 elements (16 bytes) per thread: with hd = 128 the 16 rows x 16 chunks are one item per thread,
    // and all of an item's loads (cos, sin, q, its rotation partner, k, partner, v) are independent
    const int cpr = hd >> 3;
    for (int i = t; i < AD_MAXM * cpr; i += AD_THREADS) {
      const int m = i / cpr, d0 = (i - m * cpr) << 3;
      uint4 qo = make_uint4(0, 0, 0, 0);
      if (m < M) {
        const int64_t row = (int64_t)b * p.rows_per_req + p.row_off + m;
        int64_t ps = p.pos ? p.pos[row] : (int64_t)past + p.pos_off + (int64_t)m * p.pos_stride;
        ps = ps < 0 ? 0 : (ps >= p.n_pos ? p.n_pos - 1 : ps);
        const bool lo = d0 < half;
        const int dp0 = lo ? d0 + half : d0 - half;
        const bf16_t* qrow = p.qkv + row * p.ld_qkv + h * hd;
        const bool kv_too = need_new || writer;
        const bf16_t* krow = p.qkv + row * p.ld_qkv + (p.H + hk) * hd;
        const bf16_t* vrow = p.qkv + row * p.ld_qkv + (p.H + p.HK + hk) * hd;
        const uint4 cv = *reinterpret_cast<const uint4*>(p.cos_t + ps * hd + d0);
        const uint4 sv = *reinterpret_cast<const uint4*>(p.sin_t + ps * hd + d0);
        const uint4 qv = *reinterpret_cast<const uint4*>(qrow + d0);
        const uint4 qp = *reinterpret_cast<const uint4*>(qrow + dp0);
        uint4 kv = qo, kp = qo, vv4 = qo;
        if (kv_too) {
          kv = *reinterpret_cast<const uint4*>(krow + d0);
          kp = *reinterpret_cast<const uint4*>(krow + dp0);
          vv4 = *reinterpret_cast<const uint4*>(vrow + d0);
        }
        const uint32_t* cw = reinterpret_cast<const uint32_t*>(&cv);
        const uint32_t* sw = reinterpret_cast<const uint32_t*>(&sv);
        auto rot = [&](const uint4& a, const uint4& pr) {
          const uint32_t* aw = reinterpret_cast<const uint32_t*>(&a);
          const uint32_t* pw = reinterpret_cast<const uint32_t*>(&pr);
          uint4 o;
          uint32_t* ow = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float r0 = lo_bf(pw[j]), r1 = hi_bf(pw[j]);
            if (lo) { r0 = -r0; r1 = -r1; }
            const float y0 = rbf(lo_bf(aw[j]) * lo_bf(cw[j])) + rbf(r0 * lo_bf(sw[j]));
            const float y1 = rbf(hi_bf(aw[j]) * hi_bf(cw[j])) + rbf(r1 * hi_bf(sw[j]));
            ow[j] = pack2bf(y0, y1);
          }
          return o;
        };
        qo = rot(qv, qp);
        if (kv_too) {
          const uint4 ke = rot(kv, kp);
          *reinterpret_cast<uint4*>(knew + m * AD_MAXHD + d0) = ke;
          *reinterpret_cast<uint4*>(vnew + m * AD_MAXHD + d0) = vv4;
          if (writer) {
            *reinterpret_cast<uint4*>(kc + (int64_t)(past + m) * hd + d0) = ke;
            *reinterpret_cast<uint4*>(vc + (int64_t)(past + m) * hd + d0) = vv4;
          }
        }
      }
      *reinterpret_cast<uint4*>(q_s + m * AD_MAXHD + d0) = qo;
    }
  } else
  for (int i = t; i < AD_MAXM * hd; i += AD_THREADS) {
    const int m = i / hd, d = i - m * hd;
    if (m < M) {
      const int64_t row = (int64_t)b * p.rows_per_req + p.row_off + m;
      int64_t ps = p.pos ? p.pos[row] : (int64_t)past + p.pos_off + (int64_t)m * p.pos_stride;
      ps = ps < 0 ? 0 : (ps >= p.n_pos ? p.n_pos - 1 : ps);
      const float c = bf2f(p.cos_t[ps * hd + d]), s = bf2f(p.sin_t[ps * hd + d]);
      const int dp = d < half ? d + half : d - half;
      const bf16_t* qrow = p.qkv + row * p.ld_qkv + h * hd;
      float qv = bf2f(qrow[d]), qr = bf2f(qrow[dp]);
      if (d < half) qr = -qr;
      q_s[m * AD_MAXHD + d] = f2bf(rbf(qv * c) + rbf(qr * s));
      if (need_new || writer) {
        const bf16_t* krow = p.qkv + row * p.ld_qkv + (p.H + hk) * hd;
        const bf16_t* vrow = p.qkv + row * p.ld_qkv + (p.H + p.HK + hk) * hd;
        float kv = bf2f(krow[d]), kr = bf2f(krow[dp]);
        if (d < half) kr = -kr;
        const bf16_t ke = f2bf(rbf(kv * c) + rbf(kr * s));
        const bf16_t ve = vrow[d];
        knew[m * AD_MAXHD + d] = ke;
        vnew[m * AD_MAXHD + d] = ve;
        if (writer) {
          kc[(int64_t)(past + m) * hd + d] = ke;
          vc[(int64_t)(past + m) * hd + d] = ve;
        }
      }
    } else {
      q_s[m * AD_MAXHD + d] = 0;
    }
  }
  __syncthreads();

  // ---- phase 1: scores for 16 keys per wave on MFMA; mask; fp32 into P ----
  {
    const int c0 = kc0;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c0 < c_end) {
#pragma unroll
      for (int s = 0; s < AD_MAXHD / 32; ++s) {
        if (s < ksteps) {
          bf16x8 qf = {0, 0, 0, 0, 0, 0, 0, 0};
          const int d0 = 32 * s + 8 * g;
          if (d0 < hd) {
            qf = *reinterpret_cast<const bf16x8*>(q_s + r16 * AD_MAXHD + d0);
            if (kcl < c_end && kcl >= past) kf[s] = *reinterpret_cast<const bf16x8*>(knew + (kcl - past) * AD_MAXHD + d0);
          }
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[s], qf, acc, 0, 0, 0);
        }
      }
    }
    const uint8_t* tm = p.tree_mask ? p.tree_mask + (int64_t)b * p.tm_bstride : nullptr;
    const uint8_t* pm = p.pad_mask ? p.pad_mask + (int64_t)b * p.pm_stride : nullptr;
    const int tree0 = total - p.T;
    const int m = r16;  // lane holds query m = r16, keys c0 + 4g + r
    float sc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cc = c0 + 4 * g + r;
      bool vis = (m < M) && (cc < c_end);
      if (p.causal && M > 1) vis = vis && (cc <= past + m);
      if (pm && vis) vis = pm[cc] != 0;
      if (tm && vis && cc >= tree0) vis = tm[m * p.T + (cc - tree0)] != 0;
      sc[r] = vis ? rbf(rbf(acc[r]) / p.sqrt_hd) : -INFINITY;
    }
    if (m < MM) *reinterpret_cast<float4*>(&P[m * AD_CH + wave * 16 + 4 * g]) = make_float4(sc[0], sc[1], sc[2], sc[3]);
  }
  __syncthreads();

  // ---- phase 2: chunk-local softmax numerators: 16 threads per row, 4 keys each ----
  {
    const int m = t >> 4, j = t & 15;
    float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (m < MM) v = *reinterpret_cast<const float4*>(&P[m * AD_CH + 4 * j]);
    float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
    float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mx > -INFINITY) {
      e.x = __expf(v.x - mx); e.y = __expf(v.y - mx); e.z = __expf(v.z - mx); e.w = __expf(v.w - mx);
    }
    float sum = e.x + e.y + e.z + e.w;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
    if (m < MM) *reinterpret_cast<float4*>(&P[m * AD_CH + 4 * j]) = e;
    if (j == 0) { rowm[m] = mx; rowl[m] = sum; }
  }
  __syncthreads();

  // ---- phase 3: O~ = P.V ; thread = (2 adjacent d) x (key group) ; V pairs already in registers ----
  {
    float a0[MM], a1[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) a0[m] = a1[m] = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {  // rows appended by this call come from LDS
      const int c = c_begin + kg * kpg + u;
      if (u < kpg && c < c_end && c >= past) vv[u] = *reinterpret_cast<const uint32_t*>(vnew + (c - past) * AD_MAXHD + 2 * dp);
    }
    if ((kpg & 3) == 0) {
#pragma unroll
      for (int m = 0; m < MM; ++m) {
        if (m < M) {
#pragma unroll
          for (int u4 = 0; u4 < 16; u4 += 4) {
            if (u4 < kpg) {
              const float4 pr = *reinterpret_cast<const float4*>(&P[m * AD_CH + kg * kpg + u4]);
              a0[m] += pr.x * lo_bf(vv[u4]) + pr.y * lo_bf(vv[u4 + 1]) + pr.z * lo_bf(vv[u4 + 2]) + pr.w * lo_bf(vv[u4 + 3]);
              a1[m] += pr.x * hi_bf(vv[u4]) + pr.y * hi_bf(vv[u4 + 1]) + pr.z * hi_bf(vv[u4 + 2]) + pr.w * hi_bf(vv[u4 + 3]);
            }
          }
        }
      }
    } else {  // small head_dim (tests): key groups narrower than a float4
      for (int m = 0; m < M; ++m) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (u < kpg) {
            const float pr = P[m * AD_CH + kg * kpg + u];
            a0[m] += pr * lo_bf(vv[u]);
            a1[m] += pr * hi_bf(vv[u]);
          }
        }
      }
    }
    // reduce the key groups through LDS (groups beyond 4 only exist for hd < 128: fold them first)
    const int slot = kg & 3;
    for (int pass = 0; pass < (ngroups + 3) / 4; ++pass) {
      if ((kg >> 2) == pass) {
#pragma unroll
        for (int m = 0; m < MM; ++m) {
          if (m < M) {
            float* r = &red[slot][m][2 * dp];
            if (pass == 0) { r[0] = a0[m]; r[1] = a1[m]; } else { r[0] += a0[m]; r[1] += a1[m]; }
          }
        }
      }
      __syncthreads();
    }
    const int ng4 = ngroups < 4 ? ngroups : 4;
    float* rec = p.ws + ((int64_t)(b * p.H + h) * p.max_splits + split) * AD_REC;
    for (int i = t; i < M * hd; i += AD_THREADS) {
      const int m = i / hd, d = i - m * hd;
      float s = 0.f;
      for (int q = 0; q < ng4; ++q) s += red[q][m][d];
      rec[m * AD_MAXHD + d] = s;
    }
    if (t < M) { rec[AD_MAXM * AD_MAXHD + 2 * t] = rowm[t]; rec[AD_MAXM * AD_MAXHD + 2 * t + 1] = rowl[t]; }
  }
}

// grid (H, B, ceil(M/2)); 256 threads = 2 rows x 128 d.  All partial loads of a row are issued
// before they are used (two independent passes over <= max_splits records).
#define AD_CMAX 8
__global__ __launch_bounds__(2 * AD_MAXHD) void attn_combine_kernel(AttnDecArgs p) {
  const int h = blockIdx.x, b = blockIdx.y;
  const int d = threadIdx.x & (AD_MAXHD - 1), m = blockIdx.z * 2 + (threadIdx.x >> 7);
  const int M = p.M, hd = p.hd;
  const int past = p.past_len[b] + p.past_off;
  const int total = past + M;
  const int ns = (total + p.split_keys - 1) / p.split_keys;
  if (past < 0 || total > p.cap || ns > p.max_splits || m >= M) return;
  const float* base = p.ws + (int64_t)(b * p.H + h) * p.max_splits * AD_REC;
  float mx = -INFINITY, l = 0.f, o = 0.f;
  for (int s0 = 0; s0 < ns; s0 += AD_CMAX) {
    float ms[AD_CMAX], ls[AD_CMAX], os[AD_CMAX];
#pragma unroll
    for (int u = 0; u < AD_CMAX; ++u) {
      const int s = s0 + u;
      ms[u] = -INFINITY; ls[u] = 0.f; os[u] = 0.f;
      if (s < ns) {
        const float* rec = base + (int64_t)s * AD_REC;
        ms[u] = rec[AD_MAXM * AD_MAXHD + 2 * m];
        ls[u] = rec[AD_MAXM * AD_MAXHD + 2 * m + 1];
        if (d < hd) os[u] = rec[m * AD_MAXHD + d];
      }
    }
    float cm = mx;
#pragma unroll
    for (int u = 0; u < AD_CMAX; ++u) cm = fmaxf(cm, ms[u]);
    if (cm > -INFINITY) {
      const float sc = (mx > -INFINITY) ? __expf(mx - cm) : 0.f;
      l *= sc; o *= sc; mx = cm;
#pragma unroll
      for (int u = 0; u < AD_CMAX; ++u) {
        const float w = (ms[u] > -INFINITY) ? __expf(ms[u] - cm) : 0.f;
        l += w * ls[u]; o += w * os[u];
      }
    }
  }
  if (d < hd)
    p.out[((int64_t)b * p.rows_per_req + p.row_off + m) * p.ld_out + h * hd + d] = f2bf(l > 0.f ? o / l : 0.f);
}

// ---------------------------------------------------------------------------------------------
// Wide batches (BASELINE configs[3]: 32 ragged requests x 32 heads): the 64-key workgroups above are chains of
// memory latencies with P.V on the VALU (PMC: 61 % VALU-busy, waves waiting half their cycles) and every (b, h, chunk)
// repeats the RoPE of its queries.  Here RoPE + append is one small pre-pass, and a workgroup takes up to 512 keys
// of one (request, head): each of its 4 waves walks its own 32-key tiles with a private online softmax --
//   K rows   HBM/L2 -> VGPR as MFMA A fragments (keys x queries, v_mfma_f32_16x16x32_bf16), one tile ahead,
//   V rows   HBM/L2 -> LDS by LDS-DMA into the wave's own double buffer (row-major, 16-byte chunks XOR-swizzled),
//            read back TRANSPOSED (ds_read_b64_tr_b16) as the A operand of O^T += V^T . P^T on MFMA,
// no workgroup barrier inside the loop; the four partial states merge once through LDS into one partial record per
// 512 keys (an eighth of the records the combine pass has to read).
// Grouped-query attention: a tree step has 1 or 4 query rows per request, so the 16 MFMA columns of a workgroup carry the
// queries of `gq` heads that share one KV head (column = head-in-group * (16 / gq) + query): the group's K and V rows are
// read once instead of once per query head, and a quarter of the workgroups do the same MFMA work.
#define AW_TK 32
#define AW_SPLIT 512
typedef __attribute__((ext_vector_type(4))) short aw_s4;
__device__ __forceinline__ int aw_off(int row, int ch) {   // byte offset of 16-byte chunk ch of row (0..31) in a V tile image
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// grid (ceil(rows*(H+2HK)*hd/8 / 256)): RoPE of q -> q_roped, RoPE of k and v -> cache rows [past, past+M)
__global__ __launch_bounds__(256) void attn_rope_append_kernel(AttnDecArgs p) {
  const int hd = p.hd, cpr = hd >> 3, nh = p.H + 2 * p.HK, half = hd >> 1;
  const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per_req = (int64_t)p.M * nh * cpr;
  if (item >= per_req * p.B) return;
  const int b = (int)(item / per_req);
  int rem = (int)(item - (int64_t)b * per_req);
  const int m = rem / (nh * cpr); rem -= m * nh * cpr;
  const int hh = rem / cpr, d0 = (rem - hh * cpr) << 3;
  const int past = p.past_len[b] + p.past_off;
  if (past < 0 || past + p.M > p.cap || (past + p.M + p.split_keys - 1) / p.split_keys > p.max_splits) {
    if (p.err_flag) atomicExch(p.err_flag, 1);
    return;
  }
  const int64_t row = (int64_t)b * p.rows_per_req + p.row_off + m;
  const bf16_t* src = p.qkv + row * p.ld_qkv + (int64_t)hh * hd;
  const uint4 xv = *reinterpret_cast<const uint4*>(src + d0);
  uint4 o = xv;
  if (hh < p.H + p.HK) {                     // q and k heads: x*cos + rotate_half(x)*sin with the reference's bf16 roundings
    int64_t ps = p.pos ? p.pos[row] : (int64_t)past + p.pos_off + (int64_t)m * p.pos_stride;
    ps = ps < 0 ? 0 : (ps >= p.n_pos ? p.n_pos - 1 : ps);
    const bool lo = d0 < half;
    const uint4 xp = *reinterpret_cast<const uint4*>(src + (lo ? d0 + half : d0 - half));
    const uint4 cv = *reinterpret_cast<const uint4*>(p.cos_t + ps * hd + d0);
    const uint4 sv = *reinterpret_cast<const uint4*>(p.sin_t + ps * hd + d0);
    const uint32_t* aw = reinterpret_cast<const uint32_t*>(&xv);
    const uint32_t* pw = reinterpret_cast<const uint32_t*>(&xp);
    const uint32_t* cw = reinterpret_cast<const uint32_t*>(&cv);
    const uint32_t* sw = reinterpret_cast<const uint32_t*>(&sv);
    uint32_t* ow = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float r0 = lo_bf(pw[j]), r1 = hi_bf(pw[j]);
      if (lo) { r0 = -r0; r1 = -r1; }
      ow[j] = pack2bf(rbf(lo_bf(aw[j]) * lo_bf(cw[j])) + rbf(r0 * lo_bf(sw[j])), rbf(hi_bf(aw[j]) * hi_bf(cw[j])) + rbf(r1 * hi_bf(sw[j])));
    }
  }
  if (hh < p.H) {
    *reinterpret_cast<uint4*>(p.q_roped + ((int64_t)b * AD_MAXM + m) * p.H * hd + (int64_t)hh * hd + d0) = o;
  } else {
    const bool isv = hh >= p.H + p.HK;
    const int hk = hh - p.H - (isv ? p.HK : 0);
    bf16_t* c = (isv ? p.vcache : p.kcache) + (int64_t)b * p.c_bstride + (int64_t)hk * p.c_hstride;
    *reinterpret_cast<uint4*>(c + (int64_t)(past + m) * hd + d0) = o;
  }
}

// grid (max_splits, H / gq, B); hd = 128.  Reads q from q_roped and every key / value row (this call's included) from the cache.
__global__ __launch_bounds__(AD_THREADS, 2) void attn_wide_kernel(AttnDecArgs p) {
  // ONE shared object (a second one makes hipcc wait vmcnt(0) for the pending LDS-DMA before every ds_read):
  // per wave two V tiles (reused for the merge), then the waves' running maxima and sums
  __shared__ __attribute__((aligned(16))) char lds_all[4 * 2 * AW_TK * 256 + 2 * 4 * AD_MAXM * 4];
  char (*lds)[2][AW_TK * 256] = reinterpret_cast<char (*)[2][AW_TK * 256]>(lds_all);
  float (*wm)[AD_MAXM] = reinterpret_cast<float (*)[AD_MAXM]>(lds_all + 4 * 2 * AW_TK * 256);
  float (*wl)[AD_MAXM] = wm + 4;
  const int split = blockIdx.x, h0 = blockIdx.y * p.gq, b = blockIdx.z;
  const int M = p.M, hd = 128;
  const int rep = p.H / p.HK, hk = h0 / rep;             // gq divides rep: the group's heads share this KV head
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r16 = lane & 15, g = lane >> 4;
  const int mq = 16 / p.gq;                              // MFMA columns per head
  const int past = p.past_len[b] + p.past_off;
  const int total = past + M;
  if (past < 0 || total > p.cap || (total + AW_SPLIT - 1) / AW_SPLIT > p.max_splits) return;   // flagged by the pre-pass
  const int c_begin = split * AW_SPLIT;
  if (c_begin >= total) return;
  const int c_end = min(c_begin + AW_SPLIT, total);
  const int n_wt = (c_end - c_begin + AW_TK - 1) / AW_TK;
  const bf16_t* kc = p.kcache + (int64_t)b * p.c_bstride + (int64_t)hk * p.c_hstride;
  const bf16_t* vc = p.vcache + (int64_t)b * p.c_bstride + (int64_t)hk * p.c_hstride;
  const uint8_t* tm = p.tree_mask ? p.tree_mask + (int64_t)b * p.tm_bstride : nullptr;
  const uint8_t* pm = p.pad_mask ? p.pad_mask + (int64_t)b * p.pm_stride : nullptr;
  const int tree0 = total - p.T;
  const int m = r16 % mq, h = h0 + r16 / mq;  // this lane's MFMA column: query m of head h

  bf16x8 qf[4];
  {
    const bf16_t* qrow = p.q_roped + ((int64_t)b * AD_MAXM + (m < M ? m : 0)) * p.H * hd + (int64_t)h * hd;
#pragma unroll
    for (int s = 0; s < 4; ++s) {                      // rows >= M: masked below.  Hidden from hipcc's wait counting like the
      const bf16_t* src = qrow + 32 * s + 8 * g;        // K loads: a pending plain load would make it drain the DMA at every QK^T
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qf[s]) : "v"(src) : "memory");
    }
    // no wait here: the first tile's loads go out behind these, and vector-memory operations complete in order, so the
    // counted wait for that tile (AW_WAIT_TILE, ahead of the first QK^T) covers the q fragments too
  }
  f32x4 o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mrun = -INFINITY, lrun = 0.f;
  const int tq = (lane & 15) >> 2, tp = lane & 3;

  // V tile -> this wave's LDS buffer (8 wave-instructions of 1 KB = 4 rows each), then the K fragments of the tile
  auto issue = [&](bf16x8 (&kf)[2][4], int buf, int wt) {
    const int k0 = c_begin + wt * AW_TK;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int trow = 4 * i + (lane >> 4), slot = lane & 15;
      const int ch = slot ^ (((trow & 3) << 2) | ((trow >> 2) & 3));
      const int row = min(k0 + trow, total - 1);                       // never predicated: rows past the end are masked
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vc + (int64_t)row * hd + ch * 8),
                                       (__attribute__((address_space(3))) void*)&lds[wave][buf][i * 1024], 16, 0, 0);
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int row = min(k0 + 16 * kt + r16, total - 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        // hidden from hipcc's wait counting: beside the LDS-DMA above it would wait vmcnt(0) for a plain load before the first
        // use (draining the tile just issued: no prefetch at all); AW_WAIT_TILE names these registers at the counted wait
        const bf16_t* src = kc + (int64_t)row * hd + 32 * s + 8 * g;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(kf[kt][s]) : "v"(src) : "memory");
      }
    }
  };
  // counted wait for a tile's loads (issued before the N youngest vector-memory operations); every K register is an in/out
  // operand, so no MFMA that reads them is scheduled above the wait
#define AW_WAIT_TILE(N, KF)                                                                                          \
  asm volatile("s_waitcnt vmcnt(" #N ")"                                                                             \
               : "+v"(KF[0][0]), "+v"(KF[0][1]), "+v"(KF[0][2]), "+v"(KF[0][3]), "+v"(KF[1][0]), "+v"(KF[1][1]),     \
                 "+v"(KF[1][2]), "+v"(KF[1][3])                                                                      \
               :: "memory")
  // A tile is computed in two parts so that the loads of the tile TWO ahead can be issued in between: part 1 consumes the
  // tile's K registers (QK^T) and its V buffer (the transposing reads), which are then free for the refill; part 2 (softmax,
  // P.V) runs with two tiles in flight.
  f32x4 sc[2];
  aw_s4 vlo[8], vhi[8];
  auto part1 = [&](const bf16x8 (&kf)[2][4], int buf) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      sc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][s], qf[s], sc[kt], 0, 0, 0);
    }
    // V^T fragments (the tile has landed: the caller waited); requested ahead of the softmax arithmetic
    // The transposing reads are inline asm: hipcc cannot tell a ds_read from the LDS bytes a pending LDS-DMA will write and
    // waits vmcnt(0) before the first one -- which drains the NEXT tile's loads, i.e. the whole prefetch.  This tile's DMA
    // has landed (AW_WAIT_TILE in the caller); the reads are waited for by the lgkmcnt statement below, which names them.
    const uint32_t Vt = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)&lds[wave][buf][0]);
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      const int r0 = 4 * g + tq;
      const uint32_t a0 = Vt + aw_off(r0, 2 * dt + (tp >> 1)) + 8 * (tp & 1);
      const uint32_t a1 = Vt + aw_off(r0 + 16, 2 * dt + (tp >> 1)) + 8 * (tp & 1);
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vlo[dt]) : "v"(a0) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vhi[dt]) : "v"(a1) : "memory");
    }
    // the reads are done (named: nothing that uses them moves above) and so are the QK^T reads of the K registers: the
    // caller may refill both
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(vlo[0]), "+v"(vlo[1]), "+v"(vlo[2]), "+v"(vlo[3]), "+v"(vlo[4]), "+v"(vlo[5]), "+v"(vlo[6]), "+v"(vlo[7]),
                   "+v"(vhi[0]), "+v"(vhi[1]), "+v"(vhi[2]), "+v"(vhi[3]), "+v"(vhi[4]), "+v"(vhi[5]), "+v"(vhi[6]), "+v"(vhi[7]),
                   "+v"(sc[0]), "+v"(sc[1])
                 :: "memory");
  };
  auto part2 = [&](int wt) {
    const int k0 = c_begin + wt * AW_TK;
    float x[8], mx = mrun;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cc = k0 + 16 * kt + 4 * g + r;
        bool vis = (m < M) && (cc < c_end);
        if (p.causal && M > 1) vis = vis && (cc <= past + m);
        if (pm && vis) vis = pm[cc] != 0;
        if (tm && vis && cc >= tree0) vis = tm[m * p.T + (cc - tree0)] != 0;
        x[4 * kt + r] = vis ? rbf(rbf(sc[kt][r]) / p.sqrt_hd) : -INFINITY;     // the reference's bf16 roundings of the scores
        mx = fmaxf(mx, x[4 * kt + r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float ps = 0.f;
    f32x4 p0, p1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p0[r] = (mx > -INFINITY) ? __expf(x[r] - mx) : 0.f;
      p1[r] = (mx > -INFINITY) ? __expf(x[4 + r] - mx) : 0.f;
      ps += p0[r] + p1[r];
    }
    if (__builtin_amdgcn_ballot_w64(mx != mrun)) {                    // wave-uniform: some query's maximum moved
      const float alpha = (mrun > -INFINITY) ? __expf(mrun - mx) : 0.f;
      lrun *= alpha;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) o[dt] *= alpha;
      mrun = mx;
    }
    lrun += ps;
    bf16x8 vfr[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      union { struct { aw_s4 a, b; } s; bf16x8 v; } u;
      u.s.a = vlo[dt]; u.s.b = vhi[dt];
      vfr[dt] = u.v;
    }
    bf16x8 pf;                                                         // k-slot (g, j) <-> key 16*(j>>2) + 4g + (j&3)
    pf[0] = (__bf16)p0[0]; pf[1] = (__bf16)p0[1]; pf[2] = (__bf16)p0[2]; pf[3] = (__bf16)p0[3];
    pf[4] = (__bf16)p1[0]; pf[5] = (__bf16)p1[1]; pf[6] = (__bf16)p1[2]; pf[7] = (__bf16)p1[3];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[dt], pf, o[dt], 0, 0, 0);
  };

  bf16x8 kfa[2][4], kfb[2][4];
  int wt = wave;                                                       // this wave's tiles: wt, wt + 4, ...  (all conditions wave-uniform)
  if (wt < n_wt) {
    issue(kfa, 0, wt);
    if (wt + 4 < n_wt) issue(kfb, 1, wt + 4);
    while (true) {
      // tile wt sits in set A; the only younger operations are those of tile wt + 4 (16, if it exists)
      if (wt + 4 < n_wt) AW_WAIT_TILE(16, kfa); else AW_WAIT_TILE(0, kfa);
      part1(kfa, 0);
      if (wt + 8 < n_wt) issue(kfa, 0, wt + 8);                         // set A is free again: two tiles in flight during part 2
      part2(wt);
      wt += 4;
      if (wt >= n_wt) break;
      if (wt + 4 < n_wt) AW_WAIT_TILE(16, kfb); else AW_WAIT_TILE(0, kfb);
      part1(kfb, 1);
      if (wt + 8 < n_wt) issue(kfb, 1, wt + 8);
      part2(wt);
      wt += 4;
      if (wt >= n_wt) break;
    }
  } else {
    // a wave without tiles never reaches a counted wait: its q loads must land before their registers are reused
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]) :: "memory");
  }
  // ---- merge the four waves' (m, l, O~) through LDS; one partial record per workgroup ----
  lrun += __shfl_xor(lrun, 16, 64);
  lrun += __shfl_xor(lrun, 32, 64);
  __syncthreads();                                                    // every wave is done with its V buffers
  float* ow = reinterpret_cast<float*>(&lds[wave][0][0]);             // [16 columns][128 d] fp32 = 8 KB
#pragma unroll
  for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f32x4*>(ow + r16 * 128 + 16 * dt + 4 * g) = o[dt];
  if (g == 0) { wm[wave][r16] = mrun; wl[wave][r16] = lrun; }
  __syncthreads();
  for (int i = t; i < 16 * 128; i += AD_THREADS) {
    const int c = i >> 7, d = i & 127;                                // column c = query mm of head h0 + c / mq
    const int mm = c % mq;
    if (mm >= M) continue;
    float mx = fmaxf(fmaxf(wm[0][c], wm[1][c]), fmaxf(wm[2][c], wm[3][c]));
    float acc = 0.f, l = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float sc = (wm[w][c] > -INFINITY) ? __expf(wm[w][c] - mx) : 0.f;
      acc += sc * reinterpret_cast<const float*>(&lds[w][0][0])[c * 128 + d];
      l += sc * wl[w][c];
    }
    float* rec = p.ws + ((int64_t)(b * p.H + h0 + c / mq) * p.max_splits + split) * AD_REC;
    rec[mm * AD_MAXHD + d] = acc;
    if (d == 0) { rec[AD_MAXM * AD_MAXHD + 2 * mm] = mx; rec[AD_MAXM * AD_MAXHD + 2 * mm + 1] = l; }
  }
}

static int g_attn_gq_off = 0;    // diagnostics (tests): 1 = one query head per workgroup of the wide-batch kernel
extern "C" void eagle_debug_set_attn(int no_head_packing) { g_attn_gq_off = no_head_packing; }

extern "C" int64_t eagle_attn_decode_workspace_bytes(int B, int H, int max_kv) {
  return (int64_t)B * H * ((max_kv + AD_CH - 1) / AD_CH) * AD_REC * sizeof(float)      // partial records (64-key form: the larger)
         + (int64_t)B * AD_MAXM * H * AD_MAXHD * sizeof(bf16_t);                        // roped queries of the wide-batch path
}

// See include/eagle_hip.h for the contract.
extern "C" int eagle_tree_attention_bf16(const void* qkv, int64_t ld_qkv, const int64_t* position_ids, int pos_off,
                                         int pos_stride, const void* cos_table, const void* sin_table, int n_pos,
                                         void* kcache, void* vcache, int64_t cache_batch_stride, int cap,
                                         const int32_t* past_len, int past_off,
                                         const uint8_t* tree_mask, int T, int64_t tree_mask_batch_stride,
                                         const uint8_t* pad_mask, int64_t pad_mask_stride,
                                         void* out, int64_t ld_out, void* workspace, int* err_flag,
                                         int B, int M, int H, int HK, int hd, int max_kv, int causal,
                                         int rows_per_req, int row_off, hipStream_t stream) {
  if (!qkv || !cos_table || !sin_table || !kcache || !vcache || !past_len || !out || !workspace) return EAGLE_ERR_ARG;
  if (rows_per_req < M || row_off < 0 || row_off + M > rows_per_req) return EAGLE_ERR_SHAPE;
  if (B < 1 || M < 1 || M > AD_MAXM || H < 1 || HK < 1 || (H % HK) || hd < 8 || hd > AD_MAXHD || (hd & 7)) return EAGLE_ERR_SHAPE;
  if (AD_THREADS % (hd / 2)) return EAGLE_ERR_SHAPE;
  if (max_kv < M || max_kv > cap || T < 0 || T > max_kv || (tree_mask && T < 1)) return EAGLE_ERR_SHAPE;
  if ((ld_qkv & 7) || (ld_out & 1)) return EAGLE_ERR_SHAPE;
  AttnDecArgs a;
  a.qkv = (const bf16_t*)qkv; a.ld_qkv = ld_qkv; a.pos = position_ids; a.pos_off = pos_off; a.pos_stride = pos_stride;
  a.cos_t = (const bf16_t*)cos_table; a.sin_t = (const bf16_t*)sin_table; a.n_pos = n_pos;
  a.kcache = (bf16_t*)kcache; a.vcache = (bf16_t*)vcache; a.c_bstride = cache_batch_stride; a.c_hstride = (int64_t)cap * hd;
  a.past_len = past_len; a.past_off = past_off;
  a.tree_mask = tree_mask; a.T = tree_mask ? T : 0; a.tm_bstride = tree_mask_batch_stride;
  a.pad_mask = pad_mask; a.pm_stride = pad_mask_stride;
  a.out = (bf16_t*)out; a.ld_out = ld_out; a.ws = (float*)workspace; a.max_splits = (max_kv + AD_CH - 1) / AD_CH;
  a.err_flag = err_flag;
  a.B = B; a.M = M; a.H = H; a.HK = HK; a.hd = hd; a.cap = cap; a.causal = causal;
  a.rows_per_req = rows_per_req; a.row_off = row_off;
  a.sqrt_hd = sqrtf((float)hd);
  a.vec16 = ((hd & 15) == 0) && ((cache_batch_stride & 7) == 0) &&
            (((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(cos_table) | reinterpret_cast<uintptr_t>(sin_table) |
               reinterpret_cast<uintptr_t>(kcache) | reinterpret_cast<uintptr_t>(vcache)) & 15) == 0);
  a.split_keys = AD_CH;
  a.q_roped = nullptr;
  a.gq = 1;
  // wide batches: enough (request, head) pairs to fill the chip without 64-key workgroups
  if (hd == 128 && a.vec16 && (int64_t)B * H >= 256) {
    a.split_keys = AW_SPLIT;
    a.max_splits = (max_kv + AW_SPLIT - 1) / AW_SPLIT;
    a.q_roped = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(workspace) +
                                          (int64_t)B * H * ((max_kv + AD_CH - 1) / AD_CH) * AD_REC * sizeof(float));
    const int64_t items = (int64_t)B * M * (H + 2 * HK) * (hd / 8);
    hipLaunchKernelGGL(attn_rope_append_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, a);
    int rc = eagle_check_launch();
    if (rc) return rc;
    const int grp = H / HK;                        // query heads per KV head
    a.gq = (M <= 4 && grp % 4 == 0) ? 4 : (M <= 8 && grp % 2 == 0) ? 2 : 1;
    if (g_attn_gq_off) a.gq = 1;
    hipLaunchKernelGGL(attn_wide_kernel, dim3(a.max_splits, H / a.gq, B), dim3(AD_THREADS), 0, stream, a);
    rc = eagle_check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(attn_combine_kernel, dim3(H, B, (M + 1) / 2), dim3(2 * AD_MAXHD), 0, stream, a);
    return eagle_check_launch();
  }
  const dim3 grid(a.max_splits, H, B), block(AD_THREADS);
  if (hd != 128) hipLaunchKernelGGL((attn_split_kernel<AD_MAXM, 0>), grid, block, 0, stream, a);
  else if (M <= 4) hipLaunchKernelGGL((attn_split_kernel<4, 128>), grid, block, 0, stream, a);
  else if (M <= 8) hipLaunchKernelGGL((attn_split_kernel<8, 128>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((attn_split_kernel<AD_MAXM, 128>), grid, block, 0, stream, a);
  int rc = eagle_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(attn_combine_kernel, dim3(H, B, (M + 1) / 2), dim3(2 * AD_MAXHD), 0, stream, a);
  return eagle_check_launch();
}
