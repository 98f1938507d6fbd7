// Causal self-attention for draft-head TRAINING (no KV cache, no padding mask): flash-style
// forward and backward on MFMA, plus the RoPE passes (gfx950).
//
// Replaces, for the training forward of eagle/train.py:105, the attention body of
// eagle/llama2.py:226-273 with the pure causal mask of _make_causal_mask (:14-28) -- training
// passes no attention_mask, so padded batches run causal-only -- and its autograd
// backward.  The [B,32,L,L] score tensor (537 MB fp32 at L=2048) is never materialised.
//
// Layout: qkv [B*L, 3*H*hd] bf16 (q | k | v sections, RoPE already applied in place), heads of
// hd = 128.  MFMA orientation is "keys x queries" for the query-stationary kernels (forward, dQ)
// and "queries x keys" for the key-stationary one (dK/dV): in both the 16x16 score accumulator
// of a lane holds 4 consecutive rows of ONE column, so two stacked 16-row tiles form, lane-locally,
// the 8-element B fragment of the following P.V / dS.K product (k-slot (g,j) <-> row
// 16*(j>>2) + 4g + (j&3)); the other operand of that product is an LDS tile stored transposed
// with the same slot order.
#include "common.h"

#define AT_THREADS 256
#define AT_HD 128
#define AT_QB 64      // rows per workgroup (16 per wave)
#define AT_KB 32      // streamed rows per step
#define AT_LDR (AT_HD + 8)   // row-major tile row (elements): 272 B, conflict-free b128 fragment reads
#define AT_LDT (AT_KB + 8)   // transposed tile row (elements): 80 B
#define LOG2E 1.4426950408889634f

__device__ __forceinline__ int at_slot(int r) {  // row 0..31 of a streamed tile -> k-slot of the transposed image
  const int t = r >> 4, w = r & 15;
  return (w >> 2) * 8 + (w & 3) + 4 * t;
}

// stage a [32][128] bf16 tile (rows r0.., stride ld) row-major and/or transposed into LDS
__device__ __forceinline__ void at_load2(const bf16_t* base, int64_t ld, int r0, int nrows, int tid, uint4* v) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * AT_THREADS + tid;      // 16-byte chunk id: row = c/16, col8 = c%16
    const int row = c >> 4, col = (c & 15) * 8;
    v[i] = (r0 + row < nrows) ? *reinterpret_cast<const uint4*>(base + (int64_t)(r0 + row) * ld + col) : make_uint4(0, 0, 0, 0);
  }
}
__device__ __forceinline__ void at_store_rm(bf16_t* tile, int tid, const uint4* v) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * AT_THREADS + tid;
    *reinterpret_cast<uint4*>(tile + (c >> 4) * AT_LDR + (c & 15) * 8) = v[i];
  }
}
__device__ __forceinline__ void at_store_tr(bf16_t* tile, int tid, const uint4* v) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * AT_THREADS + tid;
    const int slot = at_slot(c >> 4), col = (c & 15) * 8;
    const uint32_t* u = reinterpret_cast<const uint32_t*>(&v[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tile[(col + 2 * j) * AT_LDT + slot] = (bf16_t)(u[j] & 0xffff);
      tile[(col + 2 * j + 1) * AT_LDT + slot] = (bf16_t)(u[j] >> 16);
    }
  }
}
__device__ __forceinline__ bf16x8 pack_frag(const f32x4& a, const f32x4& b) {
  bf16x8 r;
  r[0] = (__bf16)a[0]; r[1] = (__bf16)a[1]; r[2] = (__bf16)a[2]; r[3] = (__bf16)a[3];
  r[4] = (__bf16)b[0]; r[5] = (__bf16)b[1]; r[6] = (__bf16)b[2]; r[7] = (__bf16)b[3];
  return r;
}

struct AttnTrainArgs {
  const bf16_t* qkv; int64_t ld;       // [B*L, 3*D]
  bf16_t* out; int64_t ldo;            // fwd: O [B*L, D]
  float* lse;                          // [B,H,L] (log2 units of the scaled scores)
  const bf16_t* dout; int64_t lddo;    // bwd: dO [B*L, D]
  const float* delta;                  // [B,H,L] rowsum(dO*O)
  bf16_t* dqkv; int64_t lddq;          // [B*L, 3*D]
  int B, L, H;
  float scale2;                        // log2(e)/sqrt(hd)
  float scale;                         // 1/sqrt(hd)
};

// ------------------------------------------------------------------ forward
__global__ __launch_bounds__(AT_THREADS) void attn_train_fwd_kernel(AttnTrainArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[AT_KB * AT_LDR];
  __shared__ __attribute__((aligned(16))) bf16_t Vt[AT_HD * AT_LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const int qb = (gridDim.x - 1 - blockIdx.x), h = blockIdx.y, b = blockIdx.z;   // heaviest (last) query blocks first
  const int D = p.H * AT_HD, L = p.L;
  const bf16_t* base = p.qkv + (int64_t)b * L * p.ld + h * AT_HD;
  const bf16_t* kbase = base + D;
  const bf16_t* vbase = base + 2 * D;
  const int q0 = qb * AT_QB + wave * 16;
  const int qi = q0 + r16;                       // this lane's query (as MFMA column)
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    qf[s] = qi < L ? *reinterpret_cast<const bf16x8*>(base + (int64_t)qi * p.ld + 32 * s + 8 * g) : z;
  }
  f32x4 o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, lsum = 0.f;
  const int kend = min(L, qb * AT_QB + AT_QB);   // causal: keys < end of this query block
  uint4 kv[2], vv[2];
  at_load2(kbase, p.ld, 0, L, tid, kv);
  at_load2(vbase, p.ld, 0, L, tid, vv);
  for (int k0 = 0; k0 < kend; k0 += AT_KB) {
    __syncthreads();                 // previous tile fully consumed
    at_store_rm(Ks, tid, kv);
    at_store_tr(Vt, tid, vv);
    __syncthreads();
    if (k0 + AT_KB < kend) {         // prefetch the next tile into registers
      at_load2(kbase, p.ld, k0 + AT_KB, L, tid, kv);
      at_load2(vbase, p.ld, k0 + AT_KB, L, tid, vv);
    }
    if (k0 > q0 + 15) continue;      // wave-uniform: whole tile above this wave's diagonal
    f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 ka = *reinterpret_cast<const bf16x8*>(Ks + r16 * AT_LDR + 32 * s + 8 * g);
      const bf16x8 kb = *reinterpret_cast<const bf16x8*>(Ks + (16 + r16) * AT_LDR + 32 * s + 8 * g);
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[s], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb, qf[s], s1, 0, 0, 0);
    }
    float x[8];
    float mx = m;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ka = k0 + 4 * g + r, kb = ka + 16;
      x[r] = (ka <= qi && ka < L) ? s0[r] * p.scale2 : -INFINITY;
      x[4 + r] = (kb <= qi && kb < L) ? s1[r] * p.scale2 : -INFINITY;
      mx = fmaxf(mx, fmaxf(x[r], x[4 + r]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float alpha = (m > -INFINITY) ? exp2f(m - mx) : 0.f;
    f32x4 p0, p1;
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p0[r] = (mx > -INFINITY) ? exp2f(x[r] - mx) : 0.f;
      p1[r] = (mx > -INFINITY) ? exp2f(x[4 + r] - mx) : 0.f;
      ps += p0[r] + p1[r];
    }
    lsum = lsum * alpha + ps;
    m = mx;
    const bf16x8 pf = pack_frag(p0, p1);
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      o[dt] *= alpha;
      const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vt + (dt * 16 + r16) * AT_LDT + 8 * g);
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
    }
  }
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  if (qi < L) {
    const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
    bf16_t* orow = p.out + ((int64_t)b * L + qi) * p.ldo + h * AT_HD;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      const uint2 w = make_uint2(pack2bf(o[dt][0] * inv, o[dt][1] * inv), pack2bf(o[dt][2] * inv, o[dt][3] * inv));
      *reinterpret_cast<uint2*>(orow + dt * 16 + 4 * g) = w;
    }
    if (g == 0) p.lse[((int64_t)b * p.H + h) * L + qi] = m + log2f(lsum);
  }
}

// delta[b,h,i] = sum_d dO[i,d] * O[i,d]   (one wave per (row, head))
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ o, int64_t ldo, const bf16_t* __restrict__ dout,
                                                        int64_t lddo, float* __restrict__ delta, int B, int L, int H) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (row, head)
  if (item >= (int64_t)B * L * H) return;
  const int64_t row = item / H;
  const int h = (int)(item - row * H);
  const uint32_t a = *reinterpret_cast<const uint32_t*>(o + row * ldo + h * AT_HD + 2 * lane);
  const uint32_t d = *reinterpret_cast<const uint32_t*>(dout + row * lddo + h * AT_HD + 2 * lane);
  float s = lo_bf(a) * lo_bf(d) + hi_bf(a) * hi_bf(d);
  s = wave_sum(s);
  if (lane == 0) {
    const int64_t bb = row / L, i = row - bb * L;
    delta[(bb * H + h) * L + i] = s;
  }
}

// ------------------------------------------------------------------ backward: dQ (query stationary)
__global__ __launch_bounds__(AT_THREADS) void attn_train_dq_kernel(AttnTrainArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[AT_KB * AT_LDR];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[AT_KB * AT_LDR];
  __shared__ __attribute__((aligned(16))) bf16_t Kt[AT_HD * AT_LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const int qb = (gridDim.x - 1 - blockIdx.x), h = blockIdx.y, b = blockIdx.z;
  const int D = p.H * AT_HD, L = p.L;
  const bf16_t* base = p.qkv + (int64_t)b * L * p.ld + h * AT_HD;
  const bf16_t* kbase = base + D;
  const bf16_t* vbase = base + 2 * D;
  const int q0 = qb * AT_QB + wave * 16;
  const int qi = q0 + r16;
  bf16x8 qf[4], dof[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    qf[s] = qi < L ? *reinterpret_cast<const bf16x8*>(base + (int64_t)qi * p.ld + 32 * s + 8 * g) : z;
    dof[s] = qi < L ? *reinterpret_cast<const bf16x8*>(p.dout + ((int64_t)b * L + qi) * p.lddo + h * AT_HD + 32 * s + 8 * g) : z;
  }
  const float lse = qi < L ? p.lse[((int64_t)b * p.H + h) * L + qi] : 0.f;
  const float dl = qi < L ? p.delta[((int64_t)b * p.H + h) * L + qi] : 0.f;
  f32x4 dq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) dq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int kend = min(L, qb * AT_QB + AT_QB);
  uint4 kv[2], vv[2];
  at_load2(kbase, p.ld, 0, L, tid, kv);
  at_load2(vbase, p.ld, 0, L, tid, vv);
  for (int k0 = 0; k0 < kend; k0 += AT_KB) {
    __syncthreads();
    at_store_rm(Ks, tid, kv);
    at_store_tr(Kt, tid, kv);
    at_store_rm(Vs, tid, vv);
    __syncthreads();
    if (k0 + AT_KB < kend) {
      at_load2(kbase, p.ld, k0 + AT_KB, L, tid, kv);
      at_load2(vbase, p.ld, k0 + AT_KB, L, tid, vv);
    }
    if (k0 > q0 + 15) continue;
    f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0, d0 = s0, d1 = s0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 ka = *reinterpret_cast<const bf16x8*>(Ks + r16 * AT_LDR + 32 * s + 8 * g);
      const bf16x8 kb = *reinterpret_cast<const bf16x8*>(Ks + (16 + r16) * AT_LDR + 32 * s + 8 * g);
      const bf16x8 va = *reinterpret_cast<const bf16x8*>(Vs + r16 * AT_LDR + 32 * s + 8 * g);
      const bf16x8 vb = *reinterpret_cast<const bf16x8*>(Vs + (16 + r16) * AT_LDR + 32 * s + 8 * g);
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[s], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb, qf[s], s1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, dof[s], d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb, dof[s], d1, 0, 0, 0);
    }
    f32x4 ds0, ds1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ka = k0 + 4 * g + r, kb = ka + 16;
      const float pa = (ka <= qi && ka < L && qi < L) ? exp2f(s0[r] * p.scale2 - lse) : 0.f;
      const float pb = (kb <= qi && kb < L && qi < L) ? exp2f(s1[r] * p.scale2 - lse) : 0.f;
      ds0[r] = pa * (d0[r] - dl);
      ds1[r] = pb * (d1[r] - dl);
    }
    const bf16x8 dsf = pack_frag(ds0, ds1);
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      const bf16x8 kt = *reinterpret_cast<const bf16x8*>(Kt + (dt * 16 + r16) * AT_LDT + 8 * g);
      dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf, dq[dt], 0, 0, 0);
    }
  }
  if (qi < L) {
    bf16_t* drow = p.dqkv + ((int64_t)b * L + qi) * p.lddq + h * AT_HD;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      const uint2 w = make_uint2(pack2bf(dq[dt][0] * p.scale, dq[dt][1] * p.scale), pack2bf(dq[dt][2] * p.scale, dq[dt][3] * p.scale));
      *reinterpret_cast<uint2*>(drow + dt * 16 + 4 * g) = w;
    }
  }
}

// ------------------------------------------------------------------ backward: dK, dV (key stationary)
__global__ __launch_bounds__(AT_THREADS) void attn_train_dkv_kernel(AttnTrainArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t Qs[AT_KB * AT_LDR];
  __shared__ __attribute__((aligned(16))) bf16_t Os[AT_KB * AT_LDR];   // dO row-major
  __shared__ __attribute__((aligned(16))) bf16_t Qt[AT_HD * AT_LDT];
  __shared__ __attribute__((aligned(16))) bf16_t Ot[AT_HD * AT_LDT];   // dO transposed
  __shared__ float lse_s[AT_KB], dl_s[AT_KB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const int kb_ = blockIdx.x, h = blockIdx.y, b = blockIdx.z;           // key block 0 is the heaviest: first
  const int D = p.H * AT_HD, L = p.L;
  const bf16_t* base = p.qkv + (int64_t)b * L * p.ld + h * AT_HD;
  const bf16_t* dobase = p.dout + (int64_t)b * L * p.lddo + h * AT_HD;
  const int k0w = kb_ * AT_QB + wave * 16;
  const int ki = k0w + r16;                       // this lane's key (as MFMA column)
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    kf[s] = ki < L ? *reinterpret_cast<const bf16x8*>(base + D + (int64_t)ki * p.ld + 32 * s + 8 * g) : z;
    vf[s] = ki < L ? *reinterpret_cast<const bf16x8*>(base + 2 * D + (int64_t)ki * p.ld + 32 * s + 8 * g) : z;
  }
  f32x4 dk[8], dv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[i] = dk[i]; }
  const int qstart = (kb_ * AT_QB / AT_KB) * AT_KB;      // causal: queries >= first key of the block
  uint4 qv[2], ov[2];
  at_load2(base, p.ld, qstart, L, tid, qv);
  at_load2(dobase, p.lddo, qstart, L, tid, ov);
  for (int q0 = qstart; q0 < L; q0 += AT_KB) {
    __syncthreads();
    at_store_rm(Qs, tid, qv);
    at_store_tr(Qt, tid, qv);
    at_store_rm(Os, tid, ov);
    at_store_tr(Ot, tid, ov);
    if (tid < AT_KB) {
      const int qi = q0 + tid;
      lse_s[tid] = qi < L ? p.lse[((int64_t)b * p.H + h) * L + qi] : 0.f;
      dl_s[tid] = qi < L ? p.delta[((int64_t)b * p.H + h) * L + qi] : 0.f;
    }
    __syncthreads();
    if (q0 + AT_KB < L) {
      at_load2(base, p.ld, q0 + AT_KB, L, tid, qv);
      at_load2(dobase, p.lddo, q0 + AT_KB, L, tid, ov);
    }
    if (q0 + AT_KB - 1 < k0w) continue;            // wave-uniform: all queries of the tile precede this wave's keys
    f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0, d0 = s0, d1 = s0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 qa = *reinterpret_cast<const bf16x8*>(Qs + r16 * AT_LDR + 32 * s + 8 * g);
      const bf16x8 qb2 = *reinterpret_cast<const bf16x8*>(Qs + (16 + r16) * AT_LDR + 32 * s + 8 * g);
      const bf16x8 oa = *reinterpret_cast<const bf16x8*>(Os + r16 * AT_LDR + 32 * s + 8 * g);
      const bf16x8 ob = *reinterpret_cast<const bf16x8*>(Os + (16 + r16) * AT_LDR + 32 * s + 8 * g);
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[s], s0, 0, 0, 0);     // S[q][key]: rows = queries
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qb2, kf[s], s1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa, vf[s], d0, 0, 0, 0);     // dP[q][key] = dO . V^T
      d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ob, vf[s], d1, 0, 0, 0);
    }
    f32x4 p0, p1, ds0, ds1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ra = 4 * g + r, rb = ra + 16;     // query rows inside the tile
      const int qa = q0 + ra, qb2 = q0 + rb;
      p0[r] = (ki <= qa && qa < L && ki < L) ? exp2f(s0[r] * p.scale2 - lse_s[ra]) : 0.f;
      p1[r] = (ki <= qb2 && qb2 < L && ki < L) ? exp2f(s1[r] * p.scale2 - lse_s[rb]) : 0.f;
      ds0[r] = p0[r] * (d0[r] - dl_s[ra]);
      ds1[r] = p1[r] * (d1[r] - dl_s[rb]);
    }
    const bf16x8 pf = pack_frag(p0, p1), dsf = pack_frag(ds0, ds1);
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      const bf16x8 ot = *reinterpret_cast<const bf16x8*>(Ot + (dt * 16 + r16) * AT_LDT + 8 * g);
      const bf16x8 qt = *reinterpret_cast<const bf16x8*>(Qt + (dt * 16 + r16) * AT_LDT + 8 * g);
      dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ot, pf, dv[dt], 0, 0, 0);
      dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf, dk[dt], 0, 0, 0);
    }
  }
  if (ki < L) {
    bf16_t* drow = p.dqkv + ((int64_t)b * L + ki) * p.lddq + h * AT_HD;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      *reinterpret_cast<uint2*>(drow + D + dt * 16 + 4 * g) =
          make_uint2(pack2bf(dk[dt][0] * p.scale, dk[dt][1] * p.scale), pack2bf(dk[dt][2] * p.scale, dk[dt][3] * p.scale));
      *reinterpret_cast<uint2*>(drow + 2 * D + dt * 16 + 4 * g) =
          make_uint2(pack2bf(dv[dt][0], dv[dt][1]), pack2bf(dv[dt][2], dv[dt][3]));
    }
  }
}

// ------------------------------------------------------------------ RoPE over the q and k sections, in place
// forward: x' = bf16(bf16(x*cos) + bf16(rot(x)*sin)) (eagle/llama2.py:56-69); inverse != 0 applies
// the transposed rotation (x*cos - rot(x)*sin): the backward of the forward pass.
__global__ __launch_bounds__(256) void rope_qk_kernel(bf16_t* __restrict__ qkv, int64_t ld, const int64_t* __restrict__ pos,
                                                     const bf16_t* __restrict__ cos_t, const bf16_t* __restrict__ sin_t, int n_pos,
                                                     int64_t nrows, int L, int heads2, int hd, int inverse) {
  // one thread per (row, head-of-q-or-k, d < hd/2): handles the pair (d, d + hd/2)
  const int half = hd >> 1;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per_row = (int64_t)heads2 * half;
  if (idx >= nrows * per_row) return;
  const int64_t row = idx / per_row;
  const int rem = (int)(idx - row * per_row);
  const int hh = rem / half, d = rem - hh * half;
  int64_t ps = pos ? pos[row] : (row % L);
  ps = ps < 0 ? 0 : (ps >= n_pos ? n_pos - 1 : ps);
  bf16_t* x = qkv + row * ld + (int64_t)hh * hd;
  const float a = bf2f(x[d]), bb = bf2f(x[d + half]);
  const float c0 = bf2f(cos_t[ps * hd + d]), s0 = bf2f(sin_t[ps * hd + d]);
  const float c1 = bf2f(cos_t[ps * hd + d + half]), s1 = bf2f(sin_t[ps * hd + d + half]);
  float y0, y1;
  if (!inverse) {
    y0 = rbf(a * c0) + rbf(-bb * s0);
    y1 = rbf(bb * c1) + rbf(a * s1);
  } else {
    y0 = a * c0 + bb * s1;
    y1 = bb * c1 - a * s0;
  }
  x[d] = f2bf(y0);
  x[d + half] = f2bf(y1);
}

static int at_check(const void* qkv, int64_t ld, int B, int L, int H, int hd) {
  if (!qkv) return EAGLE_ERR_ARG;
  if (B < 1 || L < 1 || H < 1 || hd != AT_HD || (ld & 7) || ld < 3 * H * hd) return EAGLE_ERR_SHAPE;
  return EAGLE_OK;
}

extern "C" int eagle_rope_qk_bf16(void* qkv, int64_t ld, const int64_t* position_ids, const void* cos_table,
                                  const void* sin_table, int n_pos, int64_t nrows, int L, int n_heads_qk, int hd, int inverse,
                                  hipStream_t stream) {
  if (!qkv || !cos_table || !sin_table) return EAGLE_ERR_ARG;
  if (nrows < 1 || n_heads_qk < 1 || hd < 2 || (hd & 1) || L < 1 || (!position_ids && n_pos < L)) return EAGLE_ERR_SHAPE;
  const int64_t total = nrows * n_heads_qk * (hd / 2);
  hipLaunchKernelGGL(rope_qk_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (bf16_t*)qkv, ld, position_ids,
                     (const bf16_t*)cos_table, (const bf16_t*)sin_table, n_pos, nrows, L, n_heads_qk, hd, inverse);
  return eagle_check_launch();
}

extern "C" int eagle_attn_train_fwd_bf16(const void* qkv, int64_t ld, void* out, int64_t ldo, float* lse, int B, int L, int H,
                                         int hd, hipStream_t stream) {
  int rc = at_check(qkv, ld, B, L, H, hd);
  if (rc) return rc;
  if (!out || !lse || ldo < H * hd || (ldo & 3)) return EAGLE_ERR_SHAPE;
  AttnTrainArgs a = {};
  a.qkv = (const bf16_t*)qkv; a.ld = ld; a.out = (bf16_t*)out; a.ldo = ldo; a.lse = lse;
  a.B = B; a.L = L; a.H = H; a.scale = 1.0f / sqrtf((float)hd); a.scale2 = a.scale * LOG2E;
  hipLaunchKernelGGL(attn_train_fwd_kernel, dim3((L + AT_QB - 1) / AT_QB, H, B), dim3(AT_THREADS), 0, stream, a);
  return eagle_check_launch();
}

extern "C" int eagle_attn_train_bwd_bf16(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout,
                                         int64_t lddo, const float* lse, float* delta_ws, void* dqkv, int64_t lddq, int B, int L,
                                         int H, int hd, hipStream_t stream) {
  int rc = at_check(qkv, ld, B, L, H, hd);
  if (rc) return rc;
  if (!out || !dout || !lse || !delta_ws || !dqkv) return EAGLE_ERR_ARG;
  if (ldo < H * hd || lddo < H * hd || lddq < 3 * H * hd || (lddq & 3) || (lddo & 7)) return EAGLE_ERR_SHAPE;
  const int64_t items = (int64_t)B * L * H;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)out, ldo,
                     (const bf16_t*)dout, lddo, delta_ws, B, L, H);
  rc = eagle_check_launch();
  if (rc) return rc;
  AttnTrainArgs a = {};
  a.qkv = (const bf16_t*)qkv; a.ld = ld; a.lse = (float*)lse; a.dout = (const bf16_t*)dout; a.lddo = lddo; a.delta = delta_ws;
  a.dqkv = (bf16_t*)dqkv; a.lddq = lddq; a.B = B; a.L = L; a.H = H;
  a.scale = 1.0f / sqrtf((float)hd); a.scale2 = a.scale * LOG2E;
  const dim3 grid((L + AT_QB - 1) / AT_QB, H, B);
  hipLaunchKernelGGL(attn_train_dq_kernel, grid, dim3(AT_THREADS), 0, stream, a);
  rc = eagle_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(attn_train_dkv_kernel, grid, dim3(AT_THREADS), 0, stream, a);
  return eagle_check_launch();
}
