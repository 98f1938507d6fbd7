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
#define LOG2E 1.4426950408889634f

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

// Workgroup -> (pair index, head, batch) for the second-generation kernels: a 1-D grid whose ids
// b, b+8, ... share an XCD (round-robin dispatch) is remapped so that every XCD owns a CONTIGUOUS
// range of (batch, head) groups: the workgroups that re-read one head's K/V (or Q/dO) tiles then
// hit the same 4 MB L2 instead of pulling the head through all eight (speed only, never correctness).
__device__ __forceinline__ void a2_work(int npairs, int H, int& pair, int& h, int& b) {
  const int n = gridDim.x, id = blockIdx.x;
  int w = id;
  if ((n & 7) == 0) w = (id & 7) * (n >> 3) + (id >> 3);
  pair = w % npairs;
  const int bh = w / npairs;
  h = bh % H;
  b = bh / H;
}

// ------------------------------------------------------------------ forward, second generation
// 128 queries per workgroup (32 per wave = two 16-query MFMA column groups), 64-key tiles, K and V
// staged ROW-MAJOR in LDS ([64][128] bf16, 256-byte rows, 16-byte chunks XOR-swizzled so that both
// the row reads and the transposing reads are (nearly) conflict-free); V^T fragments come straight
// from the row-major image through ds_read_b64_tr_b16, two reads per 32-key step: the lane-local
// k order of the P fragment (two stacked 16-key score tiles: slot (g,j) <-> key 16*(j>>2)+4g+(j&3))
// is exactly two 4-row blocks per 16-lane group.  Global loads of tile t+1 are issued before tile t
// is multiplied and written to the other LDS buffer after it: one barrier per tile.
#define A2_QB 128
#define A2_KB 64
typedef __attribute__((ext_vector_type(4))) short a2_s4;
// buffer resource over the rows [0, nrows) of a row-major bf16 operand (row stride ld elements): the DMA instructions address
// it with a 32-bit byte offset per lane
__device__ __forceinline__ __amdgpu_buffer_rsrc_t a2_rsrc(const bf16_t* base, int64_t ld, int nrows) {
  const uint64_t bytes = (uint64_t)nrows * (uint64_t)ld * 2ull;
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)bytes), 0x00020000);
}
__device__ __forceinline__ int a2_off(int row, int ch) {   // byte offset of 16-byte chunk ch of a tile row
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

template <bool PAIR>
__global__ __launch_bounds__(AT_THREADS, 2) void attn_train_fwd2_kernel(AttnTrainArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[2][2][A2_KB * 256];   // [buffer][K|V] 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const int D = p.H * AT_HD, L = p.L;
  // causal balance: a workgroup takes query block i AND block nqb-1-i (their key counts add up to
  // the same total for every i), heavy one first
  const int nqb = (L + A2_QB - 1) / A2_QB;
  int bx, h, b;
  a2_work(PAIR ? (nqb + 1) / 2 : nqb, p.H, bx, h, b);
  const bf16_t* base = p.qkv + (int64_t)b * L * p.ld + h * AT_HD;
  const bf16_t* kbase = base + D;
  const bf16_t* vbase = base + 2 * D;
  for (int pass = 0; pass < (PAIR ? 2 : 1); ++pass) {
  const int qb = pass == 0 ? nqb - 1 - bx : bx;
  if (pass == 1 && qb >= nqb - 1 - bx) break;
  const int qw = qb * A2_QB + wave * 32;          // first query of this wave
  bf16x8 qf[2][4];
#pragma unroll
  for (int qg = 0; qg < 2; ++qg) {
    const int qi = min(qw + 16 * qg + r16, L - 1);   // rows past L re-read the last row; never stored
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[qg][s] = *reinterpret_cast<const bf16x8*>(base + (int64_t)qi * p.ld + 32 * s + 8 * g);
  }
  f32x4 o[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i) { o[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; o[i][1] = o[i][0]; }
  float m[2] = {-INFINITY, -INFINITY}, lsum[2] = {0.f, 0.f};
  const int kend = min(L, qb * A2_QB + A2_QB);   // causal: keys below the end of this query block
  const int nt = (kend + A2_KB - 1) / A2_KB;

  // K/V tiles go HBM/L2 -> LDS by LDS-DMA (no staging registers, no ds_write): a wave-instruction
  // fills 1 KB = 4 rows x 256 B lane-linearly, so the chunk swizzle of a2_off() is applied to the
  // per-lane SOURCE address (lane l: row l>>4, slot l&15 holds chunk slot ^ f(row)).  Never
  // predicated: keys past L re-read the last key row and are masked out of the scores.
  // (buffer-addressed DMA: beside a pending global_load_lds hipcc waits vmcnt(0) before the first ds_read that follows,
  // i.e. for the tile just requested -- the prefetch then overlaps nothing; it does not do that for buffer_load ... lds)
  const __amdgpu_buffer_rsrc_t krs = a2_rsrc(kbase, p.ld, L), vrs = a2_rsrc(vbase, p.ld, L);
  auto stage = [&](int k0, int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = i * AT_THREADS + tid;                 // 16-byte slot of the [64][16] tile image
      const int trow = c >> 4, slot = c & 15;
      const int ch = slot ^ (((trow & 3) << 2) | ((trow >> 2) & 3));
      const int row = min(k0 + trow, L - 1);
      const uint32_t off = (uint32_t)(((int64_t)row * p.ld + ch * 8) * 2);
      char* dk = &lds[buf][0][(i * AT_THREADS + wave * 64) * 16];   // wave-uniform base; HW adds lane*16
      char* dv = &lds[buf][1][(i * AT_THREADS + wave * 64) * 16];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (__attribute__((address_space(3))) void*)dk, 16, off, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (__attribute__((address_space(3))) void*)dv, 16, off, 0, 0, 0);
    }
  };
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // transposing-read lane geometry: lane i = 4q+p of a 16-lane group addresses row q, columns 4p..4p+3
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  for (int t = 0; t < nt; ++t) {
    const int k0 = t * A2_KB;
    stage(k0 + A2_KB, (t + 1) & 1);   // unconditional (past the end: clamped rows, never read)
    const char* Kt = &lds[t & 1][0][0];
    const char* Vt = &lds[t & 1][1][0];
    if (k0 <= qw + 31) {         // wave-uniform: at least one visible key in this tile
      f32x4 sc[4][2];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) { sc[kt][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; sc[kt][1] = sc[kt][0]; }
      // K fragments one 32-wide d-step at a time, right before their MFMAs (round 3): 222 registers instead of 364, so two
      // workgroups share a CU and one's barrier / softmax time lies under the other's MFMAs (177 -> 106 us per launch)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 kfr[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) kfr[kt] = *reinterpret_cast<const bf16x8*>(Kt + a2_off(16 * kt + r16, 4 * s + g));
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          sc[kt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[kt], qf[0][s], sc[kt][0], 0, 0, 0);
          sc[kt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[kt], qf[1][s], sc[kt][1], 0, 0, 0);
        }
      }
      const bool need_mask = (k0 + A2_KB - 1 > qw) || (k0 + A2_KB > L);   // wave-uniform
      bf16x8 pf[2][2];
#pragma unroll
      for (int qg = 0; qg < 2; ++qg) {
        // VALU diet (this loop, not the MFMAs, bounds the kernel): the running maximum is kept on the
        // RAW scores (the scale is positive), each probability is one fma + one v_exp, masking only
        // runs on tiles that touch the diagonal or the end of the sequence, and the 64-register
        // rescale of O is skipped when no lane's maximum moved.  Key 0 is visible to every query,
        // so after the first tile every maximum is finite: no -inf guards.
        if (need_mask) {
          const int qi = qw + 16 * qg + r16;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = k0 + 16 * kt + 4 * g + r;
              sc[kt][qg][r] = (key <= qi && key < L) ? sc[kt][qg][r] : -INFINITY;
            }
        }
        float mx = m[qg];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
          mx = fmaxf(fmaxf(mx, fmaxf(sc[kt][qg][0], sc[kt][qg][1])), fmaxf(sc[kt][qg][2], sc[kt][qg][3]));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float neg = -mx * p.scale2;
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(sc[kt][qg][r], p.scale2, neg));
            sc[kt][qg][r] = e;
            ps += e;
          }
        if (__builtin_amdgcn_ballot_w64(mx != m[qg])) {         // wave-uniform: some row's maximum moved
          const float alpha = __builtin_amdgcn_exp2f(fmaf(m[qg], p.scale2, neg));   // m = -inf (first tile) -> 0
          lsum[qg] *= alpha;
#pragma unroll
          for (int dt = 0; dt < 8; ++dt) o[dt][qg] *= alpha;
          m[qg] = mx;
        }
        lsum[qg] += ps;
        pf[qg][0] = pack_frag(sc[0][qg], sc[1][qg]);
        pf[qg][1] = pack_frag(sc[2][qg], sc[3][qg]);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
          const int r0 = 32 * ks + 4 * g + tq;
          const a2_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) a2_s4*)(Vt + a2_off(r0, 2 * dt + (tp >> 1)) + 8 * (tp & 1)));
          const a2_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) a2_s4*)(Vt + a2_off(r0 + 16, 2 * dt + (tp >> 1)) + 8 * (tp & 1)));
          union { struct { a2_s4 a, b; } s; bf16x8 v; } u;
          u.s.a = lo; u.s.b = hi;
          o[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.v, pf[0][ks], o[dt][0], 0, 0, 0);
          o[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.v, pf[1][ks], o[dt][1], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // next tile landed (this thread's share) ...
    __builtin_amdgcn_s_barrier();                                   // ... and everybody's; this tile's reads are done
  }
#pragma unroll
  for (int qg = 0; qg < 2; ++qg) {
    float ls = lsum[qg];
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    const int qi = qw + 16 * qg + r16;
    if (qi < L) {
      const float inv = ls > 0.f ? 1.f / ls : 0.f;
      bf16_t* orow = p.out + ((int64_t)b * L + qi) * p.ldo + h * AT_HD;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        const uint2 w = make_uint2(pack2bf(o[dt][qg][0] * inv, o[dt][qg][1] * inv), pack2bf(o[dt][qg][2] * inv, o[dt][qg][3] * inv));
        *reinterpret_cast<uint2*>(orow + dt * 16 + 4 * g) = w;
      }
      if (g == 0) p.lse[((int64_t)b * p.H + h) * L + qi] = m[qg] * p.scale2 + log2f(ls);
    }
  }
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

// ------------------------------------------------------------------ backward, second generation
// Same building blocks as attn_train_fwd2_kernel: [64][128] bf16 tiles staged by LDS-DMA with the
// a2_off() chunk swizzle on the source address, one image serving the row reads (ds_read_b128) and
// the transposing reads (ds_read_b64_tr_b16), every fragment of a tile requested before the MFMAs
// that use it, no predicated loads (rows past L re-read row L-1 and are masked out of P), a
// workgroup takes block i and block n-1-i of the causal triangle.
__device__ __forceinline__ void a2_stage_tile(const bf16_t* gbase, int64_t ld, int row0, int nrows, char* dst, int tid) {
  const int wave = tid >> 6;
  const __amdgpu_buffer_rsrc_t rs = a2_rsrc(gbase, ld, nrows);       // buffer-addressed: see attn_train_fwd2_kernel
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = i * AT_THREADS + tid;
    const int trow = c >> 4, slot = c & 15;
    const int ch = slot ^ (((trow & 3) << 2) | ((trow >> 2) & 3));
    const int row = min(row0 + trow, nrows - 1);
    char* d = dst + (i * AT_THREADS + wave * 64) * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)d, 16,
                                             (uint32_t)(((int64_t)row * ld + ch * 8) * 2), 0, 0, 0);
  }
}
__device__ __forceinline__ bf16x8 a2_row_frag(const char* tile, int row, int ch) {
  return *reinterpret_cast<const bf16x8*>(tile + a2_off(row, ch));
}
// transposed fragment for k-step ks (32 tile rows) and 16-column block dt: element j of lane group g
// = tile[32*ks + 16*(j>>2) + 4g + (j&3)][16*dt + (lane&15)]
__device__ __forceinline__ bf16x8 a2_tr_frag(const char* tile, int ks, int dt, int g, int tq, int tp) {
  const int r0 = 32 * ks + 4 * g + tq;
  const a2_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) a2_s4*)(tile + a2_off(r0, 2 * dt + (tp >> 1)) + 8 * (tp & 1)));
  const a2_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) a2_s4*)(tile + a2_off(r0 + 16, 2 * dt + (tp >> 1)) + 8 * (tp & 1)));
  union { struct { a2_s4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// dQ (query stationary): 128 queries per pass (32 per wave), 64-key tiles of K and V.
__global__ __launch_bounds__(AT_THREADS, 2) void attn_train_dq2_kernel(AttnTrainArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[2][2][A2_KB * 256];   // [buffer][K|V]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  const int D = p.H * AT_HD, L = p.L;
  const int nqb = (L + A2_QB - 1) / A2_QB;
  int bx, h, b;
  a2_work((nqb + 1) / 2, p.H, bx, h, b);
  const bf16_t* base = p.qkv + (int64_t)b * L * p.ld + h * AT_HD;
  const bf16_t* kbase = base + D;
  const bf16_t* vbase = base + 2 * D;
  for (int pass = 0; pass < 2; ++pass) {
    const int qb = pass == 0 ? nqb - 1 - bx : bx;
    if (pass == 1 && qb >= nqb - 1 - bx) break;
    const int qw = qb * A2_QB + wave * 32;
    bf16x8 qf[2][4], dof[2][4];
    float nlse[2], dl[2];
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
      const int qi = min(qw + 16 * qg + r16, L - 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        qf[qg][s] = *reinterpret_cast<const bf16x8*>(base + (int64_t)qi * p.ld + 32 * s + 8 * g);
        dof[qg][s] = *reinterpret_cast<const bf16x8*>(p.dout + ((int64_t)b * L + qi) * p.lddo + h * AT_HD + 32 * s + 8 * g);
      }
      nlse[qg] = -p.lse[((int64_t)b * p.H + h) * L + qi];
      dl[qg] = p.delta[((int64_t)b * p.H + h) * L + qi];
    }
    f32x4 dq[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i) { dq[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; dq[i][1] = dq[i][0]; }
    const int kend = min(L, qb * A2_QB + A2_QB);
    const int nt = (kend + A2_KB - 1) / A2_KB;
    a2_stage_tile(kbase, p.ld, 0, L, &lds[0][0][0], tid);
    a2_stage_tile(vbase, p.ld, 0, L, &lds[0][1][0], tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < nt; ++t) {
      const int k0 = t * A2_KB;
      a2_stage_tile(kbase, p.ld, k0 + A2_KB, L, &lds[(t + 1) & 1][0][0], tid);
      a2_stage_tile(vbase, p.ld, k0 + A2_KB, L, &lds[(t + 1) & 1][1][0], tid);
      const char* Kt = &lds[t & 1][0][0];
      const char* Vt = &lds[t & 1][1][0];
      if (k0 <= qw + 31) {
        // The 64-key tile in two halves of 32 keys (= one 32-wide k step of the dQ product each): scores, dP, dS and the
        // dQ update of a half before the next half starts, fragments read right before their MFMAs.  Half the live score /
        // dP registers and no fragment arrays: 2 workgroups per CU instead of 1 (the all-at-once form needs 400 registers).
        const bool need_mask = (k0 + A2_KB - 1 > qw) || (k0 + A2_KB > L);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          f32x4 sc[2][2], dp[2][2];
#pragma unroll
          for (int j = 0; j < 2; ++j) { sc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; sc[j][1] = sc[j][0]; dp[j][0] = sc[j][0]; dp[j][1] = sc[j][0]; }
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            bf16x8 kfr[2], vrr[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              kfr[j] = a2_row_frag(Kt, 16 * (2 * ks + j) + r16, 4 * s + g);
              vrr[j] = a2_row_frag(Vt, 16 * (2 * ks + j) + r16, 4 * s + g);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              sc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[j], qf[0][s], sc[j][0], 0, 0, 0);
              sc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[j], qf[1][s], sc[j][1], 0, 0, 0);
              dp[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vrr[j], dof[0][s], dp[j][0], 0, 0, 0);
              dp[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vrr[j], dof[1][s], dp[j][1], 0, 0, 0);
            }
          }
          bf16x8 dsf[2];
#pragma unroll
          for (int qg = 0; qg < 2; ++qg) {
            const int qi = qw + 16 * qg + r16;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float pr = __builtin_amdgcn_exp2f(fmaf(sc[j][qg][r], p.scale2, nlse[qg]));
                if (need_mask) { const int key = k0 + 16 * (2 * ks + j) + 4 * g + r; pr = (key <= qi && key < L) ? pr : 0.f; }
                sc[j][qg][r] = pr * (dp[j][qg][r] - dl[qg]);
              }
            dsf[qg] = pack_frag(sc[0][qg], sc[1][qg]);
          }
#pragma unroll
          for (int dt = 0; dt < 8; ++dt) {
            const bf16x8 ktf = a2_tr_frag(Kt, ks, dt, g, tq, tp);
            dq[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[0], dq[dt][0], 0, 0, 0);
            dq[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[1], dq[dt][1], 0, 0, 0);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
      const int qi = qw + 16 * qg + r16;
      if (qi < L) {
        bf16_t* drow = p.dqkv + ((int64_t)b * L + qi) * p.lddq + h * AT_HD;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
          const uint2 w = make_uint2(pack2bf(dq[dt][qg][0] * p.scale, dq[dt][qg][1] * p.scale),
                                     pack2bf(dq[dt][qg][2] * p.scale, dq[dt][qg][3] * p.scale));
          *reinterpret_cast<uint2*>(drow + dt * 16 + 4 * g) = w;
        }
      }
    }
  }
}

// dK, dV (key stationary): 128 keys per pass (32 per wave), 64-query tiles of Q and dO; the rows'
// -lse and delta ride along in the same LDS array (two 4-byte LDS-DMA pieces per tile).
#define A2_TILE_BYTES (A2_KB * 256)
#define A2_BUF_BYTES (2 * A2_TILE_BYTES + 2 * A2_KB * 4)
__global__ __launch_bounds__(AT_THREADS) void attn_train_dkv2_kernel(AttnTrainArgs p) {
  __shared__ __attribute__((aligned(16))) char lds[2 * A2_BUF_BYTES];   // [buffer]{Q tile, dO tile, lse[64], delta[64]}
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  const int D = p.H * AT_HD, L = p.L;
  int bx, h, b;
  a2_work(((L + A2_QB - 1) / A2_QB + 1) / 2, p.H, bx, h, b);
  const bf16_t* base = p.qkv + (int64_t)b * L * p.ld + h * AT_HD;
  const bf16_t* dobase = p.dout + (int64_t)b * L * p.lddo + h * AT_HD;
  const float* lse_g = p.lse + ((int64_t)b * p.H + h) * L;
  const float* dl_g = p.delta + ((int64_t)b * p.H + h) * L;
  const int nkb = (L + A2_QB - 1) / A2_QB;
  auto stage = [&](int q0, int buf) {
    char* bb = lds + buf * A2_BUF_BYTES;
    a2_stage_tile(base, p.ld, q0, L, bb, tid);
    a2_stage_tile(dobase, p.lddo, q0, L, bb + A2_TILE_BYTES, tid);
    if (wave < 2) {   // wave-uniform: one 4-byte piece per lane, lane-linear destination
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(wave == 0 ? lse_g : dl_g), 0, L * 4, 0x00020000);
      char* d = bb + 2 * A2_TILE_BYTES + wave * (A2_KB * 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)d, 4, (uint32_t)(min(q0 + lane, L - 1) * 4), 0, 0, 0);
    }
  };
  for (int pass = 0; pass < 2; ++pass) {
    const int kb_ = pass == 0 ? bx : nkb - 1 - bx;    // key block 0 is the heaviest
    if (pass == 1 && kb_ <= bx) break;
    const int kw = kb_ * A2_QB + wave * 32;        // first key of this wave
    bf16x8 kf[2][4], vf[2][4];
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      const int ki = min(kw + 16 * kg + r16, L - 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        kf[kg][s] = *reinterpret_cast<const bf16x8*>(base + D + (int64_t)ki * p.ld + 32 * s + 8 * g);
        vf[kg][s] = *reinterpret_cast<const bf16x8*>(base + 2 * D + (int64_t)ki * p.ld + 32 * s + 8 * g);
      }
    }
    f32x4 dk[8][2], dv[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i) { dk[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[i][1] = dk[i][0]; dv[i][0] = dk[i][0]; dv[i][1] = dk[i][0]; }
    const int t0 = (kb_ * A2_QB) / A2_KB;          // causal: query tiles from the one holding the block's first key
    const int nt = (L + A2_KB - 1) / A2_KB;
    stage(t0 * A2_KB, t0 & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int t = t0; t < nt; ++t) {
      const int q0 = t * A2_KB;
      stage(q0 + A2_KB, (t + 1) & 1);
      const char* Qt = lds + (t & 1) * A2_BUF_BYTES;
      const char* Ot = Qt + A2_TILE_BYTES;
      const float* lse_s = reinterpret_cast<const float*>(Qt + 2 * A2_TILE_BYTES);
      const float* dl_s = lse_s + A2_KB;
      if (q0 + A2_KB - 1 >= kw) {                  // wave-uniform: some query of the tile sees this wave's keys
        // The 64-query tile in two halves of 32 queries (= one 32-wide k step of the dV / dK products each): scores, dP, P, dS
        // and the two updates of a half before the next half starts, fragments read right before their MFMAs (holding every
        // fragment of the tile spilled at 256 registers and needs 450 without a cap): 2 workgroups per CU.
        const bool need_mask = (q0 < kw + 31) || (q0 + A2_KB > L);   // tile touches the diagonal / the end
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          f32x4 sc[2][2], dp[2][2];
#pragma unroll
          for (int j = 0; j < 2; ++j) { sc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; sc[j][1] = sc[j][0]; dp[j][0] = sc[j][0]; dp[j][1] = sc[j][0]; }
          f32x4 lser[2], dlr[2];                     // this lane's rows 4g..4g+3 of each 16-query tile of the half
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            lser[j] = *reinterpret_cast<const f32x4*>(lse_s + 16 * (2 * ks + j) + 4 * g);
            dlr[j] = *reinterpret_cast<const f32x4*>(dl_s + 16 * (2 * ks + j) + 4 * g);
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            bf16x8 qr[2], orw[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              qr[j] = a2_row_frag(Qt, 16 * (2 * ks + j) + r16, 4 * s + g);
              orw[j] = a2_row_frag(Ot, 16 * (2 * ks + j) + r16, 4 * s + g);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              sc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qr[j], kf[0][s], sc[j][0], 0, 0, 0);   // S[q][key]: rows = queries
              sc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qr[j], kf[1][s], sc[j][1], 0, 0, 0);
              dp[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(orw[j], vf[0][s], dp[j][0], 0, 0, 0);  // dP = dO . V^T
              dp[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(orw[j], vf[1][s], dp[j][1], 0, 0, 0);
            }
          }
          bf16x8 pf[2], dsf[2];
#pragma unroll
          for (int kg = 0; kg < 2; ++kg) {
            const int ki = kw + 16 * kg + r16;
            f32x4 pr[2], dsv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float e = __builtin_amdgcn_exp2f(fmaf(sc[j][kg][r], p.scale2, -lser[j][r]));
                if (need_mask) { const int qi = q0 + 16 * (2 * ks + j) + 4 * g + r; e = (ki <= qi && qi < L && ki < L) ? e : 0.f; }
                pr[j][r] = e;
                dsv[j][r] = e * (dp[j][kg][r] - dlr[j][r]);
              }
            pf[kg] = pack_frag(pr[0], pr[1]);
            dsf[kg] = pack_frag(dsv[0], dsv[1]);
          }
#pragma unroll
          for (int dt = 0; dt < 8; ++dt) {
            const bf16x8 otf = a2_tr_frag(Ot, ks, dt, g, tq, tp), qtf = a2_tr_frag(Qt, ks, dt, g, tq, tp);
            dv[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(otf, pf[0], dv[dt][0], 0, 0, 0);
            dv[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(otf, pf[1], dv[dt][1], 0, 0, 0);
            dk[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsf[0], dk[dt][0], 0, 0, 0);
            dk[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsf[1], dk[dt][1], 0, 0, 0);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      const int ki = kw + 16 * kg + r16;
      if (ki < L) {
        bf16_t* drow = p.dqkv + ((int64_t)b * L + ki) * p.lddq + h * AT_HD;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
          *reinterpret_cast<uint2*>(drow + D + dt * 16 + 4 * g) =
              make_uint2(pack2bf(dk[dt][kg][0] * p.scale, dk[dt][kg][1] * p.scale), pack2bf(dk[dt][kg][2] * p.scale, dk[dt][kg][3] * p.scale));
          *reinterpret_cast<uint2*>(drow + 2 * D + dt * 16 + 4 * g) =
              make_uint2(pack2bf(dv[dt][kg][0], dv[dt][kg][1]), pack2bf(dv[dt][kg][2], dv[dt][kg][3]));
        }
      }
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

// The same rotation, eight pairs per thread: (d .. d+7, d+half .. d+half+7) as 16-byte loads and stores (hd % 16 == 0, ld % 8 == 0,
// 16-byte aligned bases).  Identical arithmetic per element.
__global__ __launch_bounds__(256) void rope_qk_vec_kernel(bf16_t* __restrict__ qkv, int64_t ld, const int64_t* __restrict__ pos,
                                                         const bf16_t* __restrict__ cos_t, const bf16_t* __restrict__ sin_t, int n_pos,
                                                         int64_t nrows, int L, int heads2, int hd, int inverse) {
  const int half = hd >> 1, per_head = half >> 3;            // threads per head
  const int per_row = heads2 * per_head;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nrows * per_row) return;
  const int64_t row = idx / per_row;
  const int rem = (int)(idx - row * per_row);
  const int hh = rem / per_head, d = (rem - hh * per_head) << 3;
  int64_t ps = pos ? pos[row] : (row % L);
  ps = ps < 0 ? 0 : (ps >= n_pos ? n_pos - 1 : ps);
  bf16_t* x = qkv + row * ld + (int64_t)hh * hd + d;
  const uint4 xa = *reinterpret_cast<const uint4*>(x), xb = *reinterpret_cast<const uint4*>(x + half);
  const uint4 c0v = *reinterpret_cast<const uint4*>(cos_t + ps * hd + d), c1v = *reinterpret_cast<const uint4*>(cos_t + ps * hd + d + half);
  const uint4 s0v = *reinterpret_cast<const uint4*>(sin_t + ps * hd + d), s1v = *reinterpret_cast<const uint4*>(sin_t + ps * hd + d + half);
  const uint32_t* ua = reinterpret_cast<const uint32_t*>(&xa); const uint32_t* ub = reinterpret_cast<const uint32_t*>(&xb);
  const uint32_t* uc0 = reinterpret_cast<const uint32_t*>(&c0v); const uint32_t* uc1 = reinterpret_cast<const uint32_t*>(&c1v);
  const uint32_t* us0 = reinterpret_cast<const uint32_t*>(&s0v); const uint32_t* us1 = reinterpret_cast<const uint32_t*>(&s1v);
  uint32_t oa[4], ob[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float y0[2], y1[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float a = e ? hi_bf(ua[j]) : lo_bf(ua[j]), bb = e ? hi_bf(ub[j]) : lo_bf(ub[j]);
      const float c0 = e ? hi_bf(uc0[j]) : lo_bf(uc0[j]), c1 = e ? hi_bf(uc1[j]) : lo_bf(uc1[j]);
      const float s0 = e ? hi_bf(us0[j]) : lo_bf(us0[j]), s1 = e ? hi_bf(us1[j]) : lo_bf(us1[j]);
      if (!inverse) {
        y0[e] = rbf(a * c0) + rbf(-bb * s0);
        y1[e] = rbf(bb * c1) + rbf(a * s1);
      } else {
        y0[e] = a * c0 + bb * s1;
        y1[e] = bb * c1 - a * s0;
      }
    }
    oa[j] = pack2bf(y0[0], y0[1]);
    ob[j] = pack2bf(y1[0], y1[1]);
  }
  *reinterpret_cast<uint4*>(x) = make_uint4(oa[0], oa[1], oa[2], oa[3]);
  *reinterpret_cast<uint4*>(x + half) = make_uint4(ob[0], ob[1], ob[2], ob[3]);
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
  const bool vec = (hd % 16 == 0) && (ld % 8 == 0) &&
                   (((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(cos_table) | reinterpret_cast<uintptr_t>(sin_table)) & 15) == 0);
  if (vec) {
    const int64_t threads = nrows * n_heads_qk * (hd / 16);
    hipLaunchKernelGGL(rope_qk_vec_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, (bf16_t*)qkv, ld, position_ids,
                       (const bf16_t*)cos_table, (const bf16_t*)sin_table, n_pos, nrows, L, n_heads_qk, hd, inverse);
    return eagle_check_launch();
  }
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
  hipLaunchKernelGGL(attn_train_fwd2_kernel<true>, dim3(((L + A2_QB - 1) / A2_QB + 1) / 2 * H * B), dim3(AT_THREADS), 0, stream, a);
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
  const dim3 grid2(((L + A2_QB - 1) / A2_QB + 1) / 2 * H * B);
  hipLaunchKernelGGL(attn_train_dq2_kernel, grid2, dim3(AT_THREADS), 0, stream, a);
  rc = eagle_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(attn_train_dkv2_kernel, grid2, dim3(AT_THREADS), 0, stream, a);
  return eagle_check_launch();
}
