// Bandwidth-bound element-wise pieces of the draft-head training step (gfx950): all bf16 rows,
// 16-byte accesses, fp32 math.
#include "common.h"

__device__ __forceinline__ void ld8(const bf16_t* p, float* f) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  const uint32_t* u = reinterpret_cast<const uint32_t*>(&v);
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[2 * j] = lo_bf(u[j]); f[2 * j + 1] = hi_bf(u[j]); }
}
__device__ __forceinline__ void st8(bf16_t* p, const float* f) {
  uint4 v;
  uint32_t* u = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
  for (int j = 0; j < 4; ++j) u[j] = pack2bf(f[2 * j], f[2 * j + 1]);
  *reinterpret_cast<uint4*>(p) = v;
}

// x0[n, 0:D] = embed[ids[n]], x0[n, D:2D] = hidden[n]      (eagle/llama2.py:515, :537-542)
__global__ __launch_bounds__(256) void embed_concat_kernel(const bf16_t* __restrict__ emb, const int64_t* __restrict__ ids,
                                                          const bf16_t* __restrict__ hid, int64_t ldh, bf16_t* __restrict__ out,
                                                          int D, int V) {
  const int64_t n = blockIdx.x;
  int64_t id = ids[n];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  const uint4* e = reinterpret_cast<const uint4*>(emb + id * D);
  const uint4* h = reinterpret_cast<const uint4*>(hid + n * ldh);
  uint4* o = reinterpret_cast<uint4*>(out + n * 2 * D);
  const int nc = D >> 3;
  for (int c = threadIdx.x; c < nc; c += blockDim.x) { o[c] = e[c]; o[nc + c] = h[c]; }
}

// act[n, i] = bf16(silu(bf16 g)) * u with gu = [g | u] ([N, 2I])   (eagle/llama2.py:317)
__global__ __launch_bounds__(256) void silu_mul_fwd_kernel(const bf16_t* __restrict__ gu, int64_t ld, bf16_t* __restrict__ act,
                                                          int64_t lda, int I) {
  const int64_t n = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c * 8 >= I) return;
  float g[8], u[8], y[8];
  ld8(gu + n * ld + c * 8, g);
  ld8(gu + n * ld + I + c * 8, u);
#pragma unroll
  for (int j = 0; j < 8; ++j) y[j] = rbf(g[j] / (1.f + __expf(-g[j]))) * u[j];
  st8(act + n * lda + c * 8, y);
}

// dgu = [dact*u*silu'(g) | dact*silu(g)]
__global__ __launch_bounds__(256) void silu_mul_bwd_kernel(const bf16_t* __restrict__ gu, int64_t ld, const bf16_t* __restrict__ dact,
                                                          int64_t lda, bf16_t* __restrict__ dgu, int64_t ldd, int I) {
  const int64_t n = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c * 8 >= I) return;
  float g[8], u[8], d[8], dg[8], du[8];
  ld8(gu + n * ld + c * 8, g);
  ld8(gu + n * ld + I + c * 8, u);
  ld8(dact + n * lda + c * 8, d);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float sg = 1.f / (1.f + __expf(-g[j]));
    const float si = g[j] * sg;
    dg[j] = d[j] * u[j] * (sg * (1.f + g[j] * (1.f - sg)));
    du[j] = d[j] * si;
  }
  st8(dgu + n * ldd + c * 8, dg);
  st8(dgu + n * ldd + I + c * 8, du);
}

// out[c] (+)= sum_n x[n, c]  (bias gradient); two stages through a [nb, C] fp32 workspace
__global__ __launch_bounds__(256) void colsum_bf16_stage1(const bf16_t* __restrict__ x, int64_t ld, float* __restrict__ part, int N,
                                                         int C, int rows_per_block) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c * 8 >= C) return;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(N, r0 + rows_per_block);
  for (int r = r0; r < r1; ++r) {
    float v[8];
    ld8(x + (int64_t)r * ld + c * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += v[j];
  }
  float* o = part + (int64_t)blockIdx.y * C + c * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = acc[j];
}
// 64 columns per workgroup: thread (cx, ry) sums the partial rows ry, ry + 16, ... of its 4 columns (coalesced 256-byte row
// segments), the 16 row groups meet in LDS.  C % 4 == 0.
__global__ __launch_bounds__(256) void colsum_stage2(const float* __restrict__ part, int nb, int C, bf16_t* __restrict__ out_bf,
                                                    float* __restrict__ out_f, int accumulate) {
  __shared__ float4 sh[16][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cx * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C) {
    for (int b = ry; b < nb; b += 16) {
      const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)b * C + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  sh[ry][cx] = s;
  __syncthreads();
  if (ry != 0 || c >= C) return;
  for (int r = 1; r < 16; ++r) { const float4 v = sh[r][cx]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  const float y[4] = {s.x, s.y, s.z, s.w};
  for (int j = 0; j < 4; ++j) {
    if (out_f) out_f[c + j] = accumulate ? out_f[c + j] + y[j] : y[j];
    if (out_bf) out_bf[c + j] = f2bf(accumulate ? bf2f(out_bf[c + j]) + y[j] : y[j]);
  }
}

// AdamW (decoupled weight decay, torch.optim.AdamW semantics: eagle/train.py:66) over one flat
// parameter buffer: bf16 params, bf16 grads, fp32 moments.  grad_scale multiplies the gradient
// first (e.g. clip_grad_norm_'s coefficient, read from device memory so no host sync is needed);
// clip_value > 0 clamps each element to [-clip, clip] (clip_grad_value_, eagle/train.py:127).
__global__ __launch_bounds__(256) void adamw_kernel(bf16_t* __restrict__ p, const bf16_t* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2, const float* __restrict__ grad_scale,
                                                   float clip_value) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= n) return;
  const float gs = grad_scale ? *grad_scale : 1.f;
  if (i + 8 <= n) {
    float pf[8], gf[8];
    ld8(p + i, pf);
    ld8(g + i, gf);
    float4 m0 = *reinterpret_cast<float4*>(m + i), m1 = *reinterpret_cast<float4*>(m + i + 4);
    float4 v0 = *reinterpret_cast<float4*>(v + i), v1 = *reinterpret_cast<float4*>(v + i + 4);
    float mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
    float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float gr = gf[j] * gs;
      if (clip_value > 0.f) gr = fminf(fmaxf(gr, -clip_value), clip_value);
      mm[j] = b1 * mm[j] + (1.f - b1) * gr;
      vv[j] = b2 * vv[j] + (1.f - b2) * gr * gr;
      const float upd = (mm[j] / bc1) / (sqrtf(vv[j] / bc2) + eps);
      pf[j] = pf[j] * (1.f - lr * wd) - lr * upd;
    }
    st8(p + i, pf);
    *reinterpret_cast<float4*>(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
    *reinterpret_cast<float4*>(m + i + 4) = make_float4(mm[4], mm[5], mm[6], mm[7]);
    *reinterpret_cast<float4*>(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    *reinterpret_cast<float4*>(v + i + 4) = make_float4(vv[4], vv[5], vv[6], vv[7]);
  } else {
    for (int64_t k = i; k < n; ++k) {
      float gr = bf2f(g[k]) * gs;
      if (clip_value > 0.f) gr = fminf(fmaxf(gr, -clip_value), clip_value);
      const float mk = b1 * m[k] + (1.f - b1) * gr, vk = b2 * v[k] + (1.f - b2) * gr * gr;
      m[k] = mk; v[k] = vk;
      p[k] = f2bf(bf2f(p[k]) * (1.f - lr * wd) - lr * (mk / bc1) / (sqrtf(vk / bc2) + eps));
    }
  }
}

// sum of squares of a bf16 buffer -> partial[blockIdx.x]; second stage turns it into the
// clip_grad_norm_ coefficient min(1, max_norm / (norm + 1e-6)) (eagle/train_tp.py:109)
__global__ __launch_bounds__(256) void sumsq_stage1(const bf16_t* __restrict__ g, int64_t n, float* __restrict__ part) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * blockDim.x * 8) {
    if (i + 8 <= n) {
      float f[8];
      ld8(g + i, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += f[j] * f[j];
    } else {
      for (int64_t k = i; k < n; ++k) { const float f = bf2f(g[k]); s += f * f; }
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ part, int nb, float max_norm, float* __restrict__ out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = sqrtf(sh[0] + sh[1] + sh[2] + sh[3]);
    out[0] = fminf(1.f, max_norm / (norm + 1e-6f));
    out[1] = norm;
  }
}

// g = clamp(g, -c, c) in place (clip_grad_value_ between micro-batches, eagle/train.py:127)
__global__ __launch_bounds__(256) void clip_value_kernel(bf16_t* __restrict__ g, int64_t n, float c) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= n) return;
  if (i + 8 <= n) {
    float f[8];
    ld8(g + i, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(f[j], -c), c);
    st8(g + i, f);
  } else {
    for (int64_t k = i; k < n; ++k) g[k] = f2bf(fminf(fmaxf(bf2f(g[k]), -c), c));
  }
}

// g *= *scale in place (clip_grad_norm_ between micro-batches, eagle/train_tp.py:109; the
// coefficient stays in device memory)
__global__ __launch_bounds__(256) void scale_kernel(bf16_t* __restrict__ g, int64_t n, const float* __restrict__ scale) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= n) return;
  const float c = *scale;
  if (i + 8 <= n) {
    float f[8];
    ld8(g + i, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] *= c;
    st8(g + i, f);
  } else {
    for (int64_t k = i; k < n; ++k) g[k] = f2bf(bf2f(g[k]) * c);
  }
}

// out = bf16(a_f32 + b_bf16)   (vocab-sharded loss: summed fp32 dpredict partials + smooth-L1 part)
__global__ __launch_bounds__(256) void add_f32_bf16_kernel(const float* __restrict__ a, const bf16_t* __restrict__ b,
                                                          bf16_t* __restrict__ out, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= n) return;
  if (i + 8 <= n) {
    const float4 a0 = *reinterpret_cast<const float4*>(a + i), a1 = *reinterpret_cast<const float4*>(a + i + 4);
    float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (b) ld8(b + i, f);
    f[0] += a0.x; f[1] += a0.y; f[2] += a0.z; f[3] += a0.w; f[4] += a1.x; f[5] += a1.y; f[6] += a1.z; f[7] += a1.w;
    st8(out + i, f);
  } else {
    for (int64_t k = i; k < n; ++k) out[k] = f2bf(a[k] + (b ? bf2f(b[k]) : 0.f));
  }
}

// _make_eagle_input (eagle/train.py:256-269) given the verifier's last hidden state:
//   noisy = hidden + noise ; hidden_out = noisy ; target[:, :-1] = noisy[:, 1:], last row 0 ;
//   ids_out = ids shifted left with trailing 0.   One block per (b, l) row.
__global__ __launch_bounds__(256) void make_eagle_input_kernel(const bf16_t* __restrict__ hid, const bf16_t* __restrict__ noise,
                                                              const int64_t* __restrict__ ids, bf16_t* __restrict__ hid_out,
                                                              bf16_t* __restrict__ target, int64_t* __restrict__ ids_out, int L, int D) {
  const int64_t row = blockIdx.x;
  const int l = (int)(row % L);
  for (int c = threadIdx.x; c < (D >> 3); c += blockDim.x) {
    float h[8], z[8];
    ld8(hid + row * D + c * 8, h);
    if (noise) {
      ld8(noise + row * D + c * 8, z);
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = h[j] + z[j];
    }
    st8(hid_out + row * D + c * 8, h);
    if (l > 0) st8(target + (row - 1) * D + c * 8, h);
    if (l == L - 1) { float zero[8] = {0, 0, 0, 0, 0, 0, 0, 0}; st8(target + row * D + c * 8, zero); }
  }
  if (threadIdx.x == 0) ids_out[row] = (l < L - 1) ? ids[row + 1] : 0;
}

// ---- fused input path: _make_eagle_input (eagle/train.py:256-275) + the draft head's embedding gather / concat
// (eagle/llama2.py:515, :537-542) in ONE pass over the verifier's hidden state ----
// Philox4x32-10 (Salmon et al., SC'11): counter = (element index / 4, stream offset), key = seed; word j of the
// output belongs to element 4*(index/4) + j.  u = (x >> 8) * 2^-24 in [0, 1).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// One block per (b, l) row.  noisy = bf16(hidden + bf16(low + (high-low)*u)) (the reference draws the noise in the
// hidden state's dtype and adds it in that dtype, train.py:272-275); x0[row] = cat(emb[ids[row+1] or 0], noisy[row]);
// target[row-1] = noisy[row], target[last row of a sequence] = 0; ids_out = ids shifted left with trailing 0.
__global__ __launch_bounds__(256) void eagle_input_fused_kernel(const bf16_t* __restrict__ hid, int64_t ldh_b, int64_t ldh_l,
                                                               const int64_t* __restrict__ ids, int64_t ldi,
                                                               const bf16_t* __restrict__ emb, int V, float low, float high,
                                                               uint64_t seed, uint64_t offset, bf16_t* __restrict__ x0,
                                                               bf16_t* __restrict__ target, int64_t* __restrict__ ids_out,
                                                               bf16_t* __restrict__ noise_out, int L, int D) {
  const int64_t row = blockIdx.x;
  const int b = (int)(row / L), l = (int)(row - (int64_t)b * L);
  int64_t id = (l < L - 1) ? ids[(int64_t)b * ldi + l + 1] : 0;
  if (threadIdx.x == 0 && ids_out) ids_out[row] = id;
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  const bf16_t* hrow = hid + (int64_t)b * ldh_b + (int64_t)l * ldh_l;
  const bool draw = high != low;
  for (int c = threadIdx.x; c < (D >> 3); c += blockDim.x) {
    float h[8];
    ld8(hrow + c * 8, h);
    if (draw) {
      const uint64_t e = (uint64_t)row * (uint64_t)D + (uint64_t)c * 8;      // first of this thread's 8 elements
      uint32_t r[8];
      philox4x32_10((uint32_t)(e >> 2), (uint32_t)(e >> 34), (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
      philox4x32_10((uint32_t)((e >> 2) + 1), (uint32_t)(((e >> 2) + 1) >> 32), (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed,
                    (uint32_t)(seed >> 32), r + 4);
      float z[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        z[j] = rbf(fmaf((float)(r[j] >> 8) * 5.9604644775390625e-8f, high - low, low));
        h[j] = h[j] + z[j];
      }
      if (noise_out) st8(noise_out + row * D + c * 8, z);
    }
    st8(x0 + row * 2 * D + D + c * 8, h);
    if (l > 0) st8(target + (row - 1) * D + c * 8, h);
    if (l == L - 1) { float zero[8] = {0, 0, 0, 0, 0, 0, 0, 0}; st8(target + row * D + c * 8, zero); }
    *reinterpret_cast<uint4*>(x0 + row * 2 * D + c * 8) = *reinterpret_cast<const uint4*>(emb + id * D + c * 8);
  }
}

extern "C" int eagle_input_embed_concat_bf16(const void* hidden, int64_t ldh_b, int64_t ldh_l, const int64_t* ids, int64_t ldi,
                                             const void* emb, int V, float low, float high, uint64_t seed, uint64_t offset,
                                             void* x0, void* target, int64_t* ids_out, void* noise_out, int B, int L, int D,
                                             hipStream_t stream) {
  if (!hidden || !ids || !emb || !x0 || !target) return EAGLE_ERR_ARG;
  if (B < 1 || L < 1 || D < 8 || (D & 7) || V < 1 || (ldh_b & 7) || (ldh_l & 7) || ldh_l < D || ldi < L || !(high >= low)) return EAGLE_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(hidden) | reinterpret_cast<uintptr_t>(emb) | reinterpret_cast<uintptr_t>(x0) |
       reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(noise_out)) & 15) return EAGLE_ERR_SHAPE;
  hipLaunchKernelGGL(eagle_input_fused_kernel, dim3(B * L), dim3(256), 0, stream, (const bf16_t*)hidden, ldh_b, ldh_l, ids, ldi,
                     (const bf16_t*)emb, V, low, high, seed, offset, (bf16_t*)x0, (bf16_t*)target, ids_out, (bf16_t*)noise_out, L, D);
  return eagle_check_launch();
}

extern "C" int eagle_embed_concat_bf16(const void* emb, const int64_t* ids, const void* hidden, int64_t ldh, void* out, int N, int D,
                                       int V, hipStream_t stream) {
  if (!emb || !ids || !hidden || !out) return EAGLE_ERR_ARG;
  if (N < 1 || D < 8 || (D & 7) || (ldh & 7) || V < 1) return EAGLE_ERR_SHAPE;
  hipLaunchKernelGGL(embed_concat_kernel, dim3(N), dim3(256), 0, stream, (const bf16_t*)emb, ids, (const bf16_t*)hidden, ldh,
                     (bf16_t*)out, D, V);
  return eagle_check_launch();
}

extern "C" int eagle_silu_mul_fwd_bf16(const void* gu, int64_t ld, void* act, int64_t lda, int N, int I, hipStream_t stream) {
  if (!gu || !act) return EAGLE_ERR_ARG;
  if (N < 1 || I < 8 || (I & 7) || (ld & 7) || (lda & 7) || ld < 2 * I || lda < I) return EAGLE_ERR_SHAPE;
  hipLaunchKernelGGL(silu_mul_fwd_kernel, dim3((I / 8 + 255) / 256, N), dim3(256), 0, stream, (const bf16_t*)gu, ld, (bf16_t*)act, lda, I);
  return eagle_check_launch();
}

extern "C" int eagle_silu_mul_bwd_bf16(const void* gu, int64_t ld, const void* dact, int64_t lda, void* dgu, int64_t ldd, int N,
                                       int I, hipStream_t stream) {
  if (!gu || !dact || !dgu) return EAGLE_ERR_ARG;
  if (N < 1 || I < 8 || (I & 7) || (ld & 7) || (lda & 7) || (ldd & 7) || ld < 2 * I || ldd < 2 * I || lda < I) return EAGLE_ERR_SHAPE;
  hipLaunchKernelGGL(silu_mul_bwd_kernel, dim3((I / 8 + 255) / 256, N), dim3(256), 0, stream, (const bf16_t*)gu, ld,
                     (const bf16_t*)dact, lda, (bf16_t*)dgu, ldd, I);
  return eagle_check_launch();
}

#define CS_ROWS 16                      // rows per stage-1 workgroup: 512 workgroups at 4096 tokens x 4096 columns (64 rows: 128 workgroups of serial loads)
extern "C" int64_t eagle_colsum_workspace_bytes(int N, int C) { return (int64_t)((N + CS_ROWS - 1) / CS_ROWS) * C * 4; }
extern "C" int eagle_colsum_bf16(const void* x, int64_t ld, int N, int C, void* out_bf16, float* out_f32, int accumulate,
                                 void* workspace, hipStream_t stream) {
  if (!x || (!out_bf16 && !out_f32) || !workspace) return EAGLE_ERR_ARG;
  if (N < 1 || C < 8 || (C & 7) || (ld & 7)) return EAGLE_ERR_SHAPE;
  const int nb = (N + CS_ROWS - 1) / CS_ROWS;
  hipLaunchKernelGGL(colsum_bf16_stage1, dim3((C / 8 + 255) / 256, nb), dim3(256), 0, stream, (const bf16_t*)x, ld,
                     (float*)workspace, N, C, CS_ROWS);
  int rc = eagle_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_stage2, dim3((C + 63) / 64), dim3(256), 0, stream, (const float*)workspace, nb, C, (bf16_t*)out_bf16,
                     out_f32, accumulate);
  return eagle_check_launch();
}

extern "C" int eagle_adamw_bf16(void* param, const void* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int step, const float* grad_scale, float clip_value,
                                hipStream_t stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq) return EAGLE_ERR_ARG;
  if (n < 1 || step < 1) return EAGLE_ERR_SHAPE;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  const int64_t nthreads = (n + 7) / 8;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, stream, (bf16_t*)param, (const bf16_t*)grad,
                     exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, clip_value);
  return eagle_check_launch();
}

#define SS_BLOCKS 1024
extern "C" int64_t eagle_grad_norm_workspace_bytes() { return SS_BLOCKS * 4; }
extern "C" int eagle_grad_clip_coef_bf16(const void* grad, int64_t n, float max_norm, float* coef_and_norm, void* workspace,
                                         hipStream_t stream) {
  if (!grad || !coef_and_norm || !workspace) return EAGLE_ERR_ARG;
  if (n < 1) return EAGLE_ERR_SHAPE;
  hipLaunchKernelGGL(sumsq_stage1, dim3(SS_BLOCKS), dim3(256), 0, stream, (const bf16_t*)grad, n, (float*)workspace);
  int rc = eagle_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, stream, (const float*)workspace, SS_BLOCKS, max_norm, coef_and_norm);
  return eagle_check_launch();
}

extern "C" int eagle_make_eagle_input_bf16(const void* hidden, const void* noise, const int64_t* ids, void* hidden_out,
                                           void* target, int64_t* ids_out, int B, int L, int D, hipStream_t stream) {
  if (!hidden || !ids || !hidden_out || !target || !ids_out) return EAGLE_ERR_ARG;
  if (B < 1 || L < 1 || D < 8 || (D & 7)) return EAGLE_ERR_SHAPE;
  hipLaunchKernelGGL(make_eagle_input_kernel, dim3(B * L), dim3(256), 0, stream, (const bf16_t*)hidden, (const bf16_t*)noise, ids,
                     (bf16_t*)hidden_out, (bf16_t*)target, ids_out, L, D);
  return eagle_check_launch();
}

extern "C" int eagle_clip_value_bf16(void* grad, int64_t n, float clip, hipStream_t stream) {
  if (!grad) return EAGLE_ERR_ARG;
  if (n < 1 || !(clip > 0.f) || (reinterpret_cast<uintptr_t>(grad) & 15)) return EAGLE_ERR_SHAPE;
  const int64_t nthreads = (n + 7) / 8;
  hipLaunchKernelGGL(clip_value_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, stream, (bf16_t*)grad, n, clip);
  return eagle_check_launch();
}

extern "C" int eagle_add_f32_bf16(const float* a, const void* b, void* out, int64_t n, hipStream_t stream) {
  if (!a || !out) return EAGLE_ERR_ARG;
  if (n < 1 || ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15))
    return EAGLE_ERR_SHAPE;
  const int64_t nthreads = (n + 7) / 8;
  hipLaunchKernelGGL(add_f32_bf16_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, stream, a, (const bf16_t*)b,
                     (bf16_t*)out, n);
  return eagle_check_launch();
}

extern "C" int eagle_scale_bf16(void* x, int64_t n, const float* scale, hipStream_t stream) {
  if (!x || !scale) return EAGLE_ERR_ARG;
  if (n < 1 || (reinterpret_cast<uintptr_t>(x) & 15)) return EAGLE_ERR_SHAPE;
  const int64_t nthreads = (n + 7) / 8;
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, stream, (bf16_t*)x, n, scale);
  return eagle_check_launch();
}
