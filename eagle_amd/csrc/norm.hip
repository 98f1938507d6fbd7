// LlamaRMSNorm forward/backward (eagle/llama2.py:321-335), bf16 rows, fp32 statistics.
//   y = w * bf16( x_f32 * rsqrt(mean(x_f32^2) + eps) )
// HBM-bound: one block per row, 16-byte loads, one pass (the row stays in registers).
#include "common.h"

#define NORM_THREADS 256
#define NORM_MAX_CHUNKS 4  // D <= 256*8*4 = 8192

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(NORM_THREADS) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                                  bf16_t* __restrict__ y, float* __restrict__ rstd_out,
                                                                  int D, float eps) {
  __shared__ float sh[4];
  const int64_t row = blockIdx.x;
  const int nch = D >> 3;
  uint4 v[NORM_MAX_CHUNKS];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NORM_MAX_CHUNKS; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < nch) {
      v[i] = *reinterpret_cast<const uint4*>(x + row * D + c * 8);
      const uint32_t* u = reinterpret_cast<const uint32_t*>(&v[i]);
#pragma unroll
      for (int j = 0; j < 4; ++j) { float a = lo_bf(u[j]), b = hi_bf(u[j]); ss += a * a + b * b; }
    }
  }
  ss = block_sum_256(ss, sh);
  const float rstd = rsqrtf(ss / (float)D + eps);
  if (rstd_out && threadIdx.x == 0) rstd_out[row] = rstd;
#pragma unroll
  for (int i = 0; i < NORM_MAX_CHUNKS; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < nch) {
      const uint4 wv = *reinterpret_cast<const uint4*>(w + c * 8);
      const uint32_t* u = reinterpret_cast<const uint32_t*>(&v[i]);
      const uint32_t* wu = reinterpret_cast<const uint32_t*>(&wv);
      uint4 o;
      uint32_t* ou = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        ou[j] = pack2bf(lo_bf(wu[j]) * rbf(lo_bf(u[j]) * rstd), hi_bf(wu[j]) * rbf(hi_bf(u[j]) * rstd));
      *reinterpret_cast<uint4*>(y + row * D + c * 8) = o;
    }
  }
}

// backward: given dy, x, w, rstd:  xhat = x*rstd ; g = dy*w
//   dx = rstd * (g - xhat * mean(g*xhat)) ;  dw partial per row block -> dw_part[blk][D] (fp32)
// one block handles ROWS_PER_BLOCK consecutive rows so the dw partials stay small.
__global__ __launch_bounds__(NORM_THREADS) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                  const bf16_t* __restrict__ w, const float* __restrict__ rstd,
                                                                  const bf16_t* __restrict__ res, bf16_t* __restrict__ dx,
                                                                  float* __restrict__ dw_part, int N, int D, int rows_per_block) {
  __shared__ float sh[4];
  const int nch = D >> 3;
  float dwacc[NORM_MAX_CHUNKS][8];
#pragma unroll
  for (int i = 0; i < NORM_MAX_CHUNKS; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) dwacc[i][j] = 0.f;
  const int r0 = blockIdx.x * rows_per_block;
  for (int rr = 0; rr < rows_per_block; ++rr) {
    const int64_t row = r0 + rr;
    if (row >= N) break;
    const float rs = rstd[row];
    float g[NORM_MAX_CHUNKS][8], xh[NORM_MAX_CHUNKS][8];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NORM_MAX_CHUNKS; ++i) {
      const int c = threadIdx.x + i * NORM_THREADS;
      if (c < nch) {
        const uint4 dv = *reinterpret_cast<const uint4*>(dy + row * D + c * 8);
        const uint4 xv = *reinterpret_cast<const uint4*>(x + row * D + c * 8);
        const uint4 wv = *reinterpret_cast<const uint4*>(w + c * 8);
        const uint32_t* du = reinterpret_cast<const uint32_t*>(&dv);
        const uint32_t* xu = reinterpret_cast<const uint32_t*>(&xv);
        const uint32_t* wu = reinterpret_cast<const uint32_t*>(&wv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float d0 = lo_bf(du[j]), d1 = hi_bf(du[j]);
          xh[i][2 * j] = lo_bf(xu[j]) * rs; xh[i][2 * j + 1] = hi_bf(xu[j]) * rs;
          g[i][2 * j] = d0 * lo_bf(wu[j]); g[i][2 * j + 1] = d1 * hi_bf(wu[j]);
          dwacc[i][2 * j] += d0 * xh[i][2 * j]; dwacc[i][2 * j + 1] += d1 * xh[i][2 * j + 1];
          dot += g[i][2 * j] * xh[i][2 * j] + g[i][2 * j + 1] * xh[i][2 * j + 1];
        }
      }
    }
    dot = block_sum_256(dot, sh) / (float)D;
#pragma unroll
    for (int i = 0; i < NORM_MAX_CHUNKS; ++i) {
      const int c = threadIdx.x + i * NORM_THREADS;
      if (c < nch) {
        uint4 o, rv = make_uint4(0, 0, 0, 0);
        if (res) rv = *reinterpret_cast<const uint4*>(res + row * D + c * 8);   // gradient arriving over the residual branch
        const uint32_t* ru = reinterpret_cast<const uint32_t*>(&rv);
        uint32_t* ou = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          ou[j] = pack2bf(rs * (g[i][2 * j] - xh[i][2 * j] * dot) + lo_bf(ru[j]),
                          rs * (g[i][2 * j + 1] - xh[i][2 * j + 1] * dot) + hi_bf(ru[j]));
        *reinterpret_cast<uint4*>(dx + row * D + c * 8) = o;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NORM_MAX_CHUNKS; ++i) {
    const int c = threadIdx.x + i * NORM_THREADS;
    if (c < nch) {
      float* o = dw_part + (int64_t)blockIdx.x * D + c * 8;
      *reinterpret_cast<float4*>(o) = make_float4(dwacc[i][0], dwacc[i][1], dwacc[i][2], dwacc[i][3]);
      *reinterpret_cast<float4*>(o + 4) = make_float4(dwacc[i][4], dwacc[i][5], dwacc[i][6], dwacc[i][7]);
    }
  }
}

// dw[d] (+)= sum_b dw_part[b][d]
// A workgroup takes 64 columns: thread (cx, ry) sums the partial rows ry, ry + 16, ... of its 4 columns (256-byte coalesced row
// segments, 16 loads in flight per column group), the 16 row groups meet in LDS.  (One thread per column walking all nb rows --
// the first form -- ran 16 workgroups of serial loads: 60 us for 4 MB.)  D % 4 == 0.
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ part, float* __restrict__ out, int nb, int D, int accumulate) {
  __shared__ float4 sh[16][16];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int d = blockIdx.x * 64 + cx * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (d < D) {
    for (int b = ry; b < nb; b += 16) {
      const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)b * D + d);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  sh[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && d < D) {
    for (int r = 1; r < 16; ++r) { const float4 v = sh[r][cx]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    float4* o = reinterpret_cast<float4*>(out + d);
    if (accumulate) { const float4 old = *o; s.x += old.x; s.y += old.y; s.z += old.z; s.w += old.w; }
    *o = s;
  }
}

extern "C" int eagle_rmsnorm_fwd_bf16(const void* x, const void* w, void* y, float* rstd_out, int N, int D, float eps,
                                      hipStream_t stream) {
  if (!x || !w || !y) return EAGLE_ERR_ARG;
  if (N < 1 || D < 8 || (D & 7) || D > NORM_THREADS * 8 * NORM_MAX_CHUNKS) return EAGLE_ERR_SHAPE;
  hipLaunchKernelGGL(rmsnorm_fwd_kernel, dim3(N), dim3(NORM_THREADS), 0, stream, (const bf16_t*)x, (const bf16_t*)w,
                     (bf16_t*)y, rstd_out, D, eps);
  return eagle_check_launch();
}

#define NORM_BWD_ROWS 4                 // rows per workgroup: 1024 workgroups at 4096 tokens (16 rows: 256 workgroups walking rows serially, 70 us)
extern "C" int64_t eagle_rmsnorm_bwd_workspace_bytes(int N, int D) {
  return (int64_t)((N + NORM_BWD_ROWS - 1) / NORM_BWD_ROWS) * D * sizeof(float);
}

extern "C" int eagle_rmsnorm_bwd_bf16(const void* dy, const void* x, const void* w, const float* rstd, const void* res, void* dx,
                                      float* dw, int accumulate_dw, void* workspace, int N, int D, hipStream_t stream) {
  if (!dy || !x || !w || !rstd || !dx || !dw || !workspace) return EAGLE_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(dw) | reinterpret_cast<uintptr_t>(workspace)) & 15) return EAGLE_ERR_SHAPE;   // float4 sums
  if (N < 1 || D < 8 || (D & 7) || D > NORM_THREADS * 8 * NORM_MAX_CHUNKS) return EAGLE_ERR_SHAPE;
  const int nb = (N + NORM_BWD_ROWS - 1) / NORM_BWD_ROWS;
  hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3(nb), dim3(NORM_THREADS), 0, stream, (const bf16_t*)dy, (const bf16_t*)x,
                     (const bf16_t*)w, rstd, (const bf16_t*)res, (bf16_t*)dx, (float*)workspace, N, D, NORM_BWD_ROWS);
  int rc = eagle_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((D + 63) / 64), dim3(256), 0, stream, (const float*)workspace, dw, nb, D,
                     accumulate_dw);
  return eagle_check_launch();
}
