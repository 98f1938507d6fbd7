// Large-M bf16 GEMM on MFMA for the training step and prompt prefill (gfx950).
//
//   C[M,N] = epilogue( A[M,K] . B[N,K]^T )       both operands K-contiguous ("NT"), fp32 accumulate
//
// Replaces the nn.Linear forward calls of eagle/llama2.py (fc :542, q/k/v/o :222-224/:280, MLP
// :317) and of the frozen lm_head (eagle/train.py:107,:110) at training sizes, and -- with
// transposed operands prepared by eagle_transpose_bf16 -- their autograd backward GEMMs
// (dgrad: dX = dY . W ; wgrad: dW = dY^T . X).
//
// Structure: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave as 4x4
// v_mfma_f32_16x16x32_bf16 tiles), BK = 64, operands staged HBM -> LDS with 16-byte
// global_load_lds (no VGPR round trip), double buffered; the LDS image is XOR-swizzled on the
// SOURCE address (LDS destination of an LDS-DMA is lane-linear) with the same involution on the
// ds_read_b128 side, which makes the fragment reads bank-conflict free.  The weight tile is the
// MFMA A operand and the activation tile the B operand, so each lane ends up with 4 consecutive
// output columns of one row (8-byte bf16 / 16-byte fp32 stores).  Workgroup ids are remapped so
// that consecutive ids on one XCD walk down M within a fixed N panel (L2 reuse of the B panel).
#include "common.h"
#include <stdlib.h>

#define GM_BM 128
#define GM_BN 128
#define GM_BK 64
#define GM_THREADS 256

struct GemmArgs {
  const bf16_t* A; int64_t lda;
  const bf16_t* B; int64_t ldb;
  void* C; int64_t ldc;
  const bf16_t* bias;          // [N] or null
  const bf16_t* res; int64_t ldr;  // [M,N] bf16 or null (added after bias, each rounded like eager ops)
  int M, N, K;
  int out_f32;                 // 0: bf16 C, 1: fp32 C
  int accumulate;              // C += result (same dtype as C)
  int tiles_m, tiles_n;
};

__device__ __forceinline__ void gm_stage(const GemmArgs& p, bf16_t* ldsA, bf16_t* ldsB, int m0, int n0, int kt, int tid) {
  const int wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int L = i * GM_THREADS + tid;       // 16-byte slot in the [128][8] image
    const int row = L >> 3, slot = L & 7;
    const int chunk = slot ^ (row & 7);
    int ra = m0 + row; ra = ra < p.M ? ra : p.M - 1;
    int rb = n0 + row; rb = rb < p.N ? rb : p.N - 1;
    const bf16_t* ga = p.A + (int64_t)ra * p.lda + kt * GM_BK + chunk * 8;
    const bf16_t* gb = p.B + (int64_t)rb * p.ldb + kt * GM_BK + chunk * 8;
    bf16_t* da = ldsA + (i * GM_THREADS + wave * 64) * 8;   // wave-uniform base; HW adds lane*16
    bf16_t* db = ldsB + (i * GM_THREADS + wave * 64) * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,
                                     (__attribute__((address_space(3))) void*)da, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gb,
                                     (__attribute__((address_space(3))) void*)db, 16, 0, 0);
  }
}

__global__ __launch_bounds__(GM_THREADS) void gemm_nt_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][GM_BM * GM_BK];  // [buf][A|B] 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware remap: ids b, b+8, ... share an XCD; give each XCD a contiguous run of tiles,
  // and walk M fastest inside the run so that neighbours reuse the same weight (B) panel.
  const int nwg = p.tiles_m * p.tiles_n;
  const int orig = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tn = wgid / p.tiles_m, tm = wgid - tn * p.tiles_m;
  const int m0 = tm * GM_BM, n0 = tn * GM_BN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / GM_BK;
  gm_stage(p, lds[0][0], lds[0][1], m0, n0, 0, tid);
  __syncthreads();  // hipcc drains the LDS-DMA (vmcnt(0)) in front of the barrier
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gm_stage(p, lds[cur ^ 1][0], lds[cur ^ 1][1], m0, n0, kt + 1, tid);
    const bf16_t* la = lds[cur][0];
    const bf16_t* lb = lds[cur][1];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 xf[4], wf[4];
      const int c = 4 * s + g;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra = wm * 64 + i * 16 + r16;
        const int rb = wn * 64 + i * 16 + r16;
        xf[i] = *reinterpret_cast<const bf16x8*>(la + ra * GM_BK + ((c ^ (ra & 7)) << 3));
        wf[i] = *reinterpret_cast<const bf16x8*>(lb + rb * GM_BK + ((c ^ (rb & 7)) << 3));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: acc[i][j][r] = C[m0 + wm*64 + i*16 + r16][n0 + wn*64 + j*16 + 4g + r]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + r16;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nb = n0 + wn * 64 + j * 16 + 4 * g;
      if (nb >= p.N) continue;
      float y[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int n = nb + rr;
        float v = acc[i][j][rr];
        if (p.bias && n < p.N) v += bf2f(p.bias[n]);
        if (p.res && n < p.N) v = rbf(v) + bf2f(p.res[(int64_t)m * p.ldr + n]);
        y[rr] = v;
      }
      const bool full = (nb + 3 < p.N) && ((p.ldc & 3) == 0);
      if (p.out_f32) {
        float* o = (float*)p.C + (int64_t)m * p.ldc + nb;
        if (full) {
          float4 v = make_float4(y[0], y[1], y[2], y[3]);
          if (p.accumulate) { const float4 old = *reinterpret_cast<const float4*>(o); v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w; }
          *reinterpret_cast<float4*>(o) = v;
        } else {
          for (int rr = 0; rr < 4; ++rr) if (nb + rr < p.N) o[rr] = p.accumulate ? o[rr] + y[rr] : y[rr];
        }
      } else {
        bf16_t* o = (bf16_t*)p.C + (int64_t)m * p.ldc + nb;
        if (full) {
          if (p.accumulate) {
            const uint2 old = *reinterpret_cast<const uint2*>(o);
            y[0] += lo_bf(old.x); y[1] += hi_bf(old.x); y[2] += lo_bf(old.y); y[3] += hi_bf(old.y);
          }
          *reinterpret_cast<uint2*>(o) = make_uint2(pack2bf(y[0], y[1]), pack2bf(y[2], y[3]));
        } else {
          for (int rr = 0; rr < 4; ++rr) if (nb + rr < p.N) o[rr] = f2bf(p.accumulate ? bf2f(o[rr]) + y[rr] : y[rr]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Decode batches (32-192 rows against a very tall weight matrix: the lm_head of a batch-32 tree step, eagle/train.py:
// 215-228 applied to draft rows): the launch is a weight stream, and the kernel above keeps only ONE K step in flight
// per workgroup and drains it at every barrier, so a K step costs a memory latency (1.9 us measured: 4.5 TB/s).
// Same tile math, different pipeline: BM = 32 / 64 / 128 activation rows (no MFMA work on padding rows), BN = 128
// weight rows, S LDS stages filled by LDS-DMA that stay in flight across the (single) barrier of a K step behind a
// counted vmcnt -- S-1 K steps of weights in flight per workgroup.
template <int BM, int S, int BN = GM_BN, int NW = 4>
__global__ __launch_bounds__(NW * 64) void gemm_nt_rows_kernel(GemmArgs p) {
  constexpr int NT = NW * 64;
  constexpr int WM = BM == 128 ? 2 : 1, WN = NW / WM;           // wave grid
  constexpr int I = BM / WM / 16, J = BN / WN / 16;              // 16x16 tiles per wave
  constexpr int A_OPS = BM * 8 / NT, B_OPS = BN * 8 / NT, OPS = A_OPS + B_OPS;
  constexpr int STAGE = (BM + BN) * GM_BK;                       // elements
  extern __shared__ __attribute__((aligned(16))) bf16_t dyn_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int wm = WM == 2 ? wave / WN : 0, wn = WM == 2 ? (wave % WN) : wave;
  const int tn = blockIdx.x / p.tiles_m, tm = blockIdx.x - tn * p.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;

  auto stage = [&](int buf, int kt) {
    bf16_t* la = dyn_lds + buf * STAGE;
    bf16_t* lb = la + BM * GM_BK;
#pragma unroll
    for (int i = 0; i < A_OPS; ++i) {
      const int L = i * NT + tid, row = L >> 3, slot = L & 7, chunk = slot ^ (row & 7);
      int ra = m0 + row; ra = ra < p.M ? ra : p.M - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.A + (int64_t)ra * p.lda + kt * GM_BK + chunk * 8),
                                       (__attribute__((address_space(3))) void*)(la + (i * NT + wave * 64) * 8), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_OPS; ++i) {
      const int L = i * NT + tid, row = L >> 3, slot = L & 7, chunk = slot ^ (row & 7);
      int rb = n0 + row; rb = rb < p.N ? rb : p.N - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.B + (int64_t)rb * p.ldb + kt * GM_BK + chunk * 8),
                                       (__attribute__((address_space(3))) void*)(lb + (i * NT + wave * 64) * 8), 16, 0, 0);
    }
  };

  f32x4 acc[I][J];
#pragma unroll
  for (int i = 0; i < I; ++i)
#pragma unroll
    for (int j = 0; j < J; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nk = p.K / GM_BK;
  // prologue: S-1 stages in flight (past the end: re-reads of the last K step, never multiplied)
#pragma unroll
  for (int s = 0; s < S - 1; ++s) stage(s, s < nk ? s : nk - 1);
  for (int kt = 0; kt < nk; ++kt) {
    // this thread's share of stage kt has landed when at most S-2 younger stages are outstanding ...
    if (S == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (S == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * OPS) : "memory");
    __builtin_amdgcn_s_barrier();      // ... everybody's has, and everybody is done reading the buffer refilled next
    const int nxt = kt + S - 1;
    stage(nxt % S, nxt < nk ? nxt : nk - 1);
    const bf16_t* la = dyn_lds + (kt % S) * STAGE;
    const bf16_t* lb = la + BM * GM_BK;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 xf[I], wf[J];
      const int c = 4 * s + g;
#pragma unroll
      for (int i = 0; i < I; ++i) {
        const int ra = wm * (I * 16) + i * 16 + r16;
        xf[i] = *reinterpret_cast<const bf16x8*>(la + ra * GM_BK + ((c ^ (ra & 7)) << 3));
      }
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int rb = wn * (J * 16) + j * 16 + r16;
        wf[j] = *reinterpret_cast<const bf16x8*>(lb + rb * GM_BK + ((c ^ (rb & 7)) << 3));
      }
#pragma unroll
      for (int i = 0; i < I; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (unused) refills land before the LDS is released
  // epilogue: acc[i][j][r] = C[m0 + wm*I*16 + i*16 + r16][n0 + wn*J*16 + j*16 + 4g + r]; bf16 output (+bias) (+res)
#pragma unroll
  for (int i = 0; i < I; ++i) {
    const int m = m0 + wm * (I * 16) + i * 16 + r16;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int nb = n0 + wn * (J * 16) + j * 16 + 4 * g;
      if (nb >= p.N) continue;
      float y[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int n = nb + rr;
        float v = acc[i][j][rr];
        if (p.bias && n < p.N) v += bf2f(p.bias[n]);
        if (p.res && n < p.N) v = rbf(v) + bf2f(p.res[(int64_t)m * p.ldr + n]);
        y[rr] = v;
      }
      bf16_t* o = (bf16_t*)p.C + (int64_t)m * p.ldc + nb;
      if ((nb + 3 < p.N) && ((p.ldc & 3) == 0)) *reinterpret_cast<uint2*>(o) = make_uint2(pack2bf(y[0], y[1]), pack2bf(y[2], y[3]));
      else for (int rr = 0; rr < 4; ++rr) if (nb + rr < p.N) o[rr] = f2bf(y[rr]);
    }
  }
}

static int g_rows_cfg = 0;     // 0: by shape; else (BM << 4 | S) forced (eagle_debug_set_gemm8 values >= 0x100 carry it)
template <int BM, int S, int BN = GM_BN, int NW = 4>
static int launch_rows(GemmArgs& a, hipStream_t stream) {
  constexpr int bytes = S * (BM + BN) * GM_BK * 2;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gemm_nt_rows_kernel<BM, S, BN, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return EAGLE_ERR_LAUNCH;
    attr = true;
  }
  a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm_nt_rows_kernel<BM, S, BN, NW>), dim3(a.tiles_m * a.tiles_n), dim3(NW * 64), bytes, stream, a);
  return eagle_check_launch();
}

static inline bool gm_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// gemm_mfma8.hip: 256x256 tiles, DMA in flight across barriers
int eagle_launch_gemm_nt8(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* bias,
                          const void* res, int64_t ldr, int M, int N, int K, int out_f32, int accumulate, int ksplit,
                          void* workspace, hipStream_t stream);
static int g_gemm8_mode = 2;    // 0: never, 1: whenever legal, 2: by shape (default); eagle_debug_set_gemm8
int g_gemm8_sched = 2;          // diagnostics: main-loop schedule of the 256x256 NT kernel (gemm_mfma8.hip); mode 0x20000 | sched
int g_gemm8_nt = 0;             // diagnostics: non-temporal output stores of the 256x256 kernel, 0 by size, 1 always, 2 never; mode 0x30000 | v
int g_gemm8_gm = 0;             // diagnostics: row tiles per group of the 256x256 kernel's tile order (0: default 8); mode 0x10000 | gm
extern "C" void eagle_debug_set_gemm8(int mode) {
  if (mode >= 0x30000) { g_gemm8_nt = mode & 0xff; return; }
  if (mode >= 0x20000) { g_gemm8_sched = mode & 0xff; return; }
  if (mode >= 0x10000) { g_gemm8_gm = mode & 0xff; return; }
  if (mode >= 0x100) { g_rows_cfg = mode & 0xfff; return; }     // 0x100 | BM<<4... see the header
  g_gemm8_mode = mode;
}
static bool use_gemm8(int M, int N, int K) {
  if (g_gemm8_mode == 0 || K < 2 * GM_BK) return false;
  if (g_gemm8_mode == 1) return true;
  // decode batches (a row tile that is mostly padding) stream the weights once whatever the tile: the 128x128
  // kernel's two workgroups per CU and finer tail win there (lm_head 151936 x 4096 at 128 rows: 345 vs 540 us,
  // at 32 rows 304 vs 449 us; tools/lm_head_bench.py)
  if (M <= 192) return false;
  const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
  return t256 >= 256;            // at least one full round of 256x256 tiles over the 256 CUs (measured: tools/gemm_bench.py)
}

// See include/eagle_hip.h for the contract.
extern "C" int eagle_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                  const void* bias, const void* res, int64_t ldr, int M, int N, int K,
                                  int out_f32, int accumulate, hipStream_t stream) {
  if (!A || !B || !C) return EAGLE_ERR_ARG;
  if (M < 1 || N < 1 || K < GM_BK || (K % GM_BK)) return EAGLE_ERR_SHAPE;
  if ((lda & 7) || (ldb & 7) || lda < K || ldb < K || ldc < N || (res && ldr < N)) return EAGLE_ERR_SHAPE;
  if (!gm_al16(A) || !gm_al16(B) || !gm_al16(C)) return EAGLE_ERR_SHAPE;
  if (use_gemm8(M, N, K))
    return eagle_launch_gemm_nt8(A, lda, B, ldb, C, ldc, bias, res, ldr, M, N, K, out_f32, accumulate, 1, nullptr, stream);
  GemmArgs a;
  // decode batches against a tall weight matrix: the deep-pipelined row kernels (bf16 output, no accumulate)
  if (g_gemm8_mode == 2 && (M <= 32 || (g_rows_cfg & 0xff)) && M <= 192 && N >= 8192 && K >= 4 * GM_BK && !out_f32 && !accumulate &&
      g_rows_cfg != 0xfff) {
    a.A = (const bf16_t*)A; a.lda = lda; a.B = (const bf16_t*)B; a.ldb = ldb; a.C = C; a.ldc = ldc;
    a.bias = (const bf16_t*)bias; a.res = (const bf16_t*)res; a.ldr = ldr;
    a.M = M; a.N = N; a.K = K; a.out_f32 = 0; a.accumulate = 0;
    const int cfg = g_rows_cfg & 0xff;
    switch (cfg) {                                   // diagnostics: forced configuration (BM / 16) << 4 | S
      case 0x23: return launch_rows<32, 3>(a, stream);
      case 0x24: return launch_rows<32, 4>(a, stream);
      case 0x43: return launch_rows<64, 3>(a, stream);
      case 0x83: return launch_rows<128, 3>(a, stream);
      case 0x93: return launch_rows<128, 3, 256, 8>(a, stream);      // 128 x 256 tiles, 8 waves
      default: break;
    }
    // measured on the 151936 x 4096 head from cold caches (tools/rows_sweep.py): 32 rows 253 us against 265 us for the
    // 128x128 kernel; at 64+ rows no configuration (deeper pipelines, 128x256 tiles) beats the 128x128 kernel's 295 us
    if (M <= 32) return launch_rows<32, 3>(a, stream);
  }
  a.A = (const bf16_t*)A; a.lda = lda; a.B = (const bf16_t*)B; a.ldb = ldb; a.C = C; a.ldc = ldc;
  a.bias = (const bf16_t*)bias; a.res = (const bf16_t*)res; a.ldr = ldr;
  a.M = M; a.N = N; a.K = K; a.out_f32 = out_f32; a.accumulate = accumulate;
  a.tiles_m = (M + GM_BM - 1) / GM_BM; a.tiles_n = (N + GM_BN - 1) / GM_BN;
  hipLaunchKernelGGL(gemm_nt_kernel, dim3(a.tiles_m * a.tiles_n), dim3(GM_THREADS), 0, stream, a);
  return eagle_check_launch();
}

// Split-K form for few output tiles and a very long K (the lm_head dgrad: M = tokens, N = D, K = V):
// ksplit slices of K run as separate workgroups of the 256x256 kernel into fp32 partials
// [ksplit][M][N] in `workspace` (eagle_gemm_nt_splitk_workspace_bytes), a second launch sums them
// and applies the epilogue.  ksplit <= 1 is eagle_gemm_nt_bf16.
extern "C" int64_t eagle_gemm_nt_splitk_workspace_bytes(int M, int N, int ksplit) {
  return ksplit > 1 ? (int64_t)ksplit * M * N * 4 : 0;
}
extern "C" int eagle_gemm_nt_splitk_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                         const void* bias, const void* res, int64_t ldr, int M, int N, int K, int out_f32,
                                         int accumulate, int ksplit, void* workspace, hipStream_t stream) {
  if (ksplit <= 1) return eagle_gemm_nt_bf16(A, lda, B, ldb, C, ldc, bias, res, ldr, M, N, K, out_f32, accumulate, stream);
  if (!A || !B || !C || !workspace) return EAGLE_ERR_ARG;
  if (M < 1 || N < 1 || K < 2 * GM_BK || (K % GM_BK)) return EAGLE_ERR_SHAPE;
  if ((lda & 7) || (ldb & 7) || lda < K || ldb < K || ldc < N || (res && ldr < N)) return EAGLE_ERR_SHAPE;
  if (!gm_al16(A) || !gm_al16(B) || !gm_al16(C) || !gm_al16(workspace)) return EAGLE_ERR_SHAPE;
  return eagle_launch_gemm_nt8(A, lda, B, ldb, C, ldc, bias, res, ldr, M, N, K, out_f32, accumulate, ksplit, workspace, stream);
}

// out[C, R_pad] = in[R, C]^T (bf16), 64x64 tiles through LDS; columns [R, R_pad) are zero-filled so
// that the result can be a K-padded GEMM operand (wgrad over a token count that is no multiple of 64).
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, int64_t ldi, bf16_t* __restrict__ out,
                                                            int64_t ldo, int R, int C, int R_pad) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? in[(int64_t)r * ldi + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < R_pad) out[(int64_t)c * ldo + r] = tile[tx][i];
  }
}

extern "C" int eagle_transpose_bf16(const void* in, int64_t ldi, void* out, int64_t ldo, int R, int C, int R_pad,
                                    hipStream_t stream) {
  if (!in || !out) return EAGLE_ERR_ARG;
  if (R_pad < R) R_pad = R;
  if (R < 1 || C < 1 || ldi < C || ldo < R_pad) return EAGLE_ERR_SHAPE;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (R_pad + 63) / 64), dim3(256), 0, stream, (const bf16_t*)in, ldi,
                     (bf16_t*)out, ldo, R, C, R_pad);
  return eagle_check_launch();
}
