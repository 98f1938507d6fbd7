// 256x256x64 bf16 "NT" GEMM for the big training GEMMs (lm_head forward on target and predict,
// gate|up, their wgrad shapes): 8 waves, LDS-DMA staging kept in flight ACROSS barriers.
//
//   C[M,N] = epilogue( A[M,K] . B[N,K]^T )      same contract as gemm_nt_kernel (gemm_mfma.hip)
//
// Why a second kernel: the 128x128 kernel drains its LDS-DMA (vmcnt(0)) at one barrier per K step
// and tops out near 0.9 PFLOP/s whatever is tuned around it.  The basic structure (PINGPONG = false, the first form of round 1,
// not launched any more; every launch uses the ping-pong schedules described inside the kernel): every K tile is cut into four
// phases; each phase issues ONE 16 KB half-tile of a later K tile, multiplies one quadrant of the
// wave's 128x64 output (16 MFMAs) and ends in a raw s_barrier that waits for LDS traffic only.  The
// DMA is retired once per K tile by a counted vmcnt that leaves the two youngest half-tiles in
// flight, so HBM/L2 latency is hidden behind four phases of MFMA work instead of being exposed at
// a barrier.
//
// Wave (wr, wc), wr in {0,1}, wc in {0..3}: rows wr*128.., columns wc*64.. of the 256x256 tile, so
// a wave reads A half-tile `wr` and B half-tile `wc>>1` only.  Phases of K tile t (buffer b = t&1):
//   P1  read A(mh=0) 8x b128 + B(nh=0) 4x   stage A0(t+1) -> buffer b^1     MFMA quadrant (0,0)
//   P2  read B(nh=1) 4x                      stage A1(t+1) -> buffer b^1     MFMA quadrant (0,1)
//   P3  read A(mh=1) 8x                      stage B0(t+2) -> buffer b       MFMA quadrant (1,1)
//   P4  (B(nh=0) is still in registers)      stage B1(t+2) -> buffer b       MFMA quadrant (1,0)
//       s_waitcnt vmcnt(4): everything but the P3/P4 half-tiles has landed = all of tile t+1
// WAR: a B half of buffer b is last read in P2, an A half in P3; each is restaged one barrier later
// at the earliest.  RAW: tile t+1 is complete at P4's wait (per thread), published by P4's barrier,
// first read in P1 of t+1.  Staging past the last K tile re-reads the last tile (never consumed):
// the loop body stays branch-free so the vmcnt arithmetic holds.
// LDS image of a half-tile: [128 rows][8 x 16 B], 16-byte chunk XOR (row & 7) -- applied on the
// per-lane SOURCE address of the DMA (its LDS destination is lane-linear) and on the ds_read side.
#include "common.h"
#include <stdlib.h>

#define G8_BM 256
#define G8_BN 256
#define G8_BK 64
#define G8_THREADS 512
#ifndef G8_PRIO_M
#define G8_PRIO_M 1
#define G8_PRIO_R 0
#endif
#ifndef G8_SCHED
#define G8_SCHED 2                       // main-loop schedule: 2 = four slots per K tile (see the kernel); 0 / 1 = the eight-slot forms
#endif
#define G8_HALF (128 * G8_BK)           // elements per half-tile (16 KB)

struct Gemm8Args {
  const bf16_t* A; int64_t lda;
  const bf16_t* B; int64_t ldb;
  void* C; int64_t ldc;
  const bf16_t* bias;
  const bf16_t* res; int64_t ldr;
  int M, N, K;
  int out_f32, accumulate;
  int tiles_m, tiles_n;
  int gm;              // row tiles per group of the tile order
  int ksplit;          // > 1: grid = tiles x ksplit, slice s multiplies K tiles [s*kchunk, ...) into partial[s]
  int kchunk;          // K tiles per slice
  float* partial;      // [ksplit][M][N] fp32 (split-K only)
  int tn;              // operands stored K-major: A [K, M] (row stride lda), B [K, N] (row stride ldb)
  int nt_store;        // bf16 output written with non-temporal stores (outputs too large to be re-read from cache)
};

// Operand staging by buffer_load ... lds: the row/chunk part of every lane's source address never
// changes along K, so it is computed once per thread (32-bit byte offsets from the block's first row,
// the XOR chunk swizzle and the row clamp folded in) and a K tile only moves the scalar offset --
// no per-piece address arithmetic in the main loop, and half the address bits of a 64-bit global load.
struct G8Src {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff[2][2];                 // [half-tile][piece]
};
__device__ __forceinline__ G8Src g8_src(const bf16_t* X, int64_t ld, int nrows, int r0, int K, int tid) {
  G8Src s;
  const int avail = nrows - r0;                                  // >= 1 (launcher: every tile has a valid first row)
  const int last = avail < 256 ? avail - 1 : 255;
  const bf16_t* base = X + (int64_t)r0 * ld;
  const uint64_t bytes = ((uint64_t)last * (uint64_t)ld + (uint64_t)K) * 2ull;
  s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)bytes), 0x00020000);
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int L = i * G8_THREADS + tid;      // 16-byte slot in the [128][8] image
      const int row = L >> 3, slot = L & 7;
      const int chunk = slot ^ (row & 7);
      int gr = h * 128 + row;
      gr = gr < last ? gr : last;
      s.voff[h][i] = (uint32_t)(((uint64_t)gr * (uint64_t)ld + (uint64_t)(chunk * 8)) * 2ull);
    }
  return s;
}
// K-major operands ("TN": C = A^T . B with A [K, M], B [K, N] -- the weight-gradient GEMMs dW = dY^T . X read dY and X as
// they lie in memory, no transpose pass): a half-tile is [64 k rows][128 columns] = 64 rows of 256 B, staged row-major with
// the 16-byte chunk swizzle of the training attention's V tile (attn_train.hip a2_off); the MFMA fragments (8 consecutive k
// of one column per lane) come out of it through transposing LDS reads, ds_read_b64_tr_b16, two per fragment.
__device__ __forceinline__ int g8_tn_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
__device__ __forceinline__ G8Src g8_src_tn(const bf16_t* X, int64_t ld, int c0, int K, int tid) {
  G8Src s;
  const bf16_t* base = X + c0;
  const uint64_t bytes = ((uint64_t)(K - 1) * (uint64_t)ld + 256ull) * 2ull;
  s.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)bytes), 0x00020000);
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int L = i * G8_THREADS + tid;      // 16-byte slot in the [64][16] image
      const int row = L >> 4, slot = L & 15;
      const int chunk = slot ^ (((row & 3) << 2) | ((row >> 2) & 3));
      s.voff[h][i] = (uint32_t)(((uint64_t)row * (uint64_t)ld + (uint64_t)(h * 128 + chunk * 8)) * 2ull);
    }
  return s;
}
__device__ __forceinline__ void g8_stage_tn(const G8Src& s, int h, int kt, uint32_t ktile_bytes, bf16_t* lds, int tid) {
  const int wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    bf16_t* d = lds + (i * G8_THREADS + wave * 64) * 8;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rsrc, (__attribute__((address_space(3))) void*)d, 16, s.voff[h][i],
                                             kt * ktile_bytes, 0, 0);
  }
}
typedef __attribute__((ext_vector_type(4))) short g8_s4;
// NATURAL = false: lane group g holds k = 32s + {4g..4g+3, 16+4g..16+4g+3} (both operands K-major: any common k order will do);
// NATURAL = true: k = 32s + 8g .. 8g+7, the order of a row-major operand's ds_read_b128 fragment (the other operand is row-major).
template <bool NATURAL>
__device__ __forceinline__ bf16x8 g8_tn_frag(const bf16_t* img, int tile, int s, int g, int tq, int tp) {
  const char* b = reinterpret_cast<const char*>(img);
  const int r0 = 32 * s + (NATURAL ? 8 : 4) * g + tq, r1 = r0 + (NATURAL ? 4 : 16);
  const g8_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) g8_s4*)(b + g8_tn_off(r0, 2 * tile + (tp >> 1)) + 8 * (tp & 1)));
  const g8_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) g8_s4*)(b + g8_tn_off(r1, 2 * tile + (tp >> 1)) + 8 * (tp & 1)));
  union { struct { g8_s4 a, b; } q; bf16x8 v; } u;
  u.q.a = lo; u.q.b = hi;
  return u.v;
}

// half-tile `h` at K tile kt -> lds (wave-uniform LDS base in M0; the hardware adds lane*16)
__device__ __forceinline__ void g8_stage(const G8Src& s, int h, int kt, bf16_t* lds, int tid) {
  const int wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    bf16_t* d = lds + (i * G8_THREADS + wave * 64) * 8;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rsrc, (__attribute__((address_space(3))) void*)d, 16, s.voff[h][i],
                                             kt * (G8_BK * 2), 0, 0);
  }
}

// TA / TB: operand A / B is stored K-major ([K, M] / [K, N]) and read through transposing LDS reads
template <bool PINGPONG, int DBG = 0, bool TA = false, bool TB = false, int SCHED = G8_SCHED>
__global__ __launch_bounds__(G8_THREADS) void gemm_nt8_kernel(Gemm8Args p) {
  // ONE shared array (a second LDS object makes hipcc drain the DMA before every ds_read):
  // [buffer 2][A0, A1, B0, B1][128 x 64]  = 128 KB
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * 4 * G8_HALF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;

  const int ntile = p.tiles_m * p.tiles_n;
  const int nwg = ntile * p.ksplit;
  const int orig = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int wgid0 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int kslice = wgid0 / ntile;                 // slices of one tile sit ntile ids apart
  const int wgid = wgid0 - kslice * ntile;
  // grouped order: consecutive ids sweep `gm` row tiles, then move one column tile on, so the 32 tiles an
  // XCD runs at a time form a gm x (32/gm) block (fewer distinct A/B slices per K step in its L2)
  const int per_group = p.gm * p.tiles_n;
  const int grp = wgid / per_group, in_grp = wgid - grp * per_group;
  const int first_m = grp * p.gm;
  const int gsz = min(p.tiles_m - first_m, p.gm);
  const int tn = in_grp / gsz, tm = first_m + (in_grp - tn * gsz);
  const int m0 = tm * G8_BM, n0 = tn * G8_BN;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk_all = p.K / G8_BK;
  const int kt0 = kslice * p.kchunk;
  const int nk = min(p.kchunk, nk_all - kt0);       // K tiles of this slice (>= 1: launcher)
  auto kt_of = [&](int t) { return (DBG & 8) ? kt0 + (t & 1) : kt0 + (t < nk ? t : nk - 1); };
  auto hbuf = [&](int buf, int which) { return lds + (buf * 4 + which) * G8_HALF; };   // which: 0 A0, 1 A1, 2 B0, 3 B1
  const G8Src srcA = TA ? g8_src_tn(p.A, p.lda, m0, p.K, tid) : g8_src(p.A, p.lda, p.M, m0, p.K, tid);
  const G8Src srcB = TB ? g8_src_tn(p.B, p.ldb, n0, p.K, tid) : g8_src(p.B, p.ldb, p.N, n0, p.K, tid);
  const uint32_t ktA = (uint32_t)(p.lda * (G8_BK * 2)), ktB = (uint32_t)(p.ldb * (G8_BK * 2));   // TN: bytes per K tile of rows
  auto stage_a = [&](int h, int t) {
    if (TA) g8_stage_tn(srcA, h, kt_of(t), ktA, hbuf(t & 1, h), tid); else g8_stage(srcA, h, kt_of(t), hbuf(t & 1, h), tid);
  };
  auto stage_b = [&](int h, int t) {
    if (TB) g8_stage_tn(srcB, h, kt_of(t), ktB, hbuf(t & 1, 2 + h), tid); else g8_stage(srcB, h, kt_of(t), hbuf(t & 1, 2 + h), tid);
  };
  const int tq = (lane & 15) >> 2, tp = lane & 3;        // transposing-read lane geometry (TN)

  // prologue: tile 0 complete, B halves of tile 1 in flight
  stage_b(0, 0); stage_b(1, 0); stage_a(0, 0); stage_a(1, 0);
  stage_b(0, 1); stage_b(1, 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  bf16x8 af[4][2], bf0[2][2], bf1[2][2];     // A: 4 m-tiles x 2 k-steps; B quadrant nh: 2 n-tiles x 2 k-steps
  const int a_row0 = r16;                                  // + mh*64 + i*16 within the wave's A half
  const int b_row0 = (wc & 1) * 64 + r16;                  // + nh*32 + j*16 within the wave's B half
  auto read_a = [&](const bf16_t* la, int mh) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = a_row0 + mh * 64 + i * 16;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (TA) af[i][s] = g8_tn_frag<!TB>(la, mh * 4 + i, s, g, tq, tp);
        else af[i][s] = *reinterpret_cast<const bf16x8*>(la + row * G8_BK + (((4 * s + g) ^ (row & 7)) << 3));
      }
    }
  };
  auto read_b = [&](const bf16_t* lb, int nh, bf16x8 (&bf)[2][2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = b_row0 + nh * 32 + j * 16;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (TB) bf[j][s] = g8_tn_frag<!TA>(lb, (wc & 1) * 4 + nh * 2 + j, s, g, tq, tp);
        else bf[j][s] = *reinterpret_cast<const bf16x8*>(lb + row * G8_BK + (((4 * s + g) ^ (row & 7)) << 3));
      }
    }
  };

#define G8_MFMA(MH, NH, BF)                                                                           \
  __builtin_amdgcn_s_setprio(G8_PRIO_M);                                                                   \
  _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                       \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                       \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
    acc[(MH) * 4 + i][(NH) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[j][s], af[i][s], acc[(MH) * 4 + i][(NH) * 2 + j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(G8_PRIO_R);

  if (PINGPONG) {
    // Ping-pong schedule: the two waves of a SIMD (wr = 0 / wr = 1 share it) run the same program one
    // barrier apart, so that while one multiplies a quadrant (16 MFMAs) the other issues its LDS
    // fragment reads and its share of the DMA -- the reads no longer sit in front of the MFMAs of
    // BOTH waves.  Rounds 1-2 (SCHED 0 / 1, kept for A/B): eight slots per K tile (R = reads/DMA, M = MFMA); group wr=1 enters one barrier late:
    //   S1 R  B(nh0) B(nh1) A(mh0) reads; stage A0(t+1) -> buffer b^1   S2 M  quadrant (0,0)
    //   S3 R  stage A1(t+1) -> buffer b^1                               S4 M  quadrant (0,1)
    //   S5 R  A(mh1) reads; stage B0(t+2) -> buffer b                   S6 M  quadrant (1,1)
    //   S7 R  stage B1(t+2) -> buffer b; vmcnt(4): tile t+1 landed      S8 M  quadrant (1,0)
    // Two DMA pieces per R slot: a piece costs its wave ~60-185 issue cycles (MI355X_MICROARCH.md,
    // 'LDS-DMA piece issue cost'), so four in one slot outlast the partner's 16 MFMAs.
    // WAR (a region is restaged >= 2 slots after its last read by EITHER group, whose reads retire
    // at the top of the following M slot): A halves of buffer b^1 were last read in S5(t-1), i.e. 4
    // (3 for the late group) slots before S1(t); B halves of buffer b are last read in S1(t), 4 (3)
    // slots before S5(t).  RAW: each wave retires its pieces of tile t+1 at its S7(t); the late group's
    // S7(t) ends one barrier before the early group's S1(t+1).
    if (wr == 1) __builtin_amdgcn_s_barrier();
    // DBG (timing experiments only, results invalid): bit 0 = no DMA in the loop, bit 1 = no LDS reads,
    // bit 2 = no barriers, bit 3 = DMA always from the first two K tiles (cache-hot), bit 4 = no vmcnt wait, bit 6 = no output stores
#define G8_BAR() do { if (!(DBG & 4)) __builtin_amdgcn_s_barrier(); } while (0)
    if (DBG & 2) { read_b(hbuf(0, 2 + (wc >> 1)), 0, bf0); read_a(hbuf(0, wr), 0); read_b(hbuf(0, 2 + (wc >> 1)), 1, bf1); }
    if (SCHED == 2) {
      // The shipped schedule (round 3): FOUR slots per K tile -- 32 MFMAs per M slot, half the barriers of the form above
      // (1.31-1.42 -> 1.39-1.50 PFLOP/s on the step's shapes, profiles/r03_gemm8_schedules.txt):
      //   S1 R  B(nh0) A(mh0) B(nh1) reads (16); stage A0, A1(t+1) -> buffer b^1; lgkmcnt(0)       S2 M  quadrants (0,0) (0,1)
      //   S3 R  A(mh1) reads (8); stage B0, B1(t+2) -> buffer b; vmcnt(4) lgkmcnt(0): t+1 landed    S4 M  quadrants (1,1) (1,0)
      // The partner's M slot is now 512 cycles, long enough for four DMA pieces in one R slot.  Every R slot ends with its
      // OWN LDS reads retired (lgkmcnt(0) before the barrier), which makes the hazards exact, with interval k = the time
      // between barriers k-1 and k, early group's S1(t) = interval 4t, late group's = 4t+1:
      // WAR: the A halves of buffer b^1 were last read in S3(t-1) (early 4t-2, late 4t-1, retired before barrier 4t-1 ends)
      // and are restaged in S1(t) (early 4t at the earliest); the B halves of buffer b are last read in S1(t) (early 4t, late
      // 4t+1, retired by the end of 4t+1) and restaged in S3(t) (early 4t+2).  RAW: every wave retires its pieces of tile
      // t+1 in its S3(t) (late group: interval 4t+3); the first read of tile t+1 is the early group's S1(t+1) = interval 4t+4.
      for (int t = 0; t < nk; ++t) {
        const bf16_t* la = hbuf(t & 1, wr);
        const bf16_t* lb = hbuf(t & 1, 2 + (wc >> 1));
        if (!(DBG & 2)) { read_b(lb, 0, bf0); read_a(la, 0); if (!(DBG & 32)) read_b(lb, 1, bf1); }
        if (!(DBG & 1)) { stage_a(0, t + 1); stage_a(1, t + 1); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        G8_BAR();
        G8_MFMA(0, 0, bf0)
        G8_MFMA(0, 1, bf1)
        G8_BAR();
        if (!(DBG & 2)) read_a(la, 1);
        if (DBG & 32) read_b(lb, 1, bf1);            // timing experiment (wrong results): 12 reads in either R slot
        if (!(DBG & 1)) { stage_b(0, t + 2); stage_b(1, t + 2); }
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        G8_BAR();
        G8_MFMA(1, 1, bf1)
        G8_MFMA(1, 0, bf0)
        G8_BAR();
      }
    } else
    for (int t = 0; t < nk; ++t) {
      const bf16_t* la = hbuf(t & 1, wr);
      const bf16_t* lb = hbuf(t & 1, 2 + (wc >> 1));
      if (!(DBG & 2)) { read_b(lb, 0, bf0); read_a(la, 0); if (SCHED == 0) read_b(lb, 1, bf1); }
      if (!(DBG & 1)) stage_a(0, t + 1);
      G8_BAR();
      G8_MFMA(0, 0, bf0)
      G8_BAR();
      if (SCHED == 1 && !(DBG & 2)) read_b(lb, 1, bf1);
      if (!(DBG & 1)) stage_a(1, t + 1);
      if (SCHED == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      G8_BAR();
      G8_MFMA(0, 1, bf1)
      G8_BAR();
      if (!(DBG & 2)) read_a(la, 1);
      if (!(DBG & 1)) stage_b(0, t + 2);
      G8_BAR();
      G8_MFMA(1, 1, bf1)
      G8_BAR();
      if (!(DBG & 1)) { stage_b(1, t + 2); if (!(DBG & 16)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
      G8_BAR();
      G8_MFMA(1, 0, bf0)
      G8_BAR();
    }
#undef G8_BAR
    if (wr == 0) __builtin_amdgcn_s_barrier();
  } else
  for (int t = 0; t < nk; ++t) {
    const bf16_t* la = hbuf(t & 1, wr);
    const bf16_t* lb = hbuf(t & 1, 2 + (wc >> 1));
    // P1
    read_b(lb, 0, bf0);
    read_a(la, 0);
    stage_a(0, t + 1);
    G8_MFMA(0, 0, bf0)
    __builtin_amdgcn_s_barrier();
    // P2
    read_b(lb, 1, bf1);
    stage_a(1, t + 1);
    G8_MFMA(0, 1, bf1)
    __builtin_amdgcn_s_barrier();
    // P3
    read_a(la, 1);
    stage_b(0, t + 2);
    G8_MFMA(1, 1, bf1)
    __builtin_amdgcn_s_barrier();
    // P4
    stage_b(1, t + 2);
    G8_MFMA(1, 0, bf0)
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the never-consumed tail stages must land before the LDS is released
#undef G8_MFMA

  if (DBG & 64) {      // timing experiment: no output stores (one impossible store keeps the accumulators alive)
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 12345.678f) ((float*)p.C)[0] = sum;
    return;
  }
  // epilogue: acc[i][j][r] = C[m0 + wr*128 + i*16 + r16][n0 + wc*64 + j*16 + 4g + r]
  if (p.ksplit > 1) {                               // split-K: raw fp32 partial sums; gemm8_reduce_kernel finishes
    float* part = p.partial + (int64_t)kslice * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + wr * 128 + i * 16 + r16;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nb = n0 + wc * 64 + j * 16 + 4 * g;
        float* o = part + (int64_t)m * p.N + nb;
        if (nb + 3 < p.N && (p.N & 3) == 0) *reinterpret_cast<float4*>(o) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        else for (int rr = 0; rr < 4; ++rr) if (nb + rr < p.N) o[rr] = acc[i][j][rr];
      }
    }
    return;
  }
  // Interior tiles with bf16 output: the accumulators go through the (now free) LDS so that every store
  // instruction writes whole 128-byte row segments (8 rows x 128 B per wave-instruction, 16 B per lane)
  // instead of 16 rows x 32 B -- a quarter of the write requests, and the residual is read the same way.
  const bool coalesced = !p.out_f32 && !p.accumulate && (m0 + G8_BM <= p.M) && (n0 + G8_BN <= p.N) && ((p.ldc & 7) == 0) &&
                         ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                         (!p.res || (((p.ldr & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.res) & 15) == 0)));
  if (coalesced) {
    __builtin_amdgcn_s_barrier();                   // every wave's tail DMA has landed (vmcnt(0) above): the LDS is free
    constexpr int LDW = 72;                         // 64 columns + 8 of padding: 144-byte rows keep b128 reads aligned
    bf16_t* st = lds + wave * (64 * LDW);           // 64 rows per pass, private to the wave
    const int lrow = lane >> 3, lch = lane & 7;
    float bv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) bv[j][rr] = p.bias ? bf2f(p.bias[n0 + wc * 64 + j * 16 + 4 * g + rr]) : 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 a = acc[half * 4 + i][j];
          *reinterpret_cast<uint2*>(st + (i * 16 + r16) * LDW + j * 16 + 4 * g) =
              make_uint2(pack2bf(a[0] + bv[j][0], a[1] + bv[j][1]), pack2bf(a[2] + bv[j][2], a[3] + bv[j][3]));
        }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q8 = 0; q8 < 8; ++q8) {
        const int row = q8 * 8 + lrow;
        uint4 v = *reinterpret_cast<const uint4*>(st + row * LDW + lch * 8);
        const int64_t grow = m0 + wr * 128 + half * 64 + row;
        const int gcol = n0 + wc * 64 + lch * 8;
        if (p.res) {
          const uint4 rv = *reinterpret_cast<const uint4*>(p.res + grow * p.ldr + gcol);
          v.x = pack2bf(lo_bf(v.x) + lo_bf(rv.x), hi_bf(v.x) + hi_bf(rv.x));
          v.y = pack2bf(lo_bf(v.y) + lo_bf(rv.y), hi_bf(v.y) + hi_bf(rv.y));
          v.z = pack2bf(lo_bf(v.z) + lo_bf(rv.z), hi_bf(v.z) + hi_bf(rv.z));
          v.w = pack2bf(lo_bf(v.w) + lo_bf(rv.w), hi_bf(v.w) + hi_bf(rv.w));
        }
        if (p.nt_store) {
          typedef __attribute__((ext_vector_type(4))) unsigned int g8_u32x4;
          g8_u32x4 w = {v.x, v.y, v.z, v.w};
          __builtin_nontemporal_store(w, reinterpret_cast<g8_u32x4*>((bf16_t*)p.C + grow * p.ldc + gcol));
        } else {
          *reinterpret_cast<uint4*>((bf16_t*)p.C + grow * p.ldc + gcol) = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  // The same for C += result (weight gradients of the second and later micro-batches): fp32 staging, 32 rows
  // per pass, so that the sum is rounded once exactly as in the generic path below.
  const bool coalesced_acc = !p.out_f32 && p.accumulate && !p.res && !p.bias && (m0 + G8_BM <= p.M) && (n0 + G8_BN <= p.N) &&
                             ((p.ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0);
  if (coalesced_acc) {
    __builtin_amdgcn_s_barrier();
    constexpr int LDF = 68;                         // 64 floats + 4 of padding (272-byte rows)
    float* stf = reinterpret_cast<float*>(lds) + wave * (32 * LDF);
    const int lrow = lane >> 3, lch = lane & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4*>(stf + (i * 16 + r16) * LDF + j * 16 + 4 * g) = acc[q * 2 + i][j];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q8 = 0; q8 < 4; ++q8) {
        const int row = q8 * 8 + lrow;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(stf + row * LDF + lch * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(stf + row * LDF + lch * 8 + 4);
        bf16_t* o = (bf16_t*)p.C + (int64_t)(m0 + wr * 128 + q * 32 + row) * p.ldc + n0 + wc * 64 + lch * 8;
        const uint4 old = *reinterpret_cast<const uint4*>(o);
        uint4 v;
        v.x = pack2bf(lo[0] + lo_bf(old.x), lo[1] + hi_bf(old.x));
        v.y = pack2bf(lo[2] + lo_bf(old.y), lo[3] + hi_bf(old.y));
        v.z = pack2bf(hi[0] + lo_bf(old.z), hi[1] + hi_bf(old.z));
        v.w = pack2bf(hi[2] + lo_bf(old.w), hi[3] + hi_bf(old.w));
        *reinterpret_cast<uint4*>(o) = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wr * 128 + i * 16 + r16;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nb = n0 + wc * 64 + j * 16 + 4 * g;
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

// C = epilogue(sum_s partial[s]) with the epilogue of the GEMM kernels: (+bias), (bf16 round, +res),
// bf16 or fp32 (optionally accumulated) output.  4 columns per thread.
__global__ __launch_bounds__(256) void gemm8_reduce_kernel(Gemm8Args p) {
  const int64_t q4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int n4 = (p.N + 3) >> 2;
  if (q4 >= (int64_t)p.M * n4) return;
  const int m = (int)(q4 / n4), nb = (int)(q4 - (int64_t)m * n4) * 4;
  float y[4] = {0.f, 0.f, 0.f, 0.f};
  const bool vec = (nb + 3 < p.N) && ((p.N & 3) == 0);
  for (int s = 0; s < p.ksplit; ++s) {
    const float* src = p.partial + ((int64_t)s * p.M + m) * p.N + nb;
    if (vec) { const float4 v = *reinterpret_cast<const float4*>(src); y[0] += v.x; y[1] += v.y; y[2] += v.z; y[3] += v.w; }
    else for (int rr = 0; rr < 4; ++rr) if (nb + rr < p.N) y[rr] += src[rr];
  }
  for (int rr = 0; rr < 4; ++rr) {
    const int n = nb + rr;
    if (n >= p.N) break;
    float v = y[rr];
    if (p.bias) v += bf2f(p.bias[n]);
    if (p.res) v = rbf(v) + bf2f(p.res[(int64_t)m * p.ldr + n]);
    if (p.out_f32) { float* o = (float*)p.C + (int64_t)m * p.ldc + n; *o = p.accumulate ? *o + v : v; }
    else { bf16_t* o = (bf16_t*)p.C + (int64_t)m * p.ldc + n; *o = f2bf(p.accumulate ? bf2f(*o) + v : v); }
  }
}

extern int g_gemm8_gm;
extern int g_gemm8_sched;
extern int g_gemm8_nt;            // 0: by size (default), 1: always, 2: never
static int g8_nt_store(int64_t out_bytes) { return g_gemm8_nt == 1 ? 1 : g_gemm8_nt == 2 ? 0 : out_bytes > (256ll << 20); }
static int g8_gm(int tiles_m) { const int want = g_gemm8_gm > 0 ? g_gemm8_gm : 8; return tiles_m < want ? tiles_m : want; }
// Called by eagle_gemm_nt_bf16 (gemm_mfma.hip) after its argument checks.
int eagle_launch_gemm_nt8(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* bias,
                          const void* res, int64_t ldr, int M, int N, int K, int out_f32, int accumulate, int ksplit,
                          void* workspace, hipStream_t stream) {
  if (lda > (1ll << 22) || ldb > (1ll << 22)) return EAGLE_ERR_SHAPE;   // 32-bit byte offsets span 256 rows of a tile
  Gemm8Args a;
  const int nk = K / G8_BK;
  if (ksplit < 1) ksplit = 1;
  if (ksplit > nk) ksplit = nk;
  a.kchunk = (nk + ksplit - 1) / ksplit;
  a.ksplit = (nk + a.kchunk - 1) / a.kchunk;        // every slice non-empty
  a.partial = (float*)workspace;
  if (a.ksplit > 1 && !workspace) return EAGLE_ERR_ARG;
  a.A = (const bf16_t*)A; a.lda = lda; a.B = (const bf16_t*)B; a.ldb = ldb; a.C = C; a.ldc = ldc;
  a.bias = (const bf16_t*)bias; a.res = (const bf16_t*)res; a.ldr = ldr;
  a.M = M; a.N = N; a.K = K; a.out_f32 = out_f32; a.accumulate = accumulate;
  a.tiles_m = (M + G8_BM - 1) / G8_BM; a.tiles_n = (N + G8_BN - 1) / G8_BN;
  a.gm = g8_gm(a.tiles_m);                          // row tiles per group of the grouped tile order
  a.tn = 0;
  a.nt_store = g8_nt_store((int64_t)M * N * (out_f32 ? 4 : 2));
  const dim3 grid(a.tiles_m * a.tiles_n * a.ksplit);
  if (g_gemm8_sched == 0x12) hipLaunchKernelGGL((gemm_nt8_kernel<true, 1, false, false, 2>), grid, dim3(G8_THREADS), 0, stream, a);
  else if (g_gemm8_sched == 0x22) hipLaunchKernelGGL((gemm_nt8_kernel<true, 2, false, false, 2>), grid, dim3(G8_THREADS), 0, stream, a);
  else if (g_gemm8_sched == 0x32) hipLaunchKernelGGL((gemm_nt8_kernel<true, 3, false, false, 2>), grid, dim3(G8_THREADS), 0, stream, a);
  else if (g_gemm8_sched == 0x42) hipLaunchKernelGGL((gemm_nt8_kernel<true, 7, false, false, 2>), grid, dim3(G8_THREADS), 0, stream, a);
  else if (g_gemm8_sched == 0x62) hipLaunchKernelGGL((gemm_nt8_kernel<true, 32, false, false, 2>), grid, dim3(G8_THREADS), 0, stream, a);
  else if (g_gemm8_sched == 0x52) hipLaunchKernelGGL((gemm_nt8_kernel<true, 64, false, false, 2>), grid, dim3(G8_THREADS), 0, stream, a);
  else if (g_gemm8_sched == 0) hipLaunchKernelGGL((gemm_nt8_kernel<true, 0, false, false, 0>), grid, dim3(G8_THREADS), 0, stream, a);
  else if (g_gemm8_sched == 2) hipLaunchKernelGGL((gemm_nt8_kernel<true, 0, false, false, 2>), grid, dim3(G8_THREADS), 0, stream, a);
  else hipLaunchKernelGGL((gemm_nt8_kernel<true, 0, false, false, 1>), grid, dim3(G8_THREADS), 0, stream, a);
  int rc = eagle_check_launch();
  if (rc || a.ksplit == 1) return rc;
  const int64_t q4 = (int64_t)M * ((N + 3) / 4);
  hipLaunchKernelGGL(gemm8_reduce_kernel, dim3((unsigned)((q4 + 255) / 256)), dim3(256), 0, stream, a);
  return eagle_check_launch();
}

// C[M,N] (+)= A^T . B with K-major operands: A [K, M] (row stride lda), B [K, N] (row stride ldb) -- the weight-gradient GEMMs
// behind loss.backward() (eagle/train.py:126): dW[out, in] = dY[tokens, out]^T . X[tokens, in], both read as they lie in memory.
// M % 256 == 0, N % 256 == 0, K % 64 == 0, K >= 128; lda / ldb % 8 == 0; 16-byte aligned bases.
extern "C" int eagle_gemm_tn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                                  int out_f32, int accumulate, hipStream_t stream) {
  if (!A || !B || !C) return EAGLE_ERR_ARG;
  if (M < 256 || N < 256 || (M & 255) || (N & 255) || K < 2 * G8_BK || (K % G8_BK)) return EAGLE_ERR_SHAPE;
  if ((lda & 7) || (ldb & 7) || lda < M || ldb < N || ldc < N) return EAGLE_ERR_SHAPE;
  if (((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C)) & 15)) return EAGLE_ERR_SHAPE;
  if ((uint64_t)K * (uint64_t)(lda > ldb ? lda : ldb) * 2ull > 0xffffffffull) return EAGLE_ERR_SHAPE;   // 32-bit byte offsets over K
  Gemm8Args a;
  a.A = (const bf16_t*)A; a.lda = lda; a.B = (const bf16_t*)B; a.ldb = ldb; a.C = C; a.ldc = ldc;
  a.bias = nullptr; a.res = nullptr; a.ldr = 0;
  a.M = M; a.N = N; a.K = K; a.out_f32 = out_f32; a.accumulate = accumulate;
  a.tiles_m = M / G8_BM; a.tiles_n = N / G8_BN;
  a.ksplit = 1; a.kchunk = K / G8_BK; a.partial = nullptr;
  a.gm = g8_gm(a.tiles_m);
  a.tn = 1;
  a.nt_store = g8_nt_store((int64_t)M * N * (out_f32 ? 4 : 2));
  hipLaunchKernelGGL((gemm_nt8_kernel<true, 0, true, true>), dim3(a.tiles_m * a.tiles_n), dim3(G8_THREADS), 0, stream, a);
  return eagle_check_launch();
}

// C[M,N] = A . B (+res) with B stored K-major: A [M, K] row-major (row stride lda), B [K, N] row-major (row stride ldb) --
// the data gradients behind loss.backward() (eagle/train.py:126): dX[tokens, in] = dY[tokens, out] . W[out, in] reads the
// weight as the state_dict stores it (no transposed copy per optimizer step).
// N % 256 == 0, K % 64 == 0, K >= 128; lda / ldb % 8 == 0; 16-byte aligned bases; K * ldb * 2 < 4 GiB.
extern "C" int eagle_gemm_nn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* res,
                                  int64_t ldr, int M, int N, int K, int out_f32, int accumulate, hipStream_t stream) {
  if (!A || !B || !C) return EAGLE_ERR_ARG;
  if (M < 1 || N < 256 || (N & 255) || K < 2 * G8_BK || (K % G8_BK)) return EAGLE_ERR_SHAPE;
  if ((lda & 7) || (ldb & 7) || lda < K || ldb < N || ldc < N || (res && ldr < N)) return EAGLE_ERR_SHAPE;
  if (((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C)) & 15)) return EAGLE_ERR_SHAPE;
  if (lda > (1ll << 22) || (uint64_t)K * (uint64_t)ldb * 2ull > 0xffffffffull) return EAGLE_ERR_SHAPE;   // 32-bit byte offsets
  Gemm8Args a;
  a.A = (const bf16_t*)A; a.lda = lda; a.B = (const bf16_t*)B; a.ldb = ldb; a.C = C; a.ldc = ldc;
  a.bias = nullptr; a.res = (const bf16_t*)res; a.ldr = ldr;
  a.M = M; a.N = N; a.K = K; a.out_f32 = out_f32; a.accumulate = accumulate;
  a.tiles_m = (M + G8_BM - 1) / G8_BM; a.tiles_n = N / G8_BN;
  a.ksplit = 1; a.kchunk = K / G8_BK; a.partial = nullptr;
  a.gm = g8_gm(a.tiles_m);
  a.tn = 1;
  a.nt_store = g8_nt_store((int64_t)M * N * (out_f32 ? 4 : 2));
  hipLaunchKernelGGL((gemm_nt8_kernel<true, 0, false, true>), dim3(a.tiles_m * a.tiles_n), dim3(G8_THREADS), 0, stream, a);
  return eagle_check_launch();
}
