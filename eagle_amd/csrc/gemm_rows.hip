// Weight-streaming GEMM for decode batches of 17..192 rows (batch-32 tree steps: 32 x {1,4} rows per forward; the catch-up
// forward over the accepted tokens: 32 x <= 6).
//
//   out[M,N] = epilogue( x[M,K] . W[N,K]^T )        bf16 in, fp32 accumulate, same epilogues as gemm_skinny.hip
//
// Replaces the nn.Linear calls of eagle/llama2.py at draft time (fc :542, q/k/v/o_proj :222-224/:280, gate/up/down_proj
// :317) and the frozen lm_head (eagle/train.py:215-228) when a batch of requests steps through the tree together.  At that
// size the op is still a weight stream (a 128-row step multiplies 8 MFMAs per KB of weights: a third of the MFMA rate the
// HBM stream could feed), so the two operands travel by different roads:
//   * W is the MFMA A operand and is used by ONE wave only: HBM -> VGPR directly (non-temporal 16-B loads, S register
//     stages of one 128-wide K super-step each), as in gemm_skinny_keepx_kernel -- no LDS round trip;
//   * x is shared by the four consumer waves and re-read by every column tile: it goes through LDS in full 256-byte
//     rows, filled by a fifth, loader-only wave with LDS-DMA (4 rows per instruction, 16-byte chunk XOR (row & 15) on
//     the SOURCE address so that the consumers' ds_read_b128 fragment reads are conflict-free).  The loader's DMA has a
//     vmcnt of its own: the consumers' counted waits see weight loads only (hipcc drains everything when plain loads
//     and LDS-DMA share a loop), and the x prefetch distance is independent of the weight stages;
//   * one raw s_barrier per super-step (it does not drain VMEM): it publishes x chunk j and frees the buffer of j-1.
// Few column tiles (N = 4096: 32 tiles of 128 columns) are cut along K into `ks` slices, one workgroup each, that write
// fp32 slabs in accumulator order (1 KB per wave-store); gemm_rows_reduce_kernel, a second launch, sums them in slice
// order and applies the epilogue.  The in-launch form of that sum is kept behind eagle_debug_set_rows: write-through
// (sc1) slab stores -- at the memory side when counted off vmcnt, no L2 write-back fence (a release fence per workgroup
// doubled the launch) -- then a ticket from a per-tile counter; the workgroup that draws the last one reads ALL slabs
// back with sc1 loads (past its own L1/L2, wherever the other slices ran), sums them in the same slice order and leaves
// the counter at zero.  It is the slower of the two at these slab sizes (see g_rows_two_launch).
#include "common.h"
#include "topk_select.h"

#define GR_NW 4                         // consumer waves per workgroup
#define GR_COUNTER_BYTES 16384          // head of the workspace: one arrival counter per column tile (<= 4096 tiles split)
#define GR_MAX_ROWS 192
#define GR_EPI_LINEAR 0
#define GR_EPI_SILU_MUL 1
#define GR_KEEP_RESIDENT 0x100          // accepted and ignored: this path always streams W non-temporally
#define GR_W_PACKED 0x200               // W / W2 are eagle_pack_weight_bf16 images (gemm_skinny.hip): whole-line weight reads

typedef __attribute__((ext_vector_type(4))) unsigned int gr_u32x4;

struct RowsArgs {
  const bf16_t* x0; const int64_t* idx0; int64_t ld0; int K0;
  const int64_t* ind0; int ind_n; int64_t idx0_bstride;   // idx0 row = idx0[(m/ind_n)*bstride + ind0[m%ind_n]]
  const bf16_t* x1; const int64_t* idx1; int64_t ld1;
  const bf16_t* W; const bf16_t* W2; int64_t ldw;
  const bf16_t* bias; const bf16_t* res; int64_t ldr;
  bf16_t* out; int64_t ldo;
  int M, N, K;             // N = output columns (SiLU*up: rows of W and of W2 each)
  int tiles;               // column tiles
  int ks, kchunk;          // K slices; super-steps (128 k) per slice
  float* slab;             // ks > 1: [ks][tiles][GR_NW][MT*RT][64] f32x4
  int* counters;           // ks > 1 and in-launch reduction: [tiles], zero on entry, zero on exit
  int rotate;              // K-walk stagger between neighbouring column tiles, in super-steps (0: none)
  unsigned long long* cand; int topk;   // gemm_rows_wide_kernel<MT, true>: [M][tiles][topk] candidate records instead of `out`
  int tile_cols;           // gemm_rows_stream_kernel: output columns per column tile: BN, or 3/4 BN with the fourth consumer wave idle
  int packed;              // host-side only: selects the instantiations that read packed weight images
};

// lane's 16-byte piece of 16-row tile T, k-block k / 32 in a packed image (gemm_skinny.hip: [N/16][K/32][64 lanes][8])
__device__ __forceinline__ const bf16_t* gr_pk(const bf16_t* base, int T, int K, int64_t k, int lane) {
  return base + ((int64_t)T * (K >> 5) + (k >> 5)) * 512 + lane * 8;
}

template <int RT, int EPI> struct RowsGeom {
  static constexpr int RTW = (EPI == GR_EPI_SILU_MUL) ? RT / 2 : RT;   // 16-column output tiles per wave
  static constexpr int BN = GR_NW * RTW * 16;                          // output columns per workgroup
};

// acc[t][r] = y[m = mt*16 + r16][n = tile*BN + (w*RTW + u)*16 + 4g + r]  (SiLU: t = 2u gate, 2u+1 up)
template <int RT, int EPI>
__device__ __forceinline__ void rows_epilogue(const RowsArgs& p, int tile, int w, int mt, const f32x4 (&acc)[RT], int lane) {
  using G = RowsGeom<RT, EPI>;
  const int r16 = lane & 15, g = lane >> 4;
  const int m = mt * 16 + r16;
  if (m >= p.M || w * G::RTW * 16 >= p.tile_cols) return;   // (a narrower tile: its last wave's columns belong to the next tile)
#pragma unroll
  for (int u = 0; u < G::RTW; ++u) {
    const int nb = tile * p.tile_cols + (w * G::RTW + u) * 16 + 4 * g;
    if (nb >= p.N) continue;
    float y[4];
    if (EPI == GR_EPI_SILU_MUL) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gt = rbf(acc[2 * u][r]), up = rbf(acc[2 * u + 1][r]);
        y[r] = rbf(gt / (1.0f + __expf(-gt))) * up;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nb + r;
        float v = acc[u][r];
        if (p.bias && n < p.N) v += bf2f(p.bias[n]);
        v = rbf(v);
        if (p.res && n < p.N) v = v + bf2f(p.res[(int64_t)m * p.ldr + n]);
        y[r] = v;
      }
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

// MT 16-row activation tiles (M <= 16*MT), RT 16-row weight tiles per consumer wave, S weight register stages,
// XS activation LDS buffers of MT*16 rows x 128 k.
template <int MT, int RT, int EPI, int S, int XS, int NL, bool PK = false>
__global__ __launch_bounds__((GR_NW + NL) * 64) void gemm_rows_stream_kernel(RowsArgs p) {
  using G = RowsGeom<RT, EPI>;
  constexpr int XBYTES = MT * 16 * 256;                  // one x chunk: MT*16 rows of 128 bf16
  constexpr int XOPS = MT * 4 / NL;                      // DMA instructions per chunk and loader wave (4 rows each)
  static_assert((MT * 4) % NL == 0, "loader waves share a chunk evenly");
  static_assert((XS - 2) * XOPS <= 63, "vmcnt immediate");
  __shared__ __attribute__((aligned(256))) unsigned char xl[XS * XBYTES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // consecutive ids run on ONE XCD (dispatch is round-robin over the 8 XCDs): the slices of a tile share an L2, which is
  // where the last arriver reads their slabs fastest; a speed choice only
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = orig & 7;
  const int id = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (orig >> 3);
  const int slice = id % p.ks, tile = id / p.ks;
  const int nj_all = p.K >> 7;
  const int j_beg = slice * p.kchunk;
  const int nj = min(p.kchunk, nj_all - j_beg);          // >= 1 (launcher)
  // Column tiles walk their K slice from staggered starting points (and wrap): workgroups that run side by side then
  // read different 256-byte columns of the weight rows instead of marching through the same HBM channels in step.
  const int rot = p.rotate ? (tile * p.rotate) % nj : 0;
  auto kstep = [&](int j) { const int jj = j + rot; return jj >= nj ? jj - nj : jj; };

  const bool loader = wave >= GR_NW;
  f32x4 acc[MT][RT];
  if (loader) {
    // ---------------- loader wave(s): x chunks -> LDS ----------------
    const int lrow = lane >> 4, slot = lane & 15, li = wave - GR_NW;
    const bf16_t* pa[XOPS];
    const bf16_t* pb[XOPS];
#pragma unroll
    for (int i = 0; i < XOPS; ++i) {
      const int row = 4 * (li * XOPS + i) + lrow;
      const int m = row < p.M ? row : p.M - 1;
      int64_t ra, rb;
      if (p.ind0) ra = p.idx0[(int64_t)(m / p.ind_n) * p.idx0_bstride + p.ind0[m % p.ind_n]];
      else ra = p.idx0 ? p.idx0[m] : m;
      rb = p.idx1 ? p.idx1[m] : m;
      const int ch = slot ^ (row & 15);
      pa[i] = p.x0 + ra * p.ld0 + ch * 8;
      pb[i] = p.x1 ? (p.x1 + rb * p.ld1 + ch * 8 - p.K0) : pa[i];
    }
    auto issue = [&](int j) {
      const int k = (j_beg + kstep(j)) << 7;
      const bool second = k >= p.K0;
      unsigned char* dst = xl + (j % XS) * XBYTES;
#pragma unroll
      for (int i = 0; i < XOPS; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((second ? pb[i] : pa[i]) + k),
                                         (__attribute__((address_space(3))) void*)(dst + (li * XOPS + i) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int c = 0; c < XS - 1; ++c) if (c < nj) issue(c);
    for (int j = 0; j < nj; ++j) {
      // chunk j has landed when only the younger chunks j+1 .. j+XS-2 are outstanding (tail: wait for everything)
      if (XS > 2 && j + XS - 2 < nj) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((XS - 2) * XOPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // publishes chunk j; every consumer is done with chunk j-1
      if (j + XS - 1 < nj) issue(j + XS - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (wave * G::RTW * 16 >= p.tile_cols) {
    // ---------------- a consumer wave past a narrower tile: it only keeps the barrier count ----------------
    for (int j = 0; j < nj; ++j) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  } else {
  // ---------------- consumer waves: W -> VGPR, MFMA against x fragments from LDS ----------------
  const int r16 = lane & 15, g = lane >> 4;
  const bf16_t* wp[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const int u = (EPI == GR_EPI_SILU_MUL) ? (t >> 1) : t;
    const int nb = tile * p.tile_cols + (wave * G::RTW + u) * 16;      // a multiple of 16: one packed tile
    int n = nb + r16;
    n = n < p.N ? n : p.N - 1;                           // clamped rows compute values that are never stored
    const bf16_t* base = (EPI == GR_EPI_SILU_MUL && (t & 1)) ? p.W2 : p.W;
    wp[t] = PK ? gr_pk(base, min(nb >> 4, (p.N - 1) >> 4), p.K, (int64_t)j_beg << 7, lane)
               : base + (int64_t)n * p.ldw + 8 * g + ((int64_t)j_beg << 7);
  }
  constexpr int KS = PK ? 16 : 1, SS = PK ? 512 : 32;     // element steps per k and per 32-k sub-step (packed: 1 KiB blocks)
  // Every stage refill is issued unconditionally (past the slice's last super-step it reads one hot line of W, the
  // same address in every lane) and the main loop body is branch-free: hipcc's counted vmcnt waits are exact only
  // when the number of loads between a stage's issue and its use is the same on every path.
  bf16x8 a[S][4][RT];
  auto load_w = [&](bf16x8 (&st)[4][RT], int j) {
    const bool live = j < nj;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const bf16_t* src = live ? wp[t] + (kstep(j) << 7) * KS + s * SS : p.W;
        st[s][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(src));
      }
  };
#pragma unroll
  for (int st = 0; st < S; ++st) load_w(a[st], st);

#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto step = [&](const bf16x8 (&st)[4][RT], int j) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned char* xb = xl + (j % XS) * XBYTES + r16 * 256;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 xf[MT];
      const int off = ((4 * s + g) ^ r16) << 4;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xf[mt] = *reinterpret_cast<const bf16x8*>(xb + mt * 4096 + off);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st[s][t], xf[mt], acc[mt][t], 0, 0, 0);
    }
  };
  const int n_full = nj / S * S;
  for (int j0 = 0; j0 < n_full; j0 += S) {
#pragma unroll
    for (int st = 0; st < S; ++st) {
      step(a[st], j0 + st);
      load_w(a[st], j0 + st + S);
    }
  }
#pragma unroll
  for (int st = 0; st < S - 1; ++st)
    if (n_full + st < nj) step(a[st], n_full + st);

  }   // consumer

  if (p.ks == 1) {
    if (!loader) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) rows_epilogue<RT, EPI>(p, tile, wave, mt, acc[mt], lane);
    }
    return;
  }
  const int64_t slab_wave = (int64_t)GR_NW * (MT * RT) * 64;                     // f32x4 per (slice, tile)
  constexpr int WAVE_SLAB = MT * RT * 1024;               // bytes one consumer wave writes
  if (!loader && wave * G::RTW * 16 < p.tile_cols) {
    f32x4* sl = reinterpret_cast<f32x4*>(p.slab) + ((int64_t)slice * p.tiles + tile) * slab_wave + wave * (MT * RT) * 64;
    if (p.counters) {                                    // write-through stores
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)sl, 0, WAVE_SLAB, 0x00020000);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          union { f32x4 f; gr_u32x4 u; } cv; cv.f = acc[mt][t];
          __builtin_amdgcn_raw_buffer_store_b128(cv.u, rs, ((mt * RT + t) * 64 + lane) * 16, 0, 16 /* sc1 */);
        }
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < RT; ++t) sl[(mt * RT + t) * 64 + lane] = acc[mt][t];
    }
  }
  if (!p.counters) return;                               // gemm_rows_reduce_kernel follows
  // ---- in-launch reduction: ticket per tile, the last arriver sums every slab in slice order ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave: its slab is written through
  __syncthreads();                                       // ... all of them; all x reads are done too
  int* tick = reinterpret_cast<int*>(xl);                // the one LDS array (a second object makes hipcc drain the DMA)
  if (threadIdx.x == 0) *tick = __hip_atomic_fetch_add(&p.counters[tile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*tick != p.ks - 1) return;
  if (threadIdx.x == 0) __hip_atomic_store(&p.counters[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (loader) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the slab loads below the ticket
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const f32x4* sb = reinterpret_cast<const f32x4*>(p.slab) + (int64_t)tile * slab_wave + wave * (MT * RT) * 64;
  for (int s2 = 0; s2 < p.ks; ++s2) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(sb + (int64_t)s2 * p.tiles * slab_wave), 0, WAVE_SLAB, 0x00020000);
    gr_u32x4 v[MT * RT];                                   // the whole slab of this wave in flight, then added
#pragma unroll
    for (int i = 0; i < MT * RT; ++i)
      v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (i * 64 + lane) * 16, 0, 16 /* sc1: EVERY slab load */);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        union { f32x4 f; gr_u32x4 u; } cv; cv.u = v[mt * RT + t];
        acc[mt][t] += cv.f;
      }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) rows_epilogue<RT, EPI>(p, tile, wave, mt, acc[mt], lane);
}

// grid (tiles, MT): block (tile, mt) sums the ks slabs of one 16-row block of a column tile and applies the epilogue
template <int MT, int RT, int EPI>
__global__ __launch_bounds__(GR_NW * 64) void gemm_rows_reduce_kernel(RowsArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x, mt = blockIdx.y;
  if (mt * 16 >= p.M || wave * RowsGeom<RT, EPI>::RTW * 16 >= p.tile_cols) return;
  f32x4 acc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int64_t slab_wave = (int64_t)GR_NW * (MT * RT) * 64;
  const f32x4* sb = reinterpret_cast<const f32x4*>(p.slab) + (int64_t)tile * slab_wave + (wave * (MT * RT) + mt * RT) * 64 + lane;
  const int64_t step = (int64_t)p.tiles * slab_wave;
  int s = 0;
  for (; s + 4 <= p.ks; s += 4) {                          // four slices in flight, added in slice order
    f32x4 v[4][RT];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < RT; ++t) v[u][t] = sb[(s + u) * step + t * 64];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] += v[u][t];
  }
  for (; s < p.ks; ++s)
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] += sb[s * step + t * 64];
  rows_epilogue<RT, EPI>(p, tile, wave, mt, acc, lane);
}

// ---------------------------------------------------------------------------------------------
// The many-tile, 81..128-row case (the lm_head of a 128-row tree step: 151936 x 4096 weights against 128 rows).
// Measured (tools/rows_wide_bench.py): with 128-column tiles the x rows every tile re-reads through L2 and the
// MFMA work sit on top of the weight stream instead of under it.  This form gives a workgroup up to 384 columns: eight
// consumer waves and NO loader waves (2 waves per SIMD: 256 registers each; 96 accumulators and two weight stages of 48 at
// three 16-column groups per wave), column tiles of equal width in 16-column groups sized so that the grid is a whole number
// of rounds of one workgroup per CU (151936 columns: 512 tiles of 18-19 groups, two rounds).  A wave owns 1, 2 or 3
// consecutive groups and runs the loop compiled for that count: an unused slot would have to load SOMETHING to keep the
// counted waits uniform, and both choices cost -- one shared dead address queues every wave of the chip on one L2 channel
// (264 vs 249 us at 128256 columns), re-reading the wave's own lines fetches them again (non-temporal lines do not stay: 314).
// With the LDS-DMA of x issued by the same waves that stream W, nothing here may be visible to hipcc's wait counting (it
// answers a plain load or a ds_read beside a pending LDS-DMA with vmcnt(0)): the weight loads and the x fragment reads are
// inline asm, every wait is counted by hand and followed by empty asm statements that name the registers it releases.
// Per super-step j a wave issues, in this order, its share of x chunk j+2 (XOPS DMA instructions) and weight stage j+2
// (4 x RT loads, s-major); vector-memory operations complete in order, so at the top of step j "x chunk j has landed" is
// vmcnt(2 x 4RT + XOPS) and sub-step s of weight stage j is vmcnt((3 - s) RT + 4RT + XOPS).  (Refilling each sub-step's
// registers as soon as they are consumed was 2-3 % slower.)
#define GW_NW 8
#define GW_RT 3                         // most 16-column groups per wave
template <int N> struct gw_int { static constexpr int value = N; };
#define GW_KSTRIDE 800                  // bytes per row of the 16-bit key image (TOPK): 4 rows read side by side hit 4 x 8 banks
// TOPK: no logits leave the kernel.  A tile's outputs are rounded to bf16, turned into 16-bit monotone keys and laid out in
// LDS (the x buffers are free by then); every 16-lane group then selects the p.topk best of one row of the tile (lane l holds
// columns l, l+16, ...: keys = value << 16 | 0xffff - column, the lowest column wins a tie) and writes them as candidate
// records [row][tile][topk] in topk.hip's format -- eagle_topk_merge_candidates finishes the job.  Against logits + the
// two-launch top-k this saves the 39 MB store burst of a 128 x 151936 tile row and the 39 MB pass of stage 1.
template <int MT, bool TOPK = false, bool PK = false>
__global__ __launch_bounds__(GW_NW * 64) void gemm_rows_wide_kernel(RowsArgs p) {
  constexpr int S = 2, XS = S + 1;
  constexpr int XBYTES = MT * 16 * 256;
  constexpr int LDS_BYTES = (TOPK && MT * 16 * GW_KSTRIDE > XS * XBYTES) ? MT * 16 * GW_KSTRIDE : XS * XBYTES;
  constexpr int XOPS = MT * 4 / GW_NW;
  static_assert((MT * 4) % GW_NW == 0, "the waves share a chunk evenly");
  static_assert(MT % 2 == 0, "x fragments in two halves");
  constexpr int MH = MT / 2;
  __shared__ __attribute__((aligned(256))) unsigned char xl[LDS_BYTES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = orig & 7;
  const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (orig >> 3);
  const int g16 = (p.N + 15) >> 4;
  const int gb = (int)(((int64_t)tile * g16) / p.tiles);                // this tile's 16-column groups: [gb, gb + ng)
  const int ng = (int)(((int64_t)(tile + 1) * g16) / p.tiles) - gb;     // GW_NW <= ng <= GW_NW * GW_RT (launcher)
  const int wg0 = (wave * ng) / GW_NW, wgn = ((wave + 1) * ng) / GW_NW - wg0;   // this wave's groups: 1..GW_RT of them
  const int nj = p.K >> 7;
  const int rot = p.rotate ? (tile * p.rotate) % nj : 0;
  auto kstep = [&](int j) { const int jj = j + rot; return jj >= nj ? jj - nj : jj; };
  const int r16 = lane & 15, g = lane >> 4;

  // x rows of this wave's DMA instructions (4 rows each; 16-byte chunk XOR (row & 15) on the source side); no second,
  // concatenated x operand in this form: the launcher keeps those on gemm_rows_stream_kernel
  const bf16_t* pa[XOPS];
#pragma unroll
  for (int i = 0; i < XOPS; ++i) {
    const int row = 4 * (wave * XOPS + i) + g;
    const int m = row < p.M ? row : p.M - 1;
    int64_t ra;
    if (p.ind0) ra = p.idx0[(int64_t)(m / p.ind_n) * p.idx0_bstride + p.ind0[m % p.ind_n]];
    else ra = p.idx0 ? p.idx0[m] : m;
    const int ch = r16 ^ (row & 15);
    pa[i] = p.x0 + ra * p.ld0 + ch * 8;
  }
  auto issue_x = [&](int j) {                              // past the last chunk: the last chunk again, into a buffer nobody reads
    const int k = kstep(j < nj ? j : nj - 1) << 7;
    unsigned char* dst = xl + (j % XS) * XBYTES;
#pragma unroll
    for (int i = 0; i < XOPS; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[i] + k),
                                       (__attribute__((address_space(3))) void*)(dst + (wave * XOPS + i) * 1024), 16, 0, 0);
  };
  const uint32_t xl0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)xl) + r16 * 256;

  auto run = [&](auto rt_c) {
    constexpr int RT = decltype(rt_c)::value;
    const bf16_t* wp[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      int n = (gb + wg0 + t) * 16 + r16;
      n = n < p.N ? n : p.N - 1;                           // the last group of the last tile: clamped rows are never stored
      wp[t] = PK ? gr_pk(p.W, min(gb + wg0 + t, (p.N - 1) >> 4), p.K, 0, lane) : p.W + (int64_t)n * p.ldw + 8 * g;
    }
    bf16x8 a[S][4][RT];
    auto issue_w = [&](bf16x8 (&st)[4][RT], int j) {       // past the last super-step: one hot line (two stages per wave)
      const bool live = j < nj;
      const int k = live ? (kstep(j) << 7) * (PK ? 16 : 1) : 0;   // packed: a 32-k block is 512 elements on
#pragma unroll
      for (int s = 0; s < 4; ++s)                          // s-major: the order the counted waits rely on
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          const bf16_t* src = live ? wp[t] + k : p.W;
          if (PK) {                                        // the next k-block of the same tile: the next 1 KiB
            if (s == 0) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(st[s][t]) : "v"(src) : "memory");
            if (s == 1) asm volatile("global_load_dwordx4 %0, %1, off offset:1024 nt" : "=v"(st[s][t]) : "v"(src) : "memory");
            if (s == 2) asm volatile("global_load_dwordx4 %0, %1, off offset:2048 nt" : "=v"(st[s][t]) : "v"(src) : "memory");
            if (s == 3) asm volatile("global_load_dwordx4 %0, %1, off offset:3072 nt" : "=v"(st[s][t]) : "v"(src) : "memory");
          } else {
            if (s == 0) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(st[s][t]) : "v"(src) : "memory");
            if (s == 1) asm volatile("global_load_dwordx4 %0, %1, off offset:64 nt" : "=v"(st[s][t]) : "v"(src) : "memory");
            if (s == 2) asm volatile("global_load_dwordx4 %0, %1, off offset:128 nt" : "=v"(st[s][t]) : "v"(src) : "memory");
            if (s == 3) asm volatile("global_load_dwordx4 %0, %1, off offset:192 nt" : "=v"(st[s][t]) : "v"(src) : "memory");
          }
        }
    };
    f32x4 acc[MT][RT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    issue_x(0); issue_w(a[0], 0);
    issue_x(1); issue_w(a[1], 1);

    auto step = [&](bf16x8 (&st)[4][RT], int j) {
      // x chunk j (every wave's share): behind it in the queue are weight stage j and all of step j-1's issues
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * 4 * RT + XOPS) : "memory");
      const uint32_t xb = xl0 + (j % XS) * XBYTES;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((3 - s) * RT + 4 * RT + XOPS) : "memory");
#pragma unroll
        for (int t = 0; t < RT; ++t) asm volatile("" : "+v"(st[s][t]));      // released by the wait above
        const uint32_t ad = xb + (((4 * s + g) ^ r16) << 4);
        bf16x8 xf[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[mt]) : "v"(ad), "n"(mt * 4096) : "memory");
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MH) : "memory");
#pragma unroll
        for (int mt = 0; mt < MH; ++mt) asm volatile("" : "+v"(xf[mt]));
#pragma unroll
        for (int mt = 0; mt < MH; ++mt)
#pragma unroll
          for (int t = 0; t < RT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st[s][t], xf[mt], acc[mt][t], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = MH; mt < MT; ++mt) asm volatile("" : "+v"(xf[mt]));
#pragma unroll
        for (int mt = MH; mt < MT; ++mt)
#pragma unroll
          for (int t = 0; t < RT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st[s][t], xf[mt], acc[mt][t], 0, 0, 0);
      }
      issue_x(j + S);                                      // into the buffer of chunk j-1: every wave is past barrier j
      issue_w(st, j + S);
    };
    for (int j0 = 0; j0 < nj; j0 += 2) {
      step(a[0], j0);
      if (j0 + 1 < nj) step(a[1], j0 + 1);
    }
    // the dead refills (and the DMA of a chunk nobody reads) land before the registers and the LDS are given up
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int st = 0; st < S; ++st)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < RT; ++t) asm volatile("" : "+v"(a[st][s][t]));

    if (TOPK) {
      // ---- key image: keys[m][c] (16 bit) at xl + m * GW_KSTRIDE + 2c, c = column inside the tile ----
      __syncthreads();                                     // every wave is done with the x buffers
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          const int cb = (wg0 + t) * 16 + 4 * g;
          uint32_t k16[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = gb * 16 + cb + r;
            float v = acc[mt][t][r];
            if (p.bias && n < p.N) v += bf2f(p.bias[n]);
            k16[r] = n < p.N ? tk_mono_bf16(f2bf(v)) : 0u;
          }
          *reinterpret_cast<uint2*>(xl + (mt * 16 + r16) * GW_KSTRIDE + 2 * cb) = make_uint2(k16[0] | (k16[1] << 16), k16[2] | (k16[3] << 16));
        }
      if (wave == 0 && ng < GW_NW * GW_RT) {               // the 16 slots after the tile's last group: key 0 (an odd group count)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          *reinterpret_cast<uint2*>(xl + (mt * 16 + r16) * GW_KSTRIDE + 2 * (ng * 16 + 4 * g)) = make_uint2(0u, 0u);
      }
      return;
    }
    // ---- epilogue: acc[mt][t][r] = y[m = mt*16 + r16][n = (gb + wg0 + t)*16 + 4g + r] ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mt * 16 + r16;
      if (m >= p.M) continue;
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const int nb = (gb + wg0 + t) * 16 + 4 * g;
        if (nb >= p.N) continue;
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = nb + r;
          float v = acc[mt][t][r];
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
  };
  // every variant passes the same nj barriers
  if (wgn >= 3) run(gw_int<3>{});
  else if (wgn == 2) run(gw_int<2>{});
  else run(gw_int<1>{});
  if (!TOPK) return;
  // ---- selection: wave w takes rows 16w .. 16w+15, four at a time (one per 16-lane group) ----
  __syncthreads();
  if (wave >= MT) return;                                  // 96-row form: waves 6 and 7 have no rows
  const int grp = lane >> 4, l16 = lane & 15;
  for (int pass = 0; pass < 4; ++pass) {                   // (topk_select.h; the 16 slots after an odd group count hold key 0)
    const int R = 16 * wave + 4 * pass + grp;
    tk_select_group16<GW_NW * GW_RT / 2>(xl + R * GW_KSTRIDE, 16 * ng, (uint32_t)(gb * 16), p.topk,
                                         p.cand + ((int64_t)R * p.tiles + tile) * p.topk, R < p.M, l16);
  }
}

// ---------------------------------------------------------------------------------------------
static int g_rows_target_wg = 0;    // 0: by shape; diagnostics (eagle_debug_set_rows): workgroups the K split aims for
static int g_rows_min_steps = 4;    // least super-steps per K slice
static int g_rows_stages = 0;       // 0: default; 2 / 3 / 4 weight register stages (+ 0x10 * (1 + stagger): K-walk stagger override)
static int g_rows_rotate = 1;
// Default: the slabs are summed by a second launch.  Measured at the batch-32 leg's shapes (tools/rows2_trace.sh): a slice
// writes 16-64 KB, a tile's last arriver would read 128-512 KB alone while the other CUs idle -- o_proj at 32 rows 17.5 us
// in one launch against 10.2 + 4.0 in two, at 128 rows 38.6 against 22.4 + 5.1.  The in-launch form stays selectable.
static int g_rows_two_launch = 1;
static int g_rows_narrow = 1;        // 3/4-width column tiles where they fill whole rounds (bit 10 of `stages` clears it)
static int g_rows_wide = 1;          // gemm_rows_wide_kernel for the many-tile, 81..128-row case (bit 9 of `stages` clears it)
static int g_head_skip_merge = 0;    // eagle_lm_head_topk_rows_bf16 stops after the candidate GEMM (bit 11 of `stages`: timing of that launch alone)
extern "C" void eagle_debug_set_rows(int target_wg, int min_steps, int stages) {
  g_rows_target_wg = target_wg > 0 ? target_wg : 0;
  g_rows_min_steps = min_steps > 0 ? min_steps : 4;
  g_rows_two_launch = ((stages >> 8) & 1) ? 0 : 1;
  g_rows_wide = ((stages >> 9) & 1) ? 0 : 1;
  g_rows_narrow = ((stages >> 10) & 1) ? 0 : 1;
  g_head_skip_merge = (stages >> 11) & 1;
  stages &= 0xff;
  g_rows_stages = stages & 15;
  g_rows_rotate = (stages >> 4) ? (stages >> 4) - 1 : 1;
}
static int gr_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}
struct RowsPlan { int mt, tiles, ks, kchunk, wide, tile_cols; };
static RowsPlan gr_plan(int M, int N, int K, int epi, bool has_x1 = false) {
  RowsPlan pl;
  const int mt = (M + 15) / 16;
  pl.mt = mt <= 2 ? 2 : mt <= 4 ? 4 : mt <= 6 ? 6 : mt <= 8 ? 8 : 12;
  const int bn = (epi == GR_EPI_SILU_MUL) ? RowsGeom<2, GR_EPI_SILU_MUL>::BN : RowsGeom<2, GR_EPI_LINEAR>::BN;
  pl.tiles = (N + bn - 1) / bn;
  pl.tile_cols = bn;
  const int nj = K >> 7;
  const int target = g_rows_target_wg ? g_rows_target_wg : gr_num_cus() * (pl.mt >= 6 ? 1 : 2);
  const int ks_max = nj / g_rows_min_steps;
  int ks = 1;
  if (g_rows_target_wg) {
    ks = (target + pl.tiles / 2) / pl.tiles;             // diagnostics: nearest count of slices
  } else {
    // Whole rounds: `target` workgroups run at a time (one per CU from 96 rows up: 128 KB of x buffers), so 288 of them take
    // as long as 512 (q|k|v of 128 rows x 12288 columns: 96 tiles x 3 slices, 45.6 us; x 2 slices, one round).  Cost of a
    // split in super-steps: rounds x (super-steps per slice x relative ingest per step + ~4 of ramp and drain) + the slices
    // the reduction reads.  From 96 rows up, tiles of 3/4 the width (the fourth consumer wave idles) are tried too: a step
    // takes in (weight columns + x rows) x 256 bytes, and 192 tiles on 256 CUs (SiLU*up at 12288 columns) become 256.
    float best = 0.f;
    const int wcols = (epi == GR_EPI_SILU_MUL) ? 2 : 1;  // weight rows per output column
    for (int narrow = 0; narrow <= ((pl.mt >= 6 && g_rows_narrow) ? 1 : 0); ++narrow) {
      const int w = narrow ? bn * 3 / 4 : bn;
      const int tiles = (N + w - 1) / w;
      const float rel = (float)(wcols * w + pl.mt * 16) / (float)(wcols * bn + pl.mt * 16);
      for (int c = 1; c <= (ks_max < 1 ? 1 : ks_max); ++c) {
        const int kc = (nj + c - 1) / c, real = (nj + kc - 1) / kc;
        const int rounds = (tiles * real + target - 1) / target;
        const float cost = (float)rounds * ((float)kc * rel + 4.f) + 0.25f * (float)real;
        if ((c == 1 && !narrow) || cost < best) { best = cost; ks = real; pl.tiles = tiles; pl.tile_cols = w; }
      }
    }
  }
  if (ks > ks_max) ks = ks_max;
  if (ks < 1 || pl.tiles * 4 > GR_COUNTER_BYTES) ks = 1;
  pl.kchunk = (nj + ks - 1) / ks;
  pl.ks = (nj + pl.kchunk - 1) / pl.kchunk;              // every slice non-empty
  pl.wide = 0;
  const int g16 = (N + 15) / 16, cus = gr_num_cus();
  if (g_rows_wide && epi == GR_EPI_LINEAR && pl.ks == 1 && (pl.mt == 6 || pl.mt == 8) && g16 >= 8 * cus && !has_x1) {
    // whole rounds of one workgroup per CU, each tile at most GW_NW * GW_RT groups wide
    const int rounds = (g16 + cus * GW_NW * GW_RT - 1) / (cus * GW_NW * GW_RT);
    pl.wide = 1;
    pl.tiles = rounds * cus;
  }
  return pl;
}

extern "C" int64_t eagle_gemm_rows_workspace_bytes(int M, int N, int K, int epi) {
  epi &= ~(GR_KEEP_RESIDENT | GR_W_PACKED);
  if (M < 1 || M > GR_MAX_ROWS || N < 1 || K < 128 || (K & 127) || (epi != GR_EPI_LINEAR && epi != GR_EPI_SILU_MUL)) return 0;
  const RowsPlan pl = gr_plan(M, N, K, epi);
  if (pl.ks == 1) return 0;
  return GR_COUNTER_BYTES + (int64_t)pl.ks * pl.tiles * GR_NW * (pl.mt * 2) * 64 * 16;
}

template <int MT, int EPI, int S>
static int gr_launch_s(const RowsArgs& a, hipStream_t st) {
  // x chunks in LDS: the loader runs 3 super-steps ahead (an L2 hit under a full HBM stream takes > 1 us); 192 rows: 2 ahead
  // (3 x 48 KB of the 160 KB)
  constexpr int XS = MT > 8 ? 3 : 4;
  constexpr int NL = MT <= 2 ? 1 : MT <= 4 ? 2 : 4;
  if (a.packed) hipLaunchKernelGGL((gemm_rows_stream_kernel<MT, 2, EPI, S, XS, NL, true>), dim3(a.tiles * a.ks), dim3((GR_NW + NL) * 64), 0, st, a);
  else hipLaunchKernelGGL((gemm_rows_stream_kernel<MT, 2, EPI, S, XS, NL, false>), dim3(a.tiles * a.ks), dim3((GR_NW + NL) * 64), 0, st, a);
  int rc = eagle_check_launch();
  if (rc || a.ks == 1 || a.counters) return rc;
  hipLaunchKernelGGL((gemm_rows_reduce_kernel<MT, 2, EPI>), dim3(a.tiles, (a.M + 15) / 16), dim3(GR_NW * 64), 0, st, a);
  return eagle_check_launch();
}
template <int MT, int EPI>
static int gr_launch(const RowsArgs& a, hipStream_t st) {
  if constexpr (MT > 8) return gr_launch_s<MT, EPI, 2>(a, st);     // 192 rows: 96 accumulator registers leave room for 2 stages
  else {
  if (g_rows_stages == 2) return gr_launch_s<MT, EPI, 2>(a, st);
  if constexpr (MT <= 4) { if (g_rows_stages == 4) return gr_launch_s<MT, EPI, 4>(a, st); }
  return gr_launch_s<MT, EPI, 3>(a, st);
  }
}
template <int EPI>
static int gr_launch_mt(const RowsArgs& a, int mt, hipStream_t st, int wide) {
  if (wide) {
    if (a.packed) {
      if (mt == 6) hipLaunchKernelGGL((gemm_rows_wide_kernel<6, false, true>), dim3(a.tiles), dim3(GW_NW * 64), 0, st, a);
      else hipLaunchKernelGGL((gemm_rows_wide_kernel<8, false, true>), dim3(a.tiles), dim3(GW_NW * 64), 0, st, a);
    } else {
      if (mt == 6) hipLaunchKernelGGL((gemm_rows_wide_kernel<6>), dim3(a.tiles), dim3(GW_NW * 64), 0, st, a);
      else hipLaunchKernelGGL((gemm_rows_wide_kernel<8>), dim3(a.tiles), dim3(GW_NW * 64), 0, st, a);
    }
    return eagle_check_launch();
  }
  switch (mt) {
    case 2: return gr_launch<2, EPI>(a, st);
    case 4: return gr_launch<4, EPI>(a, st);
    case 6: return gr_launch<6, EPI>(a, st);
    case 8: return gr_launch<8, EPI>(a, st);
    default: return gr_launch<12, EPI>(a, st);
  }
}

static inline bool gr_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// See include/eagle_hip.h for the contract.
extern "C" int eagle_gemm_rows_bf16(const void* x0, const int64_t* idx0, int64_t ld0, int K0,
                                    const int64_t* ind0, int ind_n, int64_t idx0_bstride,
                                    const void* x1, const int64_t* idx1, int64_t ld1,
                                    const void* W, const void* W2, int64_t ldw, const void* bias,
                                    const void* res, int64_t ldr, void* out, int64_t ldo,
                                    int M, int N, int K, int epi, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  if (!x0 || !W || !out) return EAGLE_ERR_ARG;
  const bool packed = (epi & GR_W_PACKED) != 0;
  epi &= ~(GR_KEEP_RESIDENT | GR_W_PACKED);
  if (packed) ldw = K;                                   // a packed image has no row stride
  if (epi != GR_EPI_LINEAR && epi != GR_EPI_SILU_MUL) return EAGLE_ERR_ARG;
  if (ind0 && (!idx0 || ind_n < 1)) return EAGLE_ERR_ARG;
  if (epi == GR_EPI_SILU_MUL && (!W2 || bias || res)) return EAGLE_ERR_ARG;
  if (M < 1 || M > GR_MAX_ROWS || N < 1 || K < 128 || (K & 127)) return EAGLE_ERR_SHAPE;
  if (x1) { if (K0 <= 0 || K0 >= K || (K0 & 127) || (ld1 & 7) || !gr_al16(x1)) return EAGLE_ERR_SHAPE; }
  else K0 = K;
  if ((ld0 & 7) || (ldw & 7) || ldw < K || !gr_al16(x0) || !gr_al16(W) || (W2 && !gr_al16(W2))) return EAGLE_ERR_SHAPE;
  if (ld0 < K0 && !idx0 && M > 1) return EAGLE_ERR_SHAPE;
  if (ldo < N || (res && ldr < N)) return EAGLE_ERR_SHAPE;
  const RowsPlan pl = gr_plan(M, N, K, epi, x1 != nullptr);
  const int64_t need = pl.ks == 1 ? 0 : GR_COUNTER_BYTES + (int64_t)pl.ks * pl.tiles * GR_NW * (pl.mt * 2) * 64 * 16;
  if (need && (!workspace || workspace_bytes < need || !gr_al16(workspace))) return EAGLE_ERR_ARG;
  RowsArgs a;
  a.x0 = (const bf16_t*)x0; a.idx0 = idx0; a.ld0 = ld0; a.K0 = K0;
  a.ind0 = ind0; a.ind_n = ind_n; a.idx0_bstride = idx0_bstride;
  a.x1 = (const bf16_t*)x1; a.idx1 = idx1; a.ld1 = ld1;
  a.W = (const bf16_t*)W; a.W2 = (const bf16_t*)W2; a.ldw = ldw;
  a.bias = (const bf16_t*)bias; a.res = (const bf16_t*)res; a.ldr = ldr;
  a.out = (bf16_t*)out; a.ldo = ldo;
  a.M = M; a.N = N; a.K = K;
  a.rotate = pl.wide ? 2 * g_rows_rotate : g_rows_rotate;   // wide tiles: every second 256-byte column (measured: 300 vs 337 us)
  a.tiles = pl.tiles; a.ks = pl.ks; a.kchunk = pl.kchunk; a.tile_cols = pl.tile_cols;
  a.cand = nullptr; a.topk = 0; a.packed = packed ? 1 : 0;
  a.slab = workspace ? (float*)((char*)workspace + GR_COUNTER_BYTES) : nullptr;
  a.counters = (pl.ks > 1 && !g_rows_two_launch) ? (int*)workspace : nullptr;
  return epi == GR_EPI_SILU_MUL ? gr_launch_mt<GR_EPI_SILU_MUL>(a, pl.mt, stream, 0) : gr_launch_mt<GR_EPI_LINEAR>(a, pl.mt, stream, pl.wide);
}


// ---------------------------------------------------------------------------------------------
// lm_head + top-k of a 81..128-row tree step without the logits: gemm_rows_wide_kernel<MT, true> leaves k candidate records per
// (row, column tile) in the workspace, eagle_topk_merge_candidates (topk.hip) picks every row's k best.  Same ids as
// eagle_gemm_rows_bf16 + eagle_topk_vocab(_grouped) on the bf16 logits (same rounding, same order: value, then lowest index);
// the values come back from the 16-bit keys (-0 as +0).  Shapes outside the wide form's range return EAGLE_ERR_SHAPE (the
// caller then takes the two-step path); eagle_lm_head_topk_rows_workspace_bytes returns 0 for them.
extern "C" int eagle_topk_merge_candidates(const void* cand, int ncand, int rows, int k, int64_t* out_idx, void* out_val,
                                           int group, int64_t group_stride, hipStream_t stream);
static bool gr_topk_plan(int M, int N, int K, int k, RowsPlan& pl) {
  if (M < 81 || M > 128 || N < 1 || K < 128 || (K & 127) || k < 1 || k > 16) return false;
  pl = gr_plan(M, N, K, GR_EPI_LINEAR, false);
  return pl.wide && (int64_t)pl.tiles * k <= 256 * 24;
}
// 1..16 rows (every tree step of ONE request) over a packed image with K = 4096: the register-resident weight stream of
// gemm_skinny.hip with the same candidate epilogue (gemm_skinny_keepx_kernel<..., TKP>), [M][blocks][k] records.
int eagle_skinny_topk_blocks(int M, int N, int K, int k);
int eagle_topk_merge_candidates_bounded(const void* cand, int ncand, int rows, int k, int64_t* out_idx, void* out_val,
                                        int group, int64_t group_stride, int index_bound, hipStream_t stream);
int eagle_launch_skinny_topk(const void* x0, const int64_t* idx0, int64_t ld0, const int64_t* ind0, int ind_n, int64_t idx0_bstride,
                             const void* Wpacked, const void* bias, int M, int N, int K, int k, void* cand, bool nt, hipStream_t st);
extern "C" int64_t eagle_lm_head_topk_rows_workspace_bytes(int M, int N, int K, int k) {
  if (M >= 1 && M <= 16) return (int64_t)M * eagle_skinny_topk_blocks(M, N, K, k) * k * 8;
  RowsPlan pl;
  if (!gr_topk_plan(M, N, K, k, pl)) return 0;
  return (int64_t)pl.mt * 16 * pl.tiles * k * 8;
}
extern "C" int eagle_lm_head_topk_rows_bf16(const void* x0, const int64_t* idx0, int64_t ld0, const int64_t* ind0, int ind_n,
                                            int64_t idx0_bstride, const void* W, int64_t ldw, const void* bias,
                                            int M, int N, int K, int k, int64_t* out_idx, void* out_val, int group,
                                            int64_t group_stride, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  if (!x0 || !W || !out_idx || !workspace) return EAGLE_ERR_ARG;
  if (ind0 && (!idx0 || ind_n < 1)) return EAGLE_ERR_ARG;
  if (M >= 1 && M <= 16) {                               // one request's tree step: packed image only
    const int blocks = eagle_skinny_topk_blocks(M, N, K, k);
    if (!blocks || ldw != 0) return EAGLE_ERR_SHAPE;
    if ((ld0 & 7) || !gr_al16(x0) || !gr_al16(W) || (reinterpret_cast<uintptr_t>(workspace) & 7)) return EAGLE_ERR_SHAPE;
    if (ld0 < K && !idx0 && M > 1) return EAGLE_ERR_SHAPE;
    if (workspace_bytes < (int64_t)M * blocks * k * 8) return EAGLE_ERR_ARG;
    int rc = eagle_launch_skinny_topk(x0, idx0, ld0, ind0, ind_n, idx0_bstride, W, bias, M, N, K, k, workspace, true, stream);
    if (rc || g_head_skip_merge) return rc;
    return eagle_topk_merge_candidates_bounded(workspace, blocks * k, M, k, out_idx, out_val, group, group_stride, N, stream);
  }
  RowsPlan pl;
  if (!gr_topk_plan(M, N, K, k, pl)) return EAGLE_ERR_SHAPE;
  const bool packed = ldw == 0;                          // ldw == 0: W is an eagle_pack_weight_bf16 image
  if (packed) ldw = K;
  if ((ld0 & 7) || (ldw & 7) || ldw < K || !gr_al16(x0) || !gr_al16(W) || (reinterpret_cast<uintptr_t>(workspace) & 7)) return EAGLE_ERR_SHAPE;
  if (ld0 < K && !idx0 && M > 1) return EAGLE_ERR_SHAPE;
  if (workspace_bytes < (int64_t)pl.mt * 16 * pl.tiles * k * 8) return EAGLE_ERR_ARG;
  RowsArgs a;
  a.x0 = (const bf16_t*)x0; a.idx0 = idx0; a.ld0 = ld0; a.K0 = K;
  a.ind0 = ind0; a.ind_n = ind_n; a.idx0_bstride = idx0_bstride;
  a.x1 = nullptr; a.idx1 = nullptr; a.ld1 = 0;
  a.W = (const bf16_t*)W; a.W2 = nullptr; a.ldw = ldw;
  a.bias = (const bf16_t*)bias; a.res = nullptr; a.ldr = 0;
  a.out = nullptr; a.ldo = 0;
  a.M = M; a.N = N; a.K = K;
  a.rotate = 2 * g_rows_rotate;
  a.tiles = pl.tiles; a.ks = 1; a.kchunk = K >> 7; a.tile_cols = 0;
  a.slab = nullptr; a.counters = nullptr;
  a.cand = (unsigned long long*)workspace; a.topk = k; a.packed = packed ? 1 : 0;
  if (packed) {
    if (pl.mt == 6) hipLaunchKernelGGL((gemm_rows_wide_kernel<6, true, true>), dim3(a.tiles), dim3(GW_NW * 64), 0, stream, a);
    else hipLaunchKernelGGL((gemm_rows_wide_kernel<8, true, true>), dim3(a.tiles), dim3(GW_NW * 64), 0, stream, a);
  } else if (pl.mt == 6) hipLaunchKernelGGL((gemm_rows_wide_kernel<6, true>), dim3(a.tiles), dim3(GW_NW * 64), 0, stream, a);
  else hipLaunchKernelGGL((gemm_rows_wide_kernel<8, true>), dim3(a.tiles), dim3(GW_NW * 64), 0, stream, a);
  int rc = eagle_check_launch();
  if (rc || g_head_skip_merge) return rc;
  return eagle_topk_merge_candidates(workspace, pl.tiles * k, M, k, out_idx, out_val, group, group_stride, stream);
}
