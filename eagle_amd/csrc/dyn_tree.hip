// Dynamic draft tree (EAGLE-2 style, as SGLang serves it): cumulative-score frontier selection per
// step, re-ranking of the candidate pool into the verified tree, tree walk acceptance.  gfx950.
//
// NOT in /root/reference: benchmark/benchmark.py:48-56 hands `speculative_num_steps`,
// `speculative_eagle_topk`, `speculative_num_draft_tokens` to sglang.Engine (SGLang v0.4.6.post5,
// benchmark/README.md:22, neither vendored nor installed).  The algorithm restated here is SGLang's
// select_top_k_tokens / organize_draft_results / greedy verify; the frozen spec
// and its oracle are in DESIGN.md section 6b and oracle/dyn_tree.py (parity unpinned).
//
//   pool    = every candidate ever produced: k children of the root, then k*k per later step
//             (the k children of each of the k frontier nodes), in production order.
//   score   = product of softmax probabilities along the path (fp32).
//   order   = score descending, pool index ascending (a parent always precedes its children).
// One workgroup per request (grid = B independent requests, every per-request array is [B][...] contiguous); integer / small fp work.
#include "common.h"

#define DT_MAXK 16
#define DT_MAXP 1024      // pool entries: k + (S-1) k^2
#define DT_MAXT 64        // verified tree nodes incl. the root

// lse[r] = log(sum_v exp(x[r][v])) in fp32, one 1024-thread workgroup per row: 16-byte loads, an online
// (max, sum) pair per thread updated once per 8 logits, wave then workgroup reduction.
#define LSE_THREADS 1024
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
  const float M = fmaxf(m, m2);
  if (M == -INFINITY) { m = M; s = 0.f; return; }
  s = s * __expf(m - M) + s2 * __expf(m2 - M);
  m = M;
}
__global__ __launch_bounds__(LSE_THREADS) void row_lse_kernel(const bf16_t* __restrict__ x, int64_t ld, int V, float* __restrict__ lse) {
  __shared__ float sm[LSE_THREADS / 64], ss[LSE_THREADS / 64];
  const bf16_t* xr = x + (int64_t)blockIdx.x * ld;
  float m = -INFINITY, s = 0.f;
  const bool vec = ((ld & 7) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  const int nv = vec ? (V >> 3) : 0;
  for (int c = threadIdx.x; c < nv; c += LSE_THREADS) {
    const uint4 q = *reinterpret_cast<const uint4*>(xr + 8 * c);
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
    float f[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = lo_bf(w[i]); f[2 * i + 1] = hi_bf(w[i]); }
    float mx = f[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) mx = fmaxf(mx, f[i]);
    if (mx > -INFINITY) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += __expf(f[i] - mx);
      lse_merge(m, s, mx, acc);
    }
  }
  for (int v = 8 * nv + threadIdx.x; v < V; v += LSE_THREADS) {
    const float f = bf2f(xr[v]);
    if (f > -INFINITY) lse_merge(m, s, f, 1.f);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    lse_merge(m, s, m2, s2);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sm[wave] = m; ss[wave] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = -INFINITY, S = 0.f;
    for (int w = 0; w < LSE_THREADS / 64; ++w) lse_merge(M, S, sm[w], ss[w]);
    lse[blockIdx.x] = M + __logf(S);
  }
}

extern "C" int eagle_row_lse_bf16(const void* logits, int64_t ld, int rows, int V, float* lse, hipStream_t stream) {
  if (!logits || !lse) return EAGLE_ERR_ARG;
  if (rows < 1 || V < 1 || ld < V) return EAGLE_ERR_SHAPE;
  hipLaunchKernelGGL(row_lse_kernel, dim3(rows), dim3(LSE_THREADS), 0, stream, (const bf16_t*)logits, ld, V, lse);
  return eagle_check_launch();
}

// One expansion step.  step 0: R = 1 row (the root's top-k); step i >= 1: R = k rows (the frontier's).
//   candidate (j, c): token ids[j][c], score = fscore_in[j] * exp(vals[j][c] - lse[j])  (step 0: 1 * ...),
//   pool index = (step 0) c  |  k + (step-1) k^2 + j k + c, parent = (step 0) -1 | fpool_in[j].
//   unless `last`: the k best candidates become the next frontier (rank order): token, score, pool index,
//   the frontier row whose output hidden state they take (parent_row; step 0: *root_row), and their
//   attention-mask row over the tree rows appended so far: parent's row (width step*k) | own column.
__global__ __launch_bounds__(256) void dyn_step_kernel(int step, int k, int last, const bf16_t* __restrict__ vals,
                                                       const int64_t* __restrict__ ids, const float* __restrict__ lse,
                                                       const float* __restrict__ fscore_in, const int32_t* __restrict__ fpool_in,
                                                       const uint8_t* __restrict__ mask_in, const int64_t* __restrict__ root_row,
                                                       int64_t* __restrict__ pool_tok, float* __restrict__ pool_score,
                                                       int32_t* __restrict__ pool_parent, int64_t* __restrict__ ftok,
                                                       float* __restrict__ fscore_out, int32_t* __restrict__ fpool_out,
                                                       int64_t* __restrict__ parent_row, uint8_t* __restrict__ mask_out, int P) {
  __shared__ float sc[DT_MAXK * DT_MAXK];
  const int t = threadIdx.x, b = blockIdx.x;
  const int R = step == 0 ? 1 : k, n = R * k;
  const int base = step == 0 ? 0 : k + (step - 1) * k * k;
  // request b's slices
  vals += (int64_t)b * n; ids += (int64_t)b * n; lse += (int64_t)b * R;
  pool_tok += (int64_t)b * P; pool_score += (int64_t)b * P; pool_parent += (int64_t)b * P;
  if (fscore_in) fscore_in += (int64_t)b * k;
  if (fpool_in) fpool_in += (int64_t)b * k;
  if (mask_in) mask_in += (int64_t)b * k * (step * k);
  if (ftok) { ftok += (int64_t)b * k; fscore_out += (int64_t)b * k; fpool_out += (int64_t)b * k; parent_row += (int64_t)b * k;
              mask_out += (int64_t)b * k * ((step + 1) * k); }
  if (t < n) {
    const int j = t / k;
    const float p = __expf(bf2f(vals[t]) - lse[j]);
    const float s = step == 0 ? p : fscore_in[j] * p;
    sc[t] = s;
    pool_tok[base + t] = ids[t];
    pool_score[base + t] = s;
    pool_parent[base + t] = step == 0 ? -1 : fpool_in[j];
  }
  __syncthreads();
  if (last || t >= n) return;
  int rank = 0;
  const float s = sc[t];
  for (int q = 0; q < n; ++q) rank += (sc[q] > s || (sc[q] == s && q < t)) ? 1 : 0;
  if (rank >= k) return;
  const int j = t / k;
  ftok[rank] = ids[t];
  fscore_out[rank] = s;
  fpool_out[rank] = base + t;
  parent_row[rank] = step == 0 ? root_row[b] : (int64_t)b * k + j;      // row of the previous (batched) forward
  const int w_in = step * k, w_out = (step + 1) * k;
  for (int c = 0; c < w_out; ++c) {
    uint8_t v = 0;
    if (c < w_in) v = mask_in[j * w_in + c];
    else if (c == w_in + rank) v = 1;
    mask_out[rank * w_out + c] = v;
  }
}

extern "C" int eagle_dyn_step(int step, int k, int last, const void* vals, const int64_t* ids, const float* lse,
                              const float* fscore_in, const int32_t* fpool_in, const uint8_t* mask_in,
                              const int64_t* root_row, int64_t* pool_tok, float* pool_score, int32_t* pool_parent,
                              int64_t* ftok, float* fscore_out, int32_t* fpool_out, int64_t* parent_row,
                              uint8_t* mask_out, int S, int B, hipStream_t stream) {
  if (!vals || !ids || !lse || !pool_tok || !pool_score || !pool_parent) return EAGLE_ERR_ARG;
  if (step < 0 || k < 1 || k > DT_MAXK || B < 1 || S < 1 || step >= S) return EAGLE_ERR_SHAPE;
  if (step > 0 && (!fscore_in || !fpool_in)) return EAGLE_ERR_ARG;
  if (!last && (!ftok || !fscore_out || !fpool_out || !parent_row || !mask_out || (step > 0 && !mask_in) ||
                (step == 0 && !root_row))) return EAGLE_ERR_ARG;
  const int P = k + (S - 1) * k * k;
  if (P > DT_MAXP) return EAGLE_ERR_SHAPE;
  hipLaunchKernelGGL(dyn_step_kernel, dim3(B), dim3(256), 0, stream, step, k, last, (const bf16_t*)vals, ids, lse, fscore_in,
                     fpool_in, mask_in, root_row, pool_tok, pool_score, pool_parent, ftok, fscore_out, fpool_out, parent_row,
                     mask_out, P);
  return eagle_check_launch();
}

// Re-rank the pool into the tree that is verified: the T-1 best pool entries (score desc, index asc),
// kept in pool order, become nodes 1..T-1; node 0 is the root token.
//   tree_tok[T], tree_parent[T] (node ids, root -1), tree_pos[T] (depth), tree_mask[T][T] (self, root,
//   ancestors), sel[T-1] (pool index of node n+1).
__global__ __launch_bounds__(256) void dyn_finalize_kernel(int k, int P, int T, const int64_t* __restrict__ pool_tok,
                                                           const float* __restrict__ pool_score,
                                                           const int32_t* __restrict__ pool_parent,
                                                           const int64_t* __restrict__ root_token, int64_t* __restrict__ tree_tok,
                                                           int32_t* __restrict__ tree_parent, int64_t* __restrict__ tree_pos,
                                                           uint8_t* __restrict__ tree_mask, int32_t* __restrict__ sel) {
  __shared__ float sc[DT_MAXP];
  __shared__ int16_t node_of[DT_MAXP];       // 0: not selected, else node id
  __shared__ int32_t par[DT_MAXT];
  const int t = threadIdx.x, b = blockIdx.x;
  pool_tok += (int64_t)b * P; pool_score += (int64_t)b * P; pool_parent += (int64_t)b * P; root_token += b;
  tree_tok += (int64_t)b * T; tree_parent += (int64_t)b * T; tree_pos += (int64_t)b * T; tree_mask += (int64_t)b * T * T;
  sel += (int64_t)b * (T - 1);
  for (int p = t; p < P; p += 256) sc[p] = pool_score[p];
  __syncthreads();
  for (int p = t; p < P; p += 256) {
    const float s = sc[p];
    int rank = 0;
    for (int q = 0; q < P; ++q) rank += (sc[q] > s || (sc[q] == s && q < p)) ? 1 : 0;
    node_of[p] = rank < T - 1 ? 1 : 0;
  }
  __syncthreads();
  int16_t mine[DT_MAXP / 256];
  for (int i = 0, p = t; p < P; p += 256, ++i) {
    int before = 0;
    for (int q = 0; q < p; ++q) before += node_of[q] ? 1 : 0;
    mine[i] = node_of[p] ? (int16_t)(1 + before) : 0;
  }
  __syncthreads();
  for (int i = 0, p = t; p < P; p += 256, ++i) node_of[p] = mine[i];
  if (t == 0) { tree_tok[0] = root_token[0]; tree_parent[0] = -1; tree_pos[0] = 0; par[0] = -1; }
  __syncthreads();
  for (int p = t; p < P; p += 256) {
    const int n = node_of[p];
    if (!n) continue;
    const int pp = pool_parent[p];
    const int pn = pp < 0 ? 0 : node_of[pp];
    tree_tok[n] = pool_tok[p];
    tree_parent[n] = pn;
    par[n] = pn;
    tree_pos[n] = p < k ? 1 : 2 + (p - k) / (k * k);
    sel[n - 1] = p;
  }
  __syncthreads();
  for (int i = t; i < T * T; i += 256) tree_mask[i] = 0;
  __syncthreads();
  if (t < T) {
    tree_mask[t * T + t] = 1;
    tree_mask[t * T] = 1;
    for (int a = par[t]; a > 0; a = par[a]) tree_mask[t * T + a] = 1;
  }
}

extern "C" int eagle_dyn_finalize(int k, int S, int T, const int64_t* pool_tok, const float* pool_score,
                                  const int32_t* pool_parent, const int64_t* root_token, int64_t* tree_tok,
                                  int32_t* tree_parent, int64_t* tree_pos, uint8_t* tree_mask, int32_t* sel,
                                  int B, hipStream_t stream) {
  if (!pool_tok || !pool_score || !pool_parent || !root_token || !tree_tok || !tree_parent || !tree_pos || !tree_mask || !sel)
    return EAGLE_ERR_ARG;
  if (k < 1 || k > DT_MAXK || S < 1 || B < 1) return EAGLE_ERR_SHAPE;
  const int P = k + (S - 1) * k * k;
  if (P > DT_MAXP || T < 2 || T > DT_MAXT || T - 1 > P) return EAGLE_ERR_SHAPE;
  hipLaunchKernelGGL(dyn_finalize_kernel, dim3(B), dim3(256), 0, stream, k, P, T, pool_tok, pool_score, pool_parent, root_token,
                     tree_tok, tree_parent, tree_pos, tree_mask, sel);
  return eagle_check_launch();
}

// Greedy acceptance by walking the tree: from the root, follow the child whose token equals the
// verifier's argmax at the current node (siblings carry distinct tokens; should two match, the lowest
// node id wins) until no child matches or the path holds W nodes.
//   path[W] (node ids, -1 padded; path[0] = 0), result = {0, accept_length, next_token, prev_len + accept_length + 1},
//   next_ids[W] (tokens following each accepted row, zero padded), n_acc, and the draft-side state of
//   the next cycle exactly as eagle_greedy_accept leaves it (last_idx, len_cur, len_nxt, ring cursor).
__global__ __launch_bounds__(64) void dyn_accept_kernel(int T, int W, const int64_t* __restrict__ tree_tok,
                                                       const int32_t* __restrict__ tree_parent,
                                                       const int64_t* __restrict__ target_argmax,
                                                       const int32_t* __restrict__ prev_len, int64_t* __restrict__ path,
                                                       int64_t* __restrict__ result, int64_t* __restrict__ next_ids,
                                                       int32_t* __restrict__ n_acc, int64_t* __restrict__ last_idx,
                                                       int32_t* __restrict__ len_cur, int32_t* __restrict__ len_nxt,
                                                       int64_t* __restrict__ cursor) {
  __shared__ int pth[64];
  const int lane = threadIdx.x, b = blockIdx.x;
  if (cursor && lane == 0 && b == 0) { const int64_t c = cursor[0]; cursor[1] = c; cursor[0] = c + 1; }
  tree_tok += (int64_t)b * T; tree_parent += (int64_t)b * T; target_argmax += (int64_t)b * T;
  path += (int64_t)b * W; result += (int64_t)b * 4;
  if (next_ids) next_ids += (int64_t)b * W;
  const int64_t tok = lane < T ? tree_tok[lane] : -1;
  const int par = lane < T ? tree_parent[lane] : -2;
  int cur = 0, alen = 0;
  if (lane == 0) { path[0] = 0; pth[0] = 0; }
  while (alen < W - 1) {
    const int64_t want = target_argmax[cur];
    const unsigned long long hit = __ballot(lane < T && lane > 0 && par == cur && tok == want);
    if (!hit) break;
    cur = __ffsll((long long)hit) - 1;
    ++alen;
    if (lane == 0) { path[alen] = cur; pth[alen] = cur; }
  }
  if (lane == 0) {
    const int64_t nxt = target_argmax[cur];
    for (int j = alen + 1; j < W; ++j) path[j] = -1;
    result[0] = 0; result[1] = alen; result[2] = nxt; result[3] = (prev_len ? prev_len[b] : 0) + alen + 1;
    if (n_acc) n_acc[b] = alen + 1;
    if (last_idx) last_idx[b] = (int64_t)b * W + alen;                  // row of the next (batched, W rows per request) catch-up forward
    if (len_cur && len_nxt) { const int32_t l = len_nxt[b]; len_cur[b] = l; len_nxt[b] = l + alen + 1; }
  }
  if (next_ids) {
    // lane j < alen: token of the (j+1)-th accepted node; lane alen: the verifier's next token
    __syncthreads();
    if (lane < W) {
      int64_t v = 0;
      if (lane < alen) v = tree_tok[pth[lane + 1]];
      else if (lane == alen) v = target_argmax[cur];
      next_ids[lane] = v;
    }
  }
}

extern "C" int eagle_dyn_accept(int T, int W, const int64_t* tree_tok, const int32_t* tree_parent,
                                const int64_t* target_argmax, const int32_t* prev_len, int64_t* path, int64_t* result,
                                int64_t* next_ids, int32_t* n_acc, int64_t* last_idx, int32_t* len_cur, int32_t* len_nxt,
                                int64_t* cursor, int B, hipStream_t stream) {
  if (!tree_tok || !tree_parent || !target_argmax || !path || !result) return EAGLE_ERR_ARG;
  if (T < 1 || T > DT_MAXT || W < 1 || W > 64 || B < 1) return EAGLE_ERR_SHAPE;
  hipLaunchKernelGGL(dyn_accept_kernel, dim3(B), dim3(64), 0, stream, T, W, tree_tok, tree_parent, target_argmax, prev_len, path,
                     result, next_ids, n_acc, last_idx, len_cur, len_nxt, cursor);
  return eagle_check_launch();
}
