// Static-tree index kernels: draft-token selection, greedy acceptance, KV / hidden compaction.
// All integer work; results are bit-exact against oracle/tree.py.
//
// None of this is inside /root/reference (init_tree refers to undefined names,
// eagle/llama2.py:453-455); the semantics are the upstream EAGLE-1 ones frozen in DESIGN.md
// (SURVEY.md rows a19-a23).
#include "common.h"

// ids_out[b][j] = flat_ids[b][tree_idx[j]]   (input_ids of one expansion step, row a20)
__global__ void tree_gather_ids_kernel(const int64_t* __restrict__ flat_ids, int64_t flat_bstride,
                                       const int64_t* __restrict__ tree_idx, int n, int64_t* __restrict__ out, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n) return;
  const int b = i / n, j = i - b * n;
  out[i] = flat_ids[(int64_t)b * flat_bstride + tree_idx[j]];
}

extern "C" int eagle_tree_gather_ids(const int64_t* flat_ids, int64_t flat_batch_stride, const int64_t* tree_idx, int n,
                                     int64_t* out_ids, int B, hipStream_t stream) {
  if (!flat_ids || !tree_idx || !out_ids) return EAGLE_ERR_ARG;
  if (n < 1 || B < 1) return EAGLE_ERR_SHAPE;
  hipLaunchKernelGGL(tree_gather_ids_kernel, dim3((B * n + 127) / 128), dim3(128), 0, stream, flat_ids, flat_batch_stride,
                     tree_idx, n, out_ids, B);
  return eagle_check_launch();
}

// One workgroup (one wave) per request.
//   tree_tok[n]   = cat(root_token, flat_ids)[tree_indices[n]]
//   cart[p][j]    = tree_tok[retrieve[p][j]]  (-1 where retrieve is -1)
//   accept[p]     = number of leading j >= 1 with cart[p][j] == target_argmax[retrieve[p][j-1]]
//   accept_length = max_p accept[p]; best = first p reaching it (0 when nothing accepted)
//   next_token    = target_argmax[retrieve[best][accept_length]]
//   result[b]     = {best, accept_length, next_token, prev_len + accept_length + 1}
__global__ __launch_bounds__(64) void greedy_accept_kernel(const int64_t* __restrict__ root_token,
                                                          const int64_t* __restrict__ flat_ids, int64_t flat_bstride,
                                                          const int64_t* __restrict__ tree_indices, int n_nodes,
                                                          const int64_t* __restrict__ retrieve, int P, int W,
                                                          const int64_t* __restrict__ target_argmax,
                                                          const int32_t* __restrict__ prev_len,
                                                          int64_t* __restrict__ tree_tok_out, int64_t* __restrict__ cart_out,
                                                          int32_t* __restrict__ accept_out, int64_t* __restrict__ result,
                                                          int64_t* __restrict__ next_ids_out, int32_t* __restrict__ n_acc_out,
                                                          int64_t* __restrict__ last_idx_out, int64_t* __restrict__ hid_idx0_out,
                                                          int n0, int32_t* __restrict__ len_cur, int32_t* __restrict__ len_nxt,
                                                          int64_t* __restrict__ cursor) {
  extern __shared__ int64_t tok[];  // [n_nodes]
  const int b = blockIdx.x, lane = threadIdx.x;
  // ring cursor of the verifier's output buffers: kernels BEFORE this one in the chain (the argmax)
  // read cursor[0], kernels AFTER it (the accepted-hidden gather) read cursor[1] = the same value
  if (cursor && b == 0 && lane == 0) { const int64_t c = cursor[0]; cursor[1] = c; cursor[0] = c + 1; }
  const int64_t* fl = flat_ids + (int64_t)b * flat_bstride;
  const int64_t* tgt = target_argmax + (int64_t)b * n_nodes;
  for (int n = lane; n < n_nodes; n += 64) {
    const int64_t ti = tree_indices[n];
    const int64_t v = ti == 0 ? root_token[b] : fl[ti - 1];
    tok[n] = v;
    if (tree_tok_out) tree_tok_out[(int64_t)b * n_nodes + n] = v;
  }
  __syncthreads();
  int best_len = -1, best_p = 0;
  for (int p0 = 0; p0 < P; p0 += 64) {
    const int p = p0 + lane;
    int acc = -1;
    if (p < P) {
      acc = 0;
      bool alive = true;
      for (int j = 0; j < W; ++j) {
        const int64_t node = retrieve[(int64_t)p * W + j];
        const int64_t c = node >= 0 ? tok[node] : -1;
        if (cart_out) cart_out[((int64_t)b * P + p) * W + j] = c;
        if (j >= 1) {
          const int64_t pn = retrieve[(int64_t)p * W + j - 1];
          const bool ok = alive && node >= 0 && pn >= 0 && c == tgt[pn];
          alive = ok;
          acc += ok ? 1 : 0;
        }
      }
      if (accept_out) accept_out[(int64_t)b * P + p] = acc;
    }
    // wave argmax (first max): key = acc * 2^20 + (2^20-1 - p)
    long long key = acc < 0 ? -1 : (((long long)acc) << 20) | (long long)(0xfffff - p);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      long long x = __shfl_xor(key, o, 64);
      key = x > key ? x : key;
    }
    const int l = (int)(key >> 20), pp = 0xfffff - (int)(key & 0xfffff);
    if (key >= 0 && l > best_len) { best_len = l; best_p = pp; }
  }
  if (lane == 0) {
    if (best_len <= 0) { best_len = 0; best_p = 0; }
    const int64_t node = retrieve[(int64_t)best_p * W + best_len];
    result[(int64_t)b * 4 + 0] = best_p;
    result[(int64_t)b * 4 + 1] = best_len;
    result[(int64_t)b * 4 + 2] = tgt[node];
    result[(int64_t)b * 4 + 3] = (prev_len ? prev_len[b] : 0) + best_len + 1;
    // inputs of the next draft cycle: the token FOLLOWING each accepted hidden row
    // (accepted draft tokens 1..accept_length, then the verifier's next token), zero padded.
    if (next_ids_out) {
      for (int j = 0; j < W; ++j) {
        int64_t v = 0;
        if (j < best_len) v = tok[retrieve[(int64_t)best_p * W + j + 1]];
        else if (j == best_len) v = tgt[node];
        next_ids_out[(int64_t)b * W + j] = v;
      }
    }
    if (n_acc_out) n_acc_out[b] = best_len + 1;
    // draft-side state of the next cycle (all device resident, no host round trip):
    //   last_idx = row of the last valid catch-up row in the [B*W] padded layout,
    //   hid_idx0 = that row repeated for the n0 depth-1 nodes,
    //   len_cur  = draft-cache rows before the catch-up forward, len_nxt = after it.
    const int64_t last = (int64_t)b * W + best_len;
    if (last_idx_out) last_idx_out[b] = last;
    if (hid_idx0_out) for (int j = 0; j < n0; ++j) hid_idx0_out[(int64_t)b * n0 + j] = last;
    if (len_cur && len_nxt) {
      const int32_t l = len_nxt[b];
      len_cur[b] = l;
      len_nxt[b] = l + best_len + 1;
    }
  }
}

extern "C" int eagle_greedy_accept(const int64_t* root_token, const int64_t* flat_ids, int64_t flat_batch_stride,
                                   const int64_t* tree_indices, int n_nodes, const int64_t* retrieve, int P, int W,
                                   const int64_t* target_argmax, const int32_t* prev_len, int64_t* tree_tok_out,
                                   int64_t* cart_out, int32_t* accept_out, int64_t* result, int64_t* next_ids_out,
                                   int32_t* n_acc_out, int64_t* last_idx_out, int64_t* hid_idx0_out, int n0,
                                   int32_t* len_cur, int32_t* len_nxt, int64_t* cursor, int B, hipStream_t stream) {
  if (!root_token || !flat_ids || !tree_indices || !retrieve || !target_argmax || !result) return EAGLE_ERR_ARG;
  if (n_nodes < 1 || n_nodes > 4096 || P < 1 || P >= (1 << 20) || W < 1 || B < 1) return EAGLE_ERR_SHAPE;
  hipLaunchKernelGGL(greedy_accept_kernel, dim3(B), dim3(64), (size_t)n_nodes * 8, stream, root_token, flat_ids,
                     flat_batch_stride, tree_indices, n_nodes, retrieve, P, W, target_argmax, prev_len, tree_tok_out,
                     cart_out, accept_out, result, next_ids_out, n_acc_out, last_idx_out, hid_idx0_out, n0, len_cur, len_nxt, cursor);
  return eagle_check_launch();
}

// In-place compaction of accepted rows (row a23).  cache viewed as [n_slabs][cap][row_elems]
// with slab -> request b = (slab / slabs_per_batch_inner) % B is given by batch_of_slab layout:
//   slab = (outer * B + b) * inner + i      (e.g. outer = layer*2+{k,v}, inner = kv head)
// For request b: src row = prev_len[b] + retrieve[best_b][j], dst row = prev_len[b] + j,
// j = 0..accept_length_b.  src >= dst always; all sources are staged in LDS before any store.
#define KC_MAXW 16
__global__ __launch_bounds__(256) void kv_compact_kernel(bf16_t* __restrict__ cache, int64_t slab_stride, int row_elems,
                                                         int B, int inner, const int64_t* __restrict__ retrieve, int W,
                                                         const int64_t* __restrict__ result,
                                                         const int32_t* __restrict__ prev_len, int cap,
                                                         int* __restrict__ err_flag) {
  extern __shared__ __attribute__((aligned(16))) char kc_smem[];
  uint32_t* stage = reinterpret_cast<uint32_t*>(kc_smem);  // [W][row_elems/2]
  const int slab = blockIdx.x;
  const int b = (slab / inner) % B;
  const int best = (int)result[(int64_t)b * 4 + 0], alen = (int)result[(int64_t)b * 4 + 1];
  const int prev = prev_len[b];
  const int nrow = alen + 1;
  const int rw = row_elems >> 1;
  if (nrow > W || prev < 0 || prev + nrow > cap) {
    if (threadIdx.x == 0 && err_flag) atomicExch(err_flag, 2);
    return;
  }
  uint32_t* base = reinterpret_cast<uint32_t*>(cache + (int64_t)slab * slab_stride);
  for (int i = threadIdx.x; i < nrow * rw; i += blockDim.x) {
    const int j = i / rw, e = i - j * rw;
    const int64_t node = retrieve[(int64_t)best * W + j];
    const int64_t src = prev + node;
    stage[i] = (node >= 0 && src < cap) ? base[src * rw + e] : 0u;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nrow * rw; i += blockDim.x) {
    const int j = i / rw, e = i - j * rw;
    base[(int64_t)(prev + j) * rw + e] = stage[i];
  }
}

extern "C" int eagle_kv_compact_bf16(void* cache, int64_t n_slabs, int64_t slab_stride, int row_elems, int B, int inner,
                                     const int64_t* retrieve, int W, const int64_t* result, const int32_t* prev_len,
                                     int cap, int* err_flag, hipStream_t stream) {
  if (!cache || !retrieve || !result || !prev_len) return EAGLE_ERR_ARG;
  if (n_slabs < 1 || row_elems < 2 || (row_elems & 1) || B < 1 || inner < 1 || W < 1 || W > KC_MAXW) return EAGLE_ERR_SHAPE;
  if (slab_stride < (int64_t)cap * row_elems || (slab_stride & 1)) return EAGLE_ERR_SHAPE;
  const size_t lds = (size_t)W * row_elems * 2;
  if (lds > 64 * 1024) return EAGLE_ERR_SHAPE;
  hipLaunchKernelGGL(kv_compact_kernel, dim3((unsigned)n_slabs), dim3(256), lds, stream, (bf16_t*)cache, slab_stride,
                     row_elems, B, inner, retrieve, W, result, prev_len, cap, err_flag);
  return eagle_check_launch();
}

// out[b][j][:] = src[b][retrieve[best_b][j]][:] for j <= accept_length_b, zeros after (row a23:
// the accepted path's hidden states for the next draft cycle).
__global__ __launch_bounds__(256) void gather_accepted_kernel(const bf16_t* __restrict__ src, int n_nodes, int D,
                                                              const int64_t* __restrict__ retrieve, int W,
                                                              const int64_t* __restrict__ result, bf16_t* __restrict__ out,
                                                              int64_t* __restrict__ root_io, int32_t* __restrict__ prev_len_io,
                                                              const int64_t* __restrict__ src_block, int B) {
  const int b = blockIdx.y, j = blockIdx.x;
  if (src_block) src += *src_block * (int64_t)B * n_nodes * D;
  if (j == 0 && threadIdx.x == 0) {  // verifier-side state for the next cycle
    if (root_io) root_io[b] = result[(int64_t)b * 4 + 2];
    if (prev_len_io) prev_len_io[b] = (int32_t)result[(int64_t)b * 4 + 3];
  }
  const int best = (int)result[(int64_t)b * 4 + 0], alen = (int)result[(int64_t)b * 4 + 1];
  const int64_t node = retrieve[(int64_t)best * W + j];
  const bool live = j <= alen && node >= 0 && node < n_nodes;
  const uint4* s = reinterpret_cast<const uint4*>(src + ((int64_t)b * n_nodes + (live ? node : 0)) * D);
  uint4* o = reinterpret_cast<uint4*>(out + ((int64_t)b * W + j) * D);
  for (int i = threadIdx.x; i < (D >> 3); i += blockDim.x) o[i] = live ? s[i] : make_uint4(0, 0, 0, 0);
}

extern "C" int eagle_gather_accepted_bf16(const void* src, int n_nodes, int D, const int64_t* retrieve, int W,
                                          const int64_t* result, void* out, int64_t* root_io, int32_t* prev_len_io,
                                          const int64_t* src_block, int B, hipStream_t stream) {
  if (!src || !retrieve || !result || !out) return EAGLE_ERR_ARG;
  if (n_nodes < 1 || D < 8 || (D & 7) || W < 1 || B < 1) return EAGLE_ERR_SHAPE;
  hipLaunchKernelGGL(gather_accepted_kernel, dim3(W, B), dim3(256), 0, stream, (const bf16_t*)src, n_nodes, D, retrieve, W,
                     result, (bf16_t*)out, root_io, prev_len_io, src_block, B);
  return eagle_check_launch();
}
