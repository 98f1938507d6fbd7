// Static draft tree -> every index buffer of the draft side and of the verify side (a21), built on
// the device through the C ABI, so that a host in any language can set a tree up (gfx950).
//
// Llama2Model.init_tree names `mc_sim_7b_63` and `generate_tree_buffers` without defining them
// (eagle/llama2.py:453-455); the algorithm is upstream EAGLE-1's, frozen in DESIGN.md section 6.  The same
// analysis routine runs on the host (eagle_tree_layout: validation and buffer sizes) and in the kernel
// (eagle_tree_build: contents), so the two cannot disagree.  Integer work on <= 64 nodes: one workgroup, thread 0
// analyses the tree into LDS, all threads write the outputs.
#include "common.h"

#define TB_MAX_NODES 64     // including the root
#define TB_MAX_W 16         // max depth + 1

struct TreeTopo {
  int n, max_depth, n_steps, n_fwd, n_flat_rows, P, W, mask_bytes;
  int parent[TB_MAX_NODES], depth[TB_MAX_NODES], choice[TB_MAX_NODES], nkids[TB_MAX_NODES], flat_row[TB_MAX_NODES];
  int src[TB_MAX_NODES];                 // input path index of each node (root: -1)
  int step_n[TB_MAX_W], step_first[TB_MAX_W], step_mask_off[TB_MAX_W], step_cols[TB_MAX_W];
  int fwd[TB_MAX_NODES];                 // forwarded nodes in node-id order (= draft KV order); step d owns a contiguous run
  int prow[TB_MAX_NODES];                // per forwarded node: rank of its parent among the previous step's nodes
  int leaf[TB_MAX_NODES];                // leaves in retrieve-row order
};

// lexicographic compare of two root->node chains (node ids), -1 padding sorting last
__host__ __device__ static int tb_chain(const TreeTopo& t, int node, int* out) {   // out[0..depth] = root .. node
  const int d = t.depth[node];
  for (int i = d, x = node; i >= 0; --i) { out[i] = x; x = x > 0 ? t.parent[x] : 0; }
  return d + 1;
}

// Returns 0 or an EAGLE_ERR_* code.  paths [n_paths, width] int32, -1 padded.
__host__ __device__ static int tb_analyse(const int32_t* paths, int n_paths, int width, int topk, int expand_all, TreeTopo& t) {
  if (n_paths < 1 || n_paths + 1 > TB_MAX_NODES || width < 1 || width >= TB_MAX_W || topk < 1) return EAGLE_ERR_SHAPE;
  int len[TB_MAX_NODES];
  for (int i = 0; i < n_paths; ++i) {
    int l = 0;
    while (l < width && paths[i * width + l] >= 0) ++l;
    for (int j = l; j < width; ++j) if (paths[i * width + j] != -1) return EAGLE_ERR_SHAPE;
    if (l < 1) return EAGLE_ERR_SHAPE;
    for (int j = 0; j < l; ++j) if (paths[i * width + j] >= topk) return EAGLE_ERR_SHAPE;   // choice outside top-k
    len[i] = l;
  }
  // node id = 1 + rank under (length, path)
  t.n = n_paths + 1;
  t.parent[0] = -1; t.depth[0] = 0; t.choice[0] = -1; t.src[0] = -1;
  for (int i = 0; i < n_paths; ++i) {
    int rank = 0;
    for (int j = 0; j < n_paths; ++j) {
      if (j == i) continue;
      int c = len[j] - len[i];
      for (int q = 0; c == 0 && q < len[i]; ++q) c = paths[j * width + q] - paths[i * width + q];
      if (c == 0) return EAGLE_ERR_SHAPE;                    // duplicate path
      if (c < 0) ++rank;
    }
    t.src[rank + 1] = i;
    t.depth[rank + 1] = len[i];
    t.choice[rank + 1] = paths[i * width + len[i] - 1];
  }
  t.max_depth = 0;
  for (int v = 0; v < t.n; ++v) { t.nkids[v] = 0; if (t.depth[v] > t.max_depth) t.max_depth = t.depth[v]; }
  for (int v = 1; v < t.n; ++v) {
    const int d = t.depth[v], pi = t.src[v];
    int par = d == 1 ? 0 : -1;
    for (int u = 1; par < 0 && u < v; ++u) {
      if (t.depth[u] != d - 1) continue;
      bool same = true;
      for (int q = 0; same && q < d - 1; ++q) same = paths[t.src[u] * width + q] == paths[pi * width + q];
      if (same) par = u;
    }
    if (par < 0) return EAGLE_ERR_SHAPE;                      // path without its parent in the tree
    t.parent[v] = par;
    ++t.nkids[par];
  }
  // forwarded nodes and the flat top-k row each produces (row 0 = the catch-up forward)
  t.n_fwd = 0;
  t.flat_row[0] = 0;
  for (int v = 1; v < t.n; ++v) {
    const bool f = expand_all || t.nkids[v] > 0;
    t.flat_row[v] = f ? 1 + t.n_fwd : -1;
    if (f) t.fwd[t.n_fwd++] = v;
  }
  t.n_flat_rows = 1 + t.n_fwd;
  t.n_steps = expand_all ? t.max_depth : t.max_depth - 1;
  // draft side: node ids are sorted by depth first, so step d owns a contiguous run of fwd[]
  int pos = 0, cols = 0, moff = 0;
  for (int d = 1; d <= t.n_steps; ++d) {
    const int first = pos;
    while (pos < t.n_fwd && t.depth[t.fwd[pos]] == d) ++pos;
    const int nd = pos - first;
    if (nd < 1) return EAGLE_ERR_SHAPE;                       // no node to expand at depth d
    t.step_first[d - 1] = first; t.step_n[d - 1] = nd;
    cols += nd;
    t.step_cols[d - 1] = cols; t.step_mask_off[d - 1] = moff;
    moff += nd * cols;
    for (int j = first; j < pos; ++j) {
      const int par = t.parent[t.fwd[j]];
      int r = -1;
      if (d == 1) r = par == 0 ? 0 : -1;
      else for (int q = 0; q < t.step_n[d - 2]; ++q) if (t.fwd[t.step_first[d - 2] + q] == par) r = q;
      if (r < 0) return EAGLE_ERR_SHAPE;                      // a forwarded node's parent was not forwarded
      t.prow[j] = r;
    }
  }
  t.mask_bytes = moff;
  for (int v = 1; v < t.n; ++v) if (t.flat_row[t.parent[v]] < 0) return EAGLE_ERR_SHAPE;   // parent never forwarded
  // verify side: one retrieve row per leaf, rows sorted lexicographically by node ids (-1 padding last)
  t.W = t.max_depth + 1;
  t.P = 0;
  int leaves[TB_MAX_NODES];
  for (int v = 1; v < t.n; ++v) if (t.nkids[v] == 0) leaves[t.P++] = v;
  for (int a = 0; a < t.P; ++a) {
    int ca[TB_MAX_W], rank = 0;
    const int la = tb_chain(t, leaves[a], ca);
    for (int b = 0; b < t.P; ++b) {
      if (b == a) continue;
      int cb[TB_MAX_W];
      const int lb = tb_chain(t, leaves[b], cb);
      int c = 0;
      for (int q = 0; c == 0 && q < t.W; ++q) {
        const int xa = q < la ? ca[q] : TB_MAX_NODES + 5, xb = q < lb ? cb[q] : TB_MAX_NODES + 5;
        c = xb - xa;
      }
      if (c < 0) ++rank;
    }
    t.leaf[rank] = leaves[a];
  }
  return EAGLE_OK;
}

struct TreeBuildOut {
  int64_t* tree_indices; int64_t* tree_position_ids; uint8_t* tree_attn_mask; int64_t* retrieve_indices;
  int32_t* parent; int32_t* flat_row;
  int64_t* step_tree_indices; int64_t* step_parent_row; int32_t* step_repeat_nums; uint8_t* step_attn_mask;
};

__global__ __launch_bounds__(64) void tree_build_kernel(const int32_t* __restrict__ paths, int n_paths, int width, int topk,
                                                      int expand_all, TreeBuildOut o, int* __restrict__ err_flag) {
  __shared__ TreeTopo t;
  __shared__ int rc;
  const int tid = threadIdx.x;
  if (tid == 0) {
    rc = tb_analyse(paths, n_paths, width, topk, expand_all, t);
    if (rc && err_flag) *err_flag = 1;
  }
  __syncthreads();
  if (rc) return;
  const int n = t.n;
  // verify side, one node per thread
  if (tid < n) {
    const int v = tid;
    int ch[TB_MAX_W];
    const int l = tb_chain(t, v, ch);
    o.tree_indices[v] = v ? 1 + (int64_t)topk * t.flat_row[t.parent[v]] + t.choice[v] : 0;
    o.tree_position_ids[v] = t.depth[v];
    o.parent[v] = t.parent[v];
    o.flat_row[v] = t.flat_row[v];
    for (int c = 0; c < n; ++c) {
      bool on = c == 0 || c == v;
      for (int q = 1; !on && q < l; ++q) on = ch[q] == c;
      o.tree_attn_mask[v * n + c] = on ? 1 : 0;
    }
  }
  if (tid < t.P) {
    int ch[TB_MAX_W];
    const int l = tb_chain(t, t.leaf[tid], ch);
    for (int q = 0; q < t.W; ++q) o.retrieve_indices[tid * t.W + q] = q < l ? ch[q] : -1;
  }
  // draft side, one forwarded node per thread
  if (tid < t.n_fwd) {
    const int v = t.fwd[tid];
    const int d = t.depth[v] - 1, first = t.step_first[d], r = tid - first, cols = t.step_cols[d];
    o.step_tree_indices[tid] = t.choice[v] + (int64_t)topk * t.prow[tid];
    o.step_parent_row[tid] = t.prow[tid];
    int ch[TB_MAX_W];
    const int l = tb_chain(t, v, ch);
    uint8_t* m = o.step_attn_mask + t.step_mask_off[d] + r * cols;
    for (int c = 0; c < cols; ++c) {                          // column c = forwarded node fwd[c] (KV order)
      bool on = false;
      for (int q = 1; !on && q < l; ++q) on = ch[q] == t.fwd[c];
      m[c] = on ? 1 : 0;
    }
    // repeat_nums of the step: the r-th run length of equal parent rows (0 beyond the last run)
    int ri = -1, cnt = 0;
    for (int j = first; j < first + t.step_n[d]; ++j) {
      if (j == first || t.prow[j] != t.prow[j - 1]) ++ri;
      if (ri == r) ++cnt;
    }
    o.step_repeat_nums[tid] = cnt;
  }
}

// Host-side validation and sizes (paths in HOST memory).  meta[32] (int32): [0] n_nodes (incl. root), [1] max_depth,
// [2] n_steps, [3] n_fwd (forwarded nodes), [4] n_flat_rows = 1 + n_fwd, [5] P (retrieve rows), [6] W = max_depth + 1,
// [7] step_attn_mask bytes, [8 + d] nodes of draft step d, [8 + 15 + ...] unused.
extern "C" int eagle_tree_layout(const int32_t* paths_host, int n_paths, int width, int topk, int expand_all, int32_t* meta) {
  if (!paths_host || !meta) return EAGLE_ERR_ARG;
  static thread_local TreeTopo t;
  const int rc = tb_analyse(paths_host, n_paths, width, topk, expand_all ? 1 : 0, t);
  if (rc) return rc;
  for (int i = 0; i < 32; ++i) meta[i] = 0;
  meta[0] = t.n; meta[1] = t.max_depth; meta[2] = t.n_steps; meta[3] = t.n_fwd; meta[4] = t.n_flat_rows;
  meta[5] = t.P; meta[6] = t.W; meta[7] = t.mask_bytes;
  for (int d = 0; d < t.n_steps; ++d) meta[8 + d] = t.step_n[d];
  return EAGLE_OK;
}

extern "C" int eagle_tree_build(const int32_t* paths, int n_paths, int width, int topk, int expand_all,
                                int64_t* tree_indices, int64_t* tree_position_ids, uint8_t* tree_attn_mask,
                                int64_t* retrieve_indices, int32_t* parent, int32_t* flat_row,
                                int64_t* step_tree_indices, int64_t* step_parent_row, int32_t* step_repeat_nums,
                                uint8_t* step_attn_mask, int* err_flag, hipStream_t stream) {
  if (!paths || !tree_indices || !tree_position_ids || !tree_attn_mask || !retrieve_indices || !parent || !flat_row ||
      !step_tree_indices || !step_parent_row || !step_repeat_nums || !step_attn_mask) return EAGLE_ERR_ARG;
  if (n_paths < 1 || n_paths + 1 > TB_MAX_NODES || width < 1 || width >= TB_MAX_W || topk < 1) return EAGLE_ERR_SHAPE;
  TreeBuildOut o = {tree_indices, tree_position_ids, tree_attn_mask, retrieve_indices, parent, flat_row,
                    step_tree_indices, step_parent_row, step_repeat_nums, step_attn_mask};
  hipLaunchKernelGGL(tree_build_kernel, dim3(1), dim3(64), 0, stream, paths, n_paths, width, topk, expand_all ? 1 : 0, o, err_flag);
  return eagle_check_launch();
}
