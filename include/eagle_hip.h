/* eagle_hip.h -- C ABI of libeagle_hip.so: MI355X (gfx950) kernels for the EAGLE-1 draft/verify
 * hot path and the draft-head training loss.
 *
 * The reference (vladislavkruglikov/eagle) has no native layer: its hot path is PyTorch eager
 * code in eagle/llama2.py and eagle/train.py.  Each entry point below replaces the eager ops at
 * the cited reference lines; the Python host (eagle_amd/) binds them with ctypes and passes raw
 * device pointers, sizes and the current HIP stream.  No torch types cross this boundary.
 *
 * Conventions: every function returns 0 on success, EAGLE_ERR_SHAPE (1) when the operand shapes
 * are outside what the kernel and its grid assume, EAGLE_ERR_LAUNCH (2) when the HIP launch
 * failed, EAGLE_ERR_ARG (3) on null pointers / bad enums.  No exceptions cross the boundary.
 * All "bf16" buffers are raw bfloat16 bits; all pointers are device pointers; launches are
 * asynchronous on `stream`.  Nothing here allocates or synchronises.
 */
#ifndef EAGLE_HIP_H
#define EAGLE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* eagle_stream_t; /* == hipStream_t */

#define EAGLE_EPI_LINEAR 0
#define EAGLE_EPI_SILU_MUL 1
/* OR-ed into `epi` of eagle_gemm_skinny_bf16: read W (and W2) with the default cache policy so that the matrix can stay
 * in the 256 MB Infinity Cache from one draft forward to the next; without it weights stream with non-temporal loads
 * and displace nothing.  A per-call hint (the library keeps no state); it only changes speed, never results. */
#define EAGLE_GEMM_KEEP_RESIDENT 0x100
/* OR-ed into `epi` of eagle_gemm_skinny_bf16 and eagle_gemm_rows_bf16 (which then ignores ldw): W (and W2) point at eagle_pack_weight_bf16 images instead of row-major
 * matrices.  Same arithmetic, same output bits; the weight stream then reads whole cache lines (see eagle_pack_weight_bf16). */
#define EAGLE_GEMM_W_PACKED 0x200

/* out[M,N] = epilogue(x[M,K] . W[N,K]^T), weight-streaming kernel for M <= 64 rows per weight
 * pass (larger M loops).  x row m = cat(x0[idx0 ? idx0[m] : m][0:K0], x1[idx1 ? idx1[m] : m][0:K-K0])
 * (x1 may be NULL: then K0 is ignored).  EPI_LINEAR: (+bias[N]) then (+res[M,N]), each rounded to
 * bf16 like the eager ops.  EPI_SILU_MUL: out = silu(x.W^T) * (x.W2^T).
 * Replaces nn.Linear at draft time: fc(cat(embed_tokens(ids), hidden)) eagle/llama2.py:515,:542;
 * q/k/v/o_proj :222-224,:280; LlamaMLP :317; residual adds :385,:391; and the frozen lm_head
 * eagle/train.py:215-228 applied to draft rows. Requires K % 32 == 0, K0 % 32 == 0.
 * ind0 != NULL: double indirection for the x0 rows, row m = idx0[(m/ind_n)*idx0_bstride + ind0[m%ind_n]]
 * (the tree step's input_ids = flat_topk[tree_indices], SURVEY.md a20, without a gather pass).
 * norm_eps > 0: every output row m is scaled by rsqrt(mean_k(x[m,k]^2) + norm_eps), computed from
 * the x fragments the kernel streams anyway: with the RMSNorm weight folded into W's columns this
 * is post_attention_layernorm fused into the gate/up projection (eagle/llama2.py:330-335, :389-390). */
int eagle_gemm_skinny_bf16(const void* x0, const int64_t* idx0, int64_t ld0, int K0,
                           const int64_t* ind0, int ind_n, int64_t idx0_bstride, float norm_eps,
                           const void* x1, const int64_t* idx1, int64_t ld1,
                           const void* W, const void* W2, const void* bias,
                           const void* res, int64_t ldr, void* out, int64_t ldo,
                           int M, int N, int K, int epi, eagle_stream_t stream);

/* Packed image of a weight matrix W [N, K] (row stride ldw, K % 32 == 0) for the weight-streaming GEMM above: every 16-row x
 * 32-column block -- one MFMA A fragment, which the kernel otherwise gathers as 16 x 64-byte half cache lines per wave
 * instruction -- is stored as ONE contiguous 1 KiB piece in lane order, [ceil(N/16)][K/32][64][8] bf16, rows past N zero.
 * eagle_packed_weight_bytes(N, K) = its size.  Measured on MI355X (tools/stream_probe.hip: 262 MB, cold caches, non-temporal
 * loads): half-line wave reads stream at 5.4 TB/s, whole-line reads at 6.3 TB/s.  A layout of the SAME nn.Linear weights
 * (eagle/llama2.py:222-224, :280, :317, :542; lm_head eagle/train.py:215-228) chosen for HBM; built once per weight version
 * by the host, never visible in the state_dict. */
int64_t eagle_packed_weight_bytes(int N, int K);
int eagle_pack_weight_bf16(const void* W, int64_t ldw, int N, int K, void* packed, eagle_stream_t stream);

/* The same product for decode BATCHES, 1 <= M <= 192 rows (a batch of requests stepping through the draft tree together:
 * 32 requests x 1 or 4 tree rows): one weight pass for all rows.  Operands and epilogues as eagle_gemm_skinny_bf16
 * (no norm_eps: run eagle_rmsnorm_fwd_bf16 first); W [N, K] with row stride ldw (W2 too), K % 128 == 0, K0 % 128 == 0.
 * Few column tiles are cut along K; the fp32 partial sums go through `workspace`
 * (eagle_gemm_rows_workspace_bytes(M, N, K, epi) bytes, 16-byte aligned, may be NULL when that is 0) and a second launch
 * sums them in slice order (deterministic).  The first 16 KiB of the workspace are arrival counters for the in-launch
 * form of that sum (diagnostics): they must be ZERO before the first call and every call leaves them zero; calls that
 * share a workspace must not overlap (one stream, or one workspace per stream).  Replaces the same nn.Linear calls (eagle/llama2.py:222-224, :280, :317, :542; lm_head eagle/train.py:215-228)
 * at batch sizes where eagle_gemm_skinny_bf16 would stream the weights once per 32 or 64 rows. */
int64_t eagle_gemm_rows_workspace_bytes(int M, int N, int K, int epi);
int eagle_gemm_rows_bf16(const void* x0, const int64_t* idx0, int64_t ld0, int K0,
                         const int64_t* ind0, int ind_n, int64_t idx0_bstride,
                         const void* x1, const int64_t* idx1, int64_t ld1,
                         const void* W, const void* W2, int64_t ldw, const void* bias,
                         const void* res, int64_t ldr, void* out, int64_t ldo,
                         int M, int N, int K, int epi, void* workspace, int64_t workspace_bytes, eagle_stream_t stream);

/* C[M,N] = epilogue(A[M,K] . B[N,K]^T): LDS-tiled MFMA GEMM (128x128x64 tiles, LDS-DMA staging)
 * for training-size M.  (+bias[N]) (+res[M,N] bf16); C is bf16 (out_f32 = 0) or fp32 (1);
 * accumulate != 0: C += result.  K % 64 == 0, lda/ldb % 8 == 0, 16-byte aligned bases.
 * Replaces the nn.Linear forward GEMMs of eagle/llama2.py:222-224,:280,:317,:542 and of the frozen
 * lm_head (eagle/train.py:107,:110) at training sizes.  Their autograd backward GEMMs behind loss.backward()
 * (eagle/train.py:126) run on eagle_gemm_nn_bf16 / eagle_gemm_tn_bf16 below (operands as they lie in memory); this entry
 * point with eagle_transpose_bf16 on the operands serves the shapes those two do not take. */
int eagle_gemm_nt_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                       const void* bias, const void* res, int64_t ldr, int M, int N, int K,
                       int out_f32, int accumulate, eagle_stream_t stream);
/* Split-K form of the above for few output tiles and a very long K -- the lm_head dgrad behind
 * loss.backward(), eagle/train.py:110,:126 (M = tokens, N = D, K = vocabulary): ksplit K-slices run as
 * separate workgroups into fp32 partials [ksplit][M][N] in `workspace`, a second launch sums them and
 * applies the same epilogue.  ksplit <= 1: identical to eagle_gemm_nt_bf16. */
int64_t eagle_gemm_nt_splitk_workspace_bytes(int M, int N, int ksplit);
int eagle_gemm_nt_splitk_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                              const void* bias, const void* res, int64_t ldr, int M, int N, int K, int out_f32,
                              int accumulate, int ksplit, void* workspace, eagle_stream_t stream);
/* C[M,N] (+)= A^T . B with K-MAJOR operands A [K, M] (row stride lda) and B [K, N] (row stride ldb), on the 256x256 kernel
 * with transposing LDS fragment reads: the weight-gradient GEMMs behind loss.backward() (eagle/train.py:126),
 * dW[out, in] = dY[tokens, out]^T . X[tokens, in], read dY and X as they lie in memory -- no eagle_transpose_bf16 pass.
 * M % 256 == 0, N % 256 == 0, K % 64 == 0, K >= 128, K * max(lda, ldb) * 2 < 4 GiB; other shapes: EAGLE_ERR_SHAPE (the host
 * then transposes and calls eagle_gemm_nt_bf16).  C bf16 (out_f32 = 0) or fp32; accumulate != 0: C += result. */
int eagle_gemm_tn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int M, int N, int K,
                       int out_f32, int accumulate, eagle_stream_t stream);
/* C[M,N] = A . B (+res[M,N]) with B stored K-major: A [M,K] row-major (row stride lda), B [K,N] row-major (row stride ldb).
 * The data gradients behind loss.backward() (eagle/train.py:126): dX = dY . W reads the weight [out, in] as the state_dict
 * stores it -- no transposed weight copy after each optimizer step.  N % 256 == 0, K % 64 == 0, K >= 128; lda, ldb % 8 == 0;
 * 16-byte aligned bases; K * ldb * 2 < 4 GiB.  C bf16 (out_f32 = 0) or fp32; accumulate: C += . */
int eagle_gemm_nn_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* res,
                       int64_t ldr, int M, int N, int K, int out_f32, int accumulate, eagle_stream_t stream);
/* out[C,R_pad] = in[R,C]^T, bf16, columns [R, R_pad) zero-filled (R_pad < R: no padding): builds the transposed copy of
 * the frozen lm_head once, and prepares operands for the dgrad / wgrad GEMMs behind loss.backward() (eagle/train.py:126)
 * at shapes eagle_gemm_nn_bf16 / eagle_gemm_tn_bf16 do not take (no training step at the shipped dims needs it); the zero
 * padding lets a token count that is no multiple of 64 serve as the GEMM's K.  ldo >= R_pad. */
int eagle_transpose_bf16(const void* in, int64_t ldi, void* out, int64_t ldo, int R, int C, int R_pad,
                         eagle_stream_t stream);

/* y = w * bf16(x * rsqrt(mean(x^2) + eps)); optional rstd_out[N] (fp32) for the backward.
 * Replaces LlamaRMSNorm.forward, eagle/llama2.py:330-335. D % 8 == 0, D <= 8192. */
int eagle_rmsnorm_fwd_bf16(const void* x, const void* w, void* y, float* rstd_out, int N, int D,
                           float eps, eagle_stream_t stream);
/* Backward of the above (autograd of eagle/llama2.py:330-335): dx[N,D] bf16 (+ res[N,D], the
 * gradient arriving over the residual branch of eagle/llama2.py:388-391, may be NULL), dw[D] fp32
 * (accumulate_dw != 0: dw += ...).  workspace: eagle_rmsnorm_bwd_workspace_bytes(N, D). */
int64_t eagle_rmsnorm_bwd_workspace_bytes(int N, int D);
int eagle_rmsnorm_bwd_bf16(const void* dy, const void* x, const void* w, const float* rstd, const void* res, void* dx,
                           float* dw, int accumulate_dw, void* workspace, int N, int D,
                           eagle_stream_t stream);

/* Tree-masked decode attention fused with RoPE and the KV-cache append, M <= 16 query rows per
 * request, split over 64-key chunks (one workgroup per chunk x head x request) + a combine pass.
 * qkv [B*rows_per_req, (H+2*HK)*hd] (fused q|k|v projection output; request b's query m is row
 * b*rows_per_req + row_off + m, so longer inputs are processed in 16-row chunks).
 * position_ids [B*rows_per_req] int64, or NULL: position = past + pos_off + m*pos_stride
 * (pos_off 0 / pos_stride 1 = the reference's default arange, eagle/llama2.py:520-527;
 * pos_stride 0 = every row at the same position, as tree steps need).
 * cos/sin tables [n_pos, hd] bf16 (the reference's cached tables cast to bf16),
 * kcache/vcache [B, HK, cap, hd] (element stride between requests = cache_batch_stride),
 * past_len [B] (device int32) + past_off = rows already in the cache; the M new roped K rows and
 * V rows are appended at [past, past+M).  tree_mask [M,T] bytes (0 = masked) overlays the last T
 * key columns (tree_mask_batch_stride 0 = shared by all requests); pad_mask [B, >= past+M] bytes
 * (0 = padded key) or NULL; causal != 0 adds the causal mask when M > 1.  out [B*rows_per_req, H*hd].
 * max_kv >= max_b(past+M) sizes the grid and the workspace
 * (eagle_attn_decode_workspace_bytes(B, H, max_kv) bytes of fp32 partials).
 * Device-side violations (past+M > cap or > max_kv) set *err_flag = 1 and skip the request.
 * Replaces LlamaAttention.forward after the projections, eagle/llama2.py:226-273, with the mask
 * semantics of _prepare_decoder_attention_mask :460-491 (tree overlay :482-488). */
int64_t eagle_attn_decode_workspace_bytes(int B, int H, int max_kv);
int eagle_tree_attention_bf16(const void* qkv, int64_t ld_qkv, const int64_t* position_ids, int pos_off,
                              int pos_stride, const void* cos_table, const void* sin_table, int n_pos,
                              void* kcache, void* vcache, int64_t cache_batch_stride, int cap,
                              const int32_t* past_len, int past_off,
                              const uint8_t* tree_mask, int T, int64_t tree_mask_batch_stride,
                              const uint8_t* pad_mask, int64_t pad_mask_stride,
                              void* out, int64_t ld_out, void* workspace, int* err_flag,
                              int B, int M, int H, int HK, int hd, int max_kv, int causal,
                              int rows_per_req, int row_off, eagle_stream_t stream);

/* Training attention (causal, no cache, no padding mask -- eagle/train.py:105 passes no
 * attention_mask), flash-style on MFMA; hd = 128.  qkv [B*L, 3*H*hd] = q|k|v sections with RoPE
 * already applied (eagle_rope_qk_bf16).  fwd: out [B*L, H*hd], lse [B,H,L] (log2 units).
 * bwd: dqkv [B*L, 3*H*hd] (gradients w.r.t. the roped q,k and v), delta_ws [B,H,L] fp32 scratch.
 * Replaces eagle/llama2.py:247-273 with the causal mask of :14-28 and its autograd backward. */
int eagle_attn_train_fwd_bf16(const void* qkv, int64_t ld, void* out, int64_t ldo, float* lse, int B,
                              int L, int H, int hd, eagle_stream_t stream);
int eagle_attn_train_bwd_bf16(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout,
                              int64_t lddo, const float* lse, float* delta_ws, void* dqkv, int64_t lddq,
                              int B, int L, int H, int hd, eagle_stream_t stream);
/* In-place RoPE over the first n_heads_qk heads of every row (the q and k sections):
 * x' = bf16(bf16(x*cos) + bf16(rotate_half(x)*sin)) at position_ids[row] (NULL: row % L);
 * inverse != 0 applies the transposed rotation (the backward).  eagle/llama2.py:56-69. */
int eagle_rope_qk_bf16(void* qkv, int64_t ld, const int64_t* position_ids, const void* cos_table,
                       const void* sin_table, int n_pos, int64_t nrows, int L, int n_heads_qk, int hd,
                       int inverse, eagle_stream_t stream);

/* Per-row top-k over the vocabulary; order = value descending then index ascending (a stable
 * descending sort; -0.0 == +0.0; NaN largest).  dtype 0 = bf16 logits, 1 = fp32.  out_idx
 * [rows,k] int64, out_val [rows,k] (logits dtype, may be NULL), k <= 16.
 * workspace: eagle_topk_workspace_bytes(rows, V, k).
 * Replaces torch.topk(last_headout, top_k) of the expansion loop (top_k = 10, eagle/llama2.py:11;
 * SURVEY.md a20) and, with k = 1, torch.max(..., 2) index selection (eagle/train.py:115-116).
 * block_index (device int64, may be NULL) / block_rows: the rows read are
 * [*block_index * block_rows, + rows) of `logits` -- a ring of verifier output buffers whose cursor
 * lives in device memory, so that a captured graph needs no host-side pointer update. */
int64_t eagle_topk_workspace_bytes(int rows, int V, int k);
int eagle_topk_vocab(const void* logits, int64_t ld, int dtype, int rows, int V, int k,
                     int64_t* out_idx, void* out_val, void* workspace, const int64_t* block_index,
                     int64_t block_rows, eagle_stream_t stream);
/* The same top-k written straight into a batched table: logits row r = b*group + j (request b, node j of the step) goes to
 * out_idx / out_val + b*group_stride + j*k (elements) -- the flat top-k table [B, R, k] of the expansion loop (SURVEY.md
 * a20) without a copy pass per step.  rows % group == 0, group_stride >= group*k. */
int eagle_topk_vocab_grouped(const void* logits, int64_t ld, int dtype, int rows, int V, int k, int64_t* out_idx,
                             void* out_val, void* workspace, int group, int64_t group_stride, eagle_stream_t stream);
/* The final merge of the top-k alone: rows x ncand candidate records (bf16 monotone key << 48 | 0xffffffff - index; 0 = empty
 * slot) -> every row's k best ids and bf16 values (rebuilt from the keys: -0 comes back as +0), optionally into a batched table
 * as eagle_topk_vocab_grouped (group = 0: plain [rows, k]).  ncand <= 6144.  No reference line of its own: it is the second
 * half of `torch.topk(logits, k)` for producers that never materialise the logits. */
int eagle_topk_merge_candidates(const void* cand, int ncand, int rows, int k, int64_t* out_idx, void* out_val,
                                int group, int64_t group_stride, eagle_stream_t stream);
/* lm_head + top-k of a tree step without the logits, for 81..128 rows of a request batch over a tall head (N >= 32768) and for
 * 1..16 rows (every tree step of ONE request; K = 4096 and a packed image only: the register-resident weight stream of
 * eagle_gemm_skinny_bf16 with the same candidate epilogue, one record set per (row, workgroup)) -- replaces
 * `last_headout = lm_head(h)` followed by `torch.topk(last_headout, k)` of the expansion loop (SURVEY.md a11 + a20) for a
 * request batch: the weight-stream GEMM (the 8-wave form of eagle_gemm_rows_bf16: x rows gathered through idx0 / ind0 as there,
 * no second operand) rounds its outputs to bf16 and leaves k candidate records per (row, column tile) in `workspace`
 * (eagle_lm_head_topk_rows_workspace_bytes(M, N, K, k) bytes, 8-byte aligned; 0 = shape not served: the call then returns
 * EAGLE_ERR_SHAPE and the caller takes eagle_gemm_rows_bf16 + eagle_topk_vocab_grouped), eagle_topk_merge_candidates
 * finishes.  Ids identical to that two-step path (same bf16 rounding, same order: value, then lowest index).
 * ldw == 0: W is an eagle_pack_weight_bf16 image. */
int64_t eagle_lm_head_topk_rows_workspace_bytes(int M, int N, int K, int k);
int eagle_lm_head_topk_rows_bf16(const void* x0, const int64_t* idx0, int64_t ld0, const int64_t* ind0, int ind_n,
                                 int64_t idx0_bstride, const void* W, int64_t ldw, const void* bias,
                                 int M, int N, int K, int k, int64_t* out_idx, void* out_val, int group,
                                 int64_t group_stride, void* workspace, int64_t workspace_bytes, eagle_stream_t stream);
/* bf16 logits: the same top-k plus lse_out[rows] = fp32 log-sum-exp of every row out of the same pass over the logits
 * (stage 1 leaves a (max, sum-exp) pair per 1024-logit chunk, stage 2 merges them beside its final select) -- what
 * the dynamic tree's cumulative scores need (candidate probability = exp(logit - lse)) without a
 * separate eagle_row_lse_bf16 launch.  Same workspace query. */
int eagle_topk_lse_vocab(const void* logits, int64_t ld, int rows, int V, int k, int64_t* out_idx, void* out_val,
                         float* lse_out, void* workspace, eagle_stream_t stream);

/* NO COUNTERPART IN THE REFERENCE.  eagle_topk_vocab for callers that cannot allocate or hand in a workspace (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): same contract, no workspace argument -- the
 * stage-1 candidates live in a 2 MiB device array inside the library (one copy per GPU, held until the process ends).
 * Limit: eagle_topk_workspace_bytes(rows, V, k) <= 2 MiB (159 rows of a 151936-entry vocabulary at k = 10), beyond it
 * EAGLE_ERR_SHAPE.  Calls no hipMalloc / hipFree / allocating routine.  All calls on one GPU share that GPU's array, so
 * calls that overlap on one GPU race; nothing in it has to start at zero (every slot read was written by the same call).
 * `stream` must belong to the current device; the first call on a device resolves the array's address and is not
 * capturable -- make one call per device before capturing a graph. */
int eagle_topk_vocab_noalloc(const void* logits, int64_t ld, int dtype, int rows, int V, int k, int64_t* out_idx,
                             void* out_val, const int64_t* block_index, int64_t block_rows, eagle_stream_t stream);

/* Static-tree kernels (Llama2Model.init_tree / tree_mask hooks,
 * eagle/llama2.py:453-458 -- the algorithm itself is the upstream EAGLE-1 one). */
/* generate_tree_buffers, the name Llama2Model.init_tree calls without defining it (eagle/llama2.py:453-455; the tree
 * itself, mc_sim_7b_63, is the 25-path list of SURVEY.md a18).  paths [n_paths, width] int32, -1 padded, any order;
 * nodes = paths sorted by (length, path), node 0 = root; n_paths + 1 <= 64, width <= 15, every choice < topk.
 * expand_all = 0: only nodes that have children are run through the draft head (upstream), 1: every node.
 * eagle_tree_layout (paths in HOST memory; launches nothing) validates the tree -- duplicate path, missing parent,
 * choice >= topk, empty depth: EAGLE_ERR_SHAPE -- and returns the sizes in meta[32]: [0] n_nodes, [1] max_depth,
 * [2] n_steps, [3] n_fwd, [4] n_flat_rows = 1 + n_fwd, [5] P, [6] W = max_depth + 1, [7] bytes of step_attn_mask,
 * [8 + d] nodes of draft step d.
 * eagle_tree_build (paths in DEVICE memory) fills, on the device:
 *   verify side (a21): tree_indices [n] (into cat(root_token, flat_topk): 1 + topk*flat_row[parent] + choice),
 *     tree_position_ids [n] (depth), tree_attn_mask [n,n] bytes (self | root | ancestors), retrieve_indices [P,W]
 *     (root->leaf node ids, -1 padded, rows sorted), parent [n], flat_row [n] (-1: node not forwarded);
 *   draft side (a19), the steps packed back to back in step order: step_tree_indices [n_fwd] (choice + topk*rank of the
 *     parent in the previous step), step_parent_row [n_fwd], step_repeat_nums [n_fwd] (per step: run lengths of equal
 *     parents, zero padded to the step's node count), step_attn_mask (step d: [n_d, sum_{e<=d} n_e] bytes, 1 on self
 *     and forwarded ancestors).  All nodes of draft step d sit at position stable_len + d (position offsets are zero).
 * A tree the device analysis rejects sets *err_flag = 1 (cannot happen after a successful eagle_tree_layout). */
int eagle_tree_layout(const int32_t* paths_host, int n_paths, int width, int topk, int expand_all, int32_t* meta);
int eagle_tree_build(const int32_t* paths, int n_paths, int width, int topk, int expand_all,
                     int64_t* tree_indices, int64_t* tree_position_ids, uint8_t* tree_attn_mask,
                     int64_t* retrieve_indices, int32_t* parent, int32_t* flat_row,
                     int64_t* step_tree_indices, int64_t* step_parent_row, int32_t* step_repeat_nums,
                     uint8_t* step_attn_mask, int* err_flag, eagle_stream_t stream);
/* out_ids[b][j] = flat_ids[b][tree_idx[j]]: draft tokens of one expansion step. */
int eagle_tree_gather_ids(const int64_t* flat_ids, int64_t flat_batch_stride, const int64_t* tree_idx,
                          int n, int64_t* out_ids, int B, eagle_stream_t stream);
/* Greedy acceptance along the tree's root->leaf paths.  result[b] = {best_path, accept_length,
 * next_token, prev_len[b] + accept_length + 1}; optional tree_tok_out [B,n_nodes],
 * cart_out [B,P,W], accept_out [B,P]; optional next_ids_out [B,W] = the token following each
 * accepted hidden row (accepted draft tokens, then next_token; zero padded) and n_acc_out [B] =
 * accept_length + 1: the inputs of the next draft cycle; optional draft-side state for that cycle:
 * last_idx_out [B] = b*W + accept_length (last valid catch-up row), hid_idx0_out [B,n0] = the same
 * row repeated for the n0 depth-1 nodes, and the in/out length pair len_cur/len_nxt [B] int32
 * (len_cur = len_nxt; len_nxt += accept_length + 1); optional cursor [2] (device int64): ring cursor
 * of the verifier's output buffers, cursor[1] = cursor[0]; cursor[0] += 1 (the argmax before this
 * kernel reads cursor[0], the gather after it reads cursor[1]). */
int eagle_greedy_accept(const int64_t* root_token, const int64_t* flat_ids, int64_t flat_batch_stride,
                        const int64_t* tree_indices, int n_nodes, const int64_t* retrieve, int P, int W,
                        const int64_t* target_argmax, const int32_t* prev_len, int64_t* tree_tok_out,
                        int64_t* cart_out, int32_t* accept_out, int64_t* result, int64_t* next_ids_out,
                        int32_t* n_acc_out, int64_t* last_idx_out, int64_t* hid_idx0_out, int n0,
                        int32_t* len_cur, int32_t* len_nxt, int64_t* cursor, int B, eagle_stream_t stream);
/* In-place compaction of the accepted path's KV rows: cache viewed as [n_slabs][cap][row_elems]
 * with slab = (outer*B + b)*inner + i; rows prev_len[b] + retrieve[best][j] -> prev_len[b] + j. */
int eagle_kv_compact_bf16(void* cache, int64_t n_slabs, int64_t slab_stride, int row_elems, int B,
                          int inner, const int64_t* retrieve, int W, const int64_t* result,
                          const int32_t* prev_len, int cap, int* err_flag, eagle_stream_t stream);
/* out[b][j] = src[b][retrieve[best_b][j]] for j <= accept_length_b (zeros after).  Optional
 * root_io [B] / prev_len_io [B]: overwritten with next_token / new_len for the next cycle (this is
 * the last kernel of the verify chain, so earlier kernels still saw the old values).  Optional
 * src_block (device int64): src is a ring of [B, n_nodes, D] buffers and block *src_block is read. */
int eagle_gather_accepted_bf16(const void* src, int n_nodes, int D, const int64_t* retrieve, int W,
                               const int64_t* result, void* out, int64_t* root_io, int32_t* prev_len_io,
                               const int64_t* src_block, int B, eagle_stream_t stream);

/* Dynamic draft tree (EAGLE-2 as SGLang serves it; benchmark/benchmark.py:48-56 passes
 * speculative_num_steps / speculative_eagle_topk / speculative_num_draft_tokens to sglang.Engine --
 * the algorithm itself is not in the reference; spec frozen in DESIGN.md 6b, oracle/dyn_tree.py).
 * B independent requests per launch: every per-request array below is [B][...] contiguous (the shapes given are one
 * request's).  k = candidates per node (<= 16), S = steps, T = verified nodes incl. the root (<= 64),
 * pool = k + (S-1) k^2 candidates (<= 1024) in production order; score = product of softmax
 * probabilities along the path; order = score descending, pool index ascending. */
/* lse[r] = log sum_v exp(logits[r][v]) (fp32): the softmax normaliser of the draft head's top-k. */
int eagle_row_lse_bf16(const void* logits, int64_t ld, int rows, int V, float* lse, eagle_stream_t stream);
/* One expansion step: vals/ids [R,k] = top-k of the R = (step ? k : 1) rows just computed, lse [R].
 * Appends the R*k candidates to the pool (token, score, parent pool index) and, unless `last`, makes
 * the k best the next frontier: ftok [k], fscore_out [k], fpool_out [k], parent_row [k] (row of the
 * previous BATCHED forward whose hidden state each takes: b*k + j; step 0: root_row[b]) and mask_out [k, (step+1) k] =
 * parent's row of mask_in [k, step k] | own column (select_top_k_tokens of SGLang's EAGLE worker). */
int eagle_dyn_step(int step, int k, int last, const void* vals, const int64_t* ids, const float* lse,
                   const float* fscore_in, const int32_t* fpool_in, const uint8_t* mask_in,
                   const int64_t* root_row, int64_t* pool_tok, float* pool_score, int32_t* pool_parent,
                   int64_t* ftok, float* fscore_out, int32_t* fpool_out, int64_t* parent_row,
                   uint8_t* mask_out, int S, int B, eagle_stream_t stream);
/* The T-1 best pool entries, in pool order, become tree nodes 1..T-1 (organize_draft_results):
 * tree_tok [T] (node 0 = *root_token), tree_parent [T] int32 (root -1), tree_pos [T] (depth),
 * tree_mask [T,T] uint8 (self, root, ancestors), sel [T-1] (pool index of node n+1). */
int eagle_dyn_finalize(int k, int S, int T, const int64_t* pool_tok, const float* pool_score,
                       const int32_t* pool_parent, const int64_t* root_token, int64_t* tree_tok,
                       int32_t* tree_parent, int64_t* tree_pos, uint8_t* tree_mask, int32_t* sel,
                       int B, eagle_stream_t stream);
/* Greedy acceptance by walking the tree along target_argmax [T]; path [W] (node ids, -1 padded) doubles
 * as the one-row `retrieve` table of eagle_kv_compact_bf16 / eagle_gather_accepted_bf16 with
 * result = {0, accept_length, next_token, prev_len[b] + accept_length + 1}; next_ids / n_acc / last_idx (= b*W +
 * accept_length) / len_cur / len_nxt / cursor (shared by the batch) as eagle_greedy_accept. */
int eagle_dyn_accept(int T, int W, const int64_t* tree_tok, const int32_t* tree_parent,
                     const int64_t* target_argmax, const int32_t* prev_len, int64_t* path, int64_t* result,
                     int64_t* next_ids, int32_t* n_acc, int64_t* last_idx, int32_t* len_cur, int32_t* len_nxt,
                     int64_t* cursor, int B, eagle_stream_t stream);

/* Fused soft-label CE + smooth-L1 + agreement at the logits level.
 * Replaces eagle/train.py:108-123 (== eagle/train_tp.py:90-105).
 * ce_partial: one vocab shard [N, V] (columns v0..v0+V of the full vocabulary) -> stats [N]
 *   records of eagle_ce_stat_bytes() bytes; predict/target [N,D] optional (smooth-L1 row means).
 * ce_combine: merges R shards' stats [R,N] -> lse_t/lse_o [N], row_out [N,4] =
 *   {ploss_row, vloss_row, agree, 0}, sums[4] = {sum m*ploss, sum m*vloss, sum m*agree, sum m}.
 * bwd: dlogits = mask*g_p*(softmax(o)-softmax(t)) (may alias o_logits),
 *   dpredict = mask*g_v*clamp(predict-target,-1,1)/D. */
int eagle_ce_stat_bytes(void);
int eagle_ce_partial_bf16(const void* t_logits, const void* o_logits, int64_t ld, int N, int V, int v0,
                          const void* predict, const void* target, int D, void* stats,
                          eagle_stream_t stream);
int eagle_ce_combine(const void* stats, int R, int N, const float* mask, float* lse_t, float* lse_o,
                     float* row_out, float* sums, eagle_stream_t stream);
int eagle_fused_ce_smoothl1_fwd_bf16(const void* t_logits, const void* o_logits, int64_t ld, int N, int V,
                                     const void* predict, const void* target, int D, const float* mask,
                                     void* stats_ws, float* lse_t, float* lse_o, float* row_out,
                                     float* sums, eagle_stream_t stream);
int eagle_ce_smoothl1_bwd_bf16(const void* t_logits, const void* o_logits, int64_t ld, int N, int V,
                               const float* lse_t, const float* lse_o, const float* mask, float g_p,
                               float g_v, void* dlogits, int64_t ldd, const void* predict,
                               const void* target, int D, void* dpredict, eagle_stream_t stream);

/* ---- element-wise pieces of the training step ---- */
/* out[n] = cat(embed[ids[n]], hidden[n])  ([N, 2D]; eagle/llama2.py:515, :537-542) */
int eagle_embed_concat_bf16(const void* emb, const int64_t* ids, const void* hidden, int64_t ldh, void* out,
                            int N, int D, int V, eagle_stream_t stream);
/* act = silu(g) * u for gu = [g | u] ([N, 2I]); and its backward (eagle/llama2.py:317). */
int eagle_silu_mul_fwd_bf16(const void* gu, int64_t ld, void* act, int64_t lda, int N, int I,
                            eagle_stream_t stream);
int eagle_silu_mul_bwd_bf16(const void* gu, int64_t ld, const void* dact, int64_t lda, void* dgu, int64_t ldd,
                            int N, int I, eagle_stream_t stream);
/* out[c] (+)= sum_n x[n,c] (fc.bias gradient); workspace eagle_colsum_workspace_bytes(N, C). */
int64_t eagle_colsum_workspace_bytes(int N, int C);
int eagle_colsum_bf16(const void* x, int64_t ld, int N, int C, void* out_bf16, float* out_f32, int accumulate,
                      void* workspace, eagle_stream_t stream);
/* AdamW over a flat buffer (torch.optim.AdamW semantics, eagle/train.py:66): bf16 params and
 * grads, fp32 moments.  grad_scale (device float, may be NULL) multiplies the gradient first
 * (clip_grad_norm_ coefficient, eagle/train_tp.py:109); clip_value > 0 clamps each gradient
 * element (clip_grad_value_, eagle/train.py:127). */
int eagle_adamw_bf16(void* param, const void* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int step,
                     const float* grad_scale, float clip_value, eagle_stream_t stream);
/* coef_and_norm[0] = min(1, max_norm/(||g||+1e-6)), [1] = ||g||; workspace
 * eagle_grad_norm_workspace_bytes(). */
int64_t eagle_grad_norm_workspace_bytes(void);
int eagle_grad_clip_coef_bf16(const void* grad, int64_t n, float max_norm, float* coef_and_norm,
                              void* workspace, eagle_stream_t stream);
/* grad = clamp(grad, -clip, clip) in place: accelerator.clip_grad_value_ between the micro-batches
 * of one optimizer step (eagle/train.py:127).  16-byte aligned base. */
int eagle_clip_value_bf16(void* grad, int64_t n, float clip, eagle_stream_t stream);
/* x *= *scale in place (scale: device float): torch.nn.utils.clip_grad_norm_ applied to the
 * accumulated gradient after every micro-batch backward (eagle/train_tp.py:109) with the coefficient
 * of eagle_grad_clip_coef_bf16.  16-byte aligned base. */
int eagle_scale_bf16(void* x, int64_t n, const float* scale, eagle_stream_t stream);
/* out = bf16(a + b): a fp32 (the all-reduced partial dpredict of the vocab-sharded loss), b bf16 or
 * NULL (the replicated smooth-L1 part).  The sharded form of the lm_head backward behind
 * eagle/train_tp.py:105-108.  16-byte aligned bases. */
int eagle_add_f32_bf16(const float* a, const void* b, void* out, int64_t n, eagle_stream_t stream);
/* _make_eagle_input given the verifier's last hidden state (eagle/train.py:256-269, noise
 * :272-275 pre-drawn, may be NULL): hidden_out = hidden + noise; target[:, :-1] = hidden_out[:, 1:],
 * last row 0; ids_out = ids shifted left with trailing 0. */
int eagle_make_eagle_input_bf16(const void* hidden, const void* noise, const int64_t* ids, void* hidden_out,
                                void* target, int64_t* ids_out, int B, int L, int D, eagle_stream_t stream);

/* _make_eagle_input (eagle/train.py:256-275) fused with the draft head's embedding gather + concat (eagle/llama2.py:515,
 * :537-542): ONE pass over the verifier's last hidden state [B, L, D] (strides ldh_b / ldh_l elements, so a
 * [:, :max_model_len] slice needs no copy) and the unshifted ids [B, >= L] (row stride ldi).
 *   noise   = bf16(low + (high - low) * u), u ~ U[0,1) from Philox4x32-10 (counter = element index / 4, stream `offset`;
 *             key = `seed`): drawn on the fly, never stored (noise_out [B*L, D], may be NULL, returns it for tests);
 *             high == low: no noise.  The reference draws it in the hidden state's dtype and adds it in that dtype (:272-275).
 *   x0      [B*L, 2D] = cat(embed_tokens[ids shifted left, trailing 0], hidden + noise): the input of the fc GEMM
 *   target  [B*L, D]: target[:, :-1] = (hidden + noise)[:, 1:], last row of every sequence 0 (the target is noisy too)
 *   ids_out [B*L] (may be NULL) = ids shifted left with trailing 0 (:266-268).
 * embed [V, D] contiguous.  Ids outside [0, V) are clamped into it (nn.Embedding faults on them): validate upstream. */
int eagle_input_embed_concat_bf16(const void* hidden, int64_t ldh_b, int64_t ldh_l, const int64_t* ids, int64_t ldi,
                                  const void* emb, int V, float low, float high, uint64_t seed, uint64_t offset,
                                  void* x0, void* target, int64_t* ids_out, void* noise_out, int B, int L, int D,
                                  eagle_stream_t stream);

/* ---- diagnostics (no counterpart in the reference): kernel-family overrides for A/B measurements and for the tests
 * that run every selectable path; process-wide, not for production hosts. ---- */
/* eagle_gemm_nt_bf16: 0 = always the 128x128 kernel, 1 = the 256x256 kernel wherever legal, 2 = by shape (default);
 * 0x100 | cfg: configuration of the decode-batch row kernels (<= 192 rows, N >= 8192): cfg 0 = by shape, 0x23 / 0x24 /
 * 0x43 / 0x83 / 0x93 = (rows / 16) << 4 | LDS stages (0x93: 128 x 256 tiles on 8 waves), 0xfff = never;
 * 0x10000 | gm: row tiles per group of the 256x256 kernel's tile order (0 = default 8);
 * 0x20000 | s: main-loop schedule of the 256x256 NT kernel, s = 2 four slots per K tile (default), 0 / 1 the eight-slot forms of
 * rounds 1-2; s = 0x12 / 0x22 / 0x32 / 0x42 / 0x52 / 0x62 are TIMING ABLATIONS of schedule 2 whose outputs are WRONG (no DMA / no LDS
 * reads / neither / neither and no barriers / no output stores / balanced read slots): tools/gemm_lm_bench.py only;
 * 0x30000 | v: non-temporal bf16 output stores, v = 0 for outputs above 256 MB (default), 1 always, 2 never. */
void eagle_debug_set_gemm8(int mode);
/* eagle_gemm_skinny_bf16 at <= 16 rows: 0 = tile kernel, 1 = streaming kernel, 2 = register-resident-x kernel where
 * K = 4096 (streaming otherwise), 3 = by shape (default); blocks_per_cu / min_rows > 0 override the streaming
 * kernel's grid sizing. */
void eagle_debug_set_skinny_variant(int variant, int blocks_per_cu, int min_rows);
/* eagle_tree_attention_bf16, wide-batch kernel: 1 = one query head per workgroup (no packing of the query heads that
 * share a KV head into one workgroup's MFMA columns); 0 = default. */
void eagle_debug_set_attn(int no_head_packing);
/* eagle_topk_vocab*: 1 = always the two-launch form (candidates per 1024 logits, then the merge); 2 = the one-launch form
 * (a 16-wave workgroup per row, candidates in LDS) for every bf16 call of up to 65536 columns; 0 = default: one launch
 * for k <= 2 (the greedy argmax), two otherwise.  Bit 2 (value 4): eagle_topk_merge_candidates always on its four-wave kernel
 * (default: up to 16 rows of more than 1024 records take the 16-wave form). */
void eagle_debug_set_topk(int two_stage);
/* eagle_ce_partial_bf16 / eagle_ce_smoothl1_bwd_bf16: bit 0 = default-policy logit loads (default: non-temporal), bit 1 = one
 * 16-byte chunk pair per thread in flight (default: four), bit 2 = default-policy dlogits stores (default: non-temporal),
 * bit 4 = two chunk pairs in flight. */
void eagle_debug_set_loss(int variant);
/* eagle_gemm_rows_bf16: target_wg > 0 = workgroups the K split aims for (0 = by shape), min_steps = least 128-wide K
 * super-steps per slice (default 4); stages: bits 0-3 = 2 / 3 / 4 weight register stages (0 = default 3), bits 4-7 =
 * 1 + K-walk stagger of neighbouring column tiles (0 = default stagger 1), bit 8 = sum the partial sums inside the launch (last workgroup of a tile to arrive) instead of in a second launch,
 * bit 9 = keep the many-tile 81..128-row case (a 128-row lm_head) on 128-column tiles instead of the up-to-384-column form,
 * bit 10 = full-width column tiles only (no 3/4-width tiles chosen by the round count),
 * bit 11 = eagle_lm_head_topk_rows_bf16 stops after its candidate GEMM (no merge launch, tables not written: for timing that launch alone). */
void eagle_debug_set_rows(int target_wg, int min_steps, int stages);

#ifdef __cplusplus
}
#endif
#endif
