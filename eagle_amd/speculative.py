"""EAGLE-1 static-tree draft / verify cycle on one MI355X.

``EagleDrafter`` is the device-resident version of upstream EAGLE-1's ``topK_genrate`` +
``evaluate_posterior`` + ``update_inference_inputs`` that ``Llama2Model.init_tree`` /
``tree_mask`` (eagle/llama2.py:453-458, :482-488) were written for.  Everything between "the
verifier produced logits for the tree" and "the next tree is drafted" runs as HIP kernels with
no host synchronisation: lengths, accepted counts and token ids stay in device memory and all
shapes are static (the catch-up forward is padded to max_depth+1 rows; the padding rows it
appends to the KV cache lie beyond the stable length and are overwritten by the tree steps), so a
whole cycle can be captured into one HIP graph (``capture()``).

Per cycle, for each of B independent requests:
  catch-up   forward of the accepted tokens on top of the stable KV cache, lm_head on the last
             valid row, top-k(10)                                           -> flat row 0
  expansion  per tree depth d: tokens = flat_topk[tree_indices[d]], hidden = parent rows,
             tree-masked forward at position stable_len + d, lm_head, top-k -> flat rows 1..
  verify     first-argmax of the verifier's logits per tree node, greedy acceptance along the
             root->leaf paths, in-place compaction of the verifier's KV cache, gather of the
             accepted hidden states and of the tokens that follow them.  ``verify`` also advances
             the draft-side lengths, so exactly one ``draft`` must follow each ``verify``.
The verifier forward itself is out of scope: its logits, hidden states
and KV cache are inputs.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import ops
from .llama2 import Llama2Model
from .tree import TreeBuffers, mc_sim_7b_63

BF16 = torch.bfloat16


class _CacheRows:
    """One request's rows of the batched draft KV cache (same strides, batch extent 1)."""
    __slots__ = ("k", "v")

    def __init__(self, k, v):
        self.k, self.v = k, v


class EagleDrafter:
    def __init__(self, model: Llama2Model, lm_head_weight: torch.Tensor, tree: Sequence[Sequence[int]] = mc_sim_7b_63,
                 expand: str = "wchild", batch: int = 1, max_ctx: int = 2048, topk: int = 10):
        dev = lm_head_weight.device
        if lm_head_weight.dtype != BF16 or not lm_head_weight.is_cuda:
            raise TypeError("lm_head weight must be a bf16 device tensor")
        ops.topk_limit_check(lm_head_weight.shape[0], topk)
        self.model, self.lm_head, self.B, self.k = model, lm_head_weight.contiguous(), batch, topk
        self.tb = TreeBuffers.build_on_device(tree, topk=topk, expand=expand, device=dev)   # eagle_tree_build (C ABI)
        self.W = self.tb.max_depth + 1                       # catch-up rows per request (padded)
        if self.W > 16 or max(len(s.nodes) for s in self.tb.steps) > 16:
            raise ValueError("the tree attention kernel takes at most 16 query rows per request and step")
        self.n_fwd = sum(len(s.nodes) for s in self.tb.steps)
        self.R = 1 + self.n_fwd                              # flat top-k rows per request
        self.cap = (max_ctx + self.W + self.n_fwd + 255) // 256 * 256
        self.dev, self.D = dev, model.config.hidden_size
        self.wide_head_rows = 32      # lm_head row blocks from this many rows on take the MFMA GEMM (one weight pass)
        self.head_topk_fused = True   # 81-128 rows, and 1-16 rows of one request: lm_head + top-k without the logits (eagle_lm_head_topk_rows_bf16)
        self.rows_head_max = 128      # ... and up to this many the row-batch weight stream (eagle_gemm_rows_bf16; 81..128 rows: its up-to-384-column form, 292 vs 345 us cold)
        # static bound on stable_len + tree rows handed to the attention launcher (grid size); a caller
        # that knows a tighter bound for the coming cycles may lower it (must stay >= the true length)
        self.kv_bound = self.cap
        B, V = batch, lm_head_weight.shape[0]
        self.caches = [model.new_cache(B, self.cap, dev) for _ in model.layers]      # private: forward() never touches them
        # the frozen head's packed image for the weight-stream GEMM (ops.PackedWeight; the head is frozen: built once)
        self.lm_head_pk = ops.pack_weight(self.lm_head) if (model.pack_decode_weights and self.lm_head.shape[1] % 32 == 0) else self.lm_head
        model.set_cache_residency(self.lm_head_pk)
        # draft KV-cache lengths: len_cur = rows before the catch-up forward, len_nxt = after it
        self.len_cur = torch.zeros(B, dtype=torch.int32, device=dev)
        self.len_nxt = torch.zeros(B, dtype=torch.int32, device=dev)
        self.flat_ids = torch.zeros((B, self.R, topk), dtype=torch.int64, device=dev)
        self.flat_vals = torch.zeros((B, self.R, topk), dtype=BF16, device=dev)
        self._rowbase = torch.arange(B, dtype=torch.int64, device=dev) * self.W
        rows_max = B * max(self.W, self.tb.n_nodes, 16)
        self._ws = torch.empty(ops.query("eagle_topk_workspace_bytes", rows_max, V, topk), dtype=torch.uint8, device=dev)
        # candidate records of the one-request tree steps (1-16 rows): owned here, the cycle is replayed as a HIP graph
        need = ops.query("eagle_lm_head_topk_rows_workspace_bytes", 16, V, self.D, topk) if B == 1 else 0
        self._head_ws = torch.empty(need, dtype=torch.uint8, device=dev) if need > 0 else None
        ar_b = torch.arange(B, dtype=torch.int64, device=dev)[:, None]
        self._hid_idx, prev_n = [], 1
        for s in self.tb.steps:                               # static batch-major parent-row gathers
            self._hid_idx.append((ar_b * prev_n + s.parent_row[None, :]).reshape(-1).contiguous())
            prev_n = len(s.nodes)
        self.n0 = len(self.tb.steps[0].nodes)
        # state handed from verify() to the next draft() (persistent: a captured graph replays it)
        self.result = torch.zeros((B, 4), dtype=torch.int64, device=dev)
        self.next_ids = torch.zeros((B, self.W), dtype=torch.int64, device=dev)
        self.n_acc = torch.ones(B, dtype=torch.int32, device=dev)
        self.last_idx = torch.zeros(B, dtype=torch.int64, device=dev)          # last valid catch-up row
        self.hid_idx0 = torch.zeros(B * self.n0, dtype=torch.int64, device=dev)  # ... repeated per depth-1 node
        self.acc_hidden = torch.zeros((B, self.W, self.D), dtype=BF16, device=dev)
        self._tgt = None                                                         # verifier argmax per tree node

    @property
    def stable_len(self) -> torch.Tensor:
        """Final (non-speculative) rows of the draft KV cache, per request."""
        return self.len_nxt

    def reset(self):
        self.len_cur.zero_()
        self.len_nxt.zero_()

    # -------------------------------------------------------------- draft side
    def _head_topk(self, x: torch.Tensor, idx: Optional[torch.Tensor], rows: int, row0: int, n: int):
        """top-k(lm_head(x[idx])) -> flat rows [row0, row0+n) of every request."""
        if self.B == 1 and self.head_topk_fused and rows <= 16 and self._head_ws is not None and isinstance(self.lm_head_pk, ops.PackedWeight):
            # one request's tree step: the register-resident weight stream leaves k candidates per (row, block), no logits
            ops.lm_head_topk_into_table(x, self.lm_head_pk, self.k, self.flat_ids, self.flat_vals, row0, n, idx0=idx, M=rows,
                                        workspace=self._head_ws)
            return
        if 16 < rows <= 32 and self.D == 4096 and self.lm_head.shape[0] >= 8192:
            # 17-32 rows (a 32-request batch at one row each): both 16-row tiles of x stay in registers of the weight-stream
            # kernel, which reads 2 KB of every weight row per block step (in-cycle 151936 x 4096: see DESIGN.md section 10)
            logits = ops.linear_skinny(x, self.lm_head_pk, idx0=idx, M=rows)
        elif self.B > 1 and self.head_topk_fused and self.D % 128 == 0 and \
                ops.lm_head_topk_rows_ok(rows, self.lm_head.shape[0], self.D, self.k):
            # 81-128 rows of a request batch against a tall head: candidates out of the GEMM's epilogue, no logits
            ops.lm_head_topk_into_table(x, self.lm_head_pk, self.k, self.flat_ids, self.flat_vals, row0, n, idx0=idx, M=rows)
            return
        elif self.model.rows_min <= rows <= self.rows_head_max and self.D % 128 == 0:
            # decode batches: the row-batch weight stream (gather fused; 32 rows of the 151936-row head 257 vs 328 us on
            # the LDS-tiled kernels, tools/rows2_bench.py); past ~64 rows its x re-reads cost what the tiles' do
            logits = ops.linear_rows(x, self.lm_head_pk, idx0=idx, M=rows)
        elif rows >= self.wide_head_rows and self.D % 64 == 0:
            # wide batches: one weight pass on the MFMA GEMM (from 32 rows on it beats the weight-stream kernel, which
            # passes over the weights once per 32 rows: 304 vs 520 us at 32 rows of the 151936-row head)
            logits = ops.gemm_nt(x if idx is None else x.index_select(0, idx[:rows]), self.lm_head)
        else:
            logits = ops.linear_skinny(x, self.lm_head_pk, idx0=idx, M=rows, resident=self.model._keep_resident(self.lm_head_pk))
        if self.B == 1:
            ops.topk(logits, self.k, out_idx=self.flat_ids[0, row0:row0 + n], out_val=self.flat_vals[0, row0:row0 + n],
                     workspace=self._ws)
        else:                                     # straight into the [B, R, k] tables (no copy pass per step)
            ops.topk_into_table(logits, self.k, self.flat_ids, self.flat_vals, row0, n, self._ws)

    def prefill(self, hidden: torch.Tensor, input_ids: torch.Tensor):
        """First call of a request batch: hidden [B, ctx, D] = verifier hidden states of the prompt,
        input_ids [B, ctx+1] = prompt tokens plus the verifier's first sampled token.  Equal-length
        prompts (ragged batches are prefilled request by request into their cache rows)."""
        B, ctx, D = hidden.shape
        assert B == self.B and input_ids.shape[1] == ctx + 1
        m = self.model
        ids = input_ids[:, 1:].reshape(-1).contiguous()
        self.len_cur.zero_()
        x = m._forward_rows(ids, hidden.reshape(B * ctx, D).contiguous(), None, B * ctx, B, ctx, None, self.len_cur,
                            None, None, None, self.caches, past_host=0)
        self.len_nxt.fill_(ctx)
        last = torch.arange(B, dtype=torch.int64, device=self.dev) * ctx + (ctx - 1)
        self._head_topk(x, last, B, 0, 1)
        self._expand(x, last[:, None].expand(B, self.n0).reshape(-1).contiguous())
        return self.flat_ids, self.flat_vals

    def prefill_ragged(self, hiddens: Sequence[torch.Tensor], input_ids: Sequence[torch.Tensor]):
        """First call of a batch of requests with DIFFERENT prompt lengths (BASELINE configs[3]:
        ShareGPT-length prompts): hiddens[b] [ctx_b, D], input_ids[b] [ctx_b + 1].  Each prompt is
        run through the draft head into its own rows of the batched KV cache (per-request lengths
        live on the device from here on); the tree expansion that follows is batched."""
        B, m = self.B, self.model
        if len(hiddens) != B or len(input_ids) != B:
            raise ValueError(f"expected {B} requests")
        lastx = torch.empty((B, self.D), dtype=BF16, device=self.dev)
        self.len_cur.zero_()
        for b in range(B):
            h, ids = hiddens[b], input_ids[b].reshape(-1)
            ctx = h.shape[0]
            if h.dim() != 2 or ids.numel() != ctx + 1 or ctx < 1 or ctx + self.W + self.n_fwd > self.cap:
                raise ValueError(f"request {b}: hidden [ctx, D], ids [ctx+1], ctx within the cache capacity")
            views = [_CacheRows(c.k[b:b + 1], c.v[b:b + 1]) for c in self.caches]
            x = m._forward_rows(ids[1:].contiguous(), h.contiguous(), None, ctx, 1, ctx, None, self.len_cur[b:b + 1],
                                None, None, None, views, past_host=0)
            lastx[b].copy_(x[ctx - 1])
            self.len_nxt[b:b + 1].fill_(ctx)
        rows = torch.arange(B, dtype=torch.int64, device=self.dev)
        self._head_topk(lastx, rows, B, 0, 1)
        self._expand(lastx, rows[:, None].expand(B, self.n0).reshape(-1).contiguous())
        return self.flat_ids, self.flat_vals

    def set_accepted(self, n_acc: torch.Tensor):
        """Host-free state update for callers that do not go through ``verify``: n_acc [B] int32."""
        self.len_cur.copy_(self.len_nxt)
        self.len_nxt += n_acc
        self.last_idx.copy_(self._rowbase + (n_acc.long() - 1))
        self.hid_idx0.copy_(self.last_idx[:, None].expand(self.B, self.n0).reshape(-1))

    def draft(self, acc_hidden: torch.Tensor, acc_ids: torch.Tensor, n_acc: Optional[torch.Tensor] = None):
        """One expansion.  acc_hidden [B,W,D] bf16 / acc_ids [B,W] int64 hold, per request, the
        verifier hidden states of the newly accepted tokens and the token that FOLLOWS each of
        them (the draft head pairs hidden[i] with token[i+1]); rows beyond the accepted count are
        padding.  The accepted count comes from the preceding ``verify`` (device state) or, when
        given, from n_acc [B] (device int32, 1..W).  Advances the stable length by it.
        Returns (flat_ids [B,R,k], flat_vals [B,R,k]) -- persistent buffers."""
        B, W, D, m = self.B, self.W, self.D, self.model
        if n_acc is not None:
            self.set_accepted(n_acc)
        x = m._forward_rows(acc_ids.reshape(-1), acc_hidden.reshape(B * W, D), None, B * W, B, W, None,
                            self.len_cur, self.kv_bound, None, None, self.caches)
        self._head_topk(x, self.last_idx, B, 0, 1)
        self._expand(x, self.hid_idx0)
        return self.flat_ids, self.flat_vals

    def _expand(self, x: torch.Tensor, hid_idx0: torch.Tensor):
        B, m = self.B, self.model
        stride = self.R * self.k
        off, prev_row0 = 0, 0
        for d, s in enumerate(self.tb.steps):
            n = len(s.nodes)
            # tokens = flat_topk[previous rows][tree_indices] (gathered inside the fc GEMM); hidden =
            # parent rows of the previous forward; every node of depth d sits at position
            # stable_len + d (the kernel adds past = stable_len + off)
            x = m._forward_rows(self.flat_ids[:, prev_row0], x, hid_idx0 if d == 0 else self._hid_idx[d], B * n, B, n,
                                None, self.len_nxt, self.kv_bound, s.attn_mask_u8, None, self.caches, past_off0=off,
                                pos_off=d - off, pos_stride=0, emb_ind=s.tree_indices, emb_bstride=stride)
            self._head_topk(x, None, B * n, s.row_base, n)
            off += n
            prev_row0 = s.row_base

    # -------------------------------------------------------------- verify side
    def verify(self, root_token: torch.Tensor, target_logits: torch.Tensor, prev_len: torch.Tensor,
               cursor: Optional[torch.Tensor] = None):
        """target_logits [B, n_nodes, V] (bf16 or fp32): the verifier's logits at every tree node;
        root_token [B] int64; prev_len [B] int32 = verifier cache rows before the tree.
        Fills and returns result [B,4] = {best_path, accept_length, next_token, new_len}; also
        fills next_ids [B,W] and n_acc [B] (inputs of the next ``draft``).
        cursor [2] (device int64): target_logits is then a RING [slots, B, n_nodes, V] of verifier
        output buffers and slot cursor[0] is read; the chain advances the cursor on the device
        (cursor[1] keeps the slot for ``compact``), so a captured cycle needs no host update."""
        B, n = self.B, self.tb.n_nodes
        if self._tgt is None:
            self._tgt = torch.empty((B, n), dtype=torch.int64, device=self.dev)
        V = target_logits.shape[-1]
        ops.topk(target_logits.reshape(-1, V), 1, want_values=False, workspace=self._ws, out_idx=self._tgt.view(B * n, 1),
                 block_index=None if cursor is None else cursor[0:1], rows=B * n)
        return ops.greedy_accept(root_token, self.flat_ids, self.tb.tree_indices, self.tb.retrieve_indices, self._tgt,
                                 prev_len, B=B, next_ids=self.next_ids, n_acc=self.n_acc, result=self.result,
                                 last_idx=self.last_idx, hid_idx0=self.hid_idx0, len_cur=self.len_cur,
                                 len_nxt=self.len_nxt, cursor=cursor)

    def compact(self, prev_len: torch.Tensor, target_kv: Optional[torch.Tensor], target_hidden: torch.Tensor,
                kv_inner: int = 1, root_io: Optional[torch.Tensor] = None, update_prev_len: bool = False,
                cursor: Optional[torch.Tensor] = None) -> torch.Tensor:
        """In-place compaction of the verifier's KV cache ([outer, B, inner, cap, hd], the tree's rows
        at [prev_len, prev_len+n_nodes)) and gather of the accepted path's hidden states
        [B, W, D] (zeros beyond accept_length+1; a persistent buffer) for the next ``draft``.
        root_io / update_prev_len: also overwrite the caller's root-token and prev_len arrays with
        next_token / new_len, so a serving loop needs no host round trip between cycles."""
        # (with ``cursor``, target_hidden is the ring [slots, B, n_nodes, D] matching ``verify``'s)
        if target_kv is not None:
            ops.kv_compact(target_kv, self.B, kv_inner, self.tb.retrieve_indices, self.result, prev_len, self.model._err)
        return ops.gather_accepted(target_hidden, self.tb.retrieve_indices, self.result, B=self.B, root_io=root_io,
                                   prev_len_io=prev_len if update_prev_len else None, out=self.acc_hidden,
                                   src_block=None if cursor is None else cursor[1:2])

    def check_errors(self):
        """Host-side look at the device error flag (sync point; not for the hot loop)."""
        e = int(self.model._err.item()) if self.model._err is not None else 0
        if e:
            raise ops._hip.HipKernelError(f"device-side bound violation (flag {e}): KV cache capacity exceeded")
