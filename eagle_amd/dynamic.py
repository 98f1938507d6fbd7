"""Dynamic draft tree on one MI355X: the draft/verify cycle SGLang runs for the
reference's benchmark flags ``--steps S --k K --draft T`` (benchmark/benchmark.py:26-35, :48-56:
``speculative_num_steps``, ``speculative_eagle_topk``, ``speculative_num_draft_tokens``).

The algorithm is EAGLE-2's, restated in DESIGN.md section 6b (it is not in /root/reference): the
catch-up forward gives the root's top-k; each of the S-1 following draft forwards runs the k frontier
nodes, scores their k*k children by the product of softmax probabilities along the path and keeps
the k best as the next frontier; the T-1 best candidates overall form the tree that the verifier
checks; acceptance walks that tree along the verifier's argmax.

Everything between "the verifier produced logits for the tree" and "the next tree is ready" runs as
HIP kernels with device-resident state and static shapes, like ``EagleDrafter``; ``batch`` independent requests
with ragged context lengths ride in every launch (BASELINE configs[3] serves batches with these flags):
  draft forward    eagle_gemm_skinny_bf16 / eagle_tree_attention_bf16 (the tree mask of a step is the
                   device buffer the previous selection wrote)
  top-k + lse      eagle_topk_lse_vocab (one pass over the logits)
  selection        eagle_dyn_step, eagle_dyn_finalize
  verification     eagle_topk_vocab (k = 1), eagle_dyn_accept, eagle_kv_compact_bf16,
                   eagle_gather_accepted_bf16
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .llama2 import Llama2Model

BF16 = torch.bfloat16


class EagleDynamicDrafter:
    def __init__(self, model: Llama2Model, lm_head_weight: torch.Tensor, steps: int = 5, topk: int = 8,
                 num_draft_tokens: int = 32, max_ctx: int = 2048, keep_debug: bool = False, batch: int = 1):
        dev = lm_head_weight.device
        if lm_head_weight.dtype != BF16 or not lm_head_weight.is_cuda:
            raise TypeError("lm_head weight must be a bf16 device tensor")
        S, k, T, B = int(steps), int(topk), int(num_draft_tokens), int(batch)
        P = k + (S - 1) * k * k
        if not (1 <= k <= 16 and S >= 1 and 2 <= T <= 64 and T - 1 <= P and P <= 1024 and S + 1 <= 16 and B >= 1):
            raise ValueError(f"steps={S}, topk={k}, num_draft_tokens={T}: need k <= 16, T <= 64, T-1 <= k+(S-1)k^2 <= 1024")
        ops.topk_limit_check(lm_head_weight.shape[0], k)
        self.model, self.lm_head, self.S, self.k, self.T, self.P, self.B = model, lm_head_weight.contiguous(), S, k, T, P, B
        self.W = S + 1                                          # catch-up rows (padded) = longest accepted path + 1
        self.n_tree = (S - 1) * k                               # draft KV rows appended per cycle
        self.cap = (max_ctx + self.W + self.n_tree + 255) // 256 * 256
        self.kv_bound = self.cap
        self.dev, self.D = dev, model.config.hidden_size
        self.keep_debug, self.debug = keep_debug, []
        self.wide_head_rows = 32
        self.rows_head_max = 128
        V = lm_head_weight.shape[0]
        self.caches = [model.new_cache(B, self.cap, dev) for _ in model.layers]      # private: forward() never touches them
        # the frozen head's packed image for the weight-stream GEMM (ops.PackedWeight; the head is frozen: built once)
        self.lm_head_pk = ops.pack_weight(self.lm_head) if (model.pack_decode_weights and self.lm_head.shape[1] % 32 == 0) else self.lm_head
        model.set_cache_residency(self.lm_head_pk)
        i32, i64, f32 = torch.int32, torch.int64, torch.float32
        self.len_cur = torch.zeros(B, dtype=i32, device=dev)
        self.len_nxt = torch.zeros(B, dtype=i32, device=dev)
        self.pool_tok = torch.zeros((B, P), dtype=i64, device=dev)
        self.pool_score = torch.zeros((B, P), dtype=f32, device=dev)
        self.pool_parent = torch.zeros((B, P), dtype=i32, device=dev)
        self.ftok = torch.zeros((B, k), dtype=i64, device=dev)
        self.fscore = [torch.zeros((B, k), dtype=f32, device=dev) for _ in range(2)]
        self.fpool = [torch.zeros((B, k), dtype=i32, device=dev) for _ in range(2)]
        self.parent_row = torch.zeros((B, k), dtype=i64, device=dev)
        self.masks = [torch.zeros((B, k, (i + 1) * k), dtype=torch.uint8, device=dev) for i in range(max(1, S - 1))]
        self.vals = torch.zeros((B * k, k), dtype=BF16, device=dev)
        self.ids = torch.zeros((B * k, k), dtype=i64, device=dev)
        self.lse = torch.zeros(B * k, dtype=f32, device=dev)
        self._ws = torch.empty(ops.query("eagle_topk_workspace_bytes", B * max(T, k, 16), V, max(k, 1)), dtype=torch.uint8, device=dev)
        # the verified trees (inputs of the verifier forward) and the verify-side state
        self.tree_tok = torch.zeros((B, T), dtype=i64, device=dev)
        self.tree_parent = torch.zeros((B, T), dtype=i32, device=dev)
        self.tree_pos = torch.zeros((B, T), dtype=i64, device=dev)
        self.tree_mask = torch.zeros((B, T, T), dtype=torch.uint8, device=dev)
        self.sel = torch.zeros((B, T - 1), dtype=i32, device=dev)
        self.path = torch.zeros((B, self.W), dtype=i64, device=dev)          # doubles as the per-request one-row retrieve table
        self.result = torch.zeros((B, 4), dtype=i64, device=dev)
        self.next_ids = torch.zeros((B, self.W), dtype=i64, device=dev)
        self.n_acc = torch.ones(B, dtype=i32, device=dev)
        self.last_idx = torch.zeros(B, dtype=i64, device=dev)
        self.acc_hidden = torch.zeros((B, self.W, self.D), dtype=BF16, device=dev)
        self._tgt = torch.zeros((B * T, 1), dtype=i64, device=dev)
        self._root = torch.zeros(B, dtype=i64, device=dev)
        self._one = None if B > 1 else True

    @property
    def stable_len(self) -> torch.Tensor:
        return self.len_nxt

    def _v(self, t: torch.Tensor) -> torch.Tensor:
        """Batch 1 keeps the round-1 shapes of the returned buffers (no leading batch dimension)."""
        return t[0] if self.B == 1 else t

    # ------------------------------------------------------------------ draft side
    def _head(self, x: torch.Tensor, idx: Optional[torch.Tensor], rows: int):
        if 16 < rows <= 32 and self.D == 4096 and self.lm_head.shape[0] >= 8192:
            logits = ops.linear_skinny(x, self.lm_head_pk, idx0=idx, M=rows)        # both 16-row x tiles in registers (keepx)
        elif self.model.rows_min <= rows <= self.rows_head_max and self.D % 128 == 0:
            logits = ops.linear_rows(x, self.lm_head_pk, idx0=idx, M=rows)          # decode batches: row-batch weight stream
        elif rows >= self.wide_head_rows and self.D % 64 == 0:
            logits = ops.gemm_nt(x if idx is None else x.index_select(0, idx[:rows]), self.lm_head)
        else:
            logits = ops.linear_skinny(x, self.lm_head_pk, idx0=idx, M=rows, resident=self.model._keep_resident(self.lm_head_pk))
        # top-k and the rows' log-sum-exp out of one pass over the logits (two launches, not three)
        ops.topk(logits, self.k, out_idx=self.ids[:rows], out_val=self.vals[:rows], workspace=self._ws, lse_out=self.lse[:rows])
        if self.keep_debug:
            self.debug.append({"logits": logits.clone(), "vals": self.vals[:rows].clone(), "ids": self.ids[:rows].clone(),
                               "lse": self.lse[:rows].clone()})

    def _expand(self, x: torch.Tensor, root_row: torch.Tensor, root_token: torch.Tensor):
        """x: rows of the catch-up forward, root_row [B] the row whose hidden state is each request's root."""
        S, k, m, B = self.S, self.k, self.model, self.B
        self.debug = []
        self._head(x, root_row, B)
        if self.keep_debug:
            self.debug[-1]["x_root"] = x[root_row].clone()
        ops.dyn_step(0, k, S == 1, self.vals, self.ids, self.lse, None, None, None, root_row, self.pool_tok, self.pool_score,
                     self.pool_parent, self.ftok, self.fscore[0], self.fpool[0], self.parent_row, self.masks[0], S, B)
        for i in range(1, S):
            # forward of the k frontier nodes of every request (depth i): tokens ftok, hidden = the parent's output row,
            # KV rows appended at stable + (i-1) k, every node at position stable + i - 1
            off = (i - 1) * k
            x = m._forward_rows(self.ftok.reshape(-1), x, self.parent_row.reshape(-1), B * k, B, k, None, self.len_nxt, self.kv_bound,
                                self._v(self.masks[i - 1]), None, self.caches, past_off0=off, pos_off=(i - 1) - off, pos_stride=0)
            if self.keep_debug:
                self.debug[-1].update({"ftok": self.ftok.clone(), "parent_row": self.parent_row.clone(),
                                       "mask": self.masks[i - 1].clone(), "x": x.clone()})
            self._head(x, None, B * k)
            last = i == S - 1
            a, b = (i - 1) & 1, i & 1
            ops.dyn_step(i, k, last, self.vals, self.ids, self.lse, self.fscore[a], self.fpool[a], self.masks[i - 1], None,
                         self.pool_tok, self.pool_score, self.pool_parent, self.ftok, self.fscore[b], self.fpool[b],
                         self.parent_row, None if last else self.masks[i], S, B)
        self._root.copy_(root_token.reshape(B))
        ops.dyn_finalize(k, S, self.T, self.pool_tok, self.pool_score, self.pool_parent, self._root, self.tree_tok,
                         self.tree_parent, self.tree_pos, self.tree_mask, self.sel, B)
        return self._v(self.tree_tok), self._v(self.tree_pos), self._v(self.tree_mask)

    def prefill(self, hidden: torch.Tensor, input_ids: torch.Tensor):
        """hidden [B, ctx, D] = verifier hidden states of equal-length prompts, input_ids [B, ctx+1] = prompt tokens plus
        the verifier's first sampled token (the root).  Returns (tree_tok [B,T], tree_pos [B,T] = depth, tree_mask
        [B,T,T]; without the batch dimension when batch == 1): the verifier runs tree_tok at positions prev_len +
        tree_pos under tree_mask."""
        B, ctx, D = hidden.shape
        if B != self.B or input_ids.shape[1] != ctx + 1:
            raise ValueError(f"prefill: hidden [{self.B}, ctx, D], input_ids [{self.B}, ctx+1]")
        m = self.model
        self.len_cur.zero_()
        x = m._forward_rows(input_ids[:, 1:].reshape(-1).contiguous(), hidden.reshape(B * ctx, D).contiguous(), None, B * ctx, B, ctx,
                            None, self.len_cur, None, None, None, self.caches, past_host=0)
        self.len_nxt.fill_(ctx)
        last = torch.arange(B, dtype=torch.int64, device=self.dev) * ctx + (ctx - 1)
        return self._expand(x, last, input_ids[:, -1])

    def prefill_ragged(self, hiddens, input_ids):
        """Requests with different prompt lengths: hiddens[b] [ctx_b, D], input_ids[b] [ctx_b + 1].  Each prompt runs
        through the draft head into its own rows of the batched KV cache; everything after is batched."""
        from .speculative import _CacheRows
        B, m = self.B, self.model
        if len(hiddens) != B or len(input_ids) != B:
            raise ValueError(f"expected {B} requests")
        lastx = torch.empty((B, self.D), dtype=BF16, device=self.dev)
        roots = torch.empty(B, dtype=torch.int64, device=self.dev)
        self.len_cur.zero_()
        for b in range(B):
            h, ids = hiddens[b], input_ids[b].reshape(-1)
            ctx = h.shape[0]
            if h.dim() != 2 or ids.numel() != ctx + 1 or ctx < 1 or ctx + self.W + self.n_tree > self.cap:
                raise ValueError(f"request {b}: hidden [ctx, D], ids [ctx+1], ctx within the cache capacity")
            views = [_CacheRows(c.k[b:b + 1], c.v[b:b + 1]) for c in self.caches]
            x = m._forward_rows(ids[1:].contiguous(), h.contiguous(), None, ctx, 1, ctx, None, self.len_cur[b:b + 1],
                                None, None, None, views, past_host=0)
            lastx[b].copy_(x[ctx - 1])
            roots[b] = ids[-1]
            self.len_nxt[b:b + 1].fill_(ctx)
        return self._expand(lastx, torch.arange(B, dtype=torch.int64, device=self.dev), roots)

    def draft(self, acc_hidden: torch.Tensor, acc_ids: torch.Tensor):
        """Next trees after a ``verify``: acc_hidden [B,W,D] / acc_ids [B,W] as for EagleDrafter.draft
        (the accepted rows' verifier hidden states and the token following each; padded)."""
        W, D, m, B = self.W, self.D, self.model, self.B
        x = m._forward_rows(acc_ids.reshape(-1), acc_hidden.reshape(B * W, D), None, B * W, B, W, None, self.len_cur, self.kv_bound,
                            None, None, self.caches)
        return self._expand(x, self.last_idx, self.result[:, 2])

    # ------------------------------------------------------------------ verify side
    def verify(self, target_logits: torch.Tensor, prev_len: torch.Tensor, cursor: Optional[torch.Tensor] = None):
        """target_logits [B, T, V] ([T, V] at batch 1): the verifier's logits at the tree nodes; prev_len [B] int32.
        result [B,4] = {0, accept_length, next_token, new_len}; also path, next_ids, n_acc."""
        T, B = self.T, self.B
        V = target_logits.shape[-1]
        ops.topk(target_logits.reshape(-1, V), 1, want_values=False, workspace=self._ws, out_idx=self._tgt,
                 block_index=None if cursor is None else cursor[0:1], rows=B * T)
        ops.dyn_accept(T, self.W, self.tree_tok, self.tree_parent, self._tgt, prev_len, self.path, self.result, self.next_ids,
                       self.n_acc, self.last_idx, self.len_cur, self.len_nxt, cursor, B)
        return self.result

    def compact(self, prev_len: torch.Tensor, target_kv: Optional[torch.Tensor], target_hidden: torch.Tensor,
                kv_inner: int = 1, root_io: Optional[torch.Tensor] = None, update_prev_len: bool = False,
                cursor: Optional[torch.Tensor] = None) -> torch.Tensor:
        """As EagleDrafter.compact, with each request's accepted path as its one-row retrieve table."""
        if self.B > 1:
            return self._compact_batched(prev_len, target_kv, target_hidden, kv_inner, root_io, update_prev_len, cursor)
        if target_kv is not None:
            ops.kv_compact(target_kv, 1, kv_inner, self.path, self.result, prev_len, self.model._err)
        return ops.gather_accepted(target_hidden, self.path, self.result, B=1, root_io=root_io,
                                   prev_len_io=prev_len if update_prev_len else None, out=self.acc_hidden,
                                   src_block=None if cursor is None else cursor[1:2])

    def _compact_batched(self, prev_len, target_kv, target_hidden, kv_inner, root_io, update_prev_len, cursor):
        # the static-tree kernels index ONE retrieve table by result[b][0] (= best path): stack the per-request paths as
        # the table's rows and point request b at row b
        if not hasattr(self, "_res_b"):
            self._res_b = torch.zeros((self.B, 4), dtype=torch.int64, device=self.dev)
            self._rows = torch.arange(self.B, dtype=torch.int64, device=self.dev)
        self._res_b.copy_(self.result)
        self._res_b[:, 0] = self._rows
        if target_kv is not None:
            ops.kv_compact(target_kv, self.B, kv_inner, self.path, self._res_b, prev_len, self.model._err)
        return ops.gather_accepted(target_hidden, self.path, self._res_b, B=self.B, root_io=root_io,
                                   prev_len_io=prev_len if update_prev_len else None, out=self.acc_hidden,
                                   src_block=None if cursor is None else cursor[1:2])

    def check_errors(self):
        e = int(self.model._err.item()) if self.model._err is not None else 0
        if e:
            raise ops._hip.HipKernelError(f"device-side bound violation (flag {e}): KV cache capacity exceeded")
