"""Dynamic draft tree on one MI355X: the draft/verify cycle SGLang runs for the
reference's benchmark flags ``--steps S --k K --draft T`` (benchmark/benchmark.py:26-35, :48-56:
``speculative_num_steps``, ``speculative_eagle_topk``, ``speculative_num_draft_tokens``).

The algorithm is EAGLE-2's, restated in DESIGN.md section 6b (it is not in /root/reference): the
catch-up forward gives the root's top-k; each of the S-1 following draft forwards runs the k frontier
nodes, scores their k*k children by the product of softmax probabilities along the path and keeps
the k best as the next frontier; the T-1 best candidates overall form the tree that the verifier
checks; acceptance walks that tree along the verifier's argmax.

Everything between "the verifier produced logits for the tree" and "the next tree is ready" runs as
HIP kernels with device-resident state and static shapes (batch 1), like ``EagleDrafter``:
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
                 num_draft_tokens: int = 32, max_ctx: int = 2048, keep_debug: bool = False):
        dev = lm_head_weight.device
        if lm_head_weight.dtype != BF16 or not lm_head_weight.is_cuda:
            raise TypeError("lm_head weight must be a bf16 device tensor")
        S, k, T = int(steps), int(topk), int(num_draft_tokens)
        P = k + (S - 1) * k * k
        if not (1 <= k <= 16 and S >= 1 and 2 <= T <= 64 and T - 1 <= P and P <= 1024 and S + 1 <= 16):
            raise ValueError(f"steps={S}, topk={k}, num_draft_tokens={T}: need k <= 16, T <= 64, T-1 <= k+(S-1)k^2 <= 1024")
        ops.topk_limit_check(lm_head_weight.shape[0], k)
        self.model, self.lm_head, self.S, self.k, self.T, self.P = model, lm_head_weight.contiguous(), S, k, T, P
        self.W = S + 1                                          # catch-up rows (padded) = longest accepted path + 1
        self.n_tree = (S - 1) * k                               # draft KV rows appended per cycle
        self.cap = (max_ctx + self.W + self.n_tree + 255) // 256 * 256
        self.kv_bound = self.cap
        self.dev, self.D = dev, model.config.hidden_size
        self.keep_debug, self.debug = keep_debug, []
        V = lm_head_weight.shape[0]
        self.caches = [model.new_cache(1, self.cap, dev) for _ in model.layers]      # private: forward() never touches them
        model.set_cache_residency(self.lm_head)
        i32, i64, f32 = torch.int32, torch.int64, torch.float32
        self.len_cur = torch.zeros(1, dtype=i32, device=dev)
        self.len_nxt = torch.zeros(1, dtype=i32, device=dev)
        self.pool_tok = torch.zeros(P, dtype=i64, device=dev)
        self.pool_score = torch.zeros(P, dtype=f32, device=dev)
        self.pool_parent = torch.zeros(P, dtype=i32, device=dev)
        self.ftok = torch.zeros(k, dtype=i64, device=dev)
        self.fscore = [torch.zeros(k, dtype=f32, device=dev) for _ in range(2)]
        self.fpool = [torch.zeros(k, dtype=i32, device=dev) for _ in range(2)]
        self.parent_row = torch.zeros(k, dtype=i64, device=dev)
        self.masks = [torch.zeros((k, (i + 1) * k), dtype=torch.uint8, device=dev) for i in range(max(1, S - 1))]
        self.vals = torch.zeros((k, k), dtype=BF16, device=dev)
        self.ids = torch.zeros((k, k), dtype=i64, device=dev)
        self.lse = torch.zeros(k, dtype=f32, device=dev)
        self._ws = torch.empty(ops.query("eagle_topk_workspace_bytes", max(T, k, 16), V, max(k, 1)), dtype=torch.uint8, device=dev)
        # the verified tree (inputs of the verifier forward) and the verify-side state
        self.tree_tok = torch.zeros(T, dtype=i64, device=dev)
        self.tree_parent = torch.zeros(T, dtype=i32, device=dev)
        self.tree_pos = torch.zeros(T, dtype=i64, device=dev)
        self.tree_mask = torch.zeros((T, T), dtype=torch.uint8, device=dev)
        self.sel = torch.zeros(T - 1, dtype=i32, device=dev)
        self.path = torch.zeros((1, self.W), dtype=i64, device=dev)          # doubles as the one-row retrieve table
        self.result = torch.zeros((1, 4), dtype=i64, device=dev)
        self.next_ids = torch.zeros((1, self.W), dtype=i64, device=dev)
        self.n_acc = torch.ones(1, dtype=i32, device=dev)
        self.last_idx = torch.zeros(1, dtype=i64, device=dev)
        self.acc_hidden = torch.zeros((1, self.W, self.D), dtype=BF16, device=dev)
        self._tgt = torch.zeros((T, 1), dtype=i64, device=dev)
        self._root = torch.zeros(1, dtype=i64, device=dev)

    @property
    def stable_len(self) -> torch.Tensor:
        return self.len_nxt

    # ------------------------------------------------------------------ draft side
    def _head(self, x: torch.Tensor, idx: Optional[torch.Tensor], rows: int):
        logits = ops.linear_skinny(x, self.lm_head, idx0=idx, M=rows, resident=self.model._keep_resident(self.lm_head))
        # top-k and the rows' log-sum-exp out of one pass over the logits (two launches, not three)
        ops.topk(logits, self.k, out_idx=self.ids[:rows], out_val=self.vals[:rows], workspace=self._ws, lse_out=self.lse[:rows])
        if self.keep_debug:
            self.debug.append({"logits": logits.clone(), "vals": self.vals[:rows].clone(), "ids": self.ids[:rows].clone(),
                               "lse": self.lse[:rows].clone()})

    def _expand(self, x: torch.Tensor, root_row: torch.Tensor, root_token: torch.Tensor):
        """x: rows of the catch-up forward, root_row [1] the row whose hidden state is the root's."""
        S, k, m = self.S, self.k, self.model
        self.debug = []
        self._head(x, root_row, 1)
        if self.keep_debug:
            self.debug[-1]["x_root"] = x[root_row].clone()
        ops.dyn_step(0, k, S == 1, self.vals, self.ids, self.lse, None, None, None, root_row, self.pool_tok, self.pool_score,
                     self.pool_parent, self.ftok, self.fscore[0], self.fpool[0], self.parent_row, self.masks[0])
        for i in range(1, S):
            # forward of the k frontier nodes (depth i): tokens ftok, hidden = the parent's output row,
            # KV rows appended at stable + (i-1) k, every node at position stable + i - 1
            off = (i - 1) * k
            x = m._forward_rows(self.ftok, x, self.parent_row, k, 1, k, None, self.len_nxt, self.kv_bound, self.masks[i - 1],
                                None, self.caches, past_off0=off, pos_off=(i - 1) - off, pos_stride=0)
            if self.keep_debug:
                self.debug[-1].update({"ftok": self.ftok.clone(), "parent_row": self.parent_row.clone(),
                                       "mask": self.masks[i - 1].clone(), "x": x.clone()})
            self._head(x, None, k)
            last = i == S - 1
            a, b = (i - 1) & 1, i & 1
            ops.dyn_step(i, k, last, self.vals, self.ids, self.lse, self.fscore[a], self.fpool[a], self.masks[i - 1], None,
                         self.pool_tok, self.pool_score, self.pool_parent, self.ftok, self.fscore[b], self.fpool[b],
                         self.parent_row, None if last else self.masks[i])
        self._root.copy_(root_token.reshape(1))
        ops.dyn_finalize(k, S, self.T, self.pool_tok, self.pool_score, self.pool_parent, self._root, self.tree_tok,
                         self.tree_parent, self.tree_pos, self.tree_mask, self.sel)
        return self.tree_tok, self.tree_pos, self.tree_mask

    def prefill(self, hidden: torch.Tensor, input_ids: torch.Tensor):
        """hidden [1, ctx, D] = verifier hidden states of the prompt, input_ids [1, ctx+1] = prompt tokens
        plus the verifier's first sampled token (the root).  Returns (tree_tok [T], tree_pos [T] = depth,
        tree_mask [T,T]): the verifier runs tree_tok at positions prev_len + tree_pos under tree_mask."""
        _, ctx, D = hidden.shape
        m = self.model
        self.len_cur.zero_()
        x = m._forward_rows(input_ids[0, 1:].contiguous(), hidden.reshape(ctx, D).contiguous(), None, ctx, 1, ctx, None,
                            self.len_cur, None, None, None, self.caches, past_host=0)
        self.len_nxt.fill_(ctx)
        last = torch.full((1,), ctx - 1, dtype=torch.int64, device=self.dev)
        return self._expand(x, last, input_ids[0, -1:])

    def draft(self, acc_hidden: torch.Tensor, acc_ids: torch.Tensor):
        """Next tree after a ``verify``: acc_hidden [1,W,D] / acc_ids [1,W] as for EagleDrafter.draft
        (the accepted rows' verifier hidden states and the token following each; padded)."""
        W, D, m = self.W, self.D, self.model
        x = m._forward_rows(acc_ids.reshape(-1), acc_hidden.reshape(W, D), None, W, 1, W, None, self.len_cur, self.kv_bound,
                            None, None, self.caches)
        return self._expand(x, self.last_idx, self.result[0, 2:3])

    # ------------------------------------------------------------------ verify side
    def verify(self, target_logits: torch.Tensor, prev_len: torch.Tensor, cursor: Optional[torch.Tensor] = None):
        """target_logits [T, V]: the verifier's logits at the tree nodes; prev_len [1] int32.
        result [1,4] = {0, accept_length, next_token, new_len}; also path, next_ids, n_acc."""
        T = self.T
        V = target_logits.shape[-1]
        ops.topk(target_logits.reshape(-1, V), 1, want_values=False, workspace=self._ws, out_idx=self._tgt,
                 block_index=None if cursor is None else cursor[0:1], rows=T)
        ops.dyn_accept(T, self.W, self.tree_tok, self.tree_parent, self._tgt, prev_len, self.path, self.result, self.next_ids,
                       self.n_acc, self.last_idx, self.len_cur, self.len_nxt, cursor)
        return self.result

    def compact(self, prev_len: torch.Tensor, target_kv: Optional[torch.Tensor], target_hidden: torch.Tensor,
                kv_inner: int = 1, root_io: Optional[torch.Tensor] = None, update_prev_len: bool = False,
                cursor: Optional[torch.Tensor] = None) -> torch.Tensor:
        """As EagleDrafter.compact, with the accepted path as the one-row retrieve table."""
        if target_kv is not None:
            ops.kv_compact(target_kv, 1, kv_inner, self.path, self.result, prev_len, self.model._err)
        return ops.gather_accepted(target_hidden, self.path, self.result, B=1, root_io=root_io,
                                   prev_len_io=prev_len if update_prev_len else None, out=self.acc_hidden,
                                   src_block=None if cursor is None else cursor[1:2])

    def check_errors(self):
        e = int(self.model._err.item()) if self.model._err is not None else 0
        if e:
            raise ops._hip.HipKernelError(f"device-side bound violation (flag {e}): KV cache capacity exceeded")
