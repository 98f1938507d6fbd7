"""MI355X-native drop-in for ``eagle/llama2.py``'s ``Llama2Model`` (the EAGLE-1 draft head).

Same constructor, ``forward`` signature, return convention, attributes (``tree_mask``,
``reset()``, ``init_tree()``, ``gradient_checkpointing``, ``embed_tokens``, ``layers``, ``fc``)
and ``state_dict`` keys as the reference (eagle/llama2.py:404-587), so checkpoints in the
SGLang-compatible layout load unchanged.  The compute is different: every op of the forward runs
in hand-written HIP kernels reached through the C ABI (include/eagle_hip.h):

    fc(cat(embed_tokens[ids], hidden))      -> eagle_gemm_skinny_bf16 (row gather + concat fused)
    q|k|v projection (one packed weight)    -> eagle_gemm_skinny_bf16
    RoPE + cache append + masked attention  -> eagle_tree_attention_bf16
    o_proj + residual                       -> eagle_gemm_skinny_bf16 (residual epilogue)
    post_attention_layernorm                -> eagle_rmsnorm_fwd_bf16
    silu(gate)*up                           -> eagle_gemm_skinny_bf16 (SiLU*mul epilogue)
    down_proj + residual                    -> eagle_gemm_skinny_bf16

The KV cache is a preallocated [B, kv_heads, cap, hd] buffer per layer; ``forward`` returns
views ``[:, :, :kv_len]`` with the reference's shapes.  A view handed back as
``past_key_values`` is recognised and appended to in place (no torch.cat, eagle/llama2.py:236-239);
foreign tensors are copied in first.  Rows at or beyond a returned view's length may be
overwritten by a later call that restarts from a shorter past (tree expansion does exactly that).

bf16 on a HIP device only.  There is no CPU / eager fallback: without the built library or on
CPU tensors ``forward`` raises.
"""
from __future__ import annotations

import json
import os
from typing import List, Optional, Tuple

import torch
from torch import nn

from . import ops
from .tree import TreeBuffers, generate_tree_buffers, mc_sim_7b_63

top_k = 10  # eagle/llama2.py:11


class LlamaRMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states):
        return ops.rmsnorm(hidden_states, self.weight, self.variance_epsilon)


class LlamaAttention(nn.Module):
    """Parameter container with the reference's names (eagle/llama2.py:152-171)."""

    def __init__(self, config):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = getattr(config, "num_key_value_heads", None) or self.num_heads
        if self.head_dim * self.num_heads != self.hidden_size:
            raise ValueError(
                f"hidden_size must be divisible by num_heads (got `hidden_size`: {self.hidden_size}"
                f" and `num_heads`: {self.num_heads}).")
        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=False)
        self.k_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=False)
        self.v_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=False)
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, self.hidden_size, bias=False)


class LlamaMLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.hidden_act != "silu":
            raise NotImplementedError("only hidden_act='silu' (both shipped configs) is implemented")
        self.gate_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.up_proj = nn.Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.down_proj = nn.Linear(config.intermediate_size, config.hidden_size, bias=False)


class LlamaDecoderLayer(nn.Module):
    def __init__(self, config, index):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.self_attn = LlamaAttention(config=config)
        self.mlp = LlamaMLP(config)
        self.index = index
        if self.index != 0:  # layer 0 has no input_layernorm (eagle/llama2.py:344-345)
            self.input_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)


class _LayerCache:
    """One preallocated KV buffer.  ``forward`` hands out views of it as ``past_key_values``; each view carries a
    stamp (cache, epoch, rows) so that a view whose rows a later call has overwritten is recognised when it comes
    back (the reference returns fresh tensors from torch.cat, eagle/llama2.py:236-239, so its callers may keep any
    number of branches alive; here a branch stays alive until a call restarts the same buffer from fewer rows than
    the branch holds -- then it is stale and handing it back raises instead of attending to foreign rows)."""
    __slots__ = ("k", "v", "cap", "epoch", "truncs")

    def __init__(self, B, HK, cap, hd, device):
        self.k = torch.zeros((B, HK, cap, hd), dtype=torch.bfloat16, device=device)
        self.v = torch.zeros((B, HK, cap, hd), dtype=torch.bfloat16, device=device)
        self.cap = cap
        self.epoch = 0                    # bumped by every call that overwrites rows handed out earlier
        self.truncs = []                  # [(epoch, first row overwritten)] of those calls, oldest first

    def views(self, total: int):
        k, v = self.k[:, :, :total], self.v[:, :, :total]
        k._eagle_stamp = v._eagle_stamp = (self, self.epoch, total)
        return k, v

    def restart(self, past_len: int, rows_handed_out: int):
        """A call is about to write rows [past_len, ...): views holding more than past_len rows go stale."""
        if past_len < rows_handed_out:
            self.epoch += 1
            self.truncs.append((self.epoch, past_len))
            if len(self.truncs) > 256:                      # collapse old history conservatively
                lo = min(t for _, t in self.truncs[:128])
                self.truncs = [(self.truncs[127][0], lo)] + self.truncs[128:]

    def is_stale(self, stamp) -> bool:
        _, epoch, rows = stamp
        return any(e > epoch and t < rows for e, t in self.truncs)


class Llama2Model(nn.Module):
    def __init__(self, config, load_emb=False, path=None, bias=True):
        super().__init__()
        if getattr(config, "rope_scaling", None) not in (None, {}) and \
                getattr(config, "rope_scaling", {}).get("rope_type", "default") != "default":
            raise NotImplementedError("rope_scaling variants are outside the hot path (SURVEY.md row 6b)")
        if getattr(config, "pretraining_tp", 1) != 1:
            raise NotImplementedError("pretraining_tp > 1 is outside the hot path (SURVEY.md row 6c)")
        self.config = config
        self.gradient_checkpointing = True
        self.padding_idx = config.pad_token_id
        self.vocab_size = config.vocab_size
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, self.padding_idx)
        if load_emb:
            self.embed_tokens.weight.data = _load_verifier_tensor(path, "model.embed_tokens.weight").float()
        self.layers = nn.ModuleList([LlamaDecoderLayer(config, index) for index in range(config.num_hidden_layers)])
        self.fc = nn.Linear(2 * config.hidden_size, config.hidden_size, bias=bias)
        for param in self.embed_tokens.parameters():
            param.requires_grad = False
        self.tree_mask = None
        self._caches: List[Optional[_LayerCache]] = []   # forward()'s own buffers (drafters hold private ones)
        self._rows_out: List[int] = []                   # most rows of each buffer ever handed out as a view
        self._packed = {}
        self._weight_epoch = 0                           # part of every derived-weight cache key
        # forwards of rows_min .. 192 rows run on the row-batch weight stream (eagle_gemm_rows_bf16); below, on the
        # per-step weight-stream kernels.  Tests that compare a batch with per-request runs bit for bit set it to 1, so
        # that both sides sum K in the same order.
        self.rows_min = 17
        # the weight-stream GEMM (<= 64 rows) reads eagle_pack_weight_bf16 images of the matrices: whole cache lines per wave
        # instruction instead of 16 x 64-byte halves (+12-17 % on the stream, same output bits); derived copies, rebuilt with
        # the other derived weights, never part of the state_dict.  False: stream the row-major parameters themselves.
        self.pack_decode_weights = True
        self._resident = None                            # (epoch, {data_ptr}) of the matrices to keep cache-resident
        self._resident_args = None
        self._rope = None
        self._err = None

    # ------------------------------------------------------------------ tree hooks
    def init_tree(self, tree=mc_sim_7b_63, expand: str = "wchild"):
        self.tree = tree
        self.tree_buffers = generate_tree_buffers(self.tree, self.embed_tokens.weight.device, expand=expand)
        self.tree_buffer = self.tree_buffers.as_dict()

    def reset(self):
        self.tree_mask = None

    # ------------------------------------------------------------------ helpers
    def _dims(self):
        a = self.layers[0].self_attn
        return a.num_heads, a.num_key_value_heads, a.head_dim

    def _rope_tables(self, n_pos: int, device):
        if self._rope is None or self._rope[0].shape[0] < n_pos or self._rope[0].device != device:
            n = max(n_pos, getattr(self.config, "max_position_embeddings", 2048))
            base = float(getattr(self.config, "rope_theta", 10000.0) or 10000.0)
            self._rope = ops.rope_tables(self._dims()[2], n, base=base, device=device)
        return self._rope

    def _wkey(self, srcs):
        return (self._weight_epoch,) + tuple((w.data_ptr(), w._version) for w in srcs)

    def invalidate_weight_caches(self):
        """Drop every tensor derived from the parameters (stacked q|k|v, norm-folded gate/up, stacked gate|up, the
        training path's transposed copies, the cache-residency choice).  In-place updates through torch bump
        ``Tensor._version`` and are noticed without this; kernels that write parameters through raw pointers
        (EagleTrainer's fused AdamW) and state loaders call it."""
        self._weight_epoch += 1
        self._packed.clear()
        self.__dict__.pop("_tcache", None)
        self._resident = None

    def _packed_weights(self, li: int):
        """q|k|v stacked into one [ (H+2HK)*hd, D ] weight so the projection is one weight pass.
        Rebuilt when any source parameter changes (in-place update or reassignment)."""
        a = self.layers[li].self_attn
        srcs = (a.q_proj.weight, a.k_proj.weight, a.v_proj.weight)
        key = self._wkey(srcs)
        hit = self._packed.get(li)
        if hit is None or hit[0] != key:
            hit = (key, torch.cat([w.detach() for w in srcs], dim=0).contiguous())
            self._packed[li] = hit
        return hit[1]

    def _folded_mlp(self, li: int):
        """gate/up weights with post_attention_layernorm.weight folded into their columns
        (W' = W * w[None, :]); rebuilt when a source parameter changes."""
        layer = self.layers[li]
        srcs = (layer.mlp.gate_proj.weight, layer.mlp.up_proj.weight, layer.post_attention_layernorm.weight)
        key = self._wkey(srcs)
        hit = self._packed.get(("mlp", li))
        if hit is None or hit[0] != key:
            w = srcs[2].detach()[None, :]
            hit = (key, (srcs[0].detach() * w).contiguous(), (srcs[1].detach() * w).contiguous())
            self._packed[("mlp", li)] = hit
        return hit[1], hit[2]

    def _stacked_mlp(self, li: int):
        """gate|up stacked into one [2I, D] weight (wide-batch path); rebuilt when a source changes."""
        m = self.layers[li].mlp
        srcs = (m.gate_proj.weight, m.up_proj.weight)
        key = self._wkey(srcs)
        hit = self._packed.get(("gu", li))
        if hit is None or hit[0] != key:
            hit = (key, torch.cat([w.detach() for w in srcs], dim=0).contiguous())
            self._packed[("gu", li)] = hit
        return hit[1]

    def _pk(self, tag, w: torch.Tensor):
        """The packed image of ``w`` for the weight-stream GEMM (cached per weight version under ``tag``), or ``w`` itself."""
        if not self.pack_decode_weights or w.dim() != 2 or w.shape[1] % 32 or w.stride(1) != 1 or w.dtype != torch.bfloat16:
            return w
        key = (self._weight_epoch, w.data_ptr(), w._version, tuple(w.shape))
        hit = self._packed.get(("pk", tag))
        if hit is None or hit[0] != key:
            hit = (key, ops.pack_weight(w.detach()))
            self._packed[("pk", tag)] = hit
        return hit[1]

    def set_cache_residency(self, lm_head: Optional[torch.Tensor] = None, budget_mb: float = 192.0):
        """Choose which of this model's matrices the weight-stream GEMM keeps in the Infinity Cache between the draft
        forwards of a cycle: the smallest ones up to ``budget_mb`` (default 192 MB = 3/4 of the 256 MB cache).  The
        choice lives in the model (the library keeps no state) and reaches the kernel as a per-call flag
        (EAGLE_GEMM_KEEP_RESIDENT); it is recomputed when the derived weights are rebuilt.  Speed only."""
        self._resident_args = (lm_head, float(budget_mb))
        self._resident = None

    def _keep_resident(self, w) -> bool:
        """``w``: the object handed to the GEMM (a parameter, a derived tensor or its packed image)."""
        if self._resident_args is None:
            return False
        if self._resident is None or self._resident[0] != self._weight_epoch:
            lm_head, budget_mb = self._resident_args
            cands = [self._pk("fc", self.fc.weight)]
            for li, layer in enumerate(self.layers):
                wg, wu = self._folded_mlp(li)
                cands += [self._pk(("qkv", li), self._packed_weights(li)), self._pk(("o", li), layer.self_attn.o_proj.weight),
                          (self._pk(("wg", li), wg), self._pk(("wu", li), wu)), self._pk(("down", li), layer.mlp.down_proj.weight)]
            if lm_head is not None:
                cands.append(lm_head)
            picked = ops.pick_resident_weights(cands, int(budget_mb * 1e6))
            self._resident = (self._weight_epoch, {t.data_ptr() for t in picked})
        return w.data_ptr() in self._resident[1]

    def new_cache(self, B: int, cap: int, device) -> "_LayerCache":
        """A private KV buffer (the drafters own theirs: nothing forward() does can overwrite them)."""
        _, HK, hd = self._dims()
        return _LayerCache(B, HK, (cap + 255) // 256 * 256, hd, device)

    def _cache_for(self, li: int, B: int, need: int, device, past) -> _LayerCache:
        """The buffer this call appends to.  A ``past`` that is a stamped view of the model's current buffer for the
        layer continues in place (raises if a later call overwrote rows it holds); any other past (a view of an
        older buffer, a clone, a foreign tensor) is copied into a fresh buffer, and a call without past starts a
        fresh buffer too, so views handed out earlier keep their contents like the reference's torch.cat results."""
        while len(self._caches) <= li:
            self._caches.append(None)
            self._rows_out.append(0)
        c = self._caches[li]
        past_len = 0 if past is None else past[0].shape[2]
        stamp_k = getattr(past[0], "_eagle_stamp", None) if past is not None else None
        stamp_v = getattr(past[1], "_eagle_stamp", None) if past is not None else None
        ours = (c is not None and stamp_k is not None and stamp_k[0] is c and stamp_v is not None and stamp_v[0] is c
                and past[0].shape[0] == B and c.k.device == device)
        if ours:
            if c.is_stale(stamp_k) or c.is_stale(stamp_v):
                raise ValueError(
                    "past_key_values is a view of the model's KV buffer whose rows a later forward() overwrote "
                    "(a call restarted that buffer from fewer rows); clone() the tensors of a branch that must "
                    "outlive such a call")
            if c.cap >= need:
                c.restart(past_len, self._rows_out[li])
                self._rows_out[li] = min(self._rows_out[li], past_len)
                return c
        if past is None and c is not None and self._rows_out[li] == 0 and c.k.shape[0] == B and c.cap >= need and c.k.device == device:
            # a fresh prompt while no view of the buffer is live (nothing was ever handed out, e.g. use_cache=False loops
            # over validation batches): restart it in place -- no allocation, no memset
            return c
        cap = max(need, getattr(self.config, "max_position_embeddings", 2048))
        nc = self.new_cache(B, cap, device)
        if past is not None:
            nc.k[:, :, :past_len].copy_(past[0])
            nc.v[:, :, :past_len].copy_(past[1])
        self._caches[li] = nc
        self._rows_out[li] = 0
        return nc

    # ------------------------------------------------------------------ kernel chain
    def _attn_workspace(self, B, H, max_kv, dev):
        need = ops.query("eagle_attn_decode_workspace_bytes", B, H, max_kv)
        ws = getattr(self, "_attn_ws", None)
        if ws is None or ws.numel() < need or ws.device != dev:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            self._attn_ws = ws
        return ws

    def _forward_rows(self, emb_idx, hid, hid_idx, M, B, q_len, pos, past_dev, max_kv, tmask, pad, caches,
                      past_host=None, past_off0=0, pos_off=0, pos_stride=1, emb_ind=None, emb_bstride=0):
        """One draft-head forward over M = B*q_len rows, all in HIP kernels.

        emb_idx [M] int64 rows of embed_tokens; hid [*, D] with optional row gather hid_idx [M];
        pos [M] int64 or None (then position = past + pos_off + m*pos_stride); past_dev [B] int32 device lengths (+ static offset past_off0 + chunk offset);
        max_kv: static upper bound on past+q_len when lengths live on the device only
        (``past_host`` None), else derived from the host length."""
        dev = hid.device
        H, HK, hd = self._dims()
        cos, sin = self._rope_tables(self.config.max_position_embeddings if max_kv is None else max_kv, dev)
        if self._err is None or self._err.device != dev:
            self._err = torch.zeros(1, dtype=torch.int32, device=dev)
        emb_w = self.embed_tokens.weight
        if emb_w.dtype != torch.bfloat16:
            raise TypeError("embed_tokens must be bf16 (cast the model with .to(torch.bfloat16))")
        # decode batches (a batch of requests stepping through the tree together: 17..192 rows): every projection is one
        # weight pass of the row-batch weight stream (W to registers, x through LDS, K split where the matrix is small)
        mid = self.rows_min <= M <= ops.ROWS_MAX and self.config.hidden_size % 128 == 0 and self.config.intermediate_size % 128 == 0 \
            and hid.dim() == 2 and hid.stride(1) == 1
        if emb_ind is not None and M > 64 and not mid:
            # wide batches off the row-batch stream: resolve flat_topk[tree_indices] into a plain id array first (the fused
            # double indirection of the weight-stream GEMM covers one 64-row pass; the row-batch stream's loader takes it at
            # any row count, so a 128-row tree step needs no such launch)
            emb_idx = ops.tree_gather_ids(emb_idx, emb_ind, B=B, batch_stride=emb_bstride).reshape(-1)
            emb_ind, emb_bstride = None, 0
        if mid:
            return self._forward_rows_mid(emb_idx, hid, hid_idx, M, B, q_len, pos, past_dev, max_kv, tmask, pad, caches,
                                          past_host, past_off0, pos_off, pos_stride, emb_ind, emb_bstride, cos, sin)
        # prompts and other wide row blocks without row gathers take the LDS-tiled MFMA GEMM for every
        # projection: one weight pass instead of one per 64 rows of the weight-stream kernel
        wide_rest = M > 64 and x_width_ok(self, hid)                  # qkv / o / gate|up / down: no row gathers involved
        wide = wide_rest and hid_idx is None and emb_ind is None      # fc too
        # fc(cat(embed_tokens(ids), hidden)) -- eagle/llama2.py:515, :537-542
        if wide:
            x = _gemm_wide(ops.embed_concat(emb_w, emb_idx, hid), self.fc.weight, bias=self.fc.bias)
        else:
            wfc = self._pk("fc", self.fc.weight)
            x = ops.linear_skinny(emb_w, wfc, bias=self.fc.bias, idx0=emb_idx, x1=hid, idx1=hid_idx, M=M,
                                  ind0=emb_ind, idx0_bstride=emb_bstride, resident=self._keep_resident(wfc))
        for li, layer in enumerate(self.layers):
            a = layer.self_attn
            cache = caches[li]
            xin = x if layer.index == 0 else ops.rmsnorm(x, layer.input_layernorm.weight, layer.input_layernorm.variance_epsilon)
            wqkv = self._packed_weights(li)
            if wide_rest:
                qkv = _gemm_wide(xin, wqkv)
            else:
                wq = self._pk(("qkv", li), wqkv)
                qkv = ops.linear_skinny(xin, wq, resident=self._keep_resident(wq))
            attn = torch.empty((M, H * hd), dtype=torch.bfloat16, device=dev)
            ws = self._attn_workspace(B, H, (past_host + q_len) if past_host is not None else max_kv, dev)
            for r0 in range(0, q_len, 16):
                mq = min(16, q_len - r0)
                bound = (past_host + r0 + mq) if past_host is not None else max_kv
                ops.tree_attention(qkv, pos, cos, sin, cache.k, cache.v, past_dev, past_off0 + r0, tmask, pad, attn, ws,
                                   self._err, B, mq, H, HK, hd, bound, True, q_len, r0, pos_off, pos_stride)
            if wide_rest:
                x = _gemm_wide(attn, a.o_proj.weight, res=x)
            else:
                wo = self._pk(("o", li), a.o_proj.weight)
                x = ops.linear_skinny(attn, wo, res=x, resident=self._keep_resident(wo))
            # post_attention_layernorm fused into the gate/up projection: the norm weight is folded
            # into the weight columns, the row scale rsqrt(mean(x^2)+eps) is computed in-kernel
            if M > 64 and x.shape[1] % 64 == 0:
                # wide batches (BASELINE configs[3], batch 32): one pass of gate|up on the LDS-tiled MFMA GEMM
                # instead of one weight-stream pass per 32 rows
                n = ops.rmsnorm(x, layer.post_attention_layernorm.weight, layer.post_attention_layernorm.variance_epsilon)
                wgu = self._stacked_mlp(li)
                act = ops.silu_mul(_gemm_wide(n, wgu))
            else:
                wg, wu = self._folded_mlp(li)
                wg, wu = self._pk(("wg", li), wg), self._pk(("wu", li), wu)
                act = ops.linear_skinny(x, wg, W2=wu, norm_eps=layer.post_attention_layernorm.variance_epsilon,
                                        resident=self._keep_resident(wg))
            if wide_rest and act.shape[1] % 64 == 0:
                x = _gemm_wide(act, layer.mlp.down_proj.weight, res=x)
            else:
                wd = self._pk(("down", li), layer.mlp.down_proj.weight)
                x = ops.linear_skinny(act, wd, res=x, resident=self._keep_resident(wd))
        return x

    def _forward_rows_mid(self, emb_idx, hid, hid_idx, M, B, q_len, pos, past_dev, max_kv, tmask, pad, caches,
                          past_host, past_off0, pos_off, pos_stride, emb_ind, emb_bstride, cos, sin):
        """_forward_rows for 17..192 rows: the same chain on eagle_gemm_rows_bf16 (row gathers, the cat of fc, the residual
        adds and SiLU*up all in the GEMM's loader / epilogue; the RMSNorms as their own small kernels)."""
        dev = hid.device
        H, HK, hd = self._dims()
        emb_w = self.embed_tokens.weight
        x = ops.linear_rows(emb_w, self._pk("fc", self.fc.weight), bias=self.fc.bias, idx0=emb_idx, x1=hid, idx1=hid_idx, M=M,
                            ind0=emb_ind, idx0_bstride=emb_bstride)
        for li, layer in enumerate(self.layers):
            a = layer.self_attn
            cache = caches[li]
            xin = x if layer.index == 0 else ops.rmsnorm(x, layer.input_layernorm.weight, layer.input_layernorm.variance_epsilon)
            wqkv = self._packed_weights(li)
            if M <= 32 and wqkv.shape[1] == 4096 and wqkv.shape[0] >= 8192:
                qkv = ops.linear_skinny(xin, self._pk(("qkv", li), wqkv))   # <= 32 rows, tall matrix: x in registers (keepx, two row tiles)
            else:
                qkv = ops.linear_rows(xin, self._pk(("qkv", li), wqkv))
            attn = torch.empty((M, H * hd), dtype=torch.bfloat16, device=dev)
            ws = self._attn_workspace(B, H, (past_host + q_len) if past_host is not None else max_kv, dev)
            for r0 in range(0, q_len, 16):
                mq = min(16, q_len - r0)
                bound = (past_host + r0 + mq) if past_host is not None else max_kv
                ops.tree_attention(qkv, pos, cos, sin, cache.k, cache.v, past_dev, past_off0 + r0, tmask, pad, attn, ws,
                                   self._err, B, mq, H, HK, hd, bound, True, q_len, r0, pos_off, pos_stride)
            x = ops.linear_rows(attn, self._pk(("o", li), a.o_proj.weight), res=x)
            n = ops.rmsnorm(x, layer.post_attention_layernorm.weight, layer.post_attention_layernorm.variance_epsilon)
            wgu = self._stacked_mlp(li)
            inter = wgu.shape[0] // 2
            if M <= 32 and wgu.shape[1] == 4096:
                # <= 32 rows: gate|up as ONE tall matrix on the keepx stream, then SiLU*up (50 us; the SiLU epilogue on two
                # row tiles and two weight tiles per step spills: 74)
                act = ops.silu_mul(ops.linear_skinny(n, self._pk(("gu", li), wgu)))
            else:
                act = ops.linear_rows(n, self._pk(("g", li), wgu[:inter]), W2=self._pk(("u", li), wgu[inter:]))
            x = ops.linear_rows(act, self._pk(("down", li), layer.mlp.down_proj.weight), res=x)
        return x

    # ------------------------------------------------------------------ forward
    def forward(
        self,
        hidden_states,
        input_ids,
        attention_mask: Optional[torch.Tensor] = None,
        position_ids: Optional[torch.LongTensor] = None,
        past_key_values: Optional[List[torch.FloatTensor]] = None,
        inputs_embeds: Optional[torch.FloatTensor] = None,
        use_cache: Optional[bool] = None,
        output_attentions: Optional[bool] = None,
        output_hidden_states: Optional[bool] = None,
        return_dict: Optional[bool] = None,
        std=None,
    ):
        if torch.is_grad_enabled() and self.training:
            from .train import draft_forward_train  # autograd path (MFMA GEMMs + flash attention)
            if past_key_values is not None or use_cache:
                raise ValueError("training forward does not take a KV cache (eagle/train.py:105)")
            return draft_forward_train(self, hidden_states, input_ids, attention_mask, position_ids)
        if output_attentions:
            raise NotImplementedError("attention weights are never materialised by the fused kernel")
        ops.require_device(hidden_states, "Llama2Model.forward")
        if hidden_states.dtype != torch.bfloat16:
            raise TypeError("eagle_amd computes in bf16 only (cast the model and inputs to torch.bfloat16)")
        B, q_len, D = hidden_states.shape
        dev = hidden_states.device
        H, HK, hd = self._dims()
        past_len = 0 if past_key_values is None else past_key_values[0][0].shape[2]
        total = past_len + q_len
        if position_ids is None:
            pos = torch.arange(past_len, total, dtype=torch.long, device=dev).repeat(B)
        else:
            pos = position_ids.view(-1, q_len).long()
            pos = (pos.expand(B, q_len) if pos.shape[0] == 1 else pos).reshape(-1).contiguous()
            if pos.numel() != B * q_len:
                raise ValueError(f"position_ids has {pos.numel()} entries for {B}x{q_len} rows")
        pad = None
        if attention_mask is not None:
            if tuple(attention_mask.shape) != (B, total):
                raise ValueError(f"Attention mask should be of size {(B, total)}, but is {tuple(attention_mask.shape)}")
            pad = attention_mask.to(torch.uint8).contiguous()
        tmask = None
        if getattr(self, "tree_mask", None) is not None:
            tm = self.tree_mask
            if q_len > 16:
                raise ValueError("tree_mask steps are limited to 16 query rows")
            if tm.dim() != 4 or tm.shape[2] != q_len or tm.shape[0] not in (1, B) or tm.shape[-1] > total:
                raise IndexError(f"tree_mask {tuple(tm.shape)} does not match [B,1,{q_len},T<= {total}]")
            if tm.shape[0] == 1 and B > 1:
                raise IndexError("tree_mask must carry the batch dimension, as the reference's boolean index needs")
            tmask = (tm[:, 0] != 0).to(torch.uint8).contiguous()
            tmask = tmask[0] if B == 1 else tmask
        self._rope_tables(total, dev)
        ids = input_ids.reshape(-1)
        if ids.dtype != torch.int64:
            ids = ids.long()
        ids = ids.contiguous()
        hs = hidden_states.reshape(B * q_len, D)
        if not hs.is_contiguous():
            hs = hs.contiguous()
        M = B * q_len
        past_dev = torch.full((B,), past_len, dtype=torch.int32, device=dev)
        caches = []
        for li in range(len(self.layers)):
            past = None if past_key_values is None else past_key_values[li]
            caches.append(self._cache_for(li, B, total, dev, past))
        x = self._forward_rows(ids, hs, None, M, B, q_len, pos, past_dev, None, tmask, pad, caches, past_len)
        next_cache = None
        if use_cache:
            next_cache = tuple(c.views(total) for c in caches)
            for li in range(len(caches)):
                self._rows_out[li] = max(self._rows_out[li], total)
        out = x.view(B, q_len, D)
        if use_cache:
            return out, next_cache
        return out


def _gemm_wide(x: torch.Tensor, W: torch.Tensor, bias=None, res=None) -> torch.Tensor:
    """x[M,K] . W[N,K]^T on the 256x256 MFMA GEMM for 64 < M rows: one weight pass.  With few output tiles
    (M <= 256 rows against N = 4096 is 16 tiles for 256 CUs) K is split so that about one round of
    workgroups streams the weights, at the price of fp32 partials (ksplit * M * N * 8 bytes of traffic)."""
    M, K = x.shape
    tiles = ((M + 255) // 256) * ((W.shape[0] + 255) // 256)
    ksplit = max(1, min(8, 256 // tiles, K // 256))
    return ops.gemm_nt(x, W, bias=bias, res=res, ksplit=ksplit)


def x_width_ok(model, hid: torch.Tensor) -> bool:
    """The MFMA GEMM wants K % 64 == 0 for every projection of the layer and contiguous hidden rows."""
    D = model.config.hidden_size
    return D % 64 == 0 and hid.dim() == 2 and hid.stride(1) == 1 and hid.shape[1] == D


def _load_verifier_tensor(path, name: str) -> torch.Tensor:
    """Read one tensor of the verifier checkpoint by its weight-map entry (contract of
    eagle/llama2.py:415-434 for the embedding and eagle/train.py:215-228 for lm_head)."""
    from safetensors import safe_open

    with open(os.path.join(path, "model.safetensors.index.json"), "r") as f:
        shard = json.loads(f.read())["weight_map"][name]
    with safe_open(os.path.join(path, shard), framework="pt", device="cpu") as f:
        return f.get_tensor(name)
