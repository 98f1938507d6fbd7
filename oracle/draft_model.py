"""Oracle (TEST INFRASTRUCTURE ONLY): CPU restatement of the EAGLE-1 draft head forward.

Restates, op for op and in the same dtype order, what ``eagle/llama2.py`` computes so that on
CPU the two agree bit-for-bit (checked against ``tests/golden/`` fixtures that were generated
from the reference import).  Written functionally over a plain ``dict`` of weights that uses
the reference's ``state_dict`` keys (eagle/llama2.py:404-451):

    embed_tokens.weight [V,D]            layers.0.self_attn.{q,k,v,o}_proj.weight [D,D]
    layers.0.mlp.{gate,up}_proj.weight [I,D]   layers.0.mlp.down_proj.weight [D,I]
    layers.0.post_attention_layernorm.weight [D]   fc.weight [D,2D]   fc.bias [D]

Scope notes: rope_scaling variants (eagle/llama2.py:105-147) and
pretraining_tp>1 slicing (:204-219) are dead with the shipped configs and are not restated.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

F32_MIN = torch.finfo(torch.float32).min


@dataclass
class DraftDims:
    hidden: int = 4096
    inter: int = 11008
    heads: int = 32
    kv_heads: int = 32
    vocab: int = 32000
    max_pos: int = 2048
    eps: float = 1e-6
    rope_base: float = 10000.0

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    @staticmethod
    def from_json(path: str) -> "DraftDims":
        import json

        with open(path) as f:
            c = json.load(f)
        return DraftDims(
            hidden=c["hidden_size"],
            inter=c["intermediate_size"],
            heads=c["num_attention_heads"],
            kv_heads=c.get("num_key_value_heads", c["num_attention_heads"]),
            vocab=c.get("vocab_size", 32000),  # LlamaConfig default
            max_pos=c.get("max_position_embeddings", 2048),  # LlamaConfig default
            eps=c.get("rms_norm_eps", 1e-6),
        )


def init_weights(dims: DraftDims, seed: int = 0, dtype=torch.float32, bias: bool = True,
                 std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights, N(0, std) (norm weight = 1 + N(0, std)).  Shared recipe for
    oracle, GPU tests and bench so that every side sees identical tensors."""
    g = torch.Generator().manual_seed(seed)
    D, I, V = dims.hidden, dims.inter, dims.vocab

    def n(*shape):
        return (torch.randn(*shape, generator=g) * std).to(dtype)

    w = {
        "embed_tokens.weight": n(V, D),
        "layers.0.self_attn.q_proj.weight": n(D, D),
        "layers.0.self_attn.k_proj.weight": n(dims.kv_heads * dims.head_dim, D),
        "layers.0.self_attn.v_proj.weight": n(dims.kv_heads * dims.head_dim, D),
        "layers.0.self_attn.o_proj.weight": n(D, D),
        "layers.0.mlp.gate_proj.weight": n(I, D),
        "layers.0.mlp.up_proj.weight": n(I, D),
        "layers.0.mlp.down_proj.weight": n(D, I),
        "layers.0.post_attention_layernorm.weight": (1.0 + torch.randn(D, generator=g) * std).to(dtype),
        "fc.weight": n(D, 2 * D),
    }
    if bias:
        w["fc.bias"] = n(D)
    return w


# --------------------------------------------------------------------------------------
# RoPE tables: eagle/llama2.py:70-93 (inv_freq over arange(0,hd,2)/hd, base 10000; the
# half-split "cat(freqs, freqs)" layout; cos/sin cached in fp32 and cast on use, :100-102).
# --------------------------------------------------------------------------------------
def rope_tables(head_dim: int, n_pos: int, base: float = 10000.0) -> Tuple[torch.Tensor, torch.Tensor]:
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(n_pos, dtype=inv_freq.dtype)
    fr = torch.einsum("i,j->ij", t, inv_freq)
    ang = torch.cat((fr, fr), dim=-1)
    return ang.cos(), ang.sin()


def _rot_half(x: torch.Tensor) -> torch.Tensor:
    # eagle/llama2.py:56-60
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(q, k, cos, sin, position_ids):
    # eagle/llama2.py:61-69 ; cos/sin are [n_pos, hd] already in q's dtype
    c = cos[position_ids].unsqueeze(1)
    s = sin[position_ids].unsqueeze(1)
    return (q * c) + (_rot_half(q) * s), (k * c) + (_rot_half(k) * s)


# --------------------------------------------------------------------------------------
# Additive fp32 mask [B,1,q,kv]: eagle/llama2.py:14-43 and :460-491
# --------------------------------------------------------------------------------------
def build_mask(attn_mask_2d: torch.Tensor, q_len: int, past_len: int,
               tree_mask: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    B = attn_mask_2d.shape[0]
    kv = past_len + q_len
    combined = None
    if q_len > 1:  # causal part only when more than one query row (:464)
        tri = torch.full((q_len, q_len), F32_MIN)
        idx = torch.arange(q_len)
        tri.masked_fill_(idx < (idx + 1).view(q_len, 1), 0)
        if past_len > 0:
            tri = torch.cat([torch.zeros(q_len, past_len), tri], dim=-1)
        combined = tri[None, None].expand(B, 1, q_len, kv)
    # padding part (:473-480, _expand_mask :32-43)
    exp = attn_mask_2d[:, None, None, :].expand(B, 1, q_len, kv).to(torch.float32)
    inv = 1.0 - exp
    pad = inv.masked_fill(inv.to(torch.bool), F32_MIN)
    combined = pad if combined is None else pad + combined
    if tree_mask is not None:  # overlay on the last T x T corner (:483-488)
        T = tree_mask.size(-1)
        combined = combined.clone()
        combined[:, :, -T:, -T:][tree_mask == 0] = F32_MIN
    return combined


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    # eagle/llama2.py:330-335
    dt = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return w * xf.to(dt)


def attention_core(q, k, v, dims: DraftDims, mask, position_ids, past, use_cache, cos, sin):
    """eagle/llama2.py:226-273 from the projected (pre-RoPE) q [B,H,q,hd], k / v [B,HK,q,hd] on: RoPE at
    ``position_ids``, cache concat, repeat_kv, scores / sqrt(hd) + mask, fp32 softmax cast back to the query dtype,
    P.V.  Returns (o [B,q,H*hd] before o_proj, present)."""
    B, H, q_len, hd = q.shape
    HK = k.shape[1]
    q, k = apply_rope(q, k, cos.to(v.dtype), sin.to(v.dtype), position_ids)
    if past is not None:
        k = torch.cat([past[0], k], dim=2)
        v = torch.cat([past[1], v], dim=2)
    present = (k, v) if use_cache else None
    if HK != H:  # repeat_kv (:45-54)
        rep = H // HK
        k = k[:, :, None].expand(B, HK, rep, k.shape[2], hd).reshape(B, H, k.shape[2], hd)
        v = v[:, :, None].expand(B, HK, rep, v.shape[2], hd).reshape(B, H, v.shape[2], hd)
    s = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(hd)
    if mask is not None:
        s = s + mask
    pr = F.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(pr, v).transpose(1, 2).contiguous().reshape(B, q_len, H * hd)
    return o, present


def attention(x, w, dims: DraftDims, mask, position_ids, past, use_cache, cos, sin):
    # eagle/llama2.py:193-285
    B, q_len, _ = x.shape
    H, HK, hd = dims.heads, dims.kv_heads, dims.head_dim
    p = "layers.0.self_attn."
    q = F.linear(x, w[p + "q_proj.weight"]).view(B, q_len, H, hd).transpose(1, 2)
    k = F.linear(x, w[p + "k_proj.weight"]).view(B, q_len, HK, hd).transpose(1, 2)
    v = F.linear(x, w[p + "v_proj.weight"]).view(B, q_len, HK, hd).transpose(1, 2)
    o, present = attention_core(q, k, v, dims, mask, position_ids, past, use_cache, cos, sin)
    return F.linear(o, w[p + "o_proj.weight"]), present


def mlp(x, w):
    # eagle/llama2.py:317 (silu, the shipped hidden_act)
    p = "layers.0.mlp."
    return F.linear(F.silu(F.linear(x, w[p + "gate_proj.weight"])) * F.linear(x, w[p + "up_proj.weight"]),
                    w[p + "down_proj.weight"])


def forward(w: Dict[str, torch.Tensor], dims: DraftDims, hidden_states: torch.Tensor,
            input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
            position_ids: Optional[torch.Tensor] = None, past_key_values=None,
            use_cache: bool = False, tree_mask: Optional[torch.Tensor] = None,
            rope_len: Optional[int] = None):
    """Llama2Model.forward, eagle/llama2.py:493-587 (eval path, one decoder layer without
    input_layernorm, :344-345/:373-374).  Returns h or (h, ((K,V),))."""
    B, q_len, _ = hidden_states.shape
    emb = F.embedding(input_ids, w["embed_tokens.weight"])
    past_len = 0 if past_key_values is None else past_key_values[0][0].shape[2]
    if position_ids is None:
        position_ids = torch.arange(past_len, past_len + q_len, dtype=torch.long).unsqueeze(0).view(-1, q_len)
    else:
        position_ids = position_ids.view(-1, q_len).long()
    if attention_mask is None:
        attention_mask = torch.ones((B, past_len + q_len), dtype=torch.bool)
    mask = build_mask(attention_mask, q_len, past_len, tree_mask)
    emb = emb.to(hidden_states.dtype)
    x = F.linear(torch.cat((emb, hidden_states), dim=-1), w["fc.weight"], w.get("fc.bias"))
    n_pos = max(dims.max_pos, past_len + q_len) if rope_len is None else rope_len
    cos, sin = rope_tables(dims.head_dim, n_pos, dims.rope_base)
    past = None if past_key_values is None else past_key_values[0]
    a, present = attention(x, w, dims, mask, position_ids, past, use_cache, cos, sin)
    x = x + a
    x = x + mlp(rmsnorm(x, w["layers.0.post_attention_layernorm.weight"], dims.eps), w)
    if use_cache:
        return x, (present,)
    return x
