"""Draft-head TRAINING on MI355X: forward, fused loss, hand-written backward, AdamW and the
data-parallel / vocab-sharded drivers (eagle/train.py:79-131, eagle/train_tp.py:60-112).

Two ways in, both ending in the same HIP kernel chains (include/eagle_hip.h):

* drop-in autograd path -- ``Llama2Model.forward`` in training mode returns a tensor with a
  ``grad_fn`` (``DraftForwardFunction``), and ``eagle_loss`` is a differentiable scalar
  (``FusedEagleLossFunction``), so the reference's loop ``predict = model(h, input_ids=ids);
  loss = ...; loss.backward()`` (eagle/train.py:105-126) runs unchanged on the kernels;
* ``EagleTrainer`` -- the same chains without autograd: parameters, gradients and AdamW moments
  live in flat buffers (one RCCL all-reduce view per gradient bucket, launched as soon as the
  last micro-batch's weight gradients of that bucket exist, overlapping the rest of the backward).

No [B,H,L,L] score tensor, no softmax / log-softmax tensors and no fp32 mask are ever
materialised; the only [N,V] tensors are the two bf16 logit buffers, and dlogits overwrites one.

Backward GEMMs run on the 256x256 MFMA kernel's K-major operand forms: the data gradients read the weights as the
state_dict stores them (eagle_gemm_nn_bf16), the weight gradients read dY and X as they lie in memory
(eagle_gemm_tn_bf16) -- no transposed weights or activations exist (only the frozen lm_head keeps a transposed copy, built
once); shapes outside those kernels' range fall back to eagle_transpose_bf16 + the NT kernel.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops

BF16 = torch.bfloat16


# ---------------------------------------------------------------------------------------------
# weights
# ---------------------------------------------------------------------------------------------
@dataclass
class LayerWeights:
    """The trainable tensors of the draft head in the layout the kernels consume."""
    fc_w: torch.Tensor                 # [D, 2D]
    fc_b: Optional[torch.Tensor]       # [D]
    wqkv: torch.Tensor                 # [(H+2HK)*hd, D]  q|k|v stacked
    wo: torch.Tensor                   # [D, D]
    wn: torch.Tensor                   # [D] post_attention_layernorm
    wgu: torch.Tensor                  # [2I, D] gate|up stacked
    wdown: torch.Tensor                # [D, I]
    eps: float


PARAM_ORDER = ("layers.0.mlp.down_proj.weight", "layers.0.mlp.gate_proj.weight", "layers.0.mlp.up_proj.weight",
               "layers.0.post_attention_layernorm.weight", "layers.0.self_attn.o_proj.weight",
               "layers.0.self_attn.q_proj.weight", "layers.0.self_attn.k_proj.weight", "layers.0.self_attn.v_proj.weight",
               "fc.weight", "fc.bias")
"""Flat-buffer order = the order in which the backward produces the weight gradients, so that
gradient buckets become ready (and can be all-reduced) front to back; q|k|v and gate|up are
adjacent so that their stacked views cost nothing."""


def _adjacent(ts: Sequence[torch.Tensor]) -> bool:
    """Back to back in ONE storage (flatten_parameters lays q|k|v and gate|up out that way)."""
    base = ts[0].untyped_storage().data_ptr()
    return all(t.is_contiguous() and t.untyped_storage().data_ptr() == base for t in ts) and \
        all(a.data_ptr() + a.numel() * a.element_size() == b.data_ptr() for a, b in zip(ts, ts[1:]))


def _stack(ts: Sequence[torch.Tensor], cache: dict, key) -> torch.Tensor:
    """cat(ts) along dim 0; a zero-copy view when the tensors are adjacent in one buffer."""
    if _adjacent(ts):
        return torch.as_strided(ts[0].detach(), (sum(t.shape[0] for t in ts), ts[0].shape[1]), (ts[0].shape[1], 1))
    ver = tuple((t.data_ptr(), t._version) for t in ts)
    hit = cache.get(key)
    if hit is None or hit[0] != ver:
        hit = (ver, torch.cat([t.detach() for t in ts], dim=0).contiguous())
        cache[key] = hit
    return hit[1]


def layer_weights(model) -> LayerWeights:
    if len(model.layers) != 1:
        raise NotImplementedError("the EAGLE-1 draft head has one decoder layer (both shipped configs)")
    layer = model.layers[0]
    a, m = layer.self_attn, layer.mlp
    if a.num_key_value_heads != a.num_heads or a.head_dim != 128:
        raise NotImplementedError("training attention kernel: kv_heads == heads and head_dim == 128 (both shipped configs)")
    cache = model._packed
    for p in (model.fc.weight, a.q_proj.weight, m.down_proj.weight):
        if p.dtype != BF16:                                     # (ops.* reject non-device tensors: no CPU path)
            raise TypeError("training runs in bf16 (eagle/train.py --eagle-dtype bfloat16)")
    return LayerWeights(
        fc_w=model.fc.weight.detach(), fc_b=None if model.fc.bias is None else model.fc.bias.detach(),
        wqkv=_stack((a.q_proj.weight, a.k_proj.weight, a.v_proj.weight), cache, "t_qkv"),
        wo=a.o_proj.weight.detach(), wn=layer.post_attention_layernorm.weight.detach(),
        wgu=_stack((m.gate_proj.weight, m.up_proj.weight), cache, "t_gu"),
        wdown=m.down_proj.weight.detach(), eps=layer.post_attention_layernorm.variance_epsilon)


def all_reduce_(t: torch.Tensor, group, op=None) -> None:
    """dist.all_reduce in place.  Under gloo (CPU tests, and the rehearsal of N ranks on fewer GPUs) a device tensor goes
    through host memory: gloo's own CUDA staging does not return on this image when the device has queued work or two
    ranks share a GPU (tools/dp_rehearsal.py).  RCCL takes the device tensor as it is."""
    import torch.distributed as dist
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.cpu()                       # orders behind everything queued on the current stream
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=op, group=group)


class _TransposeCache:
    """W^T per weight, rebuilt when the weight's storage or version changes."""

    def __init__(self):
        self._c: Dict[str, Tuple[tuple, torch.Tensor]] = {}

    def get(self, key: str, w: torch.Tensor, version=None) -> torch.Tensor:
        ver = (w.data_ptr(), tuple(w.shape), w._version if version is None else version)
        hit = self._c.get(key)
        if hit is None or hit[0] != ver:
            out = hit[1] if hit is not None and hit[1].shape == (w.shape[1], w.shape[0]) else None
            hit = (ver, ops.transpose(w, out=out))
            self._c[key] = hit
        return hit[1]


# ---------------------------------------------------------------------------------------------
# forward / backward kernel chains
# ---------------------------------------------------------------------------------------------
def draft_fwd(W: LayerWeights, emb: torch.Tensor, ids: torch.Tensor, hidden: torch.Tensor, B: int, L: int, H: int,
              hd: int, cos: torch.Tensor, sin: torch.Tensor, position_ids: Optional[torch.Tensor] = None,
              x0: Optional[torch.Tensor] = None):
    """eagle/llama2.py:511-579 for N = B*L rows, causal, no cache.  Returns (predict [N,D], saved).
    x0 [N, 2D]: cat(embed(ids), hidden) already built (the fused input path, data.make_eagle_draft_input)."""
    if x0 is None:
        x0 = ops.embed_concat(emb, ids, hidden)                   # cat(embed(ids), hidden)      :515,:537-542
    x = ops.gemm_nt(x0, W.fc_w, bias=W.fc_b)                      # fc                           :542
    qkv = ops.gemm_nt_rounds(x, W.wqkv)                           # q|k|v projections            :222-224 (1.5 rounds of tiles at Qwen3 dims: the half round split along K)
    ops.rope_qk_(qkv, cos, sin, L, 2 * H, hd, position_ids=position_ids)        # :228-233
    attn, lse = ops.attn_train_fwd(qkv, B, L, H, hd)              # causal softmax(QK^T/sqrt d)V :247-273
    h1 = ops.gemm_nt(attn, W.wo, res=x)                           # o_proj + residual            :280,:385
    n, rstd = ops.rmsnorm(h1, W.wn, W.eps, want_rstd=True)        # post_attention_layernorm     :389
    gu = ops.gemm_nt(n, W.wgu)                                    # gate|up                      :317
    act = ops.silu_mul(gu)
    out = ops.gemm_nt(act, W.wdown, res=h1)                       # down_proj + residual         :317,:391
    return out, (x0, x, qkv, attn, lse, h1, n, rstd, gu, act)


@dataclass
class GradSink:
    """Where the weight gradients go: tensors shaped like LayerWeights' fields.  ``accumulate``
    adds to what is there (gradient accumulation over micro-batches, eagle/train.py:96-101)."""
    fc_w: torch.Tensor
    fc_b: Optional[torch.Tensor]
    wqkv: torch.Tensor
    wo: torch.Tensor
    wn: torch.Tensor                   # fp32 [D]
    wgu: torch.Tensor
    wdown: torch.Tensor
    accumulate: bool = False


def draft_bwd(W: LayerWeights, saved, dout: torch.Tensor, B: int, L: int, H: int, hd: int, cos, sin,
              sink: GradSink, tc: _TransposeCache, position_ids: Optional[torch.Tensor] = None, on_ready=None):
    """Backward of draft_fwd.  ``on_ready(name)`` is called right after the kernels that complete
    the gradient of ``name`` ('wdown', 'wgu', 'wn', 'wo', 'wqkv', 'fc') have been launched."""
    x0, x, qkv, attn, lse, h1, n, rstd, gu, act = saved
    acc = sink.accumulate
    ready = on_ready or (lambda name: None)
    def T(t):                                                     # [rows, C] -> [C, rows padded to the GEMM's K step]
        return ops.transpose(t, pad_to=64)

    def wgrad(dy, xin, out):
        """out[out_f, in_f] (+)= dy[tokens, out_f]^T . xin[tokens, in_f].  Both operands are K-major as they lie in memory: the
        TN kernel (transposing LDS fragment reads) takes them directly; other shapes go through two transposes."""
        if ops.gemm_tn_ok(dy, xin):
            ops.gemm_tn(dy, xin, out=out, accumulate=acc)
        else:
            ops.gemm_nt(T(dy), T(xin), out=out, accumulate=acc)
    def dgrad(dy, key, w, res=None):
        """dy[tokens, out_f] . w[out_f, in_f] (+res): the NN kernel reads the weight as it is stored (K-major B operand); other
        shapes take the cached transposed copy."""
        if ops.gemm_nn_ok(dy, w):
            return ops.gemm_nn(dy, w, res=res)
        return ops.gemm_nt(dy, tc.get(key, w), res=res)
    # down_proj: out = act . Wdown^T + h1
    d_act = dgrad(dout, "wdown", W.wdown)
    wgrad(dout, act, sink.wdown)
    ready("wdown")
    dgu = ops.silu_mul_bwd(gu, d_act)
    del d_act
    dn = dgrad(dgu, "wgu", W.wgu)
    wgrad(dgu, n, sink.wgu)
    ready("wgu")
    del dgu
    dh1, _ = ops.rmsnorm_bwd(dn, h1, W.wn, rstd, res=dout, dw=sink.wn, accumulate_dw=acc)   # + residual branch
    ready("wn")
    d_attn = dgrad(dh1, "wo", W.wo)
    wgrad(dh1, attn, sink.wo)
    ready("wo")
    dqkv = ops.attn_train_bwd(qkv, attn, d_attn, lse, B, L, H, hd)
    ops.rope_qk_(dqkv, cos, sin, L, 2 * H, hd, position_ids=position_ids, inverse=True)
    dx = dgrad(dqkv, "wqkv", W.wqkv, res=dh1)                     # + residual branch of :385
    wgrad(dqkv, x, sink.wqkv)
    ready("wqkv")
    wgrad(dx, x0, sink.fc_w)
    if sink.fc_b is not None:
        ops.colsum(dx, sink.fc_b, accumulate=acc)
    ready("fc")


def loss_fwd_bwd(predict: torch.Tensor, target: torch.Tensor, lm_head: torch.Tensor, lm_head_t: Optional[torch.Tensor],
                 mask: torch.Tensor, v_w: float, p_w: float, scale: float, want_grad: bool = True):
    """eagle/train.py:106-124 at the logits level.  predict/target [N,D] bf16, lm_head [V,D] bf16
    (frozen), mask [N] fp32.  Returns (sums[4] = {sum m*ploss, sum m*vloss, sum m*agree, sum m},
    dpredict [N,D] = d(scale*(v_w*vloss + p_w*ploss))/d predict or None)."""
    N = predict.shape[0]
    if N >= 1024 and N % 256 == 0 and target.shape == predict.shape:
        # one GEMM over the stacked rows [target; predict]: the frozen head is streamed once for both (2 % faster than two
        # launches at 4096 tokens, tools/gemm_stack_bench.py, for two 33 MB row copies); whole rounds of tiles + a split-K tail
        stacked = torch.empty((2 * N, predict.shape[1]), dtype=predict.dtype, device=predict.device)
        stacked[:N].copy_(target)
        stacked[N:].copy_(predict)
        logits = ops.gemm_nt_rounds(stacked, lm_head)
        t_logits, o_logits = logits[:N], logits[N:]
    else:
        t_logits = ops.gemm_nt_rounds(target, lm_head)           # whole rounds of tiles + a split-K tail (ops.gemm_nt_tail_plan)
        o_logits = ops.gemm_nt_rounds(predict, lm_head)
    stats = ops.ce_partial(t_logits, o_logits, 0, predict, target)
    sums, lse_t, lse_o, _ = ops.ce_combine(stats, mask)
    if not want_grad:
        return sums, None
    dlog, dpv = ops.ce_smoothl1_bwd(t_logits, o_logits, lse_t, lse_o, mask, p_w * scale, v_w * scale, predict, target,
                                    in_place=True)
    dpredict = ops.gemm_nt(dlog, lm_head_t, res=dpv, ksplit=lm_dgrad_ksplit(dlog.shape[0], lm_head_t.shape[0], dlog.shape[1]))
    return sums, dpredict


def lm_dgrad_ksplit(M: int, N: int, K: int) -> int:
    """dpredict = dlogits[M,V] . lm_head[V,D] has few 256x256 output tiles (M/256 * D/256) and a very
    long K = V: split K only as far as one round of workgroups needs to cover the 256 CUs (tools/dgrad_ksplit_bench.py, 4096
    tokens = 256 tiles: 1 slice 3,440 us, 2 slices 3,494, 4 slices 3,649 -- every slice more is another fp32 slab to write and sum)."""
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    if tiles >= 256 or K < 16384:
        return 1
    return max(1, min(16, 256 // tiles))


# ---------------------------------------------------------------------------------------------
# autograd drop-ins
# ---------------------------------------------------------------------------------------------
class DraftForwardFunction(torch.autograd.Function):
    """Training forward of Llama2Model with a hand-written backward (replaces the autograd graph
    and the torch.utils.checkpoint recompute of eagle/llama2.py:554-568: nothing is recomputed).
    The trainable parameters ride along as trailing inputs so that autograd routes their gradients."""

    @staticmethod
    def forward(ctx, model, hidden2d, ids, B, L, position_ids, *params):
        W = layer_weights(model)
        H, _, hd = model._dims()
        cos, sin = model._rope_tables(max(L, model.config.max_position_embeddings), hidden2d.device)
        out, saved = draft_fwd(W, model.embed_tokens.weight.detach(), ids, hidden2d, B, L, H, hd, cos, sin, position_ids)
        ctx.model, ctx.W, ctx.saved, ctx.dims, ctx.rope, ctx.pos = model, W, saved, (B, L, H, hd), (cos, sin), position_ids
        ctx.param_ids = [id(p) for p in params]
        return out

    @staticmethod
    def backward(ctx, dout):
        model, W = ctx.model, ctx.W
        B, L, H, hd = ctx.dims
        dev = dout.device
        layer = model.layers[0]
        a, m = layer.self_attn, layer.mlp
        D = W.fc_w.shape[0]
        sink = GradSink(fc_w=torch.empty_like(W.fc_w), fc_b=None if W.fc_b is None else torch.empty_like(W.fc_b),
                        wqkv=torch.empty_like(W.wqkv), wo=torch.empty_like(W.wo),
                        wn=torch.empty(D, dtype=torch.float32, device=dev), wgu=torch.empty_like(W.wgu),
                        wdown=torch.empty_like(W.wdown), accumulate=False)
        tc = model.__dict__.setdefault("_tcache", _TransposeCache())
        draft_bwd(W, ctx.saved, dout.to(BF16).contiguous(), B, L, H, hd, ctx.rope[0], ctx.rope[1], sink, tc, ctx.pos)
        ctx.saved = None
        nq, nk = a.q_proj.weight.shape[0], a.k_proj.weight.shape[0]
        I = m.gate_proj.weight.shape[0]
        grads = {
            id(model.fc.weight): sink.fc_w, id(a.q_proj.weight): sink.wqkv[:nq], id(a.k_proj.weight): sink.wqkv[nq:nq + nk],
            id(a.v_proj.weight): sink.wqkv[nq + nk:], id(a.o_proj.weight): sink.wo,
            id(layer.post_attention_layernorm.weight): sink.wn.to(BF16),
            id(m.gate_proj.weight): sink.wgu[:I], id(m.up_proj.weight): sink.wgu[I:], id(m.down_proj.weight): sink.wdown,
        }
        if model.fc.bias is not None:
            grads[id(model.fc.bias)] = sink.fc_b
        return (None, None, None, None, None, None) + tuple(grads.get(i) for i in ctx.param_ids)


def draft_forward_train(model, hidden_states, input_ids, attention_mask=None, position_ids=None):
    """Llama2Model.forward for ``model.training`` (eagle/train.py:105): [B,L,D] -> [B,L,D] with grad."""
    if attention_mask is not None:
        raise NotImplementedError("the training forward is causal-only: eagle/train.py:105 passes no attention_mask")
    ops.require_device(hidden_states, "Llama2Model.forward")
    if hidden_states.dtype != BF16:
        raise TypeError("eagle_amd computes in bf16 only (cast the model and inputs to torch.bfloat16)")
    B, L, D = hidden_states.shape
    params = [p for p in model.parameters() if p.requires_grad]
    pos = None
    if position_ids is not None:
        pos = position_ids.view(-1, L).long()
        pos = (pos.expand(B, L) if pos.shape[0] == 1 else pos).reshape(-1).contiguous()
    out = DraftForwardFunction.apply(model, hidden_states.reshape(B * L, D).contiguous(),
                                     input_ids.reshape(-1).long().contiguous(), B, L, pos, *params)
    return out.view(B, L, D)


class FusedEagleLossFunction(torch.autograd.Function):
    """loss = v_w * sum(mean_D(mask * smoothL1(predict, target))) + p_w * (-sum(mask * p * log q))
    with p = softmax(lm_head(target)), q = softmax(lm_head(predict))  (eagle/train.py:106-123)."""

    @staticmethod
    def forward(ctx, predict, target, lm_head_w, loss_mask, v_w, p_w, holder):
        D = predict.shape[-1]
        p2, t2 = predict.reshape(-1, D).to(BF16).contiguous(), target.reshape(-1, D).to(BF16).contiguous()
        mask = loss_mask.reshape(-1).to(torch.float32).contiguous()
        tc = holder.setdefault("tc", _TransposeCache())
        sums, dpred = loss_fwd_bwd(p2, t2, lm_head_w, tc.get("lm_head", lm_head_w), mask, v_w, p_w, 1.0,
                                   want_grad=ctx.needs_input_grad[0])
        ctx.dpred, ctx.shape, ctx.dtype = dpred, predict.shape, predict.dtype
        holder["sums"] = sums
        return v_w * sums[1] + p_w * sums[0]

    @staticmethod
    def backward(ctx, g):
        d = ctx.dpred
        ctx.dpred = None
        return (d.float() * g).to(ctx.dtype).view(ctx.shape), None, None, None, None, None, None


def eagle_loss(predict, target, lm_head_weight, loss_mask, v_w: float = 1.0, p_w: float = 0.1, holder: Optional[dict] = None):
    """Differentiable fused loss; ``holder['sums']`` receives {sum m*ploss, sum m*vloss, agreement
    count (eagle/train.py:115-117), masked tokens} as a device tensor."""
    holder = holder if holder is not None else {}
    return FusedEagleLossFunction.apply(predict, target, lm_head_weight, loss_mask, float(v_w), float(p_w), holder)


# ---------------------------------------------------------------------------------------------
# flat-buffer trainer (data parallel)
# ---------------------------------------------------------------------------------------------
def linear_schedule_with_warmup(step: int, num_warmup_steps: int, num_training_steps: int) -> float:
    """LR multiplier of transformers.get_linear_schedule_with_warmup (eagle/train.py:69)."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))


def flatten_parameters(model) -> Tuple[torch.Tensor, Dict[str, Tuple[int, int]]]:
    """Move the trainable parameters into ONE flat bf16 buffer (PARAM_ORDER) and re-point the
    module's parameters at views of it.  Returns (flat, {name: (offset, numel)})."""
    named = dict(model.named_parameters())
    names = [n for n in PARAM_ORDER if n in named and named[n].requires_grad]
    extra = [n for n, p in named.items() if p.requires_grad and n not in names]
    if extra:
        raise ValueError(f"unexpected trainable parameters {extra}")
    offs, total = {}, 0
    for n in names:
        offs[n] = (total, named[n].numel())
        total += (named[n].numel() + 127) // 128 * 128          # 256-byte aligned segments
    dev = named[names[0]].device
    flat = torch.zeros(total, dtype=BF16, device=dev)
    for n in names:
        o, k = offs[n]
        flat[o:o + k].copy_(named[n].detach().reshape(-1))
        named[n].data = flat[o:o + k].view(named[n].shape)
    return flat, offs


class EagleTrainer:
    """Optimizer-step semantics of eagle/train.py:85-131 (``mode='dp'``: per-micro-batch
    clip_grad_value_, DDP mean all-reduce on the last micro-batch) or eagle/train_tp.py:60-112
    (``mode='tp'``: loss / num_items, clip_grad_norm_) on flat buffers.

    ``group``: torch.distributed process group for data parallelism (None = single process).
    AdamW moments are fp32 (torch keeps them in the parameter dtype, bf16, here: fp32 is closer
    to exact arithmetic and still inside the bf16 tolerance)."""

    def __init__(self, model, lm_head_weight: torch.Tensor, lr: float = 2e-4, betas=(0.9, 0.95), eps: float = 1e-8,
                 weight_decay: float = 0.01, v_w: float = 1.0, p_w: float = 0.1, grad_clip: float = 0.5,
                 num_warmup_steps: Optional[int] = None, num_training_steps: Optional[int] = None, mode: str = "dp",
                 group=None, world_size: int = 1, sharded_loss: Optional["ShardedLmHeadLoss"] = None, bucket_names: Sequence[Sequence[str]] = (("wdown",), ("wgu", "wn"), ("wo",), ("wqkv",), ("fc",))):
        if mode not in ("dp", "tp"):
            raise ValueError("mode: 'dp' (eagle/train.py) or 'tp' (eagle/train_tp.py)")
        self.model, self.mode, self.group, self.world = model, mode, group, world_size
        self.sharded_loss = sharded_loss
        if sharded_loss is None:
            self.lm_head = lm_head_weight.contiguous()
            self.lm_head_t = ops.transpose(self.lm_head)         # frozen: transposed once
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.v_w, self.p_w, self.clip = v_w, p_w, grad_clip
        self.warmup, self.total = num_warmup_steps, num_training_steps
        self.flat, self.offs = flatten_parameters(model)
        self.grad = torch.zeros_like(self.flat)
        self.m = torch.zeros(self.flat.numel(), dtype=torch.float32, device=self.flat.device)
        self.v = torch.zeros_like(self.m)
        D = model.config.hidden_size
        self._wn_f32 = torch.zeros(D, dtype=torch.float32, device=self.flat.device)
        self.tc = _TransposeCache()
        self.step_count = 0
        self.sched_step = 0
        self._pversion = 0                                       # bumped by every optimizer step (transposed-weight cache key)
        self._buckets = self._make_buckets(bucket_names)
        self.check_collectives = False                           # tests: drain the device and assert every async all-reduce completed before AdamW
        self.collectives_launched = 0
        self._sums = torch.zeros(4, dtype=torch.float32, device=self.flat.device)

    # ------------------------------------------------------------------ layout helpers
    def _seg(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        o, k = self.offs[name]
        return buf[o:o + k]

    def _grad_sink(self, accumulate: bool) -> GradSink:
        g = self.grad
        named = dict(self.model.named_parameters())
        D, I = self.model.config.hidden_size, self.model.config.intermediate_size

        def span(first, last, rows, cols):
            o0, _ = self.offs[first]
            o1, k1 = self.offs[last]
            assert o1 + k1 - o0 == rows * cols, "stacked parameters must be adjacent without padding"
            return g[o0:o0 + rows * cols].view(rows, cols)
        p = "layers.0."
        nqkv = sum(named[p + f"self_attn.{x}_proj.weight"].shape[0] for x in "qkv")
        return GradSink(
            fc_w=self._seg(g, "fc.weight").view(D, 2 * D), fc_b=self._seg(g, "fc.bias") if "fc.bias" in self.offs else None,
            wqkv=span(p + "self_attn.q_proj.weight", p + "self_attn.v_proj.weight", nqkv, D),
            wo=self._seg(g, p + "self_attn.o_proj.weight").view(D, D), wn=self._wn_f32,
            wgu=span(p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight", 2 * I, D),
            wdown=self._seg(g, p + "mlp.down_proj.weight").view(D, I), accumulate=accumulate)

    _GROUP_PARAMS = {
        "wdown": ("layers.0.mlp.down_proj.weight",), "wgu": ("layers.0.mlp.gate_proj.weight", "layers.0.mlp.up_proj.weight"),
        "wn": ("layers.0.post_attention_layernorm.weight",), "wo": ("layers.0.self_attn.o_proj.weight",),
        "wqkv": ("layers.0.self_attn.q_proj.weight", "layers.0.self_attn.k_proj.weight", "layers.0.self_attn.v_proj.weight"),
        "fc": ("fc.weight", "fc.bias"),
    }

    def _make_buckets(self, bucket_names):
        """[(last gradient group of the bucket, flat slice)] in backward order."""
        out = []
        for groups in bucket_names:
            los, his = [], []
            for gname in groups:
                for pn in self._GROUP_PARAMS[gname]:
                    if pn in self.offs:
                        o, k = self.offs[pn]
                        los.append(o)
                        his.append((o + k + 127) // 128 * 128)
            out.append((groups[-1], min(los), max(his)))
        assert out[0][1] == 0 and all(a[2] == b[1] for a, b in zip(out, out[1:])) and out[-1][2] == self.flat.numel(), \
            "gradient buckets must tile the flat buffer in backward order"
        return out

    # ------------------------------------------------------------------ one optimizer step
    def current_lr(self) -> float:
        if self.warmup is None or self.total is None:
            return self.lr
        return self.lr * linear_schedule_with_warmup(self.sched_step, self.warmup, self.total)

    def step(self, micro_batches: List[dict]) -> dict:
        """micro_batches: dicts with 'hidden_states' [B,L,D] bf16 (verifier hidden + noise),
        'input_ids' [B,L] int64 (already shifted), 'target' [B,L,D] bf16, 'loss_mask' [B,L]
        -- the output of _make_eagle_input (eagle/train.py:256-269).  Returns device-side sums
        (no host sync except the token count, which the reference also syncs: train.py:93)."""
        import torch.distributed as dist
        model = self.model
        H, _, hd = model._dims()
        dev = self.flat.device
        masks = [mb["loss_mask"].reshape(-1).to(torch.float32).contiguous() for mb in micro_batches]
        n_items = torch.stack([m.sum() for m in masks]).sum().reshape(1)
        if self.group is not None:
            all_reduce_(n_items, self.group)                                # train.py:92-93
        num_items = float(n_items.item())
        if num_items == 0:                                                 # train_tp.py:77-79 skips; train.py:124 would divide by zero
            return {"skipped": True, "num_items": 0.0}
        # train.py:124 + accelerate's /GA + DDP's mean: local gradients carry world/num_items, the
        # all-reduce averages.  train_tp.py:105-106: 1/num_items, one process.
        scale = (self.world if self.mode == "dp" else 1.0) / num_items
        W = layer_weights(model)
        self._sums.zero_()
        works = []
        last = len(micro_batches) - 1
        for i, mb in enumerate(micro_batches):
            x0 = mb.get("x0")                                               # fused input path: the fc input is already built
            B, L = mb["loss_mask"].shape
            D = model.config.hidden_size
            cos, sin = model._rope_tables(max(L, model.config.max_position_embeddings), dev)
            if mb["target"].dtype != BF16 or (x0 is None and mb["hidden_states"].dtype != BF16):
                raise TypeError("EagleTrainer.step: hidden_states and target must be bf16")
            tgt = mb["target"].reshape(B * L, D).contiguous()
            if x0 is not None:
                if x0.dtype != BF16 or tuple(x0.shape) != (B * L, 2 * D) or not x0.is_contiguous():
                    raise ValueError("EagleTrainer.step: x0 must be a contiguous bf16 [B*L, 2D] tensor")
                predict, saved = draft_fwd(W, None, None, None, B, L, H, hd, cos, sin, x0=x0)
            else:
                ids = mb["input_ids"].reshape(-1).to(torch.int64).contiguous()
                predict, saved = draft_fwd(W, model.embed_tokens.weight.detach(), ids,
                                           mb["hidden_states"].reshape(B * L, D).contiguous(), B, L, H, hd, cos, sin)
            if self.sharded_loss is not None:
                sums, dpred = self.sharded_loss.fwd_bwd(predict, tgt, masks[i], self.v_w, self.p_w, scale)
            else:
                sums, dpred = loss_fwd_bwd(predict, tgt, self.lm_head, self.lm_head_t, masks[i],
                                           self.v_w, self.p_w, scale)
            self._sums += sums
            sink = self._grad_sink(accumulate=i > 0)
            on_ready = None
            if i == last:
                on_ready = self._bucket_reducer(works)
            draft_bwd(W, saved, dpred, B, L, H, hd, cos, sin, sink, self._tc_versioned(), on_ready=on_ready)
            del saved
            if i != last and self.clip > 0:
                # the reference clips after EVERY micro-batch backward; the last micro-batch's clip comes
                # after the all-reduce and is fused into the AdamW kernel
                if self.mode == "dp":                                       # clip_grad_value_, train.py:127
                    self._wn_f32.clamp_(-self.clip, self.clip)
                    self._fold_wn()
                    ops.clip_value_(self.grad, self.clip)
                else:                                                       # clip_grad_norm_, train_tp.py:109
                    self._fold_wn()
                    coef = ops.grad_clip_coef(self.grad, self.clip)
                    ops.scale_(self.grad, coef)
                    self._wn_f32.mul_(coef[0])
        for w in works:
            w.wait()                                                        # the current stream waits for the collective: AdamW is ordered behind it
        self.collectives_launched = len(works)
        if self.check_collectives and works:
            torch.cuda.current_stream().synchronize()
            if not all(w.is_completed() for w in works):
                raise RuntimeError("a gradient all-reduce had not completed when the optimizer step was reached")
        self._optimizer_step()
        stats = self._sums.clone()
        if self.group is not None:
            all_reduce_(stats, self.group)                                  # train.py:137-148 as ONE reduce
        return {"sums": stats, "num_items": num_items, "lr": self.current_lr()}

    def _tc_versioned(self):
        tr, ver = self.tc, self._pversion

        class _V:
            @staticmethod
            def get(key, w):
                return tr.get(key, w, version=ver)
        return _V

    def _fold_wn(self):
        """The RMSNorm weight gradient is reduced in fp32 and folded into the bf16 flat buffer."""
        seg = self._seg(self.grad, "layers.0.post_attention_layernorm.weight")
        seg.copy_(self._wn_f32.to(BF16))

    def _bucket_reducer(self, works):
        import torch.distributed as dist
        buckets = list(self._buckets)

        def on_ready(name):
            if name == "wn":
                self._fold_wn()
            while buckets and buckets[0][0] == name:
                _, lo, hi = buckets.pop(0)
                if self.group is not None:
                    # RCCL takes a whole bucket in one asynchronous call that overlaps the rest of the backward.  gloo (CPU
                    # tests, and the rehearsal of N ranks on fewer GPUs) never returns from an all-reduce of a device tensor
                    # issued while this rank's kernels are still queued (measured on this image: with the device idle the
                    # same call takes 20 ms), so there the device is drained first and the pieces are blocking: the
                    # same collective sequence, no overlap -- a rehearsal of the logic, not of the timing.
                    if dist.get_backend(self.group) != "gloo":
                        works.append(dist.all_reduce(self.grad[lo:hi], op=dist.ReduceOp.AVG, group=self.group, async_op=True))
                    else:
                        for a in range(lo, hi, 16 * 1024 * 1024):
                            all_reduce_(self.grad[a:min(a + 16 * 1024 * 1024, hi)], self.group, op=dist.ReduceOp.AVG)
        return on_ready

    def _optimizer_step(self):
        self.step_count += 1
        lr = self.current_lr()
        coef = None
        clip_value = 0.0
        if self.mode == "dp":
            clip_value = self.clip                                            # train.py:127 on the reduced gradient
        elif self.clip > 0:
            coef = ops.grad_clip_coef(self.grad, self.clip)                   # train_tp.py:109
        ops.adamw_(self.flat, self.grad, self.m, self.v, lr, self.betas[0], self.betas[1], self.eps, self.wd,
                   self.step_count, grad_scale=coef, clip_value=clip_value)
        self._pversion += 1
        # the kernel above wrote the parameters through raw pointers (no Tensor._version bump): drop what the
        # model derived from the old values (stacked / norm-folded decode weights, transposed copies)
        self.model.invalidate_weight_caches()
        # accelerate's scheduler steps once per process per call (accelerate/scheduler.py:73-76)
        self.sched_step += self.world if self.mode == "dp" else 1


# ---------------------------------------------------------------------------------------------
# vocab-sharded lm_head / CE (the multi-GPU form of eagle/train_tp.py's loss)
# ---------------------------------------------------------------------------------------------
def vocab_shard_bounds(V: int, world: int, rank: int, align: int = 64) -> Tuple[int, int]:
    """Rows [v0, v1) of lm_head held by ``rank``: equal shards rounded up to ``align`` rows, the
    last rank takes what is left.  align = 64 because the shard width is the K dimension of the
    dpredict GEMM (eagle_gemm_nt_bf16: K % 64 == 0): V = 151936 over 8 ranks gives 7 x 19008 +
    18880 rather than 8 x 18992."""
    per = (V + world - 1) // world
    per = (per + align - 1) // align * align
    v0 = min(V, rank * per)
    return v0, min(V, v0 + per)


class ShardedLmHeadLoss:
    """Every rank holds lm_head rows [v0, v1) and sees the SAME predict/target rows (the draft
    layer is replicated across the group).  One exchange step forward (all-gather of the per-row
    softmax statistics, 32 B per row and rank), one backward (all-reduce of the [N, D] partial
    dpredict).  Argmax ties across shards resolve to the lowest global index, like torch.max on
    the unsharded logits (eagle/train_tp.py:97-99)."""

    def __init__(self, lm_head_shard: torch.Tensor, v0: int, group, world_size: int, rank: int):
        self.w = lm_head_shard.contiguous()
        self.wt = ops.transpose(self.w) if self.w.shape[0] else None
        self.v0, self.group, self.world, self.rank = v0, group, world_size, rank

    def fwd_bwd(self, predict, target, mask, v_w: float, p_w: float, scale: float, want_grad: bool = True):
        import torch.distributed as dist
        N, D = predict.shape
        t_logits = ops.gemm_nt(target, self.w)
        o_logits = ops.gemm_nt(predict, self.w)
        stats = ops.ce_partial(t_logits, o_logits, self.v0, predict, target)           # [N, 8] fp32 words
        if self.world > 1:
            allst = torch.empty((self.world * stats.shape[0], stats.shape[1]), dtype=stats.dtype, device=stats.device)
            dist.all_gather_into_tensor(allst, stats, group=self.group)     # rank-major concatenation
            allst = allst.view(self.world, stats.shape[0], stats.shape[1])
        else:
            allst = stats[None]
        sums, lse_t, lse_o, _ = ops.ce_combine(allst, mask)
        if not want_grad:
            return sums, None
        dlog, dpv = ops.ce_smoothl1_bwd(t_logits, o_logits, lse_t, lse_o, mask, p_w * scale, v_w * scale, predict, target,
                                        in_place=True)
        # partial dpredict over this rank's vocabulary slice in fp32, summed over ranks, then the
        # (replicated) smooth-L1 part is added once
        part = ops.gemm_nt(dlog, self.wt, out_dtype=torch.float32)
        if self.world > 1:
            all_reduce_(part, self.group)
        return sums, ops.add_f32_bf16(part, dpv)
