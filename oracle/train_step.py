"""Oracle (TEST INFRASTRUCTURE ONLY): CPU restatement of one optimizer step of the reference's
training loops, with torch autograd doing the backward.

``dp``  follows eagle/train.py:85-131 under ``torchrun`` with ``world`` ranks: token count summed
        over ranks (:92-93), per micro-batch loss * GA * world / num_items (:124), accelerate's
        backward dividing by GA (accelerate/accelerator.py ``backward``), ``no_sync`` on all but
        the last micro-batch (:97-101) so that DDP averages the ACCUMULATED gradients over ranks
        during the last backward, ``clip_grad_value_`` after every micro-batch backward (:127),
        AdamW (:66, :129), and accelerate's scheduler stepping once per process (:130,
        accelerate/scheduler.py:73-76).
``tp``  follows eagle/train_tp.py:68-113: one process, loss / num_items (:105-106),
        ``clip_grad_norm_`` after every micro-batch backward (:109), AdamW, scheduler.step().

The ranks of ``dp`` are emulated in-process: rank r's micro-batches are ``ranks[r]``.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import draft_model as dm
from . import loss as ol

TRAINABLE = ("fc.weight", "fc.bias", "layers.0.self_attn.q_proj.weight", "layers.0.self_attn.k_proj.weight",
             "layers.0.self_attn.v_proj.weight", "layers.0.self_attn.o_proj.weight", "layers.0.mlp.gate_proj.weight",
             "layers.0.mlp.up_proj.weight", "layers.0.mlp.down_proj.weight", "layers.0.post_attention_layernorm.weight")


def micro_batch_grads(w: Dict[str, torch.Tensor], dims: dm.DraftDims, lm_head: torch.Tensor, mb: dict, v_w: float,
                      p_w: float, scale: float):
    """forward (eagle/train.py:105), loss block (:106-123), ``scale * loss`` backward.
    Returns ({name: grad}, {"ploss","vloss","correct","num"}, predict)."""
    leaves = {k: v.detach().clone().requires_grad_(k in TRAINABLE) for k, v in w.items()}
    predict = dm.forward(leaves, dims, mb["hidden_states"], mb["input_ids"])
    out = ol.eager_loss(predict, mb["target"], lm_head, mb["loss_mask"], v_w, p_w)
    (out["loss"] * scale).backward()
    grads = {k: leaves[k].grad for k in TRAINABLE if k in leaves}
    stats = {"ploss": out["ploss"].detach(), "vloss": out["vloss"].detach(), "correct": out["correct"].detach(),
             "num": mb["loss_mask"].sum()}
    return grads, stats, predict.detach()


def linear_schedule(step: int, warmup: int, total: int) -> float:
    """transformers.get_linear_schedule_with_warmup's lambda (eagle/train.py:69)."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))


def adamw_update(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float, betas, eps: float,
                 wd: float):
    """torch.optim.AdamW single-tensor update (decoupled weight decay), in place on p, m, v."""
    b1, b2 = betas
    p.mul_(1.0 - lr * wd)
    m.mul_(b1).add_(g, alpha=1.0 - b1)
    v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
    denom = (v / bc2).sqrt_().add_(eps)
    p.addcdiv_(m / bc1, denom, value=-lr)


def accumulate_grads(w, dims, lm_head, ranks: Sequence[Sequence[dict]], mode: str, v_w: float, p_w: float, clip: float):
    """The gradient the optimizer sees, and the step statistics summed over ranks and micro-batches."""
    world = len(ranks)
    if mode == "tp" and world != 1:
        raise ValueError("train_tp.py is single-process")
    num_items = float(sum(float(mb["loss_mask"].sum()) for r in ranks for mb in r))
    if num_items == 0:
        return None, {"num": 0.0}
    tot = {"ploss": 0.0, "vloss": 0.0, "correct": 0.0, "num": num_items}
    per_rank = []
    for mbs in ranks:
        acc: Optional[Dict[str, torch.Tensor]] = None
        for i, mb in enumerate(mbs):
            scale = (world if mode == "dp" else 1.0) / num_items       # GA * world / num_items, then / GA
            g, st, _ = micro_batch_grads(w, dims, lm_head, mb, v_w, p_w, scale)
            for k in ("ploss", "vloss", "correct"):
                tot[k] += float(st[k])
            acc = g if acc is None else {k: acc[k] + g[k] for k in acc}
            last = i == len(mbs) - 1
            if mode == "dp":
                if not last and clip > 0:
                    acc = {k: t.clamp(-clip, clip) for k, t in acc.items()}
            elif clip > 0:                                              # train_tp.py:109, every micro-batch
                norm = torch.sqrt(sum((t.double() ** 2).sum() for t in acc.values())).float()
                coef = torch.clamp(clip / (norm + 1e-6), max=1.0)
                acc = {k: t * coef for k, t in acc.items()}
        per_rank.append(acc)
    g = {k: sum(a[k] for a in per_rank) / world for k in per_rank[0]}    # DDP mean over ranks
    if mode == "dp" and clip > 0:
        g = {k: t.clamp(-clip, clip) for k, t in g.items()}              # train.py:127 after the synced backward
    return g, tot


def optimizer_step(w, dims, lm_head, ranks, state: dict, mode: str = "dp", lr: float = 2e-4, betas=(0.9, 0.95),
                   eps: float = 1e-8, wd: float = 0.01, v_w: float = 1.0, p_w: float = 0.1, clip: float = 0.5,
                   warmup: Optional[int] = None, total: Optional[int] = None):
    """One optimizer step in place on ``w``; ``state`` = {"step", "sched", "m", "v"} (created on
    first use).  Returns (grads, stats)."""
    g, tot = accumulate_grads(w, dims, lm_head, ranks, mode, v_w, p_w, clip)
    if g is None:
        return None, tot
    state.setdefault("step", 0)
    state.setdefault("sched", 0)
    state.setdefault("m", {k: torch.zeros_like(t) for k, t in g.items()})
    state.setdefault("v", {k: torch.zeros_like(t) for k, t in g.items()})
    state["step"] += 1
    cur = lr if warmup is None or total is None else lr * linear_schedule(state["sched"], warmup, total)
    for k in g:
        adamw_update(w[k], g[k], state["m"][k], state["v"][k], state["step"], cur, betas, eps, wd)
    state["sched"] += len(ranks) if mode == "dp" else 1
    tot["lr"] = cur
    return g, tot
