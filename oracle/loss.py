"""Oracle (TEST INFRASTRUCTURE ONLY): CPU restatement of the EAGLE training loss block.

Follows eagle/train.py:105-126 (data-parallel script) and eagle/train_tp.py:87-108
(single-process script): frozen lm_head on target and predict, softmax / log_softmax over the
vocabulary, soft-label cross entropy, smooth-L1 feature loss (beta = 1, reduction "none",
train.py:65), argmax agreement under the loss mask, weighted sum.

``eager_loss`` keeps the reference's op order and dtypes (bit-matches the golden fixtures);
``exact_loss`` is the same math from the logits in fp32/fp64, i.e. what a fused kernel that
accumulates in fp32 should reproduce within the stated bf16 tolerance (rtol 1e-2).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F


def eager_loss(predict: torch.Tensor, target: torch.Tensor, lm_head_w: torch.Tensor,
               loss_mask: torch.Tensor, v_w: float = 1.0, p_w: float = 0.1) -> Dict[str, torch.Tensor]:
    """predict/target [B,L,D]; lm_head_w [V,D]; loss_mask [B,L] (int64 assistant mask).

    Op sequence of train.py:106-123 / train_tp.py:88-105."""
    hd = lm_head_w.dtype
    with torch.no_grad():
        target_head = F.linear(target.to(hd), lm_head_w)
    out_head = F.linear(predict.to(hd), lm_head_w)
    return eager_loss_from_logits(target_head, out_head, predict, target, loss_mask, v_w, p_w)


def eager_loss_from_logits(target_head: torch.Tensor, out_head: torch.Tensor, predict: torch.Tensor, target: torch.Tensor,
                           loss_mask: torch.Tensor, v_w: float = 1.0, p_w: float = 0.1) -> Dict[str, torch.Tensor]:
    """The statements of train.py:108-123 that follow the two lm_head products (what the fused loss kernels replace):
    softmax, log_softmax, agreement count, soft-label CE, smooth-L1, weighted sum -- in the reference's op order."""
    with torch.no_grad():
        target_p = torch.softmax(target_head, dim=2).detach()
    out_logp = torch.log_softmax(out_head, dim=2)
    m3 = loss_mask[:, :, None]
    _, t_tok = torch.max(target_p, 2)
    _, o_tok = torch.max(out_logp, 2)
    correct = ((t_tok == o_tok) * m3.squeeze()).sum()
    ploss = -torch.sum(torch.sum(m3 * (target_p * out_logp), 2))
    vl = F.smooth_l1_loss(predict, target, reduction="none")
    vloss = torch.sum(torch.mean(m3 * vl, 2))
    loss = v_w * vloss + p_w * ploss
    return {"ploss": ploss, "vloss": vloss, "loss": loss, "correct": correct,
            "target_logits": target_head, "out_logits": out_head}


def exact_loss_from_logits(t_logits: torch.Tensor, o_logits: torch.Tensor, predict: torch.Tensor,
                           target: torch.Tensor, loss_mask: torch.Tensor,
                           acc=torch.float64) -> Dict[str, torch.Tensor]:
    """Same quantities from given logits [N,V] with wide accumulation.

    ploss_row = logsumexp(o) - sum_v softmax(t)_v * o_v   (== -sum p*log_softmax(o), train.py:119-120)
    argmax tie-break: first (lowest) index, as torch.max (train.py:115-116).
    """
    t_logits = t_logits.reshape(-1, t_logits.shape[-1])
    o_logits = o_logits.reshape(-1, o_logits.shape[-1])
    t = t_logits.to(acc)
    o = o_logits.to(acc)
    m = loss_mask.reshape(-1).to(acc)
    lse_t = torch.logsumexp(t, dim=-1)
    lse_o = torch.logsumexp(o, dim=-1)
    p = torch.exp(t - lse_t[:, None])
    ploss_row = lse_o - (p * o).sum(-1)
    t_tok = first_argmax(t_logits)
    o_tok = first_argmax(o_logits)
    d = predict.to(acc).reshape(m.numel(), -1) - target.to(acc).reshape(m.numel(), -1)
    ad = d.abs()
    vl = torch.where(ad < 1.0, 0.5 * d * d, ad - 0.5)
    vloss_row = vl.mean(-1)
    return {
        "ploss_row": ploss_row, "vloss_row": vloss_row,
        "ploss": (m * ploss_row).sum(), "vloss": (m * vloss_row).sum(),
        "correct": ((t_tok == o_tok).to(acc) * m).sum(),
        "lse_t": lse_t, "lse_o": lse_o, "t_tok": t_tok, "o_tok": o_tok,
    }


def first_argmax(x: torch.Tensor) -> torch.Tensor:
    """Index of the first (lowest-index) maximum per row; independent of torch's internal
    argmax tie behaviour: stable descending sort => value desc, index asc."""
    return torch.sort(x.to(torch.float32), dim=-1, descending=True, stable=True).indices[..., 0]


def exact_grads_from_logits(t_logits, o_logits, predict, target, loss_mask, g_p: float, g_v: float,
                            acc=torch.float64):
    """d(g_p*ploss + g_v*vloss)/d(out_logits) and the smooth-L1 part of d/d(predict).

    dlogits[n,v] = mask[n]*g_p*(softmax(o)[n,v] - softmax(t)[n,v])
    dpredict_v[n,d] = mask[n]*g_v*clamp(predict-target, -1, 1)/D   (beta = 1)
    """
    t = t_logits.reshape(-1, t_logits.shape[-1]).to(acc)
    o = o_logits.reshape(-1, o_logits.shape[-1]).to(acc)
    m = loss_mask.reshape(-1, 1).to(acc)
    dlog = m * g_p * (torch.softmax(o, -1) - torch.softmax(t, -1))
    N = m.shape[0]
    d = predict.to(acc).reshape(N, -1) - target.to(acc).reshape(N, -1)
    dpv = m * g_v * d.clamp(-1.0, 1.0) / d.shape[-1]
    return dlog, dpv


def make_eagle_input(input_ids: torch.Tensor, loss_mask: torch.Tensor, hidden_big: torch.Tensor,
                     max_len: int, noise: Optional[torch.Tensor] = None):
    """Shift semantics of eagle/train.py:256-269 given the verifier's last hidden state
    (the verifier forward itself is out of scope).  ``noise`` is the already-drawn uniform
    tensor (train.py:272-275) so the result is deterministic.

    target[:, :-1] = noisy_hidden[:, 1:], last row zero; ids shifted left with trailing 0;
    loss_mask truncated but NOT shifted."""
    ids = input_ids[:, :max_len]
    lm = loss_mask[:, :max_len]
    h = hidden_big[:, :max_len]
    if noise is not None:
        h = h + noise[:, :max_len]
    tgt = h.new_zeros(h.shape)
    tgt[:, :-1, :] = h[:, 1:, :]
    ids = torch.cat((ids[:, 1:], torch.zeros(ids.size(0), 1, dtype=ids.dtype)), dim=1)
    return {"input_ids": ids, "hidden_states": h, "target": tgt, "loss_mask": lm}
