"""TEST-ONLY CPU stand-ins for the kernel wrappers in eagle_amd/ops.py that eagle_amd/train.py calls.

The product has no CPU path.  These torch-CPU functions restate each kernel's CONTRACT
(include/eagle_hip.h) so that the HOST logic of the training drivers -- flat-buffer layout, gradient
accumulation, bucketed all-reduce, clip ordering, LR schedule, the vocab-sharded exchange -- can run
under a world_size-2 gloo group on a machine without a GPU (tests/test_dist_gloo.py installs them
with monkeypatching).  They are never imported by eagle_amd/.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

BF16 = torch.bfloat16


def gemm_nt(A, B, bias=None, res=None, out=None, out_dtype=BF16, accumulate=False, ksplit=1):
    y = A.float() @ B.float().t()
    if bias is not None:
        y = y + bias.float()
    if res is not None:
        y = y.to(BF16).float() + res.float()
    if out is None:
        return y.to(out_dtype)
    if accumulate:
        y = y + out.float()
    out.copy_(y.to(out.dtype))
    return out


def gemm_tn_ok(A, B):
    return False            # the CPU stand-ins take the transpose + NT route


def gemm_nn_ok(A, B):
    return False


def transpose(x, out=None, pad_to=1):
    R, C = x.shape
    Rp = (R + pad_to - 1) // pad_to * pad_to
    y = torch.zeros((C, Rp), dtype=x.dtype)
    y[:, :R] = x.t()
    if out is not None:
        out.copy_(y)
        return out
    return y


def embed_concat(emb, ids, hidden):
    return torch.cat((emb[ids], hidden), dim=-1)


def _rot(x, cos, sin, inverse):
    h = x.shape[-1] // 2
    if inverse:   # transpose of the rotation
        return x * cos + torch.cat((x[..., h:] * sin[..., h:], -x[..., :h] * sin[..., :h]), dim=-1)
    return x * cos + torch.cat((-x[..., h:], x[..., :h]), dim=-1) * sin


def rope_qk_(qkv, cos, sin, L, n_heads_qk, hd, position_ids=None, inverse=False):
    rows = qkv.shape[0]
    pos = position_ids if position_ids is not None else torch.arange(rows) % L
    c, s = cos[pos].float()[:, None, :], sin[pos].float()[:, None, :]
    x = qkv[:, : n_heads_qk * hd].float().view(rows, n_heads_qk, hd)
    qkv[:, : n_heads_qk * hd] = _rot(x, c, s, inverse).reshape(rows, -1).to(qkv.dtype)
    return qkv


def _split(qkv, B, L, H, hd):
    x = qkv.float().view(B, L, 3, H, hd)
    return x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)


def attn_train_fwd(qkv, B, L, H, hd):
    q, k, v = _split(qkv, B, L, H, hd)
    s = (q @ k.transpose(2, 3)) / math.sqrt(hd)
    s = s.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool), 1), float("-inf"))
    lse = torch.logsumexp(s, -1)
    out = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, H * hd)
    return out.to(BF16), (lse / math.log(2.0)).float()


@torch.enable_grad()          # called from inside autograd.Function.backward too, where grad mode is off
def attn_train_bwd(qkv, out, dout, lse, B, L, H, hd, dqkv=None):
    x = qkv.float().detach().requires_grad_(True)
    q, k, v = _split(x, B, L, H, hd)
    s = (q @ k.transpose(2, 3)) / math.sqrt(hd)
    s = s.masked_fill(torch.triu(torch.ones(L, L, dtype=torch.bool), 1), float("-inf"))
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * L, H * hd)
    (o * dout.float()).sum().backward()
    return x.grad.to(BF16)


def rmsnorm(x, w, eps, want_rstd=False):
    xf = x.float()
    rstd = torch.rsqrt(xf.pow(2).mean(-1) + eps)
    y = w * (xf * rstd[..., None]).to(x.dtype)
    return (y, rstd.reshape(-1)) if want_rstd else y


@torch.enable_grad()
def rmsnorm_bwd(dy, x, w, rstd, res=None, dw=None, accumulate_dw=False):
    xf, wf = x.float().detach().requires_grad_(True), w.float().detach().requires_grad_(True)
    eps = 1.0 / rstd[:, None] ** 2 - xf.detach().pow(2).mean(-1, keepdim=True)       # recover eps from the saved rstd
    y = wf * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps))
    (y * dy.float()).sum().backward()
    dx = xf.grad + (res.float() if res is not None else 0.0)
    g = wf.grad
    if dw is None:
        dw = g.clone()
    elif accumulate_dw:
        dw += g
    else:
        dw.copy_(g)
    return dx.to(BF16), dw


def silu_mul(gu):
    I = gu.shape[1] // 2
    return (F.silu(gu[:, :I].float()).to(BF16).float() * gu[:, I:].float()).to(BF16)


def silu_mul_bwd(gu, dact):
    I = gu.shape[1] // 2
    g, u, d = gu[:, :I].float(), gu[:, I:].float(), dact.float()
    sg = torch.sigmoid(g)
    return torch.cat((d * u * (sg * (1 + g * (1 - sg))), d * g * sg), dim=1).to(BF16)


def colsum(x, out, accumulate=False):
    s = x.float().sum(0)
    out.copy_(((out.float() + s) if accumulate else s).to(out.dtype))
    return out


def adamw_(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=None, clip_value=0.0):
    g = grad.float()
    if grad_scale is not None:
        g = g * grad_scale[0]
    if clip_value > 0:
        g = g.clamp(-clip_value, clip_value)
    exp_avg.mul_(beta1).add_(g, alpha=1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    upd = (exp_avg / bc1) / ((exp_avg_sq / bc2).sqrt() + eps)
    param.copy_((param.float() * (1 - lr * weight_decay) - lr * upd).to(param.dtype))


def grad_clip_coef(grad, max_norm):
    n = grad.float().norm()
    return torch.stack((torch.clamp(max_norm / (n + 1e-6), max=1.0), n)).float()


def clip_value_(grad, clip):
    grad.clamp_(-clip, clip)
    return grad


def scale_(x, scale):
    x.copy_((x.float() * scale[0]).to(x.dtype))
    return x


def add_f32_bf16(a, b):
    return (a + (b.float() if b is not None else 0.0)).to(BF16)


# ------------------------------------------------------------------ loss (record layout of CeStat, loss.hip)
def ce_partial(t_logits, o_logits, v0=0, predict=None, target=None):
    t, o = t_logits.float(), o_logits.float()
    N = t.shape[0]
    st = torch.zeros((N, 8), dtype=torch.float32)
    m_t, m_o = t.max(-1).values, o.max(-1).values
    e = torch.exp(t - m_t[:, None])
    st[:, 0], st[:, 1], st[:, 2] = m_t, e.sum(-1), (e * o).sum(-1)
    st[:, 3], st[:, 4] = m_o, torch.exp(o - m_o[:, None]).sum(-1)
    first = lambda x: torch.sort(x, dim=-1, descending=True, stable=True).indices[:, 0]
    iv = st.view(torch.int32)
    iv[:, 5], iv[:, 6] = (first(t) + v0).int(), (first(o) + v0).int()
    if predict is not None:
        d = predict.float() - target.float()
        st[:, 7] = torch.where(d.abs() < 1, 0.5 * d * d, d.abs() - 0.5).mean(-1)
    return st


def ce_combine(stats, mask):
    if stats.dim() == 2:
        stats = stats[None]
    iv = stats.view(torch.int32)
    m_t, m_o = stats[..., 0].max(0).values, stats[..., 3].max(0).values
    z_t = (stats[..., 1] * torch.exp(stats[..., 0] - m_t)).sum(0)
    a = (stats[..., 2] * torch.exp(stats[..., 0] - m_t)).sum(0)
    z_o = (stats[..., 4] * torch.exp(stats[..., 3] - m_o)).sum(0)

    def arg(mcol, icol, mx):
        cand = torch.where(stats[..., mcol] == mx, iv[..., icol].long(), torch.full_like(iv[..., icol].long(), 1 << 40))
        return cand.min(0).values
    agree = (arg(0, 5, m_t) == arg(3, 6, m_o)).float()
    lse_t, lse_o = m_t + torch.log(z_t), m_o + torch.log(z_o)
    pl = lse_o - a / z_t
    vl = stats[0, :, 7]
    mk = mask if mask is not None else torch.ones_like(pl)
    rows = torch.stack((pl, vl, agree, torch.zeros_like(pl)), dim=1)
    sums = torch.stack(((mk * pl).sum(), (mk * vl).sum(), (mk * agree).sum(), mk.sum())).float()
    return sums, lse_t, lse_o, rows


def ce_smoothl1_bwd(t_logits, o_logits, lse_t, lse_o, mask, g_p, g_v, predict=None, target=None, in_place=False):
    t, o = t_logits.float(), o_logits.float()
    mk = (mask if mask is not None else torch.ones(t.shape[0]))[:, None]
    dl = (mk * g_p * (torch.exp(o - lse_o[:, None]) - torch.exp(t - lse_t[:, None]))).to(BF16)
    if in_place:
        o_logits.copy_(dl)
        dl = o_logits
    dp = None
    if predict is not None:
        d = predict.float() - target.float()
        dp = (mk * g_v * d.clamp(-1, 1) / d.shape[-1]).to(BF16)
    return dl, dp


def rope_tables(head_dim, n_pos, base=10000.0, device=None):
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = torch.outer(torch.arange(n_pos, dtype=torch.float32), inv_freq)
    ang = torch.cat((fr, fr), dim=-1)
    return ang.cos().to(BF16), ang.sin().to(BF16)


def require_device(t, what):
    """Test processes only: the host logic of the N > 1 drivers runs on CPU tensors under gloo."""
    return None


def install(monkeypatch_setattr=None):
    """Replace the kernel wrappers of eagle_amd.ops by the stand-ins above (test processes only)."""
    from eagle_amd import ops
    g = globals()
    for name in ("gemm_nt", "transpose", "embed_concat", "rope_qk_", "attn_train_fwd", "attn_train_bwd", "rmsnorm", "rmsnorm_bwd",
                 "silu_mul", "silu_mul_bwd", "colsum", "adamw_", "grad_clip_coef", "clip_value_", "scale_", "add_f32_bf16",
                 "ce_partial", "ce_combine", "ce_smoothl1_bwd", "rope_tables", "require_device", "gemm_tn_ok", "gemm_nn_ok"):
        if monkeypatch_setattr is not None:
            monkeypatch_setattr(ops, name, g[name])
        else:
            setattr(ops, name, g[name])
