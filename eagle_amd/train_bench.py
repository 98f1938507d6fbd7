"""Draft-head training throughput (BASELINE.json configs[2]): Qwen3-8B-dims draft head
(resources/eagle_config_qwen3_8b.json: D=4096, I=12288, 32 heads, V=151936), fused smooth-L1 + CE
against a frozen lm_head, sequence 2048, per rank and optimizer step `ga` micro-batches of
`micro_batch` x 2048 tokens (bench.py passes the reference's own Qwen3 recipe per GPU: micro-batch 2,
gradient accumulation 2, pages/train_qwen_3.md:19-20), data parallel over the ranks with one bucketed
RCCL all-reduce of the flat bf16 gradient per optimizer step, deferred over the accumulation
micro-batches (eagle/train.py:85-131).  Synthetic verifier hidden states / ids, random-init weights.

Called from bench.py on every rank (collectives inside).  Tokens/s follows eagle/train.py:140-141:
loss-masked tokens per step / step wall time, summed over ranks.
"""
from __future__ import annotations

import json
import os
import time
from typing import Optional

import torch

BF16 = torch.bfloat16
MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s achievable)


def loss_roofline(dev, N: int, V: int, D: int, iters: int = 8) -> dict:
    """The fused smooth-L1 + soft-label CE kernels alone (HBM-bound): HIP events on the
    launch stream around `iters` forward launches (ce_partial + ce_combine) and `iters` backward
    launches (ce_bwd), logits of the training shape already in HBM.  Algorithmic bytes: forward reads
    both logit tensors once plus predict/target (2*N*V*2 + 2*N*D*2); backward reads both again and
    writes dlogits and dpredict (3*N*V*2 + 3*N*D*2)."""
    from . import ops
    g = torch.Generator(device=dev).manual_seed(5)
    t_logits = (torch.randn((N, V), generator=g, device=dev, dtype=torch.float32) * 2).to(BF16)
    o_logits = (torch.randn((N, V), generator=g, device=dev, dtype=torch.float32) * 2).to(BF16)
    predict = torch.randn((N, D), generator=g, device=dev, dtype=torch.float32).to(BF16)
    target = torch.randn((N, D), generator=g, device=dev, dtype=torch.float32).to(BF16)
    mask = torch.ones(N, dtype=torch.float32, device=dev)

    def fwd():
        stats = ops.ce_partial(t_logits, o_logits, 0, predict, target)
        return ops.ce_combine(stats, mask)

    sums, lse_t, lse_o, _ = fwd()
    ops.ce_smoothl1_bwd(t_logits, o_logits, lse_t, lse_o, mask, 0.1 / N, 1.0 / N, predict, target)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    for _ in range(iters):
        fwd()
    ev[1].record()
    for _ in range(iters):
        ops.ce_smoothl1_bwd(t_logits, o_logits, lse_t, lse_o, mask, 0.1 / N, 1.0 / N, predict, target)
    ev[2].record()
    torch.cuda.synchronize()
    t_f, t_b = ev[0].elapsed_time(ev[1]) / iters * 1e-3, ev[1].elapsed_time(ev[2]) / iters * 1e-3
    b_f, b_b = 2.0 * N * V * 2 + 2.0 * N * D * 2, 3.0 * N * V * 2 + 3.0 * N * D * 2
    ach = (b_f + b_b) / (t_f + t_b) / 1e9
    # HBM bytes of one forward + one backward from rocprofv3 PMC passes of the same kernels at this shape
    # (profiles/r0x_loss_pmc.json, newest first: tools/gpu_profile_r03.sh + tools/pmc2_to_json.py), else null
    traffic = None
    for name in ("r03_loss_pmc.json", "r01_loss_pmc.json"):
        pj = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", name)
        if os.path.exists(pj):
            with open(pj) as f:
                pm = json.load(f)
            if pm.get("shape") == {"N": N, "V": V, "D": D}:
                traffic = pm.get("traffic_bytes_fwd_plus_bwd")
            break
    return {"bound": "hbm", "kernel": "ce_partial_kernel + ce_combine_kernel (fwd), ce_bwd_kernel (bwd)", "achieved": ach,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "bytes_fwd_plus_bwd": b_f + b_b,
            "fwd": {"us": t_f * 1e6, "bytes": b_f, "GB/s": b_f / t_f / 1e9}, "bwd": {"us": t_b * 1e6, "bytes": b_b, "GB/s": b_b / t_b / 1e9},
            "shape": {"N": N, "V": V, "D": D}, "timing": "HIP events around %d back-to-back launches each" % iters}


def flops_per_token(cfg, L: int) -> float:
    """Algorithmic flops of one training step per token (no recompute)."""
    D, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    layer = 2.0 * (2 * D * D + 4 * D * D + 3 * D * I)          # fc (K = 2D) + q,k,v,o + gate,up,down
    head = 2.0 * D * V
    attn_fwd = 4.0 * D * (L / 2.0)                             # QK^T + PV over the causal half
    fwd = layer + 2 * head + attn_fwd
    bwd = 2 * layer + head + 2.5 * attn_fwd                    # dgrad + wgrad; frozen lm_head: dgrad only
    return fwd + bwd


def train_tokens_per_s(dev, rank: int, world: int, dist, steps: int = 6, warmup: int = 2, L: int = 2048,
                       config: str = "qwen3_8b", ga: int = 1, micro_batch: int = 1) -> Optional[dict]:
    from .config import DraftConfig, LLAMA2_7B_DRAFT, QWEN3_8B_DRAFT
    from .llama2 import Llama2Model
    from .train import EagleTrainer

    cfg = DraftConfig(**(QWEN3_8B_DRAFT if config == "qwen3_8b" else LLAMA2_7B_DRAFT))
    with torch.device("meta"):
        model = Llama2Model(cfg)
    model = model.to_empty(device=dev).to(BF16)
    g = torch.Generator(device=dev).manual_seed(0)              # identical initial weights on every rank
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("layernorm.weight"):
                p.fill_(1.0)
            else:
                p.copy_((torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32) * 0.02).to(BF16))
        lm_head = (torch.randn((cfg.vocab_size, cfg.hidden_size), generator=g, device=dev, dtype=torch.float32) * 0.02).to(BF16)
    group = dist.group.WORLD if (dist is not None and world > 1) else None
    tr = EagleTrainer(model, lm_head, lr=2e-4, betas=(0.9, 0.95), grad_clip=0.5, num_warmup_steps=world * 10,
                      num_training_steps=world * 1000, mode="dp", group=group, world_size=world)
    gd = torch.Generator(device=dev).manual_seed(1000 + rank)  # different data per rank
    D, MB = cfg.hidden_size, micro_batch

    def batch():
        hid = torch.randn((MB, L, D), generator=gd, device=dev, dtype=torch.float32).to(BF16)
        noise = ((torch.rand((MB, L, D), generator=gd, device=dev) - 0.5) * 0.2).to(BF16)
        ids = torch.randint(0, cfg.vocab_size, (MB, L), generator=gd, device=dev)
        from . import ops
        h, t, i = ops.make_eagle_input(hid, ids, noise)
        return {"hidden_states": h, "target": t, "input_ids": i, "loss_mask": torch.ones((MB, L), dtype=torch.int64, device=dev)}

    pool = [[batch() for _ in range(ga)] for _ in range(2)]

    def sync():
        torch.cuda.synchronize()
        if group is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for s in range(warmup):
        tr.step(pool[s % 2])
        if rank == 0:
            import sys
            print(f"[train_bench] warm-up step {s} issued", file=sys.stderr, flush=True)
    sync()
    t0 = time.perf_counter()
    for s in range(steps):
        out = tr.step(pool[s % 2])
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if group is not None:
        from .train import all_reduce_
        all_reduce_(t, group, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    sums = out["sums"].cpu()
    toks = world * ga * MB * L * steps
    fpt = flops_per_token(cfg, L)
    tf = toks * fpt / dt / 1e12
    res = {"value": toks / dt, "unit": "loss tokens/s (whole job)", "ms_per_step": dt / steps * 1e3, "n_gpus": world,
           "config": {"workload": f"BASELINE configs[2]: {config} dims draft head training, fused smooth-L1+CE, seq {L}, "
                                  f"micro-batch {MB}x{L} per GPU, GA {ga}, DP={world} (RCCL all-reduce of the flat bf16 gradient)"},
           "dtype": "bf16", "steps": steps, "warmup": warmup,
           "roofline": {"bound": "mfma", "achieved": tf / world, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / world / MFMA_PEAK_TFLOPS,
                        "flops_per_token": fpt, "note": "whole training step (all kernels + all-reduce) per GPU, algorithmic flops"},
           "last_step": {"ploss_per_token": float(sums[0] / sums[3]), "vloss_per_token": float(sums[1] / sums[3]),
                         "accuracy": float(sums[2] / sums[3])}}
    del tr, model, lm_head, pool
    torch.cuda.empty_cache()
    if rank == 0:
        res["loss_roofline"] = loss_roofline(dev, L, cfg.vocab_size, cfg.hidden_size)
        torch.cuda.empty_cache()
    return res
