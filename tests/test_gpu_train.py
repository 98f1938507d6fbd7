"""GPU parity tests of the draft-head TRAINING step (eagle/train.py:85-131, eagle/train_tp.py:68-113):
the HIP kernel chains behind eagle_amd.train against the CPU oracle (oracle/train_step.py: the
reference's op sequence with torch autograd doing the backward) on the same seeded inputs.

Bar: bf16 rtol 1e-2 (north_star) on predict / loss; gradients are compared by relative L2 error
and cosine against the oracle run in fp32 on the same bf16-rounded weights (a bf16 eager run of the
reference differs from that fp32 truth by more than the kernels do; the test checks that too).
Counts (agreement, masked tokens) are exact."""
import copy
import os

import pytest
import torch

from oracle import draft_model as dm
from oracle import loss as ol
from oracle import train_step as ots

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DEV = "cuda:0"


def rel_l2(a, b):
    a, b = a.float().cpu().reshape(-1), b.float().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def cosine(a, b):
    a, b = a.float().cpu().reshape(-1), b.float().cpu().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


DIMS = dm.DraftDims(hidden=256, inter=512, heads=2, kv_heads=2, vocab=1024, max_pos=256)


def make_model(seed=0, dims=DIMS, std=0.05):
    from eagle_amd.config import DraftConfig
    from eagle_amd.llama2 import Llama2Model
    w32 = dm.init_weights(dims, seed=seed, std=std)
    wb = {k: v.to(BF) for k, v in w32.items()}
    cfg = DraftConfig(hidden_size=dims.hidden, intermediate_size=dims.inter, num_attention_heads=dims.heads,
                      vocab_size=dims.vocab, max_position_embeddings=dims.max_pos)
    model = Llama2Model(cfg)
    model.load_state_dict(wb)
    model = model.to(BF).to(DEV)
    g = torch.Generator().manual_seed(seed + 99)
    lm_head = (torch.randn(dims.vocab, dims.hidden, generator=g) * 0.2).to(BF)
    return model, wb, lm_head


def make_batch(seed, B, L, dims=DIMS, p_mask=0.7):
    g = torch.Generator().manual_seed(seed)
    hid = torch.randn(B, L, dims.hidden, generator=g).to(BF)
    ids = torch.randint(0, dims.vocab, (B, L), generator=g)
    tgt = (hid.float() * 0.5 + 0.7 * torch.randn(B, L, dims.hidden, generator=g)).to(BF)
    mask = (torch.rand(B, L, generator=g) < p_mask).long()
    mask[0, 0] = 1
    return {"hidden_states": hid, "input_ids": ids, "target": tgt, "loss_mask": mask}


def to_dev(mb):
    return {k: v.to(DEV) for k, v in mb.items()}


def f32(w):
    return {k: v.float() for k, v in w.items()}


# ------------------------------------------------------------------ element-wise kernels
def test_embed_concat_silu_colsum_make_input():
    from eagle_amd import ops
    g = torch.Generator().manual_seed(1)
    N, D, V, I = 37, 64, 50, 136
    emb = torch.randn(V, D, generator=g).to(BF)
    ids = torch.randint(0, V, (N,), generator=g)
    hid = torch.randn(N, D, generator=g).to(BF)
    out = ops.embed_concat(emb.to(DEV), ids.to(DEV), hid.to(DEV))
    assert torch.equal(out.cpu(), torch.cat((emb[ids], hid), dim=-1))                 # llama2.py:515,:537-542 bit-exact
    gu = (torch.randn(N, 2 * I, generator=g) * 2).to(BF)
    act = ops.silu_mul(gu.to(DEV))
    ref = torch.nn.functional.silu(gu[:, :I]) * gu[:, I:]                             # bf16 eager op order, llama2.py:317
    assert rel_l2(act, ref) < 4e-3
    dact = torch.randn(N, I, generator=g).to(BF)
    guf = gu.float().requires_grad_(True)
    (torch.nn.functional.silu(guf[:, :I]) * guf[:, I:] * dact.float()).sum().backward()
    assert rel_l2(ops.silu_mul_bwd(gu.to(DEV), dact.to(DEV)), guf.grad) < 4e-3
    x = torch.randn(300, 72, generator=g).to(BF)
    o32 = torch.full((72,), 2.0, device=DEV)
    ops.colsum(x.to(DEV), o32, accumulate=True)
    assert torch.allclose(o32.cpu(), x.float().sum(0) + 2.0, rtol=1e-5, atol=1e-4)
    obf = torch.zeros(72, dtype=BF, device=DEV)
    ops.colsum(x.to(DEV), obf)
    assert rel_l2(obf, x.float().sum(0)) < 4e-3
    # _make_eagle_input: train.py:256-269
    B, L = 3, 11
    h = torch.randn(B, L, D, generator=g).to(BF)
    nz = ((torch.rand(B, L, D, generator=g) - 0.5) * 0.2).to(BF)
    tok = torch.randint(0, V, (B, L), generator=g)
    ref = ol.make_eagle_input(tok, torch.ones(B, L, dtype=torch.long), h, L, noise=nz)
    ho, tg, io = ops.make_eagle_input(h.to(DEV), tok.to(DEV), nz.to(DEV))
    assert torch.equal(ho.cpu(), ref["hidden_states"]) and torch.equal(tg.cpu(), ref["target"]) and torch.equal(io.cpu(), ref["input_ids"])
    ho, tg, io = ops.make_eagle_input(h.to(DEV), tok.to(DEV), None)
    assert torch.equal(ho.cpu(), h) and torch.equal(tg[:, :-1].cpu(), h[:, 1:]) and not tg[:, -1].any()


@pytest.mark.parametrize("N,D", [(5, 64), (300, 256), (2048, 4096)])
def test_rmsnorm_bwd(N, D):
    from eagle_amd import ops
    g = torch.Generator().manual_seed(N)
    x = torch.randn(N, D, generator=g).to(BF)
    w = (1 + 0.1 * torch.randn(D, generator=g)).to(BF)
    dy = torch.randn(N, D, generator=g).to(BF)
    res = torch.randn(N, D, generator=g).to(BF)
    y, rstd = ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-6, want_rstd=True)
    xf, wf = x.float().requires_grad_(True), w.float().requires_grad_(True)
    yr = dm.rmsnorm(xf, wf, 1e-6)
    (yr * dy.float()).sum().backward()
    assert rel_l2(y, yr) < 4e-3
    dx, dw = ops.rmsnorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), rstd, res=res.to(DEV))
    assert rel_l2(dx, xf.grad + res.float()) < 4e-3
    assert rel_l2(dw, wf.grad) < 2e-3
    dw2 = dw.clone()
    ops.rmsnorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), rstd, dw=dw2, accumulate_dw=True)
    assert rel_l2(dw2, 2 * wf.grad) < 2e-3
    dx0, _ = ops.rmsnorm_bwd(dy.to(DEV), x.to(DEV), w.to(DEV), rstd)
    assert rel_l2(dx0, xf.grad) < 4e-3


@pytest.mark.parametrize("n", [8, 1000, 128 * 1024 + 5])
def test_adamw_clip_scale_kernels(n):
    from eagle_amd import ops
    g = torch.Generator().manual_seed(n)
    npad = (n + 7) // 8 * 8
    p = torch.randn(npad, generator=g).to(BF)[:n]
    gr = (torch.randn(npad, generator=g) * 0.7).to(BF)[:n]
    pd, gd = p.to(DEV), gr.to(DEV)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    pr, mr, vr = p.float().clone(), torch.zeros(n), torch.zeros(n)
    for step in (1, 2, 3):                                                     # torch.optim.AdamW semantics, train.py:66
        ops.adamw_(pd, gd, m, v, 1e-2, 0.9, 0.95, 1e-8, 0.01, step, clip_value=0.5)
        ots.adamw_update(pr, gr.float().clamp(-0.5, 0.5), mr, vr, step, 1e-2, (0.9, 0.95), 1e-8, 0.01)
        pr = pr.to(BF).float()                                                 # the kernel keeps bf16 parameters
        assert torch.allclose(m.cpu(), mr, rtol=1e-5, atol=1e-7) and torch.allclose(v.cpu(), vr, rtol=1e-5, atol=1e-9)
        assert float((pd.float().cpu() - pr).abs().max()) <= 2.0 ** -7 * float(pr.abs().max())   # at most one bf16 ulp
        assert rel_l2(pd, pr) < 1e-3
    # clip_grad_norm_ coefficient (train_tp.py:109) and its application
    cn = ops.grad_clip_coef(gd, 0.5)
    norm = float(gr.float().norm())
    assert abs(float(cn[1]) - norm) <= 1e-4 * norm and abs(float(cn[0]) - min(1.0, 0.5 / (norm + 1e-6))) <= 1e-5
    g2 = gd.clone()
    ops.scale_(g2, cn)
    assert rel_l2(g2, gr.float() * float(cn[0])) < 4e-3
    m2, v2, p2 = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), p.to(DEV)
    ops.adamw_(p2, gd, m2, v2, 1e-2, 0.9, 0.95, 1e-8, 0.0, 1, grad_scale=cn)
    assert torch.allclose(m2.cpu(), 0.1 * gr.float() * float(cn[0]), rtol=1e-4, atol=1e-8)
    if n % 8 == 0:
        g3 = gd.clone()
        ops.clip_value_(g3, 0.25)
        assert torch.equal(g3.cpu(), gr.clamp(-0.25, 0.25))
        a = torch.randn(n, generator=g)
        assert torch.equal(ops.add_f32_bf16(a.to(DEV), gd).cpu(), (a + gr.float()).to(BF))
        assert torch.equal(ops.add_f32_bf16(a.to(DEV), None).cpu(), a.to(BF))


# ------------------------------------------------------------------ forward / backward chains
@pytest.mark.parametrize("B,L", [(1, 64), (2, 150)])
def test_train_forward_backward_vs_oracle(B, L):
    """draft_fwd + fused loss + draft_bwd (the EagleTrainer chain) against autograd on the oracle."""
    from eagle_amd import ops, train as T
    model, wb, lm_head = make_model(seed=3)
    mb = make_batch(11, B, L)
    scale = 1.0 / float(mb["loss_mask"].sum())
    gref, st, pref = ots.micro_batch_grads(f32(wb), DIMS, lm_head.float(), {k: (v.float() if v.is_floating_point() else v) for k, v in mb.items()},
                                           1.0, 0.1, scale)
    gbf, stb, pbf = ots.micro_batch_grads(wb, DIMS, lm_head, mb, 1.0, 0.1, scale)      # the reference's own bf16 eager run
    W = T.layer_weights(model)
    H, _, hd = model._dims()
    cos, sin = model._rope_tables(DIMS.max_pos, torch.device(DEV))
    d = to_dev(mb)
    N = B * L
    predict, saved = T.draft_fwd(W, model.embed_tokens.weight.detach(), d["input_ids"].reshape(-1), d["hidden_states"].reshape(N, -1),
                                 B, L, H, hd, cos, sin)
    assert rel_l2(predict, pref) < 1e-2, rel_l2(predict, pref)
    lm_d = lm_head.to(DEV)
    mask = d["loss_mask"].reshape(-1).float()
    sums, dpred = T.loss_fwd_bwd(predict, d["target"].reshape(N, -1), lm_d, ops.transpose(lm_d), mask, 1.0, 0.1, scale)
    assert abs(float(sums[0]) - float(st["ploss"])) <= 1e-2 * abs(float(st["ploss"]))
    assert abs(float(sums[1]) - float(st["vloss"])) <= 1e-2 * abs(float(st["vloss"]))
    assert int(sums[3]) == int(mb["loss_mask"].sum())
    # agreement count: exact on the kernel's own logits (first-max tie-break, train.py:115-117)
    ex = ol.exact_loss_from_logits(ops.gemm_nt(d["target"].reshape(N, -1), lm_d).cpu(), ops.gemm_nt(predict, lm_d).cpu(),
                                   predict.cpu(), mb["target"].reshape(N, -1), mb["loss_mask"].reshape(-1))
    assert int(sums[2]) == int(ex["correct"])
    D, I = DIMS.hidden, DIMS.inter
    sink = T.GradSink(fc_w=torch.empty_like(W.fc_w), fc_b=torch.empty_like(W.fc_b), wqkv=torch.empty_like(W.wqkv), wo=torch.empty_like(W.wo),
                      wn=torch.empty(D, device=DEV), wgu=torch.empty_like(W.wgu), wdown=torch.empty_like(W.wdown))
    order = []
    T.draft_bwd(W, saved, dpred, B, L, H, hd, cos, sin, sink, T._TransposeCache(), on_ready=order.append)
    assert order == ["wdown", "wgu", "wn", "wo", "wqkv", "fc"]
    p = "layers.0."
    got = {"fc.weight": sink.fc_w, "fc.bias": sink.fc_b, p + "self_attn.q_proj.weight": sink.wqkv[:D],
           p + "self_attn.k_proj.weight": sink.wqkv[D:2 * D], p + "self_attn.v_proj.weight": sink.wqkv[2 * D:],
           p + "self_attn.o_proj.weight": sink.wo, p + "mlp.gate_proj.weight": sink.wgu[:I], p + "mlp.up_proj.weight": sink.wgu[I:],
           p + "mlp.down_proj.weight": sink.wdown, p + "post_attention_layernorm.weight": sink.wn}
    for k, ref in gref.items():
        e_k, e_ref = rel_l2(got[k], ref), rel_l2(gbf[k], ref)
        assert cosine(got[k], ref) > 0.9995 and e_k < max(2e-2, 1.5 * e_ref), (k, e_k, e_ref)
    # accumulate=True adds a second micro-batch on top
    sink.accumulate = True
    T.draft_bwd(W, saved, dpred, B, L, H, hd, cos, sin, sink, T._TransposeCache())
    assert rel_l2(sink.wdown, 2 * gref[p + "mlp.down_proj.weight"]) < 3e-2
    assert rel_l2(sink.wn, 2 * gref[p + "post_attention_layernorm.weight"]) < 3e-2


def test_autograd_drop_in_matches_reference_loop():
    """The reference's loop body, unchanged: predict = model(h, input_ids=ids); loss; loss.backward()."""
    from eagle_amd.train import eagle_loss
    model, wb, lm_head = make_model(seed=5)
    model.train()
    mb = make_batch(21, 2, 96)
    gref, st, pref = ots.micro_batch_grads(f32(wb), DIMS, lm_head.float(), {k: (v.float() if v.is_floating_point() else v) for k, v in mb.items()},
                                           1.0, 0.1, 1.0 / float(mb["loss_mask"].sum()))
    d = to_dev(mb)
    predict = model(d["hidden_states"], input_ids=d["input_ids"])
    assert predict.requires_grad and predict.shape == d["hidden_states"].shape
    holder = {}
    loss = eagle_loss(predict, d["target"], lm_head.to(DEV), d["loss_mask"], 1.0, 0.1, holder)
    ref_loss = 1.0 * float(st["vloss"]) + 0.1 * float(st["ploss"])
    assert abs(float(loss) - ref_loss) <= 1e-2 * abs(ref_loss)
    (loss / d["loss_mask"].sum()).backward()
    named = dict(model.named_parameters())
    assert named["embed_tokens.weight"].grad is None                              # frozen, llama2.py:450-451
    for k, ref in gref.items():
        assert named[k].grad is not None and named[k].grad.shape == ref.shape
        assert cosine(named[k].grad, ref) > 0.999 and rel_l2(named[k].grad, ref) < 3e-2, (k, rel_l2(named[k].grad, ref))
    # eval / no_grad still takes the decode kernels and matches the training forward
    model.eval()
    with torch.no_grad():
        out = model(d["hidden_states"], d["input_ids"])
    assert rel_l2(out, predict.detach()) < 1e-2


@pytest.mark.parametrize("mode,ga", [("dp", 1), ("dp", 3), ("tp", 1), ("tp", 2)])
def test_trainer_step_vs_oracle(mode, ga):
    """EagleTrainer.step (single process): the gradient handed to AdamW and the updated parameters."""
    from eagle_amd.train import EagleTrainer
    model, wb, lm_head = make_model(seed=7)
    mbs = [make_batch(100 + i, 2, 64, p_mask=0.6) for i in range(ga)]
    w_ref = f32(wb)
    state = {}
    hp = dict(lr=1e-3, betas=(0.9, 0.95), eps=1e-8, wd=0.01, v_w=1.0, p_w=0.1, clip=0.02, warmup=2, total=10)
    mbs32 = [{k: (v.float() if v.is_floating_point() else v) for k, v in mb.items()} for mb in mbs]
    tr = EagleTrainer(model, lm_head.to(DEV), lr=hp["lr"], betas=hp["betas"], eps=hp["eps"], weight_decay=hp["wd"], grad_clip=hp["clip"],
                      num_warmup_steps=hp["warmup"], num_training_steps=hp["total"], mode=mode)
    for step in range(3):
        g, tot = ots.optimizer_step(w_ref, DIMS, lm_head.float(), [mbs32], state, mode=mode, **hp)
        # keep the oracle on the bf16 parameter grid the kernels live on
        w_before = {k: v.clone() for k, v in model.state_dict().items()}
        out = tr.step([to_dev(mb) for mb in mbs])
        assert abs(out["lr"] - hp["lr"] * ots.linear_schedule(step + 1, hp["warmup"], hp["total"])) < 1e-12
        sums = out["sums"].cpu()
        assert abs(float(sums[0]) - tot["ploss"]) <= 1e-2 * abs(tot["ploss"]) and abs(float(sums[1]) - tot["vloss"]) <= 1e-2 * abs(tot["vloss"])
        assert float(sums[3]) == tot["num"] and abs(float(sums[2]) - tot["correct"]) <= 0.02 * tot["num"] + 2
        for k, ref in g.items():
            o, n = tr.offs[k]
            got = tr.grad[o:o + n].view(ref.shape)
            if mode == "tp":      # the last micro-batch's clip_grad_norm_ is fused into the AdamW kernel
                got = got.float() * min(1.0, hp["clip"] / (float(tr.grad.float().norm()) + 1e-6))
            assert cosine(got, ref) > 0.995 and rel_l2(got, ref) < 6e-2, (step, k, rel_l2(got, ref), cosine(got, ref))
        # the parameter update is AdamW on the trainer's own gradient
        if step == 0:
            for k in g:
                o, n = tr.offs[k]
                gk = tr.grad[o:o + n].view(g[k].shape).float().cpu()
                if mode == "dp":
                    gk = gk.clamp(-hp["clip"], hp["clip"])
                else:
                    gk = gk * min(1.0, hp["clip"] / (float(tr.grad.float().norm()) + 1e-6))
                p = w_before[k].float().cpu().clone()
                ots.adamw_update(p, gk, torch.zeros_like(p), torch.zeros_like(p), 1, tot["lr"], hp["betas"], hp["eps"], hp["wd"])
                now = dict(model.named_parameters())[k].detach().float().cpu()
                assert float((now - p).abs().max()) <= 2.0 ** -7 * float(p.abs().max()) + 1e-6, k
        # re-sync the oracle's parameters to the trainer's (bf16) so that steps compare like for like
        for k in w_ref:
            if k in g:
                w_ref[k] = dict(model.named_parameters())[k].detach().float().cpu().clone()
    assert tr.step_count == 3 and tr.sched_step == 3


def test_trainer_zero_token_step_is_skipped():
    """train_tp.py:77-79 skips a step without loss tokens (train.py:124 would divide by zero)."""
    from eagle_amd.train import EagleTrainer
    model, wb, lm_head = make_model(seed=9)
    mb = make_batch(1, 1, 64)
    mb["loss_mask"].zero_()
    tr = EagleTrainer(model, lm_head.to(DEV), mode="tp")
    before = tr.flat.clone()
    out = tr.step([to_dev(mb)])
    assert out.get("skipped") and torch.equal(before, tr.flat) and tr.step_count == 0


def test_sharded_lm_head_loss_single_rank_equals_unsharded():
    """ShardedLmHeadLoss with the whole vocabulary on one rank must reproduce loss_fwd_bwd (the
    exchange steps degenerate to identities); two in-process 'ranks' combined by hand must too."""
    from eagle_amd import ops, train as T
    g = torch.Generator().manual_seed(4)
    N, D, V = 200, 256, 1024
    lm = (torch.randn(V, D, generator=g) * 0.2).to(BF).to(DEV)
    pr = torch.randn(N, D, generator=g).to(BF).to(DEV)
    tg = torch.randn(N, D, generator=g).to(BF).to(DEV)
    mask = (torch.rand(N, generator=g) < 0.5).float().to(DEV)
    s1, d1 = T.loss_fwd_bwd(pr, tg, lm, ops.transpose(lm), mask, 1.0, 0.1, 0.01)
    sh = T.ShardedLmHeadLoss(lm, 0, None, 1, 0)
    s2, d2 = sh.fwd_bwd(pr, tg, mask, 1.0, 0.1, 0.01)
    assert torch.allclose(s1, s2, rtol=1e-5) and rel_l2(d2, d1) < 4e-3
    # the two-rank exchange done by hand: partial stats stacked, fp32 partial dpredict summed
    b = [T.vocab_shard_bounds(V, 2, r, align=64) for r in range(2)]
    stats, parts, dpv = [], [], None
    for v0, v1 in b:
        w = lm[v0:v1].contiguous()
        stats.append(ops.ce_partial(ops.gemm_nt(tg, w), ops.gemm_nt(pr, w), v0, pr, tg))
    sums, lse_t, lse_o, _ = ops.ce_combine(torch.stack(stats), mask)
    for v0, v1 in b:
        w = lm[v0:v1].contiguous()
        t_l, o_l = ops.gemm_nt(tg, w), ops.gemm_nt(pr, w)
        dlog, dpv = ops.ce_smoothl1_bwd(t_l, o_l, lse_t, lse_o, mask, 0.1 * 0.01, 1.0 * 0.01, pr, tg, in_place=True)
        parts.append(ops.gemm_nt(dlog, ops.transpose(w), out_dtype=torch.float32))
    d3 = ops.add_f32_bf16(parts[0] + parts[1], dpv)
    assert torch.allclose(s1, sums, rtol=1e-4) and rel_l2(d3, d1) < 4e-3


def test_trainer_over_a_real_rccl_group_of_one_rank():
    """The N > 1 code path on real RCCL (backend "nccl"), world size 1 -- the most this one-GPU box can
    run: process-group init on the device, the token-count all-reduce, the asynchronous bucketed
    AVG all-reduce of the flat bf16 gradient behind the last backward, the statistics all-reduce and
    the vocab-sharded loss's exchange.  With one rank every collective is the identity, so the
    updated parameters must equal those of the group-less trainer bit for bit."""
    import socket

    import torch.distributed as dist
    from eagle_amd import train as T
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        mbs = [to_dev(make_batch(300 + i, 2, 64, p_mask=0.6)) for i in range(2)]
        res = []
        for group in (None, dist.group.WORLD):
            model, wb, lm_head = make_model(seed=11)
            tr = T.EagleTrainer(model, lm_head.to(DEV), lr=1e-3, grad_clip=0.02, num_warmup_steps=2, num_training_steps=10,
                                mode="dp", group=group, world_size=1)
            tr.check_collectives = group is not None     # every async bucket all-reduce completed before AdamW
            for _ in range(2):
                out = tr.step(mbs)
            assert tr.collectives_launched == (len(tr._buckets) if group is not None else 0)
            torch.cuda.synchronize()
            res.append((tr.flat.clone(), out["sums"].clone(), out["num_items"]))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
        # vocab-sharded loss with a one-rank group == the unsharded loss
        model, wb, lm_head = make_model(seed=12)
        lm = lm_head.to(DEV)
        g = torch.Generator().manual_seed(5)
        pr, tg = torch.randn(96, DIMS.hidden, generator=g).to(BF).to(DEV), torch.randn(96, DIMS.hidden, generator=g).to(BF).to(DEV)
        mask = (torch.rand(96, generator=g) < 0.7).float().to(DEV)
        sh = T.ShardedLmHeadLoss(lm, 0, dist.group.WORLD, 1, 0)
        s1, d1 = sh.fwd_bwd(pr, tg, mask, 1.0, 0.1, 0.01)
        s0, d0 = T.loss_fwd_bwd(pr, tg, lm, lm.t().contiguous(), mask, 1.0, 0.1, 0.01)
        assert torch.allclose(s1, s0, rtol=1e-5, atol=1e-5) and rel_l2(d1, d0) < 1e-2
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------ fused input path
def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon et al., SC'11) on uint32 numpy arrays: ctr [n, 4], key (k0, k1) -> [n, 4]."""
    import numpy as np
    c = [ctr[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ k0) & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & MASK
        c = [n0, p1 & MASK, n2, p0 & MASK]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return np.stack(c, axis=1).astype(np.uint32)


def test_philox_known_answer():
    """Random123's published known-answer vectors for philox4x32-10: the numpy restatement the device kernel is checked
    against is itself pinned."""
    import numpy as np
    out = philox4x32_10(np.zeros((1, 4), dtype=np.uint32), (0, 0))[0]
    assert [hex(int(v)) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    out = philox4x32_10(np.full((1, 4), 0xFFFFFFFF, dtype=np.uint32), (0xFFFFFFFF, 0xFFFFFFFF))[0]
    assert [hex(int(v)) for v in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]


def test_fused_eagle_input_matches_oracle_and_philox():
    """eagle_input_embed_concat_bf16 = _make_eagle_input (train.py:256-275) + embedding gather / concat (llama2.py:515,
    :537-542) in one pass: the noise is Philox4x32-10 bit for bit, and GIVEN that noise x0 / target / ids equal the
    oracle's make_eagle_input + cat exactly; a [:, :max_len] slice of a longer hidden state needs no copy."""
    import numpy as np
    from eagle_amd import data
    B, Lfull, D, V, max_len = 3, 11, 64, 50, 9
    g = torch.Generator().manual_seed(3)
    emb = torch.randn(V, D, generator=g).to(BF)
    hid = torch.randn(B, Lfull, D, generator=g).to(BF)
    ids = torch.randint(0, V, (B, Lfull), generator=g)
    mask = (torch.rand(B, Lfull, generator=g) < 0.6).long()
    low, high, seed, offset = -0.1, 0.1, 0x1234567887654321, 7
    out = data.make_eagle_draft_input({"input_ids": ids, "loss_mask": mask}, hid.to(DEV), emb.to(DEV), max_len, low, high, seed=seed,
                                      offset=offset, want_noise=True)
    n = B * max_len * D
    e4 = np.arange(n // 4, dtype=np.uint64)
    ctr = np.stack([e4 & 0xFFFFFFFF, e4 >> 32, np.full_like(e4, offset & 0xFFFFFFFF), np.full_like(e4, offset >> 32)], axis=1).astype(np.uint32)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32)).reshape(-1)
    u = torch.from_numpy((r >> 8).astype(np.float32)) * 2.0 ** -24
    noise = torch.addcmul(torch.tensor(low), u, torch.tensor(high - low))         # fma in fp32 ...
    noise_ref = noise.to(BF).view(B, max_len, D)                                       # ... rounded to the hidden dtype
    got_noise = out["noise"].cpu()
    assert float((got_noise.float() - noise_ref.float()).abs().max()) <= 2.0 ** -10    # one bf16 ulp at 0.1 (fma vs mul+add rounding)
    assert float(got_noise.float().min()) >= low - 1e-3 and float(got_noise.float().max()) <= high + 1e-3
    assert abs(float(got_noise.float().mean())) < 2e-3 and abs(float(got_noise.float().std()) - 0.2 / 12 ** 0.5) < 2e-3
    ref = ol.make_eagle_input(ids, mask, hid, max_len, noise=got_noise)                # the oracle, given the device's noise
    assert torch.equal(out["input_ids"].cpu(), ref["input_ids"]) and torch.equal(out["loss_mask"].cpu(), ref["loss_mask"])
    assert torch.equal(out["target"].cpu(), ref["target"])
    assert torch.equal(out["hidden_states"].cpu(), ref["hidden_states"])
    x0_ref = torch.cat((emb[ref["input_ids"]], ref["hidden_states"]), dim=-1).view(B * max_len, 2 * D)
    assert torch.equal(out["x0"].cpu(), x0_ref)
    # no noise: high == low == 0 -> plain shift
    out0 = data.make_eagle_draft_input({"input_ids": ids, "loss_mask": mask}, hid.to(DEV), emb.to(DEV), max_len)
    assert torch.equal(out0["hidden_states"].cpu(), hid[:, :max_len])
    # a different offset draws a different stream
    out2 = data.make_eagle_draft_input({"input_ids": ids, "loss_mask": mask}, hid.to(DEV), emb.to(DEV), max_len, low, high, seed=seed, offset=8,
                                       want_noise=True)
    assert not torch.equal(out2["noise"], out["noise"])
    # no offset given: every call draws a fresh stream (the reference's torch.rand_like per call, eagle/train.py:272-275),
    # and the counter is restorable for resume
    data.set_noise_state(100)
    a = data.make_eagle_draft_input({"input_ids": ids, "loss_mask": mask}, hid.to(DEV), emb.to(DEV), max_len, low, high, seed=seed, want_noise=True)
    b = data.make_eagle_draft_input({"input_ids": ids, "loss_mask": mask}, hid.to(DEV), emb.to(DEV), max_len, low, high, seed=seed, want_noise=True)
    assert (a["noise_offset"], b["noise_offset"], data.noise_state()) == (100, 101, 102)
    assert not torch.equal(a["noise"], b["noise"]) and not torch.equal(a["target"], b["target"])
    data.set_noise_state(100)
    a2 = data.make_eagle_draft_input({"input_ids": ids, "loss_mask": mask}, hid.to(DEV), emb.to(DEV), max_len, low, high, seed=seed, want_noise=True)
    assert torch.equal(a2["noise"], a["noise"]) and torch.equal(a2["x0"], a["x0"])


def test_trainer_step_from_the_fused_input_equals_the_two_pass_input():
    from eagle_amd import data
    from eagle_amd.train import EagleTrainer
    res = []
    mb = make_batch(41, 2, 64)
    for fused in (False, True):
        model, wb, lm_head = make_model(seed=13)
        tr = EagleTrainer(model, lm_head.to(DEV), lr=1e-3, mode="dp")
        raw = {"input_ids": mb["input_ids"], "loss_mask": mb["loss_mask"]}
        if fused:
            d = data.make_eagle_draft_input(raw, mb["hidden_states"].to(DEV), model.embed_tokens.weight.detach(), 64)
            step_in = {"x0": d["x0"], "target": d["target"], "loss_mask": d["loss_mask"]}
        else:
            step_in = data.make_eagle_batch(raw, mb["hidden_states"].to(DEV), 64)
        out = tr.step([step_in])
        res.append((tr.flat.clone(), out["sums"].clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
