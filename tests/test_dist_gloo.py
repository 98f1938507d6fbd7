"""World-size-2 `gloo` tests (CPU) of the N > 1 HOST logic of the training drivers.

The kernels need an MI355X; here eagle_amd.ops is replaced, in the spawned test processes only, by
the torch-CPU contract stand-ins of tests/cpu_ops.py, so what is exercised is everything around
the kernels: num_items summed over ranks before the loss scale (eagle/train.py:92-93, :124), local
accumulation + per-micro-batch clip on all but the last micro-batch (:97-101, :127), the bucketed
mean all-reduce of the flat gradient, the fused statistics all-reduce (:137-148), the scheduler
stepping `world` times (accelerate/scheduler.py:73-76), and the vocab-sharded lm_head/CE exchange
(SURVEY.md 8e).  The expected values come from oracle/train_step.py with both ranks emulated
in one process."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF = torch.bfloat16


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(rank, world, port):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    import cpu_ops
    cpu_ops.install()
    return dist


def _tiny(seed=3):
    from oracle import draft_model as dm
    from eagle_amd.config import DraftConfig
    from eagle_amd.llama2 import Llama2Model
    dims = dm.DraftDims(hidden=256, inter=256, heads=2, kv_heads=2, vocab=512, max_pos=64)
    wb = {k: v.to(BF) for k, v in dm.init_weights(dims, seed=seed, std=0.05).items()}
    cfg = DraftConfig(hidden_size=256, intermediate_size=256, num_attention_heads=2, vocab_size=512, max_position_embeddings=64)
    model = Llama2Model(cfg)
    model.load_state_dict(wb)
    model = model.to(BF)
    g = torch.Generator().manual_seed(seed + 1)
    lm_head = (torch.randn(512, 256, generator=g) * 0.2).to(BF)
    return dims, wb, model, lm_head


def _batch(seed, dims, B=1, L=32):
    g = torch.Generator().manual_seed(seed)
    hid = torch.randn(B, L, dims.hidden, generator=g).to(BF)
    return {"hidden_states": hid, "input_ids": torch.randint(0, dims.vocab, (B, L), generator=g),
            "target": (0.5 * hid.float() + 0.7 * torch.randn(B, L, dims.hidden, generator=g)).to(BF),
            "loss_mask": (torch.rand(B, L, generator=g) < (0.4 + 0.3 * (seed % 2))).long()}


def _rel(a, b):
    a, b = a.float().reshape(-1), b.float().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def _f32(mb):
    return {k: (v.float() if v.is_floating_point() else v) for k, v in mb.items()}


def _dp_worker(rank, world, port, ga):
    dist = _setup(rank, world, port)
    try:
        from eagle_amd.train import EagleTrainer
        from oracle import train_step as ots
        dims, wb, model, lm_head = _tiny()
        all_mbs = [[_batch(10 * r + i, dims) for i in range(ga)] for r in range(world)]   # rank r sees all_mbs[r]
        hp = dict(lr=1e-3, betas=(0.9, 0.95), eps=1e-8, wd=0.01, v_w=1.0, p_w=0.1, clip=0.01, warmup=4, total=40)
        tr = EagleTrainer(model, lm_head, lr=hp["lr"], betas=hp["betas"], eps=hp["eps"], weight_decay=hp["wd"], grad_clip=hp["clip"],
                          num_warmup_steps=hp["warmup"], num_training_steps=hp["total"], mode="dp", group=dist.group.WORLD,
                          world_size=world)
        w_ref = {k: v.float() for k, v in wb.items()}
        state = {}
        for step in range(2):
            g, tot = ots.optimizer_step(w_ref, dims, lm_head.float(), [[_f32(mb) for mb in r] for r in all_mbs], state, mode="dp", **hp)
            out = tr.step(all_mbs[rank])
            assert out["num_items"] == tot["num"]                                   # summed over BOTH ranks
            s = out["sums"]
            assert abs(float(s[0]) - tot["ploss"]) <= 1e-2 * abs(tot["ploss"]) and abs(float(s[1]) - tot["vloss"]) <= 1e-2 * abs(tot["vloss"])
            assert float(s[3]) == tot["num"] and abs(float(s[2]) - tot["correct"]) <= 2
            for k, ref in g.items():
                o, n = tr.offs[k]
                got = tr.grad[o:o + n].view(ref.shape)
                assert _rel(got, ref) < 5e-2, (step, k, _rel(got, ref))
            # identical parameters on every rank after the step
            mine = tr.flat.float().clone()
            other = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(other, mine)
            assert all(torch.equal(o_, mine) for o_ in other)
            assert tr.sched_step == world * (step + 1) and abs(out["lr"] - hp["lr"] * ots.linear_schedule(world * (step + 1), 4, 40)) < 1e-12
            for k in g:                                                            # keep the oracle on the trainer's bf16 grid
                w_ref[k] = dict(model.named_parameters())[k].detach().float().clone()
    finally:
        dist.destroy_process_group()


def _sharded_worker(rank, world, port):
    dist = _setup(rank, world, port)
    try:
        from eagle_amd import train as T
        dims, wb, model, lm_head = _tiny(seed=5)
        g = torch.Generator().manual_seed(77)                                       # SAME rows on every rank
        N, D, V = 48, dims.hidden, dims.vocab
        pr, tg = torch.randn(N, D, generator=g).to(BF), torch.randn(N, D, generator=g).to(BF)
        mask = (torch.rand(N, generator=g) < 0.6).float()
        lm = lm_head.clone()
        lm[300] = lm[40]                                                            # identical rows in different shards: ties across ranks
        v0, v1 = T.vocab_shard_bounds(V, world, rank, align=64)
        assert (v0, v1) == ((0, 256), (256, 512))[rank]
        sh = T.ShardedLmHeadLoss(lm[v0:v1], v0, dist.group.WORLD, world, rank)
        sums, dpred = sh.fwd_bwd(pr, tg, mask, 1.0, 0.1, 0.02)
        full_sums, full_dpred = T.loss_fwd_bwd(pr, tg, lm, lm.t().contiguous(), mask, 1.0, 0.1, 0.02)
        assert torch.allclose(sums, full_sums, rtol=1e-4, atol=1e-4), (sums, full_sums)
        assert float(sums[2]) == float(full_sums[2])                                # agreement count: lowest GLOBAL index wins ties
        assert _rel(dpred, full_dpred) < 1e-2
        # and through the trainer in 'tp' mode: every rank ends with the same parameters
        tr = T.EagleTrainer(model, None, mode="tp", sharded_loss=sh, grad_clip=0.5)
        out = tr.step([_batch(5, dims)])
        mine = tr.flat.float().clone()
        other = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(other, mine)
        assert all(torch.equal(o_, mine) for o_ in other) and tr.step_count == 1
    finally:
        dist.destroy_process_group()


def _ddp_worker(rank, world, port):
    """The reference's own loop body (eagle/train.py:72, :92-101, :105-131) around the drop-in model: DistributedDataParallel
    (what accelerate's prepare_model wraps it in), no_sync on all but the last micro-batch, the loss scaling of :124,
    accelerate's backward (/ GA), clip_grad_value_ after every backward, AdamW."""
    import contextlib
    dist = _setup(rank, world, port)
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP
        from eagle_amd.train import eagle_loss
        from oracle import train_step as ots
        dims, wb, model, lm_head = _tiny(seed=9)
        model.train()
        ddp = DDP(model)                                                            # train.py:72 (accelerator.prepare_model)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.95))     # train.py:66
        GA, clip = 2, 0.01
        all_mbs = [[_batch(20 * r + i, dims) for i in range(GA)] for r in range(world)]
        mbs = all_mbs[rank]
        n = sum(mb["loss_mask"].sum() for mb in mbs).clone()
        dist.all_reduce(n)                                                          # train.py:92-93
        num_items = float(n)
        holder = {}
        for i, mb in enumerate(mbs):
            ctx = ddp.no_sync if (i < len(mbs) - 1 and world > 1) else contextlib.nullcontext   # train.py:97-100
            with ctx():
                predict = ddp(mb["hidden_states"], input_ids=mb["input_ids"])       # train.py:105
                loss = eagle_loss(predict, mb["target"], lm_head, mb["loss_mask"], 1.0, 0.1, holder)
                loss = (loss * GA * world) / num_items                              # train.py:124
                (loss / GA).backward()                                              # accelerator.backward divides by GA
                torch.nn.utils.clip_grad_value_(model.parameters(), clip)           # train.py:127
        g, tot = ots.accumulate_grads({k: v.float() for k, v in wb.items()}, dims, lm_head.float(),
                                      [[_f32(mb) for mb in r] for r in all_mbs], "dp", 1.0, 0.1, clip)
        assert tot["num"] == num_items
        named = dict(model.named_parameters())
        assert named["embed_tokens.weight"].grad is None
        for k, ref in g.items():
            assert named[k].grad is not None and _rel(named[k].grad, ref) < 5e-2, (k, _rel(named[k].grad, ref))
        opt.step()
        opt.zero_grad()
        mine = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()])
        other = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(other, mine)
        assert all(torch.equal(o_, mine) for o_ in other)                           # identical parameters on every rank
        moved = max(float((named[k].detach().float() - wb[k].float()).abs().max()) for k in g)
        assert moved > 0
    finally:
        dist.destroy_process_group()


def test_reference_loop_under_ddp_two_ranks_gloo():
    mp.spawn(_ddp_worker, args=(2, _free_port()), nprocs=2, join=True)


@pytest.mark.parametrize("ga", [1, 2])
def test_dp_trainer_two_ranks_gloo(ga):
    mp.spawn(_dp_worker, args=(2, _free_port(), ga), nprocs=2, join=True)


def test_vocab_sharded_loss_two_ranks_gloo():
    mp.spawn(_sharded_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_vocab_shard_bounds_cover_the_vocabulary():
    from eagle_amd.train import vocab_shard_bounds
    for V, world in [(151936, 8), (32000, 8), (32000, 3), (1000, 4), (512, 2)]:
        b = [vocab_shard_bounds(V, world, r) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == V and all(x[1] == y[0] for x, y in zip(b, b[1:]))
        assert all((x[1] - x[0]) % 64 == 0 for x in b[:-1])
    assert vocab_shard_bounds(151936, 8, 0) == (0, 19008) and vocab_shard_bounds(151936, 8, 7) == (133056, 151936)


def test_bucket_layout_tiles_the_flat_buffer():
    """The all-reduce buckets are contiguous slices of the flat gradient in backward order."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import cpu_ops
    from eagle_amd import ops
    saved = {k: getattr(ops, k) for k in ("transpose", "rope_tables")}
    try:
        cpu_ops.install()
        from eagle_amd.train import EagleTrainer, PARAM_ORDER
        dims, wb, model, lm_head = _tiny()
        tr = EagleTrainer(model, lm_head, mode="dp")
        assert [b[0] for b in tr._buckets] == ["wdown", "wn", "wo", "wqkv", "fc"]
        assert tr._buckets[0][1] == 0 and tr._buckets[-1][2] == tr.flat.numel()
        for name, p in model.named_parameters():
            if p.requires_grad:
                o, n = tr.offs[name]
                assert p.data_ptr() == tr.flat.data_ptr() + 2 * o and o % 128 == 0      # parameters are views of the flat buffer
        assert list(tr.offs) == [n for n in PARAM_ORDER]
    finally:
        importlib.reload(ops)
