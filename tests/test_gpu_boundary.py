"""GPU tests of the drop-in boundary's stateful parts: the static-tree builder behind the C ABI, the KV views
``Llama2Model.forward`` hands out (eagle/llama2.py:236-239, :581-585 return fresh tensors; here views of a buffer),
the derived-weight caches across optimizer steps, every selectable kernel family of the weight-stream GEMM, and the
fused loss with -inf logits."""
import pytest
import torch

from oracle import draft_model as dm
from oracle import loss as ol
from oracle import tree as ot

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DEV = "cuda:0"


def _close(a, b, rtol=1e-2, what=""):
    a, b = a.float().cpu(), b.float().cpu()
    tol = 2e-2 * float(b.abs().mean()) + float(b.abs().max()) / 256.0 + rtol * b.abs()
    err = (a - b).abs()
    bad = int((err > tol).sum())
    assert bad <= 1e-3 * a.numel() and float((err / tol).max()) < 3.0, f"{what}: {bad}/{a.numel()} outside tol, max err {float(err.max()):.4g}"


def _random_paths(g, topk, max_depth, max_nodes):
    paths = {(int(torch.randint(0, topk, (1,), generator=g)),)}
    for _ in range(40 * max_nodes):                 # bounded: a small top-k / depth cannot hold max_nodes paths
        if len(paths) >= max_nodes:
            break
        parent = sorted(paths)[int(torch.randint(0, len(paths), (1,), generator=g))]
        if torch.rand(1, generator=g) < 0.3:
            paths.add((int(torch.randint(0, topk, (1,), generator=g)),))
        elif len(parent) < max_depth:
            paths.add(parent + (int(torch.randint(0, topk, (1,), generator=g)),))
    return [list(p) for p in paths]


def _assert_tree_equals_oracle(h, o):
    assert (h.n_nodes, h.parent, h.depth, h.choice, h.flat_row) == (o.n_nodes, o.parent, o.depth, o.choice, o.flat_row)
    assert h.max_depth == max(o.depth) and len(h.steps) == len(o.step_nodes)
    for d, st in enumerate(h.steps):
        assert st.nodes == o.step_nodes[d]
        assert st.tree_indices.cpu().tolist() == o.step_tree_indices[d].tolist()
        assert st.parent_row.cpu().tolist() == o.step_parent_row[d].tolist()
        assert st.repeat_nums == o.step_repeat_nums[d]
        assert torch.equal(st.attn_mask.cpu().float(), o.step_attn_mask[d].float())
        assert torch.equal(st.attn_mask_u8.cpu().float(), o.step_attn_mask[d][0, 0].float())
        assert st.position_ids.cpu().tolist() == o.step_position_ids[d].tolist()
    assert torch.equal(h.tree_attn_mask.cpu().float(), o.tree_attn_mask.float())
    assert h.tree_indices.cpu().tolist() == o.tree_indices.tolist()
    assert h.tree_position_ids.cpu().tolist() == o.tree_position_ids.tolist()
    assert h.retrieve_indices.cpu().tolist() == o.retrieve_indices.tolist()


@pytest.mark.parametrize("expand", ["wchild", "all"])
def test_tree_build_abi_matches_oracle_on_the_25_path_tree(expand):
    """eagle_tree_layout + eagle_tree_build (what Llama2Model.init_tree's generate_tree_buffers resolves to,
    eagle/llama2.py:453-455) against oracle.generate_tree_buffers, every a19 / a21 buffer bit for bit."""
    from eagle_amd.tree import TreeBuffers, generate_tree_buffers
    h = TreeBuffers.build_on_device(expand=expand, device=DEV)
    _assert_tree_equals_oracle(h, ot.generate_tree_buffers(expand=expand))
    assert generate_tree_buffers(device=DEV, expand=expand).tree_indices.is_cuda


def test_tree_build_abi_matches_oracle_on_random_trees():
    from eagle_amd.tree import TreeBuffers
    for seed in range(16):
        g = torch.Generator().manual_seed(100 + seed)
        topk = int(torch.randint(3, 11, (1,), generator=g))
        paths = _random_paths(g, topk, int(torch.randint(2, 7, (1,), generator=g)), int(torch.randint(3, 60, (1,), generator=g)))
        for expand in ("wchild", "all"):
            h = TreeBuffers.build_on_device(paths, topk=topk, expand=expand, device=DEV)
            _assert_tree_equals_oracle(h, ot.generate_tree_buffers(paths, topk=topk, expand=expand))
    for bad in ([[0], [1, 0]], [[0], [10]], [[0], [0]], [[0, 0]]):
        with pytest.raises(ValueError):
            TreeBuffers.build_on_device(bad, device=DEV)


def test_init_tree_fills_device_buffers_with_the_upstream_keys():
    m, _, _ = _tiny_model()
    m.init_tree()
    o = ot.generate_tree_buffers()
    assert [t.cpu().tolist() for t in m.tree_buffer["tree_indices"]] == [t.tolist() for t in o.step_tree_indices]
    assert m.tree_buffer["repeat_nums"] == o.step_repeat_nums
    assert all(a.is_cuda for a in m.tree_buffer["attn_mask"])


# ------------------------------------------------------------------ KV views
DIMS = dm.DraftDims(hidden=256, inter=512, heads=2, kv_heads=2, vocab=500)


def _tiny_model(seed=3, dims=DIMS):
    from eagle_amd.config import DraftConfig
    from eagle_amd.llama2 import Llama2Model
    w = dm.init_weights(dims, seed=seed, std=0.05)
    cfg = DraftConfig(hidden_size=dims.hidden, intermediate_size=dims.inter, num_attention_heads=dims.heads,
                      num_key_value_heads=dims.kv_heads, vocab_size=dims.vocab)
    m = Llama2Model(cfg)
    m.load_state_dict(w, strict=True)
    return m.to(BF).to(DEV).eval(), {k: v.to(BF) for k, v in w.items()}, dims


def test_interleaved_sequences_keep_their_kv_like_the_reference():
    """Two sequences A and B advanced alternately through one model, plus a drafter created in between: every
    continuation must see its own past (the reference's torch.cat results are independent tensors)."""
    from eagle_amd.speculative import EagleDrafter
    m, wb, dims = _tiny_model()
    g = torch.Generator().manual_seed(21)
    def rnd(L):
        return torch.randn(1, L, dims.hidden, generator=g).to(BF), torch.randint(0, dims.vocab, (1, L), generator=g)
    (hA, iA), (hB, iB), (hA2, iA2), (hB2, iB2), (hA3, iA3) = rnd(9), rnd(5), rnd(3), rnd(2), rnd(1)
    with torch.no_grad():
        rA, rkvA = dm.forward(wb, dims, hA, iA, use_cache=True)
        rB, rkvB = dm.forward(wb, dims, hB, iB, use_cache=True)
        rA2, rkvA2 = dm.forward(wb, dims, hA2, iA2, past_key_values=rkvA, use_cache=True)
        rB2, rkvB2 = dm.forward(wb, dims, hB2, iB2, past_key_values=rkvB, use_cache=True)
        rA3, _ = dm.forward(wb, dims, hA3, iA3, past_key_values=rkvA2, use_cache=True)
        d = DEV
        oA, kvA = m(hA.to(d), iA.to(d), use_cache=True)
        oB, kvB = m(hB.to(d), iB.to(d), use_cache=True)              # a fresh prompt must not overwrite A's rows
        lm_head = (torch.randn(dims.vocab, dims.hidden, generator=g) * 0.2).to(BF).to(d)
        drafter = EagleDrafter(m, lm_head, max_ctx=64)               # ... nor may a drafter's cache share them
        drafter.prefill(hB.to(d), torch.cat([iB[:, :1], iB], dim=1).to(d))
        oA2, kvA2 = m(hA2.to(d), iA2.to(d), past_key_values=kvA, use_cache=True)
        oB2, kvB2 = m(hB2.to(d), iB2.to(d), past_key_values=kvB, use_cache=True)
        oA3, kvA3 = m(hA3.to(d), iA3.to(d), past_key_values=kvA2, use_cache=True)
    for got, ref, what in ((oA, rA, "A"), (oB, rB, "B"), (oA2, rA2, "A2"), (oB2, rB2, "B2"), (oA3, rA3, "A3")):
        _close(got, ref, what=what)
    _close(kvA2[0][0], rkvA2[0][0], what="A2.k"); _close(kvB2[0][1], rkvB2[0][1], what="B2.v")
    assert tuple(kvA3[0][0].shape) == (1, dims.kv_heads, 13, dims.hidden // dims.heads)


def test_stale_kv_view_raises_and_a_clone_survives():
    """Restarting a buffer from a shorter past (what tree expansion does) overwrites the rows a longer view holds:
    handing that longer view back must raise, never attend to foreign rows; a clone() taken before is copied in."""
    m, wb, dims = _tiny_model()
    g = torch.Generator().manual_seed(22)
    d = DEV
    def rnd(L):
        return torch.randn(1, L, dims.hidden, generator=g).to(BF), torch.randint(0, dims.vocab, (1, L), generator=g)
    (h0, i0), (h1, i1), (h2, i2), (h3, i3) = rnd(6), rnd(4), rnd(3), rnd(2)
    with torch.no_grad():
        r0, rkv0 = dm.forward(wb, dims, h0, i0, use_cache=True)
        r1, rkv1 = dm.forward(wb, dims, h1, i1, past_key_values=rkv0, use_cache=True)
        r2, rkv2 = dm.forward(wb, dims, h2, i2, past_key_values=rkv0, use_cache=True)      # branch from the short past
        r3, _ = dm.forward(wb, dims, h3, i3, past_key_values=rkv1, use_cache=True)         # continue the long branch
        o0, kv0 = m(h0.to(d), i0.to(d), use_cache=True)
        o1, kv1 = m(h1.to(d), i1.to(d), past_key_values=kv0, use_cache=True)
        keep = ((kv1[0][0].clone(), kv1[0][1].clone()),)
        o2, kv2 = m(h2.to(d), i2.to(d), past_key_values=kv0, use_cache=True)               # overwrites rows 6..9 of kv1
        _close(o2, r2, what="branch from the shorter past")
        with pytest.raises(ValueError, match="overwrote"):
            m(h3.to(d), i3.to(d), past_key_values=kv1, use_cache=True)
        o3, _ = m(h3.to(d), i3.to(d), past_key_values=keep, use_cache=True)
        _close(o3, r3, what="continuation from the clone")
        o2b, _ = m(h3.to(d), i3.to(d), past_key_values=kv2, use_cache=True)                # kv2 lives in an older buffer: copied
        rb, _ = dm.forward(wb, dims, h3, i3, past_key_values=rkv2, use_cache=True)
        _close(o2b, rb, what="continuation of the short branch")


def test_forward_without_cache_reuses_its_buffer_and_views_still_survive():
    """A loop of forward() calls that hand out no cache views (use_cache=False: validation batches) restarts one KV buffer
    in place -- same storage, right answers -- while a view handed out by use_cache=True still gets a fresh buffer on the
    next prompt (the reference's independent torch.cat results)."""
    m, wb, dims = _tiny_model()
    g = torch.Generator().manual_seed(23)
    d = DEV
    ptrs = set()
    with torch.no_grad():
        for L in (7, 12, 3):
            h, i = torch.randn(2, L, dims.hidden, generator=g).to(BF), torch.randint(0, dims.vocab, (2, L), generator=g)
            _close(m(h.to(d), i.to(d)), dm.forward(wb, dims, h, i), what=f"no-cache forward L={L}")
            ptrs.add(m._caches[0].k.data_ptr())
        assert len(ptrs) == 1, "forward(use_cache=False) allocated a new KV buffer per call"
        h, i = torch.randn(2, 5, dims.hidden, generator=g).to(BF), torch.randint(0, dims.vocab, (2, 5), generator=g)
        o, kv = m(h.to(d), i.to(d), use_cache=True)                     # reuses the buffer too, then hands out a view of it
        assert kv[0][0].data_ptr() in ptrs
        k_before = kv[0][0].clone()
        h2, i2 = torch.randn(2, 9, dims.hidden, generator=g).to(BF), torch.randint(0, dims.vocab, (2, 9), generator=g)
        m(h2.to(d), i2.to(d))                                           # a fresh prompt while the view is out: new buffer
        assert m._caches[0].k.data_ptr() not in ptrs and torch.equal(kv[0][0], k_before)


# ------------------------------------------------------------------ derived weights across optimizer steps
def test_eval_forward_after_trainer_step_uses_the_updated_weights():
    """EagleTrainer's fused AdamW writes the parameters through raw pointers (no Tensor._version bump): the decode
    path's stacked / norm-folded copies must be rebuilt, so that eval forward == oracle forward on the new weights."""
    from eagle_amd.config import DraftConfig
    from eagle_amd.llama2 import Llama2Model
    from eagle_amd.train import EagleTrainer
    dims = dm.DraftDims(hidden=256, inter=512, heads=2, kv_heads=2, vocab=1024, max_pos=256)
    w = dm.init_weights(dims, seed=5, std=0.05)
    cfg = DraftConfig(hidden_size=dims.hidden, intermediate_size=dims.inter, num_attention_heads=dims.heads,
                      vocab_size=dims.vocab, max_position_embeddings=dims.max_pos)
    m = Llama2Model(cfg)
    m.load_state_dict({k: v.to(BF) for k, v in w.items()})
    m = m.to(BF).to(DEV)
    g = torch.Generator().manual_seed(6)
    lm_head = (torch.randn(dims.vocab, dims.hidden, generator=g) * 0.2).to(BF).to(DEV)
    tr = EagleTrainer(m, lm_head, lr=5e-2, grad_clip=0.5, mode="dp")     # large lr: stale copies would be far off
    h = torch.randn(1, 12, dims.hidden, generator=g).to(BF)
    ids = torch.randint(0, dims.vocab, (1, 12), generator=g)

    def check(what):
        m.eval()
        with torch.no_grad():
            got, kv = m(h.to(DEV), ids.to(DEV), use_cache=True)
            wb = {k: v.detach().cpu() for k, v in m.state_dict().items()}
            ref, rkv = dm.forward(wb, dims, h, ids, use_cache=True)
        _close(got, ref, what=what)
        _close(kv[0][0], rkv[0][0], what=what + " k")
        return got.float().cpu()

    before = check("before any step")
    for step in range(2):
        mb = {"hidden_states": torch.randn(2, 64, dims.hidden, generator=g).to(BF).to(DEV),
              "input_ids": torch.randint(0, dims.vocab, (2, 64), generator=g).to(torch.int32).to(DEV),   # int32 ids are coerced
              "target": torch.randn(2, 64, dims.hidden, generator=g).to(BF).to(DEV),
              "loss_mask": torch.ones(2, 64, dtype=torch.long, device=DEV)}
        m.train()
        tr.step([mb])
        after = check(f"after step {step + 1}")
        assert float((after - before).abs().max()) > 1e-2            # the weights did move
        before = after


# ------------------------------------------------------------------ every selectable weight-stream kernel family
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(4, 4096, 4096), (8, 4096, 8192), (3, 11008, 4096), (16, 1000, 4096), (5, 300, 1024)])
def test_gemm_skinny_every_kernel_family(variant, M, N, K):
    from eagle_amd import _hip, ops
    g = torch.Generator().manual_seed(M * 7 + N + K)
    x = torch.randn(M, K, generator=g).to(BF)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF)
    W2 = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF)
    b = torch.randn(N, generator=g).to(BF)
    r = torch.randn(M, N, generator=g).to(BF)
    ref = x.float() @ W.float().t()
    ref2 = (ref + b.float()).to(BF).float() + r.float()
    gt, up = ref.to(BF), (x.float() @ W2.float().t()).to(BF)
    ref3 = torch.nn.functional.silu(gt) * up
    try:
        _hip.lib().eagle_debug_set_skinny_variant(variant, 0, 0)
        for resident in (False, True):
            _close(ops.linear_skinny(x.to(DEV), W.to(DEV), resident=resident), ref, what=f"plain v{variant}")
            _close(ops.linear_skinny(x.to(DEV), W.to(DEV), bias=b.to(DEV), res=r.to(DEV), resident=resident), ref2, what="bias+res")
            _close(ops.linear_skinny(x.to(DEV), W.to(DEV), W2=W2.to(DEV), resident=resident), ref3, what="silu*up")
    finally:
        _hip.lib().eagle_debug_set_skinny_variant(3, 0, 0)


# ------------------------------------------------------------------ fused loss with -inf logits
def test_fused_loss_with_minus_inf_columns():
    """Masked / padded vocabulary columns set to -inf in BOTH logit tensors (reachable through the C ABI): a whole
    8-wide chunk, a 1024-wide run and scattered entries.  exp(-inf) = 0 everywhere, no NaN; equals the oracle on the
    remaining columns."""
    from eagle_amd import ops
    g = torch.Generator().manual_seed(31)
    N, V, D = 6, 4099, 64
    t = (torch.randn(N, V, generator=g) * 3).to(BF)
    o = (torch.randn(N, V, generator=g) * 3).to(BF)
    dead = torch.zeros(V, dtype=torch.bool)
    dead[0:8] = True; dead[1024:2048] = True; dead[torch.randint(0, V, (200,), generator=g)] = True; dead[-3:] = True
    t[:, dead] = float("-inf"); o[:, dead] = float("-inf")
    predict = torch.randn(N, D, generator=g).to(BF)
    target = torch.randn(N, D, generator=g).to(BF)
    mask = torch.tensor([1, 1, 0, 1, 1, 1], dtype=torch.float32)
    ex = ol.exact_loss_from_logits(t[:, ~dead], o[:, ~dead], predict, target, mask)
    stats = ops.ce_partial(t.to(DEV), o.to(DEV), 0, predict.to(DEV), target.to(DEV))
    sums, lse_t, lse_o, rows = ops.ce_combine(stats, mask.to(DEV))
    assert torch.isfinite(sums).all() and torch.isfinite(rows).all()
    assert abs(float(sums[0]) - float(ex["ploss"])) <= 1e-3 * abs(float(ex["ploss"]))
    assert int(sums[2]) == int(ex["correct"])
    _close(lse_t, ex["lse_t"], rtol=1e-4, what="lse_t"); _close(lse_o, ex["lse_o"], rtol=1e-4, what="lse_o")
    dlog, _ = ops.ce_smoothl1_bwd(t.to(DEV), o.to(DEV), lse_t, lse_o, mask.to(DEV), 0.1, 1.0, predict.to(DEV), target.to(DEV))
    assert torch.isfinite(dlog.float()).all() and not dlog[:, dead.to(DEV)].any() and not dlog[2].any()
    rl, _ = ol.exact_grads_from_logits(t[:, ~dead], o[:, ~dead], predict, target, mask, 0.1, 1.0)
    _close(dlog[:, (~dead).to(DEV)], rl, what="dlogits on the live columns")
