"""GPU parity of the BATCHED draft/verify cycle (BASELINE.json configs[3]: batch of requests with
ShareGPT-like ragged prompt lengths, independent per request, no collective).

The batch-1 cycle is pinned against the oracle loop in test_gpu_parity.py.  Here every request of
a ragged batch must reproduce what a batch-1 drafter produces for that request alone: top-k ids,
tree tokens, acceptance lengths, next tokens, KV compaction -- bit-exact -- through several
closed-loop cycles with different acceptance lengths per request.  At the tiny test dims every row
of the weight-stream GEMM and of the split-KV attention accumulates in an order that does not
depend on how many other rows ride in the launch, so the comparison can be exact."""
import pytest
import torch

from oracle import draft_model as dm
from oracle import tree as ot

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def dev():
    return torch.device("cuda:0")


def close(a, b, rtol=1e-2, what=""):
    a, b = a.float().cpu(), b.float().cpu()
    tol = 2e-2 * float(b.abs().mean()) + float(b.abs().max()) / 256.0 + rtol * b.abs()
    err = (a - b).abs()
    bad = int((err > tol).sum())
    assert bad <= 1e-3 * a.numel() and float((err / tol).max()) < 3.0, f"{what}: {bad}/{a.numel()} outside tol, max err {float(err.max()):.4g}"


def _model(dims, w, rows_min=None):
    from eagle_amd.config import DraftConfig
    from eagle_amd.llama2 import Llama2Model
    cfg = DraftConfig(hidden_size=dims.hidden, intermediate_size=dims.inter, num_attention_heads=dims.heads,
                      num_key_value_heads=dims.kv_heads, vocab_size=dims.vocab)
    m = Llama2Model(cfg)
    m.load_state_dict(w, strict=True)
    m = m.to(BF).to(dev()).eval()
    if rows_min is not None:
        m.rows_min = rows_min
    return m


def _plant(tgt_logits, tree_tok, tb, depth):
    node = 0
    for _ in range(depth):
        kids = [i for i in range(tb.n_nodes) if tb.parent[i] == node]
        if not kids:
            break
        kid = kids[-1] if len(kids) > 1 and depth % 2 else kids[0]
        tgt_logits[node, int(tree_tok[kid])] = 50.0
        node = kid


@pytest.mark.parametrize("expand", ["wchild", "all"])
def test_ragged_batch_matches_per_request_drafters(expand):
    from eagle_amd.speculative import EagleDrafter
    d = dev()
    dims = dm.DraftDims(hidden=256, inter=512, heads=2, kv_heads=2, vocab=2000)
    w = dm.init_weights(dims, seed=5, std=0.08)
    g = torch.Generator().manual_seed(77)
    lm_head = (torch.randn(dims.vocab, dims.hidden, generator=g) * 0.3).to(BF).to(d)
    ctxs = [37, 9, 64, 21, 130]
    B = len(ctxs)
    tb = ot.generate_tree_buffers(expand=expand)
    n_nodes = tb.n_nodes
    # the batch (5 x 4 = 20 rows per step) runs on the row-batch weight stream; the per-request drafters are put on the
    # same kernel family (rows_min = 1), where a row's result does not depend on the rows beside it: bit-for-bit equality.
    # Across kernel families the K summation order differs; test_batch_1_kernel_families_agree bounds that difference.
    batched = EagleDrafter(_model(dims, w, 1), lm_head, expand=expand, batch=B, max_ctx=256)
    singles = [EagleDrafter(_model(dims, w, 1), lm_head, expand=expand, batch=1, max_ctx=256) for _ in range(B)]
    hid = [torch.randn(c, dims.hidden, generator=g).to(BF).to(d) for c in ctxs]
    tok = [torch.randint(0, dims.vocab, (c + 1,), generator=g).to(d) for c in ctxs]
    with torch.no_grad():
        ids_b, vals_b = batched.prefill_ragged(hid, tok)
        for b in range(B):
            i1, v1 = singles[b].prefill(hid[b][None], tok[b][None])
            assert torch.equal(ids_b[b], i1[0]), f"prefill: request {b} top-k ids differ from its batch-1 run"
            assert torch.equal(vals_b[b], v1[0])
        assert batched.stable_len.tolist() == ctxs

        prev = torch.tensor(ctxs, dtype=torch.int32, device=d)
        root = torch.stack([t[-1] for t in tok])
        kv_b = torch.randn(2, B, 2, 256, 16, generator=g).to(BF).to(d)          # [outer, B, inner, cap, hd]
        kv_s = [kv_b[:, b:b + 1].clone() for b in range(B)]
        seen = set()
        for cycle in range(4):
            tgt_logits = (torch.randn(B, n_nodes, dims.vocab, generator=g) * 2).to(BF)
            tgt_hidden = torch.randn(B, n_nodes, dims.hidden, generator=g).to(BF).to(d)
            expect = []
            for b in range(B):
                tree_tok, cart = ot.tree_candidates(int(root[b]), batched.flat_ids[b].cpu(), tb)
                _plant(tgt_logits[b], tree_tok, tb, (cycle + b) % 5)
                best, alen, nxt, _ = ot.greedy_accept(cart, ot.first_argmax(tgt_logits[b]), tb)
                expect.append([best, alen, nxt, int(prev[b]) + alen + 1])
                seen.add(alen)
            tgt_logits = tgt_logits.to(d)
            res = batched.verify(root, tgt_logits, prev)
            assert res.tolist() == expect, f"cycle {cycle}: acceptance differs from the oracle"
            acc_b = batched.compact(prev, kv_b, tgt_hidden, kv_inner=2)
            for b in range(B):
                p1 = prev[b:b + 1].clone()
                r1 = singles[b].verify(root[b:b + 1], tgt_logits[b:b + 1], p1)
                assert r1[0].tolist() == expect[b]
                a1 = singles[b].compact(p1, kv_s[b], tgt_hidden[b:b + 1], kv_inner=2)
                n = expect[b][1] + 1
                assert torch.equal(acc_b[b, :n], a1[0, :n])
                assert torch.equal(kv_b[:, b], kv_s[b][:, 0]), "verifier KV compaction differs"
                assert torch.equal(batched.next_ids[b, :n], singles[b].next_ids[0, :n])
            ids_b, vals_b = batched.draft(acc_b, batched.next_ids)
            for b in range(B):
                i1, v1 = singles[b].draft(singles[b].acc_hidden, singles[b].next_ids)
                assert torch.equal(ids_b[b], i1[0]), f"cycle {cycle}: request {b} top-k ids differ"
                assert torch.equal(vals_b[b], v1[0])
                assert int(batched.stable_len[b]) == int(singles[b].stable_len[0])
            root = res[:, 2].clone()
            prev = res[:, 3].to(torch.int32)
        assert len(seen) >= 3, "the planted verifier should exercise several acceptance lengths"
        batched.check_errors()


def test_batch_1_kernel_families_agree():
    """The same prompt and expansion through the per-step weight-stream kernels (default at batch 1) and through the
    row-batch weight stream (rows_min = 1): top-k values within the bf16 tolerance of close()."""
    from eagle_amd.speculative import EagleDrafter
    d = dev()
    dims = dm.DraftDims(hidden=256, inter=512, heads=2, kv_heads=2, vocab=2000)
    w = dm.init_weights(dims, seed=5, std=0.08)
    g = torch.Generator().manual_seed(78)
    lm_head = (torch.randn(dims.vocab, dims.hidden, generator=g) * 0.3).to(BF).to(d)
    hid = torch.randn(1, 23, dims.hidden, generator=g).to(BF).to(d)
    tok = torch.randint(0, dims.vocab, (1, 24), generator=g).to(d)
    a = EagleDrafter(_model(dims, w), lm_head, batch=1, max_ctx=128)
    b = EagleDrafter(_model(dims, w, 1), lm_head, batch=1, max_ctx=128)
    with torch.no_grad():
        ia, va = a.prefill(hid, tok)
        ib, vb = b.prefill(hid, tok)
    close(vb[0, :1], va[0, :1], what="first top-k row, two kernel families")
    assert torch.equal(ia[0, 0, :3], ib[0, 0, :3])          # well-separated leading candidates


def test_batch32_real_dims_cycle_is_self_consistent():
    """configs[3] at full size (Qwen3-8B draft dims, V = 151936, batch 32, ragged prompts).  The CPU
    oracle is too slow for this size, so the check is by properties: (1) the catch-up top-k VALUES of
    a sampled request agree with a batch-1 run within bf16 tolerance (ids can legitimately permute
    among bf16-tied logits when the accumulation order changes with the launch shape); (2) given the
    drafter's own top-k table, acceptance / next token / lengths are bit-exact against the oracle."""
    from eagle_amd.config import DraftConfig, QWEN3_8B_DRAFT
    from eagle_amd.llama2 import Llama2Model
    from eagle_amd.speculative import EagleDrafter
    d = dev()
    cfg = DraftConfig(**QWEN3_8B_DRAFT)
    with torch.device("meta"):
        model = Llama2Model(cfg)
    model = model.to_empty(device=d).to(BF).eval()
    gd = torch.Generator(device=d).manual_seed(0)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("layernorm.weight"):
                p.fill_(1.0)
            else:
                p.copy_((torch.randn(p.shape, generator=gd, device=d, dtype=torch.float32) * 0.02).to(BF))
        lm_head = (torch.randn((cfg.vocab_size, cfg.hidden_size), generator=gd, device=d) * 0.02).to(BF)
    B = 32
    g = torch.Generator().manual_seed(3)
    ctxs = [int(min(600, max(8, x))) for x in torch.exp(torch.randn(B, generator=g) * 0.8 + 4.5).tolist()]
    hid = [torch.randn(c, cfg.hidden_size, generator=g).to(BF).to(d) for c in ctxs]
    tok = [torch.randint(0, cfg.vocab_size, (c + 1,), generator=g).to(d) for c in ctxs]
    tb = ot.generate_tree_buffers()
    with torch.no_grad():
        batched = EagleDrafter(model, lm_head, batch=B, max_ctx=768)
        ids_b, vals_b = batched.prefill_ragged(hid, tok)
        ids_b, vals_b = ids_b.clone(), vals_b.clone()
        single = EagleDrafter(model, lm_head, batch=1, max_ctx=768)
        for b in (0, 17, 31):
            _, v1 = single.prefill(hid[b][None], tok[b][None])
            a, r = vals_b[b, 0].float(), v1[0, 0].float()
            assert float((a - r).abs().max()) <= 1e-2 * float(r.abs().max()) + float(r.abs().max()) / 128
        # the batched state must survive the batch-1 runs above (separate caches per drafter)
        prev = torch.tensor(ctxs, dtype=torch.int32, device=d)
        root = torch.stack([t[-1] for t in tok])
        tgt = (torch.randn(B, tb.n_nodes, cfg.vocab_size, generator=g) * 2).to(BF)
        expect = []
        for b in range(B):
            tree_tok, cart = ot.tree_candidates(int(root[b]), ids_b[b].cpu(), tb)
            _plant(tgt[b], tree_tok, tb, b % 6)
            best, alen, nxt, _ = ot.greedy_accept(cart, ot.first_argmax(tgt[b]), tb)
            expect.append([best, alen, nxt, ctxs[b] + alen + 1])
        res = batched.verify(root, tgt.to(d), prev)
        assert res.tolist() == expect
        th = torch.randn(B, tb.n_nodes, cfg.hidden_size, generator=g).to(BF).to(d)
        acc = batched.compact(prev, None, th)
        for b in (0, 5, 31):
            n = expect[b][1] + 1
            assert torch.equal(acc[b, :n].cpu(), th[b, tb.retrieve_indices[expect[b][0], :n]].cpu())
        ids2, _ = batched.draft(acc, batched.next_ids)
        assert batched.stable_len.tolist() == [c + e[1] + 1 for c, e in zip(ctxs, expect)]
        assert int(ids2.min()) >= 0 and int(ids2.max()) < cfg.vocab_size
        batched.check_errors()


def test_chain_tree_is_the_reference_benchmark_setting():
    """benchmark/README.md:52-54 runs SGLang with --steps 4 --k 1 --draft 4: one candidate per step, a
    chain.  The static-tree machinery covers it with the chain as its tree and top-k = 1; the cycle
    must match the oracle loop bit for bit (ids, acceptance, compaction)."""
    from eagle_amd import ops
    from eagle_amd.speculative import EagleDrafter
    d = dev()
    chain = [[0], [0, 0], [0, 0, 0], [0, 0, 0, 0]]
    dims = dm.DraftDims(hidden=256, inter=512, heads=2, kv_heads=2, vocab=2000)
    w = dm.init_weights(dims, seed=9, std=0.08)
    g = torch.Generator().manual_seed(12)
    lm_head = (torch.randn(dims.vocab, dims.hidden, generator=g) * 0.3).to(BF).to(d)
    m_api, m_dev = _model(dims, w), _model(dims, w)
    tb = ot.generate_tree_buffers(paths=chain, topk=1)
    assert tb.n_nodes == 5 and tb.retrieve_indices.tolist() == [[0, 1, 2, 3, 4]]
    drafter = EagleDrafter(m_dev, lm_head, tree=chain, topk=1, batch=1, max_ctx=128)

    def draft_fn(h, ids, past_key_values=None, position_ids=None, tree_mask=None, use_cache=True):
        m_api.tree_mask = None if tree_mask is None else tree_mask.to(d)
        pk = None if past_key_values is None else tuple((k.to(d), v.to(d)) for k, v in past_key_values)
        with torch.no_grad():
            o, kv = m_api(h.to(d), ids.to(d), past_key_values=pk,
                          position_ids=None if position_ids is None else position_ids.to(d), use_cache=True)
        return o.cpu(), tuple((k.cpu().clone(), v.cpu().clone()) for k, v in kv)

    def head_fn(h):
        return ops.linear_skinny(h.to(d).contiguous(), lm_head).cpu()

    ctx = 17
    tokens = torch.randint(0, dims.vocab, (1, ctx + 1), generator=g)
    hidden = torch.randn(1, ctx, dims.hidden, generator=g).to(BF)
    with torch.no_grad():
        ids_o, vals_o, stable, _ = ot.topk_generate(draft_fn, None, tb, hidden, tokens, None, head_fn=head_fn)
        ids_g, vals_g = drafter.prefill(hidden.to(d), tokens.to(d))
        assert torch.equal(ids_g[0].cpu(), ids_o) and torch.equal(vals_g[0].cpu(), vals_o)
        prev_len = ctx
        for cycle, n_agree in enumerate([4, 0, 2]):
            tgt_logits = (torch.randn(tb.n_nodes, dims.vocab, generator=g) * 2).to(BF)
            tgt_hidden = torch.randn(1, tb.n_nodes, dims.hidden, generator=g).to(BF)
            tree_tok, cart = ot.tree_candidates(int(tokens[0, -1]), ids_o, tb)
            for node in range(n_agree):
                tgt_logits[node, int(tree_tok[node + 1])] = 50.0
            best, alen, nxt, _ = ot.greedy_accept(cart, ot.first_argmax(tgt_logits), tb)
            assert alen >= n_agree
            prev_dev = torch.tensor([prev_len], dtype=torch.int32, device=d)
            res = drafter.verify(tokens[:, -1].to(d), tgt_logits.to(d)[None], prev_dev)
            assert res[0].tolist() == [best, alen, nxt, prev_len + alen + 1]
            acc_h_g = drafter.compact(prev_dev, None, tgt_hidden.to(d))
            _, acc_h = ot.compact([], tgt_hidden, tb, best, alen, prev_len)
            assert torch.equal(acc_h_g[0, : alen + 1].cpu(), acc_h[0])
            path_tokens = [int(cart[best, j]) for j in range(1, alen + 1)] + [nxt]
            tokens = torch.cat([tokens, torch.tensor([path_tokens])], dim=1)
            ids_o, vals_o, stable, _ = ot.topk_generate(draft_fn, None, tb, acc_h, tokens, stable, head_fn=head_fn)
            ids_g, vals_g = drafter.draft(acc_h_g, drafter.next_ids)
            assert torch.equal(ids_g[0].cpu(), ids_o), f"cycle {cycle}: chain ids differ"
            prev_len += alen + 1
        drafter.check_errors()


def test_kv_capacity_overflow_sets_the_error_flag_instead_of_faulting():
    """A request that outgrows the draft KV cache must surface as HipKernelError at the next
    check_errors() (the kernels bound-check against the capacity and skip the work), never as a fault."""
    from eagle_amd import ops
    from eagle_amd.speculative import EagleDrafter
    d = dev()
    dims = dm.DraftDims(hidden=256, inter=512, heads=2, kv_heads=2, vocab=500)
    w = dm.init_weights(dims, seed=2, std=0.08)
    g = torch.Generator().manual_seed(1)
    lm_head = (torch.randn(dims.vocab, dims.hidden, generator=g) * 0.3).to(BF).to(d)
    drafter = EagleDrafter(_model(dims, w), lm_head, batch=1, max_ctx=200)
    assert drafter.cap == 256
    ctx = 236                                   # + 6 catch-up rows + 11 tree rows fits 256 exactly once, not twice
    tokens = torch.randint(0, dims.vocab, (1, ctx + 1), generator=g).to(d)
    hidden = torch.randn(1, ctx, dims.hidden, generator=g).to(BF).to(d)
    tb = ot.generate_tree_buffers()
    with torch.no_grad():
        drafter.prefill(hidden, tokens)
        drafter.check_errors()                  # the first tree still fits
        for _ in range(3):                      # every cycle accepts nothing: +1 row per cycle, tree rows on top
            tgt = torch.zeros(1, tb.n_nodes, dims.vocab, dtype=BF, device=d)
            tgt[..., 0] = 1.0
            prev = torch.tensor([ctx], dtype=torch.int32, device=d)
            drafter.verify(tokens[:, -1], tgt, prev)
            acc = drafter.compact(prev, None, torch.zeros(1, tb.n_nodes, dims.hidden, dtype=BF, device=d))
            drafter.draft(acc, drafter.next_ids)
        drafter.check_errors()                  # 239 stable rows + 6 + 11 still fit
        for _ in range(30):
            drafter.verify(tokens[:, -1], tgt, prev)
            drafter.draft(acc, drafter.next_ids)
        assert int(drafter.stable_len[0]) > drafter.cap
        with pytest.raises(ops._hip.HipKernelError):
            drafter.check_errors()


def test_single_token_prompt_and_batch_of_one_ragged():
    from eagle_amd.speculative import EagleDrafter
    d = dev()
    dims = dm.DraftDims(hidden=256, inter=512, heads=2, kv_heads=2, vocab=500)
    w = dm.init_weights(dims, seed=3, std=0.08)
    g = torch.Generator().manual_seed(8)
    lm_head = (torch.randn(dims.vocab, dims.hidden, generator=g) * 0.3).to(BF).to(d)
    a = EagleDrafter(_model(dims, w), lm_head, batch=1, max_ctx=64)
    b = EagleDrafter(_model(dims, w), lm_head, batch=1, max_ctx=64)
    tok = torch.randint(0, dims.vocab, (2,), generator=g).to(d)
    hid = torch.randn(1, dims.hidden, generator=g).to(BF).to(d)
    with torch.no_grad():
        ia, va = a.prefill(hid[None], tok[None])
        ib, vb = b.prefill_ragged([hid], [tok])
    assert torch.equal(ia, ib) and torch.equal(va, vb) and a.stable_len.tolist() == [1] == b.stable_len.tolist()
    a.check_errors()
    b.check_errors()


@pytest.mark.parametrize("M,HK,use_tree", [(1, 32, False), (4, 32, True), (6, 32, False), (8, 8, True), (16, 32, False),
                                           (1, 8, False), (4, 8, True), (3, 16, True), (6, 16, False)])   # grouped-query: 4 / 2 heads per workgroup
def test_wide_batch_attention_matches_the_per_request_kernel(M, HK, use_tree):
    """The wide-batch attention (RoPE/append pre-pass + 512-key workgroups with P.V on MFMA, chosen when B*H >= 256)
    against the 64-key split kernel run request by request (B*H < 256), which the oracle fixtures pin: same cache
    contents bit for bit, outputs within the bf16 tolerance (the wide kernel rounds the probabilities to bf16 for the
    MFMA, as the reference does; the split kernel keeps them fp32).  Ragged lengths incl. empty pasts and > 512 keys."""
    from eagle_amd import ops
    d = dev()
    B, H, hd, cap = 9, 32, 128, 1024
    g = torch.Generator().manual_seed(M * 10 + HK)
    past = torch.tensor([0, 5, 63, 64, 300, 511, 512, 700, 37], dtype=torch.int32)
    qkv = (torch.randn(B * M, (H + 2 * HK) * hd, generator=g) * 0.8).to(BF).to(d)
    kc = torch.randn(B, HK, cap, hd, generator=g).to(BF).to(d)
    vc = torch.randn(B, HK, cap, hd, generator=g).to(BF).to(d)
    cos, sin = ops.rope_tables(hd, 2048, device=d)
    tmask = None
    if use_tree:
        tm = (torch.rand(M, M, generator=g) < 0.5).to(torch.uint8)
        tm[torch.arange(M), torch.arange(M)] = 1
        tmask = tm.to(d).contiguous()
    max_kv = 1024
    err = torch.zeros(1, dtype=torch.int32, device=d)

    def run(Bn, qkv_, kc_, vc_, past_):
        out = torch.zeros((Bn * M, H * hd), dtype=BF, device=d)
        ws = torch.empty(ops.query("eagle_attn_decode_workspace_bytes", Bn, H, max_kv), dtype=torch.uint8, device=d)
        ops.tree_attention(qkv_, None, cos, sin, kc_, vc_, past_, 0, tmask, None, out, ws, err, Bn, M, H, HK, hd, max_kv, True, M, 0,
                           pos_off=0, pos_stride=0 if use_tree else 1)
        return out
    kw, vw = kc.clone(), vc.clone()
    wide = run(B, qkv, kw, vw, past.to(d))
    ref = torch.empty_like(wide)
    ks, vs = kc.clone(), vc.clone()
    for b in range(B):
        ref[b * M:(b + 1) * M] = run(1, qkv[b * M:(b + 1) * M], ks[b:b + 1], vs[b:b + 1], past[b:b + 1].to(d))
    assert int(err.item()) == 0
    assert torch.equal(kw, ks) and torch.equal(vw, vs), "appended K/V rows differ"
    for b in range(B):
        close(wide[b * M:(b + 1) * M], ref[b * M:(b + 1) * M], what=f"request {b} (past {int(past[b])})")
