"""Known answers for oracle.tree (parity unpinned by the reference: hand-derived pins).

The expected tables below were derived by hand from the 25-path tree and the upstream
EAGLE-1 index formulas (see oracle/tree.py header); they pin the oracle, which in turn is the
checker for the HIP tree/accept/compact kernels."""
import torch

from oracle import tree as ot


def test_static_tree_draft_side_wchild():
    tb = ot.generate_tree_buffers()
    assert tb.n_nodes == 26
    assert [sum(1 for d in tb.depth if d == k) for k in range(1, 6)] == [4, 8, 8, 3, 2]
    assert tb.step_nodes == [[1, 2, 3, 4], [5, 6, 7, 8], [13], [21]]
    assert [t.tolist() for t in tb.step_tree_indices] == [[0, 1, 2, 3], [0, 1, 2, 10], [0], [0]]
    assert tb.step_repeat_nums == [[4], [3, 1], [1], [1]]
    assert [t.tolist() for t in tb.step_parent_row] == [[0, 0, 0, 0], [0, 0, 0, 1], [0], [0]]
    assert torch.equal(tb.step_attn_mask[0][0, 0], torch.eye(4))
    m1 = torch.zeros(4, 8)
    for r, cols in enumerate([[0, 4], [0, 5], [0, 6], [1, 7]]):
        m1[r, cols] = 1
    assert torch.equal(tb.step_attn_mask[1][0, 0], m1)
    assert tb.step_attn_mask[2][0, 0].nonzero()[:, 1].tolist() == [0, 4, 8]
    assert tb.step_attn_mask[3][0, 0].nonzero()[:, 1].tolist() == [0, 4, 8, 9]
    assert all(int(p.sum()) == 0 for p in tb.step_position_ids)


def test_static_tree_verify_side():
    tb = ot.generate_tree_buffers()
    exp_ti = [0, 1, 2, 3, 4, 11, 12, 13, 21, 22, 31, 32, 41, 51, 52, 53, 61, 62, 71, 72, 81, 91, 92, 93, 101, 102]
    assert tb.tree_indices.tolist() == exp_ti
    assert tb.tree_position_ids.tolist() == [0] + [1] * 4 + [2] * 8 + [3] * 8 + [4] * 3 + [5] * 2
    exp_ri = [
        [0, 1, 5, 13, 21, 24], [0, 1, 5, 13, 21, 25], [0, 1, 5, 13, 22, -1], [0, 1, 5, 13, 23, -1],
        [0, 1, 5, 14, -1, -1], [0, 1, 5, 15, -1, -1], [0, 1, 6, 16, -1, -1], [0, 1, 6, 17, -1, -1],
        [0, 1, 7, 18, -1, -1], [0, 1, 7, 19, -1, -1], [0, 2, 8, 20, -1, -1], [0, 2, 9, -1, -1, -1],
        [0, 3, 10, -1, -1, -1], [0, 3, 11, -1, -1, -1], [0, 4, 12, -1, -1, -1],
    ]
    assert tb.retrieve_indices.tolist() == exp_ri
    tm = tb.tree_attn_mask[0, 0]
    assert tm.shape == (26, 26) and bool((tm[:, 0] == 1).all())
    assert tm[24].nonzero().flatten().tolist() == [0, 1, 5, 13, 21, 24]
    assert tm[12].nonzero().flatten().tolist() == [0, 4, 12]
    assert bool((tm.triu(1) == 0).all())


def test_static_tree_expand_all():
    tb = ot.generate_tree_buffers(expand="all")
    assert [len(s) for s in tb.step_nodes] == [4, 8, 8, 3, 2]
    # node 9 = [1,1]: parent node 2 is row 1 of the depth-1 step
    assert tb.step_tree_indices[1].tolist() == [0, 1, 2, 10, 11, 20, 21, 30]
    assert tb.step_repeat_nums[1] == [3, 2, 2, 1]
    # flat row of a node = 1 + its rank among forwarded nodes = its own id
    assert tb.tree_indices[13].item() == 1 + 10 * 5 + 0 and tb.tree_indices[25].item() == 1 + 10 * 21 + 1


def test_stable_topk_ties():
    x = torch.tensor([[1.0, 3.0, 3.0, 2.0, 3.0, 0.0, 3.0], [0.0, -0.0, -1.0, 0.0, -0.0, 5.0, 5.0]])
    v, i = ot.stable_topk(x, 3)
    assert i.tolist() == [[1, 2, 4], [5, 6, 0]]
    assert ot.first_argmax(x).tolist() == [1, 5]
    xb = torch.tensor([[0.5, 0.50390625, 0.5, 0.50390625]], dtype=torch.bfloat16)
    assert ot.stable_topk(xb, 4)[1].tolist() == [[1, 3, 0, 2]]


def _flat_from_tokens(tb, node_tok):
    """Build a flat top-k table so that tree node n carries token node_tok[n]."""
    n_rows = 1 + sum(len(s) for s in tb.step_nodes)
    flat = torch.full((n_rows * tb.topk,), -7, dtype=torch.long)
    for n in range(1, tb.n_nodes):
        flat[tb.tree_indices[n] - 1] = node_tok[n]
    return flat.view(n_rows, tb.topk)


def test_greedy_accept_known_paths():
    tb = ot.generate_tree_buffers()
    node_tok = [100 + n for n in range(tb.n_nodes)]
    flat = _flat_from_tokens(tb, node_tok)
    tree_tok, cart = ot.tree_candidates(100, flat, tb)
    assert tree_tok.tolist() == node_tok
    assert cart[2].tolist() == [100, 101, 105, 113, 122, -1]
    # verifier agrees along root->1->5->13 then wants something else
    tgt = torch.full((26,), 9999, dtype=torch.long)
    tgt[0], tgt[1], tgt[5], tgt[13] = 101, 105, 113, 777
    best, alen, nxt, acc = ot.greedy_accept(cart, tgt, tb)
    assert (best, alen, nxt) == (0, 3, 777)
    assert acc.tolist() == [3, 3, 3, 3, 2, 2, 1, 1, 1, 1, 0, 0, 0, 0, 0]
    # nothing accepted -> best path 0, length 0, next token = verifier's token at the root
    tgt0 = torch.full((26,), 5, dtype=torch.long)
    assert ot.greedy_accept(cart, tgt0, tb)[:3] == (0, 0, 5)
    # full-depth acceptance along the second deepest path (leaf 25)
    tgt2 = torch.full((26,), 9999, dtype=torch.long)
    for a, b in [(0, 1), (1, 5), (5, 13), (13, 21), (21, 25)]:
        tgt2[a] = node_tok[b]
    tgt2[25] = 4242
    assert ot.greedy_accept(cart, tgt2, tb)[:3] == (1, 5, 4242)
    # acceptance down a short branch: root->3->11
    tgt3 = torch.full((26,), 9999, dtype=torch.long)
    tgt3[0], tgt3[3], tgt3[11] = node_tok[3], node_tok[11], 31337
    assert ot.greedy_accept(cart, tgt3, tb)[:3] == (13, 2, 31337)


def test_compact_rows():
    tb = ot.generate_tree_buffers()
    prev, n = 5, tb.n_nodes
    kv = torch.arange(2 * 3 * 40 * 4, dtype=torch.float32).view(2, 3, 40, 4)
    orig = kv.clone()
    hidden = torch.arange(n * 6, dtype=torch.float32).view(1, n, 6)
    new_len, acc_h = ot.compact([kv], hidden, tb, best=13, accept_length=2, prev_len=prev)
    assert new_len == 8
    for j, node in enumerate([0, 3, 11]):
        assert torch.equal(kv[..., prev + j, :], orig[..., prev + node, :])
        assert torch.equal(acc_h[0, j], hidden[0, node])
    assert torch.equal(kv[..., :prev, :], orig[..., :prev, :])


def test_topk_generate_shapes_and_chain():
    """Expansion loop against the oracle draft head (tiny dims): row counts, cache growth and
    that step inputs are the selected top-k tokens / repeated parent hidden rows."""
    from oracle import draft_model as dm

    dims = dm.DraftDims(hidden=64, inter=128, heads=4, kv_heads=4, vocab=100)
    w = dm.init_weights(dims, seed=0, std=0.2)
    g = torch.Generator().manual_seed(5)
    lm_head = torch.randn(100, 64, generator=g) * 0.2
    tb = ot.generate_tree_buffers()

    def draft(h, ids, past_key_values=None, position_ids=None, tree_mask=None, use_cache=True):
        return dm.forward(w, dims, h, ids, past_key_values=past_key_values, position_ids=position_ids,
                          tree_mask=tree_mask, use_cache=True)

    ctx = 9
    tokens = torch.randint(0, 100, (1, ctx + 1), generator=g)
    hidden = torch.randn(1, ctx, 64, generator=g)
    with torch.no_grad():
        ids, vals, stable, dbg = ot.topk_generate(draft, lm_head, tb, hidden, tokens, None)
    assert ids.shape == (11, 10) and vals.shape == (11, 10)
    assert stable[0][0].shape[2] == ctx
    assert dbg["steps"][1]["input_ids"].tolist() == ids.reshape(-1)[torch.tensor([10, 11, 12, 20])].tolist()
    assert dbg["steps"][0]["position_ids"].tolist() == [ctx] * 4 and dbg["steps"][3]["position_ids"].tolist() == [ctx + 3]
    # second cycle: 3 accepted tokens are caught up on top of the stable cache
    tokens2 = torch.cat([tokens, torch.randint(0, 100, (1, 3), generator=g)], dim=1)
    with torch.no_grad():
        ids2, _, stable2, _ = ot.topk_generate(draft, lm_head, tb, torch.randn(1, 3, 64, generator=g), tokens2, stable)
    assert stable2[0][0].shape[2] == ctx + 3 and ids2.shape == (11, 10)


def _random_paths(g, topk, max_depth, max_nodes):
    paths = {(int(torch.randint(0, topk, (1,), generator=g)),)}
    while len(paths) < max_nodes:
        parent = sorted(paths)[int(torch.randint(0, len(paths), (1,), generator=g))]
        if torch.rand(1, generator=g) < 0.3:
            paths.add((int(torch.randint(0, topk, (1,), generator=g)),))
        elif len(parent) < max_depth:
            paths.add(parent + (int(torch.randint(0, topk, (1,), generator=g)),))
    return [list(p) for p in paths]          # unsorted on purpose: both builders sort by (len, path)


def test_host_tree_buffers_equal_the_oracle_on_random_trees():
    """The host-side buffers the kernels consume (eagle_amd.tree.TreeBuffers, CPU tensors) against the oracle's, for
    random prefix-closed trees in both expansion modes: draft-side indices / parent rows / repeat counts / masks and
    verify-side mask, indices, positions and retrieve table."""
    from eagle_amd.tree import TreeBuffers
    for seed in range(12):
        g = torch.Generator().manual_seed(seed)
        topk = int(torch.randint(3, 11, (1,), generator=g))
        paths = _random_paths(g, topk, int(torch.randint(2, 7, (1,), generator=g)), int(torch.randint(3, 50, (1,), generator=g)))
        for expand in ("wchild", "all"):
            o = ot.generate_tree_buffers(paths, topk=topk, expand=expand)
            h = TreeBuffers(paths, topk=topk, expand=expand, device="cpu")
            assert (h.n_nodes, h.parent, h.depth, h.choice) == (o.n_nodes, o.parent, o.depth, o.choice)
            assert len(h.steps) == len(o.step_nodes)
            for d, st in enumerate(h.steps):
                assert st.nodes == o.step_nodes[d]
                assert st.tree_indices.tolist() == o.step_tree_indices[d].tolist()
                assert st.parent_row.tolist() == o.step_parent_row[d].tolist()
                assert st.repeat_nums == o.step_repeat_nums[d]
                assert torch.equal(st.attn_mask.float(), o.step_attn_mask[d].float())
                assert torch.equal(st.attn_mask_u8.float(), o.step_attn_mask[d][0, 0].float())
            assert torch.equal(h.tree_attn_mask.float(), o.tree_attn_mask.float())
            assert h.tree_indices.tolist() == o.tree_indices.tolist()
            assert h.tree_position_ids.tolist() == o.tree_position_ids.tolist()
            assert h.retrieve_indices.tolist() == o.retrieve_indices.tolist()
            assert h.flat_row == o.flat_row
