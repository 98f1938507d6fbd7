"""GPU parity of the dynamic draft tree (eagle_amd/dynamic.py, csrc/dyn_tree.hip) against oracle/dyn_tree.py.

The floating-point part (logits, their top-k values, the row log-sum-exp, the path scores) is held to
tolerance; everything downstream of the scores is integer work and must be bit-exact: given the
device's own per-step top-k and scores the oracle must reproduce the pool, every frontier (tokens,
parent rows, attention masks), the verified tree, the accepted path, the compaction and the state
handed to the next cycle.  (parity unpinned: the algorithm is SGLang's, not in /root/reference.)"""
import pytest
import torch

from oracle import draft_model as dm
from oracle import dyn_tree as od
from oracle import tree as ot

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def dev():
    return torch.device("cuda:0")


def _model(dims, w, rows_min=None):
    from eagle_amd.config import DraftConfig
    from eagle_amd.llama2 import Llama2Model
    cfg = DraftConfig(hidden_size=dims.hidden, intermediate_size=dims.inter, num_attention_heads=dims.heads,
                      num_key_value_heads=dims.kv_heads, vocab_size=dims.vocab)
    m = Llama2Model(cfg)
    m.load_state_dict(w, strict=True)
    m = m.to(BF).to(dev()).eval()
    if rows_min is not None:
        m.rows_min = rows_min      # batch and per-request runs on the same kernel family: rows are independent of their neighbours
    return m


def test_row_lse_matches_logsumexp():
    from eagle_amd import ops
    g = torch.Generator().manual_seed(4)
    for rows, V in ((1, 32000), (8, 151936), (3, 100), (5, 4097)):
        x = (torch.randn(rows, V, generator=g) * 3).to(BF)
        x[0, 0] = float("-inf")
        got = ops.row_lse(x.to(dev())).cpu()
        ref = torch.logsumexp(x.float(), dim=-1)
        assert torch.allclose(got, ref, rtol=0, atol=2e-4), (rows, V, (got - ref).abs().max())


def _request_view(dr, b):
    """Request b's slice of the (batched) debug records and state, in the batch-1 shapes the oracle uses."""
    k = dr.k
    out = []
    for i, d in enumerate(dr.debug):
        rows = slice(b, b + 1) if i == 0 else slice(b * k, (b + 1) * k)
        e = {"logits": d["logits"][rows], "vals": d["vals"][rows], "ids": d["ids"][rows], "lse": d["lse"][rows]}
        if "x_root" in d:
            e["x_root"] = d["x_root"][b:b + 1]
        if "ftok" in d:
            e.update({"ftok": d["ftok"][b], "mask": d["mask"][b], "x": d["x"][b * k:(b + 1) * k],
                      "parent_row": d["parent_row"][b] - (b * k if i > 0 else 0)})     # global row of the batched forward -> local
        out.append(e)
    return out


def _check_cycle(dr, k, S, T, root_token, b=0):
    """Oracle recomputation of one expansion of request b from the device's per-step top-k / lse."""
    dbg = _request_view(dr, b)
    assert len(dbg) == S
    steps = []
    for i, d in enumerate(dbg):
        logits = d["logits"].cpu()
        rv, ri = ot.stable_topk(logits, k)
        assert torch.equal(d["ids"].cpu(), ri) and torch.equal(d["vals"].cpu(), rv), f"step {i}: top-k differs from the stable sort"
        assert torch.allclose(d["lse"].cpu(), torch.logsumexp(logits.float(), dim=-1), rtol=0, atol=2e-4)
        steps.append((d["vals"].cpu(), d["ids"].cpu(), d["lse"].cpu()))
    pool_tok, pool_score, pool_parent = dr.pool_tok[b].cpu(), dr.pool_score[b].cpu(), dr.pool_parent[b].cpu()
    o_tok, o_score, o_parent, fronts = od.expand_reference(steps, k, S)
    assert torch.allclose(pool_score, o_score, rtol=2e-4, atol=1e-12), "path scores outside tolerance"
    # integer bookkeeping, teacher-forced on the DEVICE scores: frontier i = k best of the device's step-i scores
    for i in range(S - 1):
        base, n = (0, k) if i == 0 else (k + (i - 1) * k * k, k * k)
        pick = od.select_frontier(pool_score[base: base + n], k)
        d = dbg[i]                                     # frontier i is recorded beside the step that produced it
        assert d["ftok"].cpu().tolist() == [int(pool_tok[base + c]) for c in pick], f"frontier {i}: tokens"
        if i > 0:
            assert d["parent_row"].cpu().tolist() == [c // k for c in pick], f"frontier {i}: parent rows"
        prev_mask = dbg[i - 1]["mask"].cpu() if i > 0 else None
        assert torch.equal(d["mask"].cpu(), od.step_masks(prev_mask, [c // k for c in pick], i, k)), f"frontier {i}: mask"
        # the pool records the parent of every child of that frontier
        nb = k + i * k * k
        assert pool_parent[nb: nb + k * k].tolist() == [base + c for c in pick for _ in range(k)]
    t_tok, t_par, t_pos, t_mask, sel = od.finalize(pool_tok, pool_score, pool_parent, root_token, k, S, T)
    assert torch.equal(dr.tree_tok[b].cpu(), t_tok) and torch.equal(dr.tree_parent[b].cpu(), t_par)
    assert torch.equal(dr.tree_pos[b].cpu(), t_pos) and torch.equal(dr.tree_mask[b].cpu(), t_mask) and torch.equal(dr.sel[b].cpu(), sel)
    return t_tok, t_par


@pytest.mark.parametrize("S,k,T", [(4, 4, 12), (5, 8, 32), (3, 2, 6), (4, 1, 4)])
def test_dynamic_tree_cycles_match_oracle(S, k, T):
    from eagle_amd import ops
    from eagle_amd.dynamic import EagleDynamicDrafter
    d = dev()
    dims = dm.DraftDims(hidden=256, inter=512, heads=2, kv_heads=2, vocab=2000)
    w = dm.init_weights(dims, seed=31, std=0.08)
    g = torch.Generator().manual_seed(S * 100 + k)
    lm_head = (torch.randn(dims.vocab, dims.hidden, generator=g) * 0.3).to(BF).to(d)
    model = _model(dims, w)
    dr = EagleDynamicDrafter(model, lm_head, steps=S, topk=k, num_draft_tokens=T, max_ctx=256, keep_debug=True)
    ctx = 19
    tokens = torch.randint(0, dims.vocab, (1, ctx + 1), generator=g)
    hidden = torch.randn(1, ctx, dims.hidden, generator=g).to(BF)
    wb = {kk: v.to(BF) for kk, v in w.items()}
    with torch.no_grad():
        dr.prefill(hidden.to(d), tokens.to(d))
        # the root's logits against the CPU oracle model (bf16 tolerance)
        ref_h, _ = dm.forward(wb, dims, hidden, tokens[:, 1:], use_cache=True)
        ref_logits = torch.nn.functional.linear(ref_h[0, -1:].float(), lm_head.cpu().float())
        got = dr.debug[0]["logits"].float().cpu()
        view = _request_view(dr, 0)
        assert float((got - ref_logits).abs().max()) <= 2e-2 * float(ref_logits.abs().max()) + 0.05
        # teacher-forced forward parity of every tree step: same inputs (the device's frontier tokens,
        # parent hidden rows, mask) through the CPU oracle model with its own KV cache
        _, okv = dm.forward(wb, dims, hidden, tokens[:, 1:], use_cache=True)
        x_prev = dr.debug[0]["x_root"].cpu().repeat(k, 1)
        for i in range(1, S):
            dbg = view[i - 1]
            par = dbg["parent_row"].cpu()
            hin = x_prev if i == 1 else x_prev[par]
            tm = dbg["mask"].cpu().float()[None, None]
            pos = torch.full((k,), ctx + i - 1, dtype=torch.long)
            o, okv = dm.forward(wb, dims, hin[None].to(BF), dbg["ftok"].cpu()[None], past_key_values=okv, position_ids=pos,
                                tree_mask=tm, use_cache=True)
            got_x, ref_x = dbg["x"].float().cpu(), o[0].float()
            err = (got_x - ref_x).abs()
            tol = 1e-2 * ref_x.abs() + 3e-2 * float(ref_x.abs().mean()) + float(ref_x.abs().max()) / 128
            assert float((err > tol).float().mean()) < 5e-3 and float((err / tol).max()) < 3.0, f"tree step {i}: hidden states outside bf16 tolerance"
            x_prev = dbg["x"].cpu().to(BF)            # teacher forcing: continue from the device's rows
        root = int(tokens[0, -1])
        prev_len = ctx
        kv = torch.randn(2, 1, 2, 128, 16, generator=g).to(BF)
        kv_g = kv.to(d).clone()
        seen = set()
        for cycle in range(4):
            t_tok, t_par = _check_cycle(dr, k, S, T, root)
            assert int(dr.stable_len[0]) == prev_len
            # synthetic verifier: plant agreement `cycle` levels deep along the first child at each level
            tgt_logits = (torch.randn(T, dims.vocab, generator=g) * 2).to(BF)
            tgt_hidden = torch.randn(1, T, dims.hidden, generator=g).to(BF)
            node = 0
            for _ in range((cycle * 2) % (S + 1)):
                kids = [n for n in range(1, T) if int(t_par[n]) == node]
                if not kids:
                    break
                kid = kids[-1] if cycle % 2 else kids[0]
                tgt_logits[node, int(t_tok[kid])] = 50.0
                node = kid
            path, alen, nxt = od.accept(t_tok, t_par, ot.first_argmax(tgt_logits), dr.W)
            seen.add(alen)
            prev_dev = torch.tensor([prev_len], dtype=torch.int32, device=d)
            res = dr.verify(tgt_logits.to(d), prev_dev)
            assert res[0].tolist() == [0, alen, nxt, prev_len + alen + 1]
            assert dr.path[0, : alen + 1].tolist() == path and (dr.path[0, alen + 1:] == -1).all()
            assert dr.next_ids[0, : alen + 1].tolist() == [int(t_tok[n]) for n in path[1:]] + [nxt]
            assert int(dr.n_acc[0]) == alen + 1 and int(dr.last_idx[0]) == alen
            acc_h = dr.compact(prev_dev, kv_g, tgt_hidden.to(d), kv_inner=2)
            ref_kv = kv.clone()
            sel_rows = torch.tensor(path) + prev_len
            rows = ref_kv[..., sel_rows, :].clone()
            ref_kv[..., prev_len: prev_len + alen + 1, :] = rows
            assert torch.equal(kv_g.cpu(), ref_kv)
            assert torch.equal(acc_h[0, : alen + 1].cpu(), tgt_hidden[0, path])
            kv = ref_kv
            dr.draft(acc_h, dr.next_ids)
            prev_len += alen + 1
            root = nxt
        assert len(seen) >= 2
        dr.check_errors()


@pytest.mark.parametrize("S,k,T", [(4, 4, 12), (3, 8, 20)])
def test_batched_dynamic_tree_matches_oracle_and_batch_1_per_request(S, k, T):
    """B ragged requests in every launch (BASELINE configs[3] serves batches with the --steps/--k/--draft flags,
    benchmark/benchmark.py:48-56): every request's pool, frontiers, masks, verified tree, accepted path, compaction and
    next-cycle state against the oracle, and its top-k ids / trees against a batch-1 drafter fed the same request."""
    from eagle_amd.dynamic import EagleDynamicDrafter
    d = dev()
    dims = dm.DraftDims(hidden=256, inter=512, heads=2, kv_heads=2, vocab=2000)
    w = dm.init_weights(dims, seed=37, std=0.08)
    g = torch.Generator().manual_seed(S * 10 + k)
    lm_head = (torch.randn(dims.vocab, dims.hidden, generator=g) * 0.3).to(BF).to(d)
    ctxs = [23, 5, 70]
    B = len(ctxs)
    batched = EagleDynamicDrafter(_model(dims, w, 1), lm_head, steps=S, topk=k, num_draft_tokens=T, max_ctx=256, keep_debug=True, batch=B)
    singles = [EagleDynamicDrafter(_model(dims, w, 1), lm_head, steps=S, topk=k, num_draft_tokens=T, max_ctx=256) for _ in range(B)]
    hid = [torch.randn(c, dims.hidden, generator=g).to(BF).to(d) for c in ctxs]
    tok = [torch.randint(0, dims.vocab, (c + 1,), generator=g).to(d) for c in ctxs]
    with torch.no_grad():
        tt, tp, tm = batched.prefill_ragged(hid, tok)
        assert tuple(tt.shape) == (B, T) and tuple(tm.shape) == (B, T, T)
        for b in range(B):
            t1, p1, m1 = singles[b].prefill(hid[b][None], tok[b][None])
            assert torch.equal(tt[b], t1) and torch.equal(tp[b], p1) and torch.equal(tm[b], m1), f"request {b}: tree differs from its batch-1 run"
        prev = torch.tensor(ctxs, dtype=torch.int32, device=d)
        roots = [int(t[-1]) for t in tok]
        kv_b = torch.randn(2, B, 2, 256, 16, generator=g).to(BF).to(d)
        kv_ref = kv_b.cpu().clone()
        seen = set()
        for cycle in range(3):
            tgt_logits = (torch.randn(B, T, dims.vocab, generator=g) * 2).to(BF)
            tgt_hidden = torch.randn(B, T, dims.hidden, generator=g).to(BF)
            expect = []
            for b in range(B):
                t_tok, t_par = _check_cycle(batched, k, S, T, roots[b], b)
                node = 0
                for _ in range((cycle + b) % (S + 1)):
                    kids = [n for n in range(1, T) if int(t_par[n]) == node]
                    if not kids:
                        break
                    kid = kids[-1] if (cycle + b) % 2 else kids[0]
                    tgt_logits[b, node, int(t_tok[kid])] = 50.0
                    node = kid
                path, alen, nxt = od.accept(t_tok, t_par, ot.first_argmax(tgt_logits[b]), batched.W)
                expect.append((path, alen, nxt, t_tok))
                seen.add(alen)
            res = batched.verify(tgt_logits.to(d), prev)
            acc_h = batched.compact(prev, kv_b, tgt_hidden.to(d), kv_inner=2)
            for b, (path, alen, nxt, t_tok) in enumerate(expect):
                pl = int(prev[b])
                assert res[b].tolist() == [0, alen, nxt, pl + alen + 1]
                assert batched.path[b, : alen + 1].tolist() == path and (batched.path[b, alen + 1:] == -1).all()
                assert batched.next_ids[b, : alen + 1].tolist() == [int(t_tok[n]) for n in path[1:]] + [nxt]
                assert int(batched.n_acc[b]) == alen + 1 and int(batched.last_idx[b]) == b * batched.W + alen
                rows = kv_ref[:, b, :, torch.tensor(path) + pl, :].clone()
                kv_ref[:, b, :, pl: pl + alen + 1, :] = rows
                assert torch.equal(acc_h[b, : alen + 1].cpu(), tgt_hidden[b, path])
                p1 = prev[b:b + 1].clone()
                singles[b].verify(tgt_logits[b].to(d), p1)
                a1 = singles[b].compact(p1, None, tgt_hidden[b:b + 1].to(d))
                singles[b].draft(a1, singles[b].next_ids)
            assert torch.equal(kv_b.cpu(), kv_ref), "verifier KV compaction differs"
            tt, tp, tm = batched.draft(acc_h, batched.next_ids)
            for b in range(B):
                assert torch.equal(tt[b], singles[b].tree_tok[0]) and torch.equal(tm[b], singles[b].tree_mask[0]), f"cycle {cycle}: request {b}"
                assert int(batched.stable_len[b]) == int(singles[b].stable_len[0])
            prev = res[:, 3].to(torch.int32)
            roots = [e[2] for e in expect]
        assert len(seen) >= 2
        batched.check_errors()
