"""Oracle anchors for the kernels of the batched (batch-32) leg and the real-dims cycle.

Round-2 review: the wide-batch attention kernel, the fused lm_head + top-k of 81..128-row steps and the drafter cycle at
real dims were only ever compared with OTHER HIP kernels.  Every test here compares the HIP path (through the C ABI) with
the CPU oracle (oracle/draft_model.py restating eagle/llama2.py:226-273, :460-491; oracle/tree.py; oracle/loss.py pinned
by the reference's own loss statements, eagle/train.py:106-123) on the same seeded inputs.

Tolerances: hidden states / logits to close() (bf16 rtol 1e-2 + one bf16 ulp at the largest magnitude, see
tests/test_gpu_parity.py) with the relative L2 error bounded beside it; cache rows, ids where the oracle's logits are
untied, acceptance lengths, compaction: bit-exact."""
import os

import pytest
import torch
from safetensors.torch import load_file

from oracle import draft_model as dm
from oracle import tree as ot
from test_gpu_parity import BF, MARGINS, _model, close, dev

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def bf16_ulp(v: torch.Tensor) -> torch.Tensor:
    """Spacing of bf16 numbers at |v| (8 significant bits)."""
    return torch.exp2(torch.floor(torch.log2(v.float().abs().clamp_min(1e-30))) - 7)


# ------------------------------------------------------------------ wide-batch attention (attn_wide_kernel)
WIDE_CASES = [
    # B, M, H, HK, tree, extra tree columns (T = M + extra)
    (32, 4, 8, 8, True, 0),      # VERDICT's shape: 8 heads x hd 128, B.H = 256
    (32, 8, 8, 8, True, 4),      # a deeper step: the mask also covers 4 earlier tree rows
    (32, 6, 8, 8, False, 0),     # catch-up rows: causal, consecutive positions
    (32, 1, 8, 8, False, 0),     # q_len 1: no causal part (eagle/llama2.py:464)
    (9, 4, 32, 32, True, 0),     # the shipped head count
    (9, 16, 32, 32, False, 0),   # a full 16-row block
    (12, 4, 32, 8, True, 0),     # grouped-query: 4 query heads share a workgroup
    (16, 3, 16, 8, False, 0),    # grouped-query: 2 per workgroup
]


def _attention_against_the_oracle(B, M, H, HK, use_tree, extra, base, max_kv, what):
    """eagle_tree_attention_bf16 on seeded q|k|v rows and caches against oracle.draft_model.attention_core request by request:
    appended K/V rows bit for bit, rows outside the append untouched, outputs to close() and rel-L2 < 1e-2."""
    from eagle_amd import ops
    d = dev()
    hd, cap = 128, 1024
    g = torch.Generator().manual_seed(B * 1000 + M * 10 + H + HK + max_kv)
    past = torch.tensor([base[b % len(base)] + (b // len(base)) * 7 + extra for b in range(B)], dtype=torch.int32)
    assert int(past.max()) + M <= max_kv <= cap
    qkv = (torch.randn(B * M, (H + 2 * HK) * hd, generator=g) * 0.8).to(BF)
    kc = torch.randn(B, HK, cap, hd, generator=g).to(BF)
    vc = torch.randn(B, HK, cap, hd, generator=g).to(BF)
    cos_d, sin_d = ops.rope_tables(hd, 2048, device=d)
    cos, sin = dm.rope_tables(hd, 2048)
    T = M + extra
    tm = None
    if use_tree:
        tm = (torch.rand(M, T, generator=g) < 0.5).to(torch.uint8)
        tm[torch.arange(M), extra + torch.arange(M)] = 1            # every row sees itself
    err = torch.zeros(1, dtype=torch.int32, device=d)
    out = torch.zeros((B * M, H * hd), dtype=BF, device=d)
    ws = torch.empty(ops.query("eagle_attn_decode_workspace_bytes", B, H, max_kv), dtype=torch.uint8, device=d)
    kd, vd = kc.to(d), vc.to(d)
    ops.tree_attention(qkv.to(d), None, cos_d, sin_d, kd, vd, past.to(d), 0, None if tm is None else tm.to(d).contiguous(), None,
                       out, ws, err, B, M, H, HK, hd, max_kv, True, M, 0, pos_off=0, pos_stride=0 if use_tree else 1)
    assert int(err.item()) == 0
    out, kd, vd = out.cpu(), kd.cpu(), vd.cpu()
    dims = dm.DraftDims(hidden=H * hd, heads=H, kv_heads=HK)
    worst = 0.0
    for b in range(B):
        p = int(past[b])
        rows = qkv[b * M:(b + 1) * M]
        q = rows[:, :H * hd].view(1, M, H, hd).transpose(1, 2)
        k = rows[:, H * hd:(H + HK) * hd].view(1, M, HK, hd).transpose(1, 2)
        v = rows[:, (H + HK) * hd:].view(1, M, HK, hd).transpose(1, 2)
        pos = (torch.full((1, M), p) if use_tree else p + torch.arange(M)[None]).long()
        mask = dm.build_mask(torch.ones(1, p + M, dtype=torch.bool), M, p, None if tm is None else tm[None, None])
        o, (kk, vv) = dm.attention_core(q, k, v, dims, mask, pos, (kc[b:b + 1, :, :p], vc[b:b + 1, :, :p]), True, cos, sin)
        assert torch.equal(kd[b, :, p:p + M], kk[0, :, p:]), f"request {b}: appended K rows differ from the oracle's RoPE"
        assert torch.equal(vd[b, :, p:p + M], vv[0, :, p:]), f"request {b}: appended V rows differ"
        assert torch.equal(kd[b, :, :p], kc[b, :, :p]) and torch.equal(kd[b, :, p + M:], kc[b, :, p + M:]), "rows outside the append touched"
        close(out[b * M:(b + 1) * M], o[0], what=what)
        worst = max(worst, rel_l2(out[b * M:(b + 1) * M], o[0]))
    assert worst < 1e-2, f"relative L2 error {worst:.3g}"


@pytest.mark.parametrize("B,M,H,HK,use_tree,extra", WIDE_CASES)
def test_wide_batch_attention_against_the_oracle(B, M, H, HK, use_tree, extra):
    """eagle_tree_attention_bf16 at B.H >= 256 (RoPE/append pre-pass + attn_wide_kernel with P.V on MFMA + combine) against
    oracle.draft_model.attention_core (eagle/llama2.py:226-273 with the mask of :460-491) request by request: ragged pasts
    incl. 0, 511, 512 and > 512 keys; appended K/V rows bit for bit, outputs to close()."""
    assert B * H >= 256, "this shape would run the split-KV kernel"
    _attention_against_the_oracle(B, M, H, HK, use_tree, extra, [0, 511, 512, 700, 5, 63, 64, 300, 37, 1, 129, 513], 1024,
                                  "wide attention vs oracle")


NARROW_CASES = [
    # B, M, H, HK, tree, extra tree columns, key bound
    (1, 4, 32, 32, True, 0, 704),      # the batch-1 tree step of the bench (cycle bound 704 keys)
    (1, 6, 32, 32, False, 0, 704),     # the catch-up forward
    (1, 4, 32, 32, True, 3, 512),      # a mask that covers earlier tree rows
    (1, 8, 32, 32, True, 0, 1024),     # a dynamic-tree step: 8 rows
    (1, 1, 32, 32, False, 0, 1024),    # q_len 1
    (5, 4, 32, 8, True, 0, 1024),      # a narrow ragged batch, grouped-query heads
    (3, 16, 16, 8, False, 0, 384),     # a full 16-row block
    (2, 2, 2, 2, True, 0, 256),        # the tiny test models' head count
]


@pytest.mark.parametrize("B,M,H,HK,use_tree,extra,max_kv", NARROW_CASES)
def test_split_kv_attention_against_the_oracle(B, M, H, HK, use_tree, extra, max_kv):
    """The split-KV kernel pair of narrow batches (B.H < 256: every batch-1 tree step) against the oracle at real head
    dims, with the same bar as the wide-batch kernel above."""
    base = [p for p in [0, 5, 63, 64, 300, 37, 1, 129, 250, 500, 700, 1000] if p + M + extra + 14 <= max_kv]
    assert B * H < 256
    _attention_against_the_oracle(B, M, H, HK, use_tree, extra, base, max_kv, "split-KV attention vs oracle")


@pytest.mark.parametrize("HK", [8, 2])
def test_batch32_forward_prefill_and_tree_steps_against_the_oracle(HK):
    """Llama2Model.forward at B = 32, hidden 1024 = 8 heads x 128 (B.H = 256: the wide-batch attention), against
    oracle.draft_model.forward: a right-padded ragged prefill (attention_mask), then two tree-masked steps with explicit
    position_ids and a batched tree_mask (eagle/llama2.py:482-488) on top of the returned cache."""
    d = dev()
    B, L0 = 32, 24
    dims = dm.DraftDims(hidden=1024, inter=2048, heads=8, kv_heads=HK, vocab=700)
    w = dm.init_weights(dims, seed=40 + HK, std=0.04)
    m = _model(dims, w)
    wb = {k: v.to(BF) for k, v in w.items()}
    g = torch.Generator().manual_seed(90 + HK)
    lens = torch.tensor([1 + (7 * b + 3) % L0 for b in range(B)])
    lens[0], lens[1] = L0, 1
    amask = torch.arange(L0)[None, :] < lens[:, None]
    h = torch.randn(B, L0, dims.hidden, generator=g).to(BF)
    ids = torch.randint(0, dims.vocab, (B, L0), generator=g)
    tb = ot.generate_tree_buffers(expand="all")
    with torch.no_grad():
        ro, rkv = dm.forward(wb, dims, h, ids, attention_mask=amask, use_cache=True)
        o, kv = m(h.to(d), ids.to(d), attention_mask=amask.to(d), use_cache=True)
        valid = amask.reshape(-1)
        close(o.reshape(B * L0, -1)[valid.to(d)], ro.reshape(B * L0, -1)[valid], what="batch-32 prefill out (valid rows)")
        assert rel_l2(o.reshape(B * L0, -1)[valid.to(d)], ro.reshape(B * L0, -1)[valid]) < 1e-2
        kmask = amask[:, None, :, None].expand_as(rkv[0][0])
        close(kv[0][0].cpu()[kmask], rkv[0][0][kmask], what="batch-32 prefill k")
        close(kv[0][1].cpu()[kmask], rkv[0][1][kmask], what="batch-32 prefill v")
        total = L0
        for s in range(2):
            n = len(tb.step_nodes[s])
            hs = torch.randn(B, n, dims.hidden, generator=g).to(BF)
            ti = torch.randint(0, dims.vocab, (B, n), generator=g)
            tm = tb.step_attn_mask[s].expand(B, 1, n, tb.step_attn_mask[s].shape[-1]).contiguous()
            pos = (lens[:, None] + s).expand(B, n).contiguous()
            amask = torch.cat([amask, torch.ones(B, n, dtype=torch.bool)], dim=1)
            total += n
            ro, rkv = dm.forward(wb, dims, hs, ti, attention_mask=amask, position_ids=pos, past_key_values=rkv, use_cache=True,
                                 tree_mask=tm)
            m.tree_mask = tm.to(d)
            o, kv = m(hs.to(d), ti.to(d), attention_mask=amask.to(d), position_ids=pos.to(d), past_key_values=kv, use_cache=True)
            assert tuple(kv[0][0].shape) == tuple(rkv[0][0].shape) == (B, HK, total, 128)
            close(o, ro, what=f"batch-32 tree step {s} out")
            assert rel_l2(o, ro) < 1e-2
            close(kv[0][0][:, :, -n:], rkv[0][0][:, :, -n:], what=f"batch-32 tree step {s} k")
        m.reset()


# ------------------------------------------------------------------ fused lm_head + top-k (no logits)
@pytest.mark.parametrize("B,n,N,K", [(32, 4, 151936, 4096), (25, 5, 33000, 384), (128, 1, 40008, 128),
                                     (1, 4, 32000, 4096), (1, 1, 32000, 4096), (1, 16, 151936, 4096), (1, 3, 4100, 4096)])
def test_fused_lm_head_topk_against_the_stable_sort_oracle(B, n, N, K):
    """eagle_lm_head_topk_rows_bf16 against oracle.tree.stable_topk of the fp32 product rounded to bf16 (what the reference's
    bf16 nn.Linear + torch.topk see, with the oracle's stated tie-break).  The kernel's fp32 accumulation order differs from
    torch's, so a logit can land one bf16 step away: values must agree within one ulp, every (id, value) pair the kernel
    returns must be that id's logit, and ids must be EQUAL at every rank whose fp32 gaps to both neighbours exceed two ulps.
    B = 1 (one request's tree step, 1..16 rows): the register-resident weight stream over the packed image (batch-1 cycle)."""
    from eagle_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(B * n + N + 1)
    M, k, R = B * n, 10, 2 + n + 3
    x = torch.randn(M, K, generator=g).to(BF).to(d)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF).to(d)
    assert ops.lm_head_topk_rows_ok(M, N, K, k)
    ti = torch.full((B, R, k), -7, dtype=torch.int64, device=d)
    tv = torch.zeros((B, R, k), dtype=BF, device=d)
    ops.lm_head_topk_into_table(x, ops.pack_weight(W) if M <= 16 else W, k, ti, tv, 2, n, M=M)
    assert int((ti[:, :2] != -7).sum()) == 0 and int((ti[:, 2 + n:] != -7).sum()) == 0, "rows outside [row0, row0+n) touched"
    ids = ti[:, 2:2 + n].reshape(M, k).cpu()
    vals = tv[:, 2:2 + n].reshape(M, k).cpu()
    ref32 = (x.float() @ W.float().t()).cpu()                       # fp32 product (plumbing: torch on the device, then the CPU oracle)
    refbf = ref32.to(BF)
    ov, oi = ot.stable_topk(refbf, k)
    s32 = torch.sort(ref32, dim=1, descending=True, stable=True)
    top32 = s32.values[:, :k + 1]
    ulp = bf16_ulp(top32[:, :k])
    assert bool(((vals.float() - ov.float()).abs() <= ulp).all()), "top-k values differ from the oracle's by more than one bf16 ulp"
    at = torch.gather(refbf, 1, ids)
    assert bool(((vals.float() - at.float()).abs() <= bf16_ulp(at)).all()), "a returned value is not its id's logit"
    gap_dn = top32[:, :k] - top32[:, 1:k + 1]
    gap_up = torch.cat([torch.full((M, 1), float("inf")), gap_dn[:, :-1]], dim=1)
    untied = (gap_dn > 2 * ulp) & (gap_up > 2 * ulp)
    assert float(untied.float().mean()) > (0.25 if M >= 64 else 0.1), "the check would be vacuous"     # a handful of rows: fewer samples
    assert torch.equal(ids[untied], s32.indices[:, :k][untied]), "ids differ at ranks the oracle's logits separate by > 2 ulp"
    # the kernel's own order: value descending, index ascending
    dv = vals.float()[:, :-1] - vals.float()[:, 1:]
    assert bool((dv >= 0).all()) and bool((ids[:, :-1][dv == 0] < ids[:, 1:][dv == 0]).all())


@pytest.mark.parametrize("B,n,N,K", [(32, 4, 151936, 128), (27, 3, 70000, 256), (1, 7, 32000, 4096), (1, 16, 50000, 4096)])
def test_fused_lm_head_topk_exact_on_integer_logits(B, n, N, K):
    """Operands in {-1, 0, 1}: every product sum is a small integer, exact in fp32 and in bf16 whatever the summation order,
    so the kernel's logits equal the oracle's bit for bit and the vocabulary is full of ties -- ids and values must equal
    oracle.tree.stable_topk exactly (value descending, lowest index first)."""
    from eagle_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(N + K)
    M, k = B * n, 10
    x = torch.randint(-1, 2, (M, K), generator=g).to(BF)
    if K > 256:
        x[:, 256:] = 0                                               # |logit| <= 256: integers bf16 holds exactly
    W = torch.randint(-1, 2, (N, K), generator=g).to(BF)
    x[5] = 0                                                         # all-equal row: ids 0..9
    assert ops.lm_head_topk_rows_ok(M, N, K, k)
    ti = torch.zeros((B, n, k), dtype=torch.int64, device=d)
    tv = torch.zeros((B, n, k), dtype=BF, device=d)
    ops.lm_head_topk_into_table(x.to(d), ops.pack_weight(W.to(d)) if M <= 16 else W.to(d), k, ti, tv, 0, n, M=M)
    logits = (x.float().to(d) @ W.float().to(d).t()).cpu()
    assert float(logits.abs().max()) <= 256 and torch.equal(logits, logits.to(BF).float())
    ov, oi = ot.stable_topk(logits.to(BF), k)
    assert torch.equal(ti.reshape(M, k).cpu(), oi), "ids differ from the stable-sort oracle on exactly representable logits"
    assert torch.equal(tv.reshape(M, k).cpu(), ov)
    assert ti.reshape(M, k)[5].tolist() == list(range(10))


# ------------------------------------------------------------------ real-dims drafter cycle
def _check_table_against_oracle_logits(ids_g, vals_g, logits_o, logits_x, what):
    """Device top-k table vs the oracle's logits of the same rows (the oracle loop was fed the device's tokens, so row r of
    both sides saw the same inputs up to rounding).  logits_o: the oracle in bf16 (the reference's arithmetic), logits_x:
    the oracle in fp32 on the same bf16 weights.  Values: close() against the bf16 oracle, and no further from the fp32
    run than the bf16 oracle itself is (rel-L2).  Ids: equal at every rank where the oracle's own ordering is separated
    from both neighbours by more than the observed device-vs-oracle deviation allows."""
    R, k = ids_g.shape
    lf = logits_o.float()
    at = torch.gather(lf, 1, ids_g)
    close(vals_g, at, what=what + " top-k values vs oracle logits")
    ax = torch.gather(logits_x.float(), 1, ids_g)
    e_dev, e_ref = rel_l2(vals_g, ax), rel_l2(at, ax)
    assert e_dev <= 1.25 * e_ref + 2e-3, f"{what}: device {e_dev:.4f} vs fp32 oracle, the bf16 oracle itself {e_ref:.4f}"
    dev_err = (vals_g.float() - at).abs().max(dim=1, keepdim=True).values
    s = torch.sort(lf, dim=1, descending=True, stable=True)
    top = s.values[:, :k + 1]
    margin = 2 * dev_err + bf16_ulp(top[:, :k])
    # completeness: no vocabulary entry the device left out lies clearly above the device's k-th value
    rest = lf.scatter(1, ids_g, float("-inf")).max(dim=1).values
    assert bool((rest <= vals_g.float()[:, -1] + margin[:, -1]).all()), what + ": an entry above the device's k-th value is missing"
    gap_dn = top[:, :k] - top[:, 1:k + 1]
    gap_up = torch.cat([torch.full((R, 1), float("inf")), gap_dn[:, :-1]], dim=1)
    untied = (gap_dn > margin) & (gap_up > margin)
    assert torch.equal(ids_g[untied], s.indices[:, :k][untied]), what + ": ids differ where the oracle's logits are untied"
    return float(untied.float().mean()), float(untied[:, 0].float().mean()), e_dev, e_ref


def test_real_dims_drafter_cycles_against_the_oracle_loop():
    """configs[1] at full size (resources/eagle_config.json dims: D 4096, I 11008, 32 heads, V 32000): the device-resident
    cycle against oracle.tree.topk_generate driven by oracle.draft_model.forward ON THE CPU (bf16, the reference's op
    order; and once more in fp32 as the exact yardstick).  The oracle loop is fed the device's tokens (forced_ids) so that
    both sides expand the same tree; per flat row the device's top-k values must match the oracle's logits at those ids
    (bf16 tolerance) and its ids must equal the oracle's stable top-k wherever that is untied; acceptance, next tokens,
    lengths and compaction are bit-exact given the device's table.

    Weights: N(0, 0.02) with fc scaled by 0.4 and o_proj / down_proj by 0.5.  Unscaled, the random head is an expanding
    map -- its output, fed back as the next depth's hidden state, doubles per depth and every rounding difference with
    it: the CPU oracle's own bf16 and fp32 runs then disagree by 27 % and 50 % (rel-L2 of the logits) at depths 3 and 4,
    so no tolerance could tell a right kernel from a wrong one.  A trained head maps hidden states to hidden states at
    unit gain; the scaling gives this one that property (oracle bf16 vs fp32: 0.8-3.3 % per row)."""
    from eagle_amd.speculative import EagleDrafter
    d = dev()
    dims = dm.DraftDims()
    w = dm.init_weights(dims, seed=77)
    w["fc.weight"] *= 0.4
    w["layers.0.self_attn.o_proj.weight"] *= 0.5
    w["layers.0.mlp.down_proj.weight"] *= 0.5
    g = torch.Generator().manual_seed(5)
    lm_cpu = (torch.randn(dims.vocab, dims.hidden, generator=g) * 0.02).to(BF)
    m = _model(dims, w)
    wb = {k: v.to(BF) for k, v in w.items()}
    wx = {k: v.float() for k, v in wb.items()}
    lm_x = lm_cpu.float()
    del w
    tb = ot.generate_tree_buffers()
    drafter = EagleDrafter(m, lm_cpu.to(d), batch=1, max_ctx=256)

    def draft_fn(hh, ii, past_key_values=None, position_ids=None, tree_mask=None, use_cache=True):
        with torch.no_grad():
            ww = wb if hh.dtype == BF else wx
            return dm.forward(ww, dims, hh, ii, position_ids=position_ids, past_key_values=past_key_values, use_cache=True,
                              tree_mask=tree_mask)

    ctx = 24
    tokens = torch.randint(0, dims.vocab, (1, ctx + 1), generator=g)
    hidden = torch.randn(1, ctx, dims.hidden, generator=g).to(BF)
    with torch.no_grad():
        ids_g, vals_g = drafter.prefill(hidden.to(d), tokens.to(d))
        ids_g, vals_g = ids_g[0].cpu().clone(), vals_g[0].cpu().clone()
        _, _, stable, dbg = ot.topk_generate(draft_fn, lm_cpu, tb, hidden, tokens, None, forced_ids=ids_g)
        _, _, stable_x, dbx = ot.topk_generate(draft_fn, lm_x, tb, hidden.float(), tokens, None, forced_ids=ids_g)
        checked = [_check_table_against_oracle_logits(ids_g, vals_g, dbg["logits"], dbx["logits"], "prefill cycle")]
        prev_len = ctx
        tgt_kv = torch.randn(2, 1, 2, 128, 16, generator=g).to(BF)
        tgt_kv_g = tgt_kv.to(d).clone()
        for cycle in range(2):
            tgt_logits = (torch.randn(tb.n_nodes, dims.vocab, generator=g) * 2).to(BF)
            tgt_hidden = torch.randn(1, tb.n_nodes, dims.hidden, generator=g).to(BF)
            root = int(tokens[0, -1])
            tree_tok, cart = ot.tree_candidates(root, ids_g, tb)            # the DEVICE's table
            node = 0
            for _ in range(cycle + 2):
                kids = [i for i in range(tb.n_nodes) if tb.parent[i] == node]
                if not kids:
                    break
                tgt_logits[node, int(tree_tok[kids[0]])] = 50.0
                node = kids[0]
            best, alen, nxt, _ = ot.greedy_accept(cart, ot.first_argmax(tgt_logits), tb)
            ref_kv = tgt_kv.clone()
            new_len, acc_h = ot.compact([ref_kv], tgt_hidden, tb, best, alen, prev_len)
            prev_dev = torch.tensor([prev_len], dtype=torch.int32, device=d)
            res = drafter.verify(tokens[:, -1].to(d), tgt_logits.to(d)[None], prev_dev)
            acc_h_g = drafter.compact(prev_dev, tgt_kv_g, tgt_hidden.to(d))
            assert res[0].tolist() == [best, alen, nxt, new_len] and alen >= cycle + 2
            path_tokens = [int(cart[best, j]) for j in range(1, alen + 1)] + [nxt]
            assert drafter.next_ids[0, : alen + 1].tolist() == path_tokens
            assert torch.equal(tgt_kv_g.cpu(), ref_kv) and torch.equal(acc_h_g[0, : alen + 1].cpu(), acc_h[0])
            tgt_kv = ref_kv
            tokens = torch.cat([tokens, torch.tensor([path_tokens])], dim=1)
            ids_g, vals_g = drafter.draft(acc_h_g, drafter.next_ids)
            ids_g, vals_g = ids_g[0].cpu().clone(), vals_g[0].cpu().clone()
            _, _, stable, dbg = ot.topk_generate(draft_fn, lm_cpu, tb, acc_h, tokens, stable, forced_ids=ids_g)
            _, _, stable_x, dbx = ot.topk_generate(draft_fn, lm_x, tb, acc_h.float(), tokens, stable_x, forced_ids=ids_g)
            checked.append(_check_table_against_oracle_logits(ids_g, vals_g, dbg["logits"], dbx["logits"], f"cycle {cycle}"))
            prev_len = new_len
            assert int(drafter.stable_len[0]) == stable[0][0].shape[2] == tokens.shape[1] - 1
        drafter.check_errors()
    MARGINS["real-dims cycle: share of (row, rank) slots with ids checked | of rank-0 slots | worst rel-L2 device vs fp32 oracle"] = \
        (min(c[0] for c in checked), min(c[1] for c in checked), max(c[2] for c in checked))
    MARGINS["real-dims cycle: worst rel-L2 of the bf16 oracle itself vs the fp32 oracle"] = (0.0, 0.0, max(c[3] for c in checked))
    assert sum(c[1] for c in checked) / len(checked) >= 0.4, f"too few rows had an untied top-1 to check: {checked}"


def test_batch1_head_without_logits_equals_the_two_step_head_bit_for_bit():
    """One request's tree steps: lm_head + top-k out of the weight stream's candidate epilogue (EagleDrafter.head_topk_fused, the
    default) against the same drafter with lm_head logits + eagle_topk_vocab.  Both round the same fp32 sums to bf16 and order
    by (value, lowest index), so tables, acceptance and lengths must be EQUAL over whole cycles (real A dims)."""
    from eagle_amd.speculative import EagleDrafter
    d = dev()
    dims = dm.DraftDims()
    w = dm.init_weights(dims, seed=3)
    w["fc.weight"] *= 0.4
    w["layers.0.self_attn.o_proj.weight"] *= 0.5
    w["layers.0.mlp.down_proj.weight"] *= 0.5
    g = torch.Generator().manual_seed(6)
    lm = (torch.randn(dims.vocab, dims.hidden, generator=g) * 0.02).to(BF).to(d)
    m = _model(dims, w)
    del w
    tb = ot.generate_tree_buffers()
    ctx = 40
    tokens = torch.randint(0, dims.vocab, (1, ctx + 1), generator=g).to(d)
    hidden = torch.randn(1, ctx, dims.hidden, generator=g).to(BF).to(d)
    tgt_logits = [(torch.randn(tb.n_nodes, dims.vocab, generator=g) * 2).to(BF).to(d) for _ in range(3)]
    tgt_hidden = [torch.randn(1, tb.n_nodes, dims.hidden, generator=g).to(BF).to(d) for _ in range(3)]
    runs = []
    for fused in (True, False):
        dr = EagleDrafter(m, lm, batch=1, max_ctx=256)
        dr.head_topk_fused = fused
        assert dr._head_ws is not None, "the fused path would not be taken at these dims"
        tgt_kv = torch.zeros(2, 1, 2, 128, 16, dtype=BF, device=d)
        rec = []
        with torch.no_grad():
            ids, vals = dr.prefill(hidden, tokens)
            rec.append((ids.clone(), vals.clone()))
            prev = torch.tensor([ctx], dtype=torch.int32, device=d)
            root = tokens[:, -1].clone()
            for c in range(3):
                res = dr.verify(root, tgt_logits[c][None], prev)
                acc_h = dr.compact(prev, tgt_kv, tgt_hidden[c], root_io=root, update_prev_len=True)
                ids, vals = dr.draft(acc_h, dr.next_ids)
                rec.append((ids.clone(), vals.clone(), res.clone(), dr.next_ids.clone(), dr.stable_len.clone()))
            dr.check_errors()
        runs.append(rec)
    for a, b in zip(*runs):
        for x, y in zip(a, b):
            assert torch.equal(x, y)


# ------------------------------------------------------------------ the whole g3 fixture on the device
def test_loss_fixture_every_output_and_the_mixed_precision_case(golden_dir):
    """g3 fixture = outputs of the reference's own loss statements (eagle/train.py:106-123 executed by
    tests/golden/make_golden.py) in bf16: cases s / m / q and x (accelerate mixed precision: predict arrives as fp32,
    eagle/train.py:121-122).  Checks what test_loss_matches_reference_eager_fixture leaves out: vloss, the agreement count
    (eagle/train.py:115-117) and d loss / d predict = dlogits . lm_head + the smooth-L1 part."""
    from eagle_amd import ops
    fx = load_file(os.path.join(golden_dir, "g3_loss_bf16.safetensors"))
    d = dev()
    for c in ("s", "m", "q", "x"):
        V = fx[f"{c}.t_logits"].shape[-1]
        t = fx[f"{c}.t_logits"].reshape(-1, V)
        o = fx[f"{c}.o_logits"].reshape(-1, V)
        D = fx[f"{c}.predict"].shape[-1]
        pred = fx[f"{c}.predict"].reshape(-1, D)
        assert torch.equal(pred.to(BF).to(pred.dtype), pred), "the fp32 predict of case x holds bf16 values"
        pred_bf = pred.to(BF)
        targ = fx[f"{c}.target"].reshape(-1, D)
        mask = fx[f"{c}.mask"].reshape(-1).float()
        stats = ops.ce_partial(t.to(d), o.to(d), 0, pred_bf.to(d), targ.to(d))
        sums, lse_t, lse_o, _ = ops.ce_combine(stats, mask.to(d))
        ploss, vloss, correct = float(sums[0]), float(sums[1]), int(sums[2])
        assert abs(ploss - float(fx[f"{c}.ploss"])) <= 1e-2 * abs(float(fx[f"{c}.ploss"])), c
        assert abs(vloss - float(fx[f"{c}.vloss"])) <= 1e-2 * abs(float(fx[f"{c}.vloss"])), (c, vloss, float(fx[f"{c}.vloss"]))
        assert abs(vloss + 0.1 * ploss - float(fx[f"{c}.loss"])) <= 1e-2 * abs(float(fx[f"{c}.loss"])), c
        assert correct == int(fx[f"{c}.correct"]), (c, correct)          # agreement count, eagle/train.py:115-117
        assert int(sums[3]) == int(mask.sum())
        dlog, dpv = ops.ce_smoothl1_bwd(t.to(d), o.to(d), lse_t, lse_o, mask.to(d), 0.1, 1.0, pred_bf.to(d), targ.to(d))
        ref_dl = fx[f"{c}.dlogits"].reshape(-1, V)
        close(dlog, ref_dl, rtol=2e-2, atol=float(ref_dl.float().abs().max()) * 8e-3, what="dlogits vs eager bf16")
        if f"{c}.W" in fx:
            # d predict: the lm_head data gradient of the HIP dlogits (fp32 torch here: the fixture's D = 32 / 64 is below the
            # MFMA GEMM's K step; the GEMM itself is covered by test_gemm_nt / test_gpu_train) + the HIP smooth-L1 part
            dpred = dlog.float() @ fx[f"{c}.W"].float().to(d) + dpv.float()
            ref_dp = fx[f"{c}.dpredict"].reshape(-1, D)
            close(dpred, ref_dp, rtol=2e-2, atol=float(ref_dp.float().abs().max()) * 1e-2, what=f"dpredict vs eager ({c})")
