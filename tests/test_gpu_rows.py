"""eagle_gemm_rows_bf16 (decode batches of up to 128 rows in one weight pass) against fp32 torch on the same inputs, through
the C ABI: every row-tile count (2, 4, 6, 8, 12 tiles of 16 rows), split and unsplit K, both epilogues, gathered / concatenated / doubly indirected rows."""
import pytest
import torch

from test_gpu_parity import BF, close, dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(17, 4096, 4096), (32, 4096, 4096), (33, 6144, 4096), (64, 4096, 8192), (84, 4096, 4096),
                                   (96, 4096, 12288), (128, 4096, 4096), (128, 24576, 4096), (128, 151936, 4096), (32, 151936, 4096),
                                   (1, 128, 128), (20, 1000, 256), (128, 200, 384), (50, 72, 1024), (192, 4096, 4096), (150, 12288, 4096), (161, 4096, 8192),
                                   (192, 600, 256),
                                   # the many-tile 81..128-row form (gemm_rows_wide_kernel): whole rounds of up to 384-column tiles
                                   (96, 151936, 4096), (100, 70000, 256), (128, 40000, 128), (81, 33000, 384), (128, 128256, 1024),
                                   # 3/4-width column tiles chosen by the round count (q|k|v at 128 / 192 rows) and odd widths beside them
                                   (128, 12288, 4096), (192, 12288, 4096), (96, 6000, 1024), (128, 12200, 512)])
def test_gemm_rows_linear(M, N, K):
    from eagle_amd import ops
    g = torch.Generator().manual_seed(M * 131 + N + K)
    x = (torch.randn(M, K, generator=g)).to(BF).to(dev())
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF).to(dev())
    b = torch.randn(N, generator=g).to(BF).to(dev())
    r = torch.randn(M, N, generator=g).to(BF).to(dev())
    ref = x.float() @ W.float().t()
    close(ops.linear_rows(x, W), ref, what="rows plain")
    ref2 = (ref + b.float()).to(BF).float() + r.float()
    close(ops.linear_rows(x, W, bias=b, res=r), ref2, what="rows bias+res")


@pytest.mark.parametrize("M,I,K", [(32, 12288, 4096), (128, 12288, 4096), (84, 11008, 4096), (24, 200, 128), (100, 64, 512), (192, 12288, 4096), (130, 96, 256)])
def test_gemm_rows_silu_mul(M, I, K):
    from eagle_amd import ops
    g = torch.Generator().manual_seed(7 + M)
    x = torch.randn(M, K, generator=g).to(BF).to(dev())
    Wgu = (torch.randn(2 * I, K, generator=g) / K ** 0.5).to(BF).to(dev())     # stacked gate|up: W2 is a view of it
    gt = (x.float() @ Wgu[:I].float().t()).to(BF)
    up = (x.float() @ Wgu[I:].float().t()).to(BF)
    ref = torch.nn.functional.silu(gt) * up
    close(ops.linear_rows(x, Wgu[:I], W2=Wgu[I:]), ref, what="rows silu*up")


def test_gemm_rows_gather_concat_and_double_indirection():
    from eagle_amd import ops
    g = torch.Generator().manual_seed(11)
    V, D, M = 1000, 256, 70
    emb = torch.randn(V, D, generator=g).to(BF)
    hid = torch.randn(90, D, generator=g).to(BF)
    W = (torch.randn(D, 2 * D, generator=g) / (2 * D) ** 0.5).to(BF)
    bias = torch.randn(D, generator=g).to(BF)
    ids = torch.randint(0, V, (M,), generator=g)
    prow = torch.randint(0, 90, (M,), generator=g)
    ref = torch.cat([emb[ids], hid[prow]], -1).float() @ W.float().t() + bias.float()
    y = ops.linear_rows(emb.to(dev()), W.to(dev()), bias=bias.to(dev()), idx0=ids.to(dev()), x1=hid.to(dev()),
                        idx1=prow.to(dev()), M=M)
    close(y, ref, what="rows gather+concat")
    B, n = 24, 4
    hid2 = torch.randn(B * n, D, generator=g).to(BF)
    flat = torch.randint(0, V, (B, 7, 10), generator=g)           # flat top-k table per request
    tree_idx = torch.tensor([0, 1, 2, 10])
    row0 = 2
    ids2 = flat[:, row0:].reshape(B, -1)[:, tree_idx]             # [B, n]
    ref = torch.cat([emb[ids2.reshape(-1)], hid2], -1).float() @ W.float().t()
    fd = flat.to(dev())
    y = ops.linear_rows(emb.to(dev()), W.to(dev()), idx0=fd[:, row0], x1=hid2.to(dev()), M=B * n,
                        ind0=tree_idx.to(dev()), idx0_bstride=7 * 10)
    close(y, ref, what="rows double indirection")


def test_gemm_rows_wide_form_gathered_rows_and_switch():
    """The 8-wave form takes gathered x rows too, and bit 9 of the diagnostic word routes the same call to the 128-column form:
    both against fp32 torch (the two walk K from different staggered starting points, so they agree to rounding, not bit for bit)."""
    from eagle_amd import ops
    from eagle_amd._hip import lib
    g = torch.Generator().manual_seed(23)
    M, N, K = 120, 48000, 512
    table = torch.randn(500, K, generator=g).to(BF).to(dev())
    ids = torch.randint(0, 500, (M,), generator=g).to(dev())
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF).to(dev())
    ref = table[ids].float() @ W.float().t()
    try:
        y_wide = ops.linear_rows(table, W, idx0=ids, M=M)
        lib().eagle_debug_set_rows(0, 0, 0x200)
        y_narrow = ops.linear_rows(table, W, idx0=ids, M=M)
    finally:
        lib().eagle_debug_set_rows(0, 0, 0)
    close(y_wide, ref, what="rows wide gather")
    close(y_narrow, ref, what="rows narrow gather")
    assert (y_wide.float() - y_narrow.float()).abs().max().item() <= 2 ** -5


@pytest.mark.parametrize("B,n,N,K", [(32, 4, 151936, 4096), (32, 3, 70000, 256), (25, 5, 33000, 384), (128, 1, 40008, 128)])
def test_lm_head_topk_without_logits(B, n, N, K):
    """eagle_lm_head_topk_rows_bf16 (candidates out of the GEMM epilogue + the final merge) against the two-step path
    (eagle_gemm_rows_bf16 + eagle_topk_vocab_grouped) on the same rows: identical ids -- bf16 logits over a big vocabulary are full
    of ties -- and identical values, written into the same [B, R, k] table slots; gathered x rows; a planted all-equal row."""
    from eagle_amd import ops
    g = torch.Generator().manual_seed(B * n + N)
    M, k, R = B * n, 10, 7
    xs = torch.randn(M + 9, K, generator=g).to(BF).to(dev())
    idx = torch.randperm(M + 9, generator=g)[:M].to(dev())
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF).to(dev())
    xs[idx[3]] = 0                                         # a row of all-zero logits: ids 0..9
    assert ops.lm_head_topk_rows_ok(M, N, K, k)
    ti = torch.full((B, R, k), -7, dtype=torch.int64, device=dev()); tv = torch.zeros((B, R, k), dtype=BF, device=dev())
    ops.lm_head_topk_into_table(xs, W, k, ti, tv, 2, n, idx0=idx, M=M)
    ri = torch.full((B, R, k), -7, dtype=torch.int64, device=dev()); rv = torch.zeros((B, R, k), dtype=BF, device=dev())
    logits = ops.linear_rows(xs, W, idx0=idx, M=M)
    ws = torch.empty(ops.query("eagle_topk_workspace_bytes", M, N, k), dtype=torch.uint8, device=dev())
    ops.topk_into_table(logits, k, ri, rv, 2, n, ws)
    assert torch.equal(ti, ri), "fused lm_head top-k ids differ from the two-step path"
    assert torch.equal(tv, rv)
    flat = ti[:, 2:2 + n].reshape(M, k)
    assert flat[3].tolist() == list(range(10))
    ev, ei = torch.topk(logits.float(), k, dim=1)          # values: the k largest logits of every row (ids may tie)
    assert torch.equal(tv[:, 2:2 + n].reshape(M, k).float(), ev)


@pytest.mark.parametrize("M,N,K", [(17, 4096, 4096), (32, 12288, 4096), (64, 4096, 8192), (96, 4096, 12288), (128, 4096, 4096), (128, 151936, 4096),
                                   (96, 70000, 256), (20, 1000, 256), (128, 200, 384), (50, 72, 1024), (192, 4096, 4096), (150, 12288, 4096),
                                   (128, 12200, 512), (100, 33001, 384)])
def test_gemm_rows_packed_weights_are_bit_equal(M, N, K):
    """eagle_gemm_rows_bf16 over eagle_pack_weight_bf16 images (whole-cache-line weight reads) against the same call over
    the row-major matrix: both forms of the kernel (4 + loader waves; the 8-wave many-tile form), split and unsplit K, ragged
    N, bias + residual, SiLU*up, gathered rows -- identical output bits (same tiles, same K order)."""
    from eagle_amd import ops
    g = torch.Generator().manual_seed(M * 31 + N + K)
    x = torch.randn(M, K, generator=g).to(BF).to(dev())
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF).to(dev())
    b = torch.randn(N, generator=g).to(BF).to(dev())
    r = torch.randn(M, N, generator=g).to(BF).to(dev())
    Wp = ops.pack_weight(W)
    assert torch.equal(ops.linear_rows(x, Wp), ops.linear_rows(x, W))
    assert torch.equal(ops.linear_rows(x, Wp, bias=b, res=r), ops.linear_rows(x, W, bias=b, res=r))
    idx = torch.randperm(M, generator=g).to(dev())
    assert torch.equal(ops.linear_rows(x, Wp, idx0=idx, M=M), ops.linear_rows(x, W, idx0=idx, M=M))
    if N % 32 == 0 and N <= 24576:
        I = N // 2                                                    # stacked gate|up: W2 a view of the row-major matrix
        gp, up = ops.pack_weight(W[:I]), ops.pack_weight(W[I:])
        assert torch.equal(ops.linear_rows(x, gp, W2=up), ops.linear_rows(x, W[:I], W2=W[I:]))
    if ops.lm_head_topk_rows_ok(M, N, K, 10):
        B, n = (M // 4, 4) if M % 4 == 0 else (M, 1)
        t0 = torch.zeros((B, n, 10), dtype=torch.int64, device=dev()); v0 = torch.zeros((B, n, 10), dtype=BF, device=dev())
        t1, v1 = t0.clone(), v0.clone()
        ops.lm_head_topk_into_table(x, W, 10, t0, v0, 0, n, M=M)
        ops.lm_head_topk_into_table(x, Wp, 10, t1, v1, 0, n, M=M)
        assert torch.equal(t0, t1) and torch.equal(v0, v1)


def test_lm_head_topk_shapes_outside_the_wide_form_are_refused():
    from eagle_amd import ops
    assert not ops.lm_head_topk_rows_ok(32, 151936, 4096, 10)      # 32 rows: the register-resident stream's case
    assert not ops.lm_head_topk_rows_ok(128, 4096, 4096, 10)       # a short head
    assert not ops.lm_head_topk_rows_ok(128, 151936, 4000, 10)     # K % 128


def test_gemm_rows_narrow_tiles_on_and_off():
    """3/4-width column tiles (picked where they fill whole rounds of resident workgroups) and full-width tiles: the same product,
    linear with bias + residual and SiLU*up, at 128 and 192 rows."""
    from eagle_amd import ops
    from eagle_amd._hip import lib
    g = torch.Generator().manual_seed(31)
    for M in (128, 192):
        N, K = 12288, 1024
        x = torch.randn(M, K, generator=g).to(BF).to(dev())
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF).to(dev())
        b = torch.randn(N, generator=g).to(BF).to(dev())
        r = torch.randn(M, N, generator=g).to(BF).to(dev())
        ref = ((x.float() @ W.float().t()) + b.float()).to(BF).float() + r.float()
        Wgu = (torch.randn(2 * 6144, K, generator=g) / K ** 0.5).to(BF).to(dev())
        refs = torch.nn.functional.silu((x.float() @ Wgu[:6144].float().t()).to(BF)) * (x.float() @ Wgu[6144:].float().t()).to(BF)
        try:
            for knob in (0, 0x400):
                lib().eagle_debug_set_rows(0, 0, knob)
                close(ops.linear_rows(x, W, bias=b, res=r), ref, what="rows narrow/full tiles")
                close(ops.linear_rows(x, Wgu[:6144], W2=Wgu[6144:]), refs, what="rows narrow/full tiles silu")
        finally:
            lib().eagle_debug_set_rows(0, 0, 0)


def test_gemm_rows_every_split_and_stage_setting():
    """The diagnostic knobs select the K split and the register stages: every combination gives the same product."""
    from eagle_amd import ops
    from eagle_amd._hip import lib
    g = torch.Generator().manual_seed(5)
    M, N, K = 100, 512, 2048
    x = torch.randn(M, K, generator=g).to(BF).to(dev())
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(BF).to(dev())
    r = torch.randn(M, N, generator=g).to(BF).to(dev())
    ref = (x.float() @ W.float().t()).to(BF).float() + r.float()
    try:
        for target in (1, 8, 16, 64):
            for steps in (1, 2, 4):
                for stages in (2, 3, 4, 0x103, 0x13, 0x33):
                    lib().eagle_debug_set_rows(target, steps, stages)
                    close(ops.linear_rows(x, W, res=r), ref, what="rows knobs")
                    close(ops.linear_rows(x[:30], W, res=r[:30]), ref[:30], what="rows knobs")
    finally:
        lib().eagle_debug_set_rows(0, 0, 0)


def test_gemm_rows_rejects_bad_shapes():
    from eagle_amd import ops
    from eagle_amd._hip import HipKernelError
    W = torch.zeros(16, 192, dtype=BF, device=dev())
    with pytest.raises(HipKernelError):
        ops.linear_rows(torch.zeros(2, 192, dtype=BF, device=dev()), W)       # K % 128 != 0
    W = torch.zeros(16, 256, dtype=BF, device=dev())
    with pytest.raises(HipKernelError):
        ops.linear_rows(torch.zeros(193, 256, dtype=BF, device=dev()), W)     # more than 192 rows
