"""The entry point that allocates nothing (eagle_topk_vocab_noalloc, include/eagle_hip.h) against eagle_topk_vocab,
through the C ABI, bit for bit, on two calls in a row (the second with a smaller input), plus its size limit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(logits, k):
    from eagle_amd import _hip
    rows, V = logits.shape
    dt = 0 if logits.dtype == torch.bfloat16 else 1
    d = logits.device
    ws = torch.empty(_hip.query("eagle_topk_workspace_bytes", rows, V, k), dtype=torch.uint8, device=d)
    i0 = torch.full((rows, k), -7, dtype=torch.int64, device=d)
    v0 = torch.zeros((rows, k), dtype=logits.dtype, device=d)
    i1, v1 = i0.clone(), v0.clone()
    _hip.call("eagle_topk_vocab", _hip.ptr(logits), logits.stride(0), dt, rows, V, k, _hip.ptr(i0), _hip.ptr(v0), _hip.ptr(ws),
              None, rows, _hip.stream())
    import gc
    gc.collect()                                     # garbage of earlier tests must not be freed between the two readings
    before = torch.cuda.memory_allocated()
    _hip.call("eagle_topk_vocab_noalloc", _hip.ptr(logits), logits.stride(0), dt, rows, V, k, _hip.ptr(i1), _hip.ptr(v1),
              None, rows, _hip.stream())
    assert torch.cuda.memory_allocated() <= before     # nothing was allocated through torch either
    torch.cuda.synchronize()
    # ... nor behind torch's back: once the scratch array's address is resolved (the call above), another call must leave
    # the device's free memory as it is (torch.cuda.memory_allocated cannot see a hipMalloc made inside the library)
    i2, v2 = torch.full_like(i1, -7), torch.zeros_like(v1)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    _hip.call("eagle_topk_vocab_noalloc", _hip.ptr(logits), logits.stride(0), dt, rows, V, k, _hip.ptr(i2), _hip.ptr(v2),
              None, rows, _hip.stream())
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before, "eagle_topk_vocab_noalloc changed the device's free memory"
    assert torch.equal(i2, i1) and torch.equal(v2, v1)
    return i0, v0, i1, v1


def test_noalloc_topk_matches_workspace_form_twice_then_smaller():
    from oracle.tree import stable_topk
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77)
    big = (torch.randn(26, 151936, generator=g) * 3).to(torch.bfloat16)
    big[3, 100:140] = big[3].max()                      # ties: lowest index first
    small = (torch.randn(5, 4099, generator=g) * 3).to(torch.bfloat16)
    f32 = torch.randn(3, 32000, generator=g)
    for x, k in ((big, 10), (small, 10), (f32, 1), (big[:2, :1000].contiguous(), 16)):
        i0, v0, i1, v1 = _both(x.to(d), k)
        assert torch.equal(i0, i1) and torch.equal(v0, v1)
        rv, ri = stable_topk(x.float(), k)
        assert torch.equal(i1.cpu(), ri) and torch.equal(v1.float().cpu(), rv)


def test_noalloc_topk_rejects_sizes_beyond_its_scratch():
    from eagle_amd import _hip
    d = torch.device("cuda:0")
    x = torch.zeros((160, 151936), dtype=torch.bfloat16, device=d)   # 160 rows x 149 chunks x 11 x 8 B > 2 MiB
    out = torch.empty((160, 10), dtype=torch.int64, device=d)
    rc = _hip.lib().eagle_topk_vocab_noalloc(_hip.ptr(x), x.stride(0), 0, 160, 151936, 10, _hip.ptr(out), None, None, 160,
                                             _hip.stream())
    assert rc == 1                                                     # EAGLE_ERR_SHAPE
    with pytest.raises(_hip.HipKernelError):
        _hip.call("eagle_topk_vocab_noalloc", _hip.ptr(x), x.stride(0), 0, 160, 151936, 10, _hip.ptr(out), None, None, 160,
                  _hip.stream())
