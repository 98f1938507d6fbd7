"""What does a fork / join inside a captured HIP graph cost, and do two branches really overlap?  (round 3, batch-1 cycle)

Three graphs of `reps` repetitions each, A dims, 4 rows:
  serial   lm_head GEMM -> top-k (2 launches) -> 33.5 MB weight stream (the hidden half of fc), one stream
  forked   lm_head GEMM -> { top-k | 33.5 MB weight stream on a second captured stream } -> join
  chain    the same launches with the fork / join events but nothing to overlap (top-k on the side stream, stream after join)
and the pure boundary cost: a chain of 64 one-workgroup kernels on one stream vs alternating between two captured streams."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops  # noqa: E402

BF = torch.bfloat16
d = torch.device("cuda:0")
g = torch.Generator(device=d).manual_seed(0)


def rnd(*s):
    return (torch.randn(s, generator=g, device=d, dtype=torch.float32) * 0.02).to(BF)


V, D, M, k = 32000, 4096, 4, 10
W_lm, W_hid, W_dn = rnd(V, D), rnd(D, D), rnd(D, 11008)
x = rnd(M, D) * 50
act = rnd(M, 11008) * 50
logits = torch.empty((M, V), dtype=BF, device=d)
part = torch.empty((M, D), dtype=BF, device=d)
xo = torch.empty((M, D), dtype=BF, device=d)
oi = torch.empty((M, k), dtype=torch.int64, device=d)
ov = torch.empty((M, k), dtype=BF, device=d)
ws = torch.empty(ops.query("eagle_topk_workspace_bytes", M, V, k), dtype=torch.uint8, device=d)
tiny_in = torch.zeros((4, 40), dtype=torch.int64, device=d)
tidx = torch.arange(4, dtype=torch.int64, device=d)
tiny_out = torch.empty((1, 4), dtype=torch.int64, device=d)
side = torch.cuda.Stream()


def body(mode):
    ops.linear_skinny(act, W_dn, out=xo)                      # down projection: the predecessor of lm_head
    ops.linear_skinny(x, W_lm, out=logits)
    if mode == "serial":
        ops.topk(logits, k, out_idx=oi, out_val=ov, workspace=ws)
        ops.linear_skinny(x, W_hid, out=part)
        return
    main = torch.cuda.current_stream()
    e0 = torch.cuda.Event()
    e0.record(main)
    side.wait_event(e0)
    with torch.cuda.stream(side):
        if mode == "forked":
            ops.linear_skinny(x, W_hid, out=part)
        else:
            ops.topk(logits, k, out_idx=oi, out_val=ov, workspace=ws)
        e1 = torch.cuda.Event()
        e1.record(side)
    if mode == "forked":
        ops.topk(logits, k, out_idx=oi, out_val=ov, workspace=ws)
    main.wait_event(e1)
    if mode == "chain":
        ops.linear_skinny(x, W_hid, out=part)


def tiny_chain(two_streams, n=64):
    main = torch.cuda.current_stream()
    for i in range(n):
        if two_streams and i % 2:
            e0 = torch.cuda.Event()
            e0.record(main)
            side.wait_event(e0)
            with torch.cuda.stream(side):
                ops.tree_gather_ids(tiny_in, tidx, B=1, out=tiny_out)
                e1 = torch.cuda.Event()
                e1.record(side)
            main.wait_event(e1)
        else:
            ops.tree_gather_ids(tiny_in, tidx, B=1, out=tiny_out)


def timed(fn, reps, replays=30):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(replays):
        t0 = time.perf_counter()
        gr.replay()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] / reps * 1e6, ts[0] / reps * 1e6


if __name__ == "__main__":
    for rnd_ in range(2):
        for mode in ("serial", "forked", "chain"):
            med, lo = timed(lambda: body(mode), 10)
            print(f"{mode:8s} per repetition: median {med:7.2f} us  min {lo:7.2f} us", flush=True)
        for two in (False, True):
            med, lo = timed(lambda: tiny_chain(two), 4)
            print(f"tiny chain of 64, {'two streams' if two else 'one stream '}: median {med / 64:6.2f} us per kernel  min {lo / 64:6.2f}", flush=True)
