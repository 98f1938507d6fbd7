"""PMC target (round 3): is the lm_head stream's FETCH_SIZE excess (1.085 x algorithmic in r02) real traffic or a counting artefact?
Launches, each behind a 1 GiB scrub (Infinity Cache cold), 3 per variant, in this order:
  0  default   keepx<RT=1>, staggered K walk, non-temporal loads          (the in-cycle lm_head)
  1  RT=2      two 16-row weight tiles per tile-step (round 1's form), staggered walk
  2  RT=1      no stagger
  3  RT=2      no stagger                                                  (round 1's kernel)
  4  default-policy loads (EAGLE_GEMM_KEEP_RESIDENT), RT=1, staggered
  5  reference: torch's sum over the same 262 MB as int32 (wide coalesced reads of a known byte count)
  6  the packed weight image (ops.pack_weight: every wave-instruction reads 1 KiB contiguous), non-temporal -- what the drafters stream
Run under rocprofv3 --pmc <counters>; tools/pmc_variants_table.py turns the CSV into a table (variant = dispatch order)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import ops  # noqa: E402
from eagle_amd._hip import lib  # noqa: E402

V, D, rows = 32000, 4096, 4
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
W = (torch.randn((V, D), generator=g, device=dev) * 0.02).to(torch.bfloat16)
x = torch.randn((rows, D), generator=g, device=dev).to(torch.bfloat16)
out = torch.empty((rows, V), device=dev, dtype=torch.bfloat16)
scrub = torch.empty(1 << 28, dtype=torch.float32, device=dev)
Wp = ops.pack_weight(W)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(21)]
n = 0
setv = lib().eagle_debug_set_skinny_variant
for variant, (word, resident) in enumerate([(3, False), (3 | (2 << 12), False), (3 | (1 << 15), False), (3 | (2 << 12) | (1 << 15), False),
                                            (3, True), (None, False), ("packed", False)]):
    if word == "packed":
        setv(3, 0, 0)
    elif word is not None:
        setv(word, 0, 0)
    for i in range(3):
        scrub.fill_(float(i))
        ev[n][0].record()
        if word is None:
            s = W.view(torch.int32).sum()
        elif word == "packed":
            ops.linear_skinny(x, Wp, out=out)
        else:
            ops.linear_skinny(x, W, out=out, resident=resident)
        ev[n][1].record()
        n += 1
setv(3, 0, 0)
torch.cuda.synchronize()
for v in range(7):
    print("variant", v, "us:", " ".join("%.1f" % (ev[3 * v + i][0].elapsed_time(ev[3 * v + i][1]) * 1e3) for i in range(3)))
