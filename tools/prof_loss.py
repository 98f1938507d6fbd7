"""PMC target: a few launches of the fused loss kernels (ce_partial + ce_combine forward, ce_bwd backward) at the
training shape N = 2048, V = 151936, D = 4096 and nothing else of note.
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/X/pmc_loss_fetch -- python3 tools/prof_loss.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd.train_bench import loss_roofline  # noqa: E402

r = loss_roofline(torch.device("cuda:0"), 2048, 151936, 4096, iters=4)
print("fwd %.0f GB/s (%.0f us)  bwd %.0f GB/s (%.0f us)" % (r["fwd"]["GB/s"], r["fwd"]["us"], r["bwd"]["GB/s"], r["bwd"]["us"]), flush=True)
