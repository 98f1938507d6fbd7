"""Run only bench.py's batch-32 leg (BASELINE configs[3]); target for a rocprofv3 kernel trace."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
print(json.dumps(bench.batched_leg(torch.device("cuda:0"))), flush=True)
