"""bench.py --gpus N: the launcher's argument and exit-code logic (CPU; no GPU call is made by the parent).

The reference is launched as `torchrun --nproc_per_node=N eagle/train.py` (pages/reproduce_eagle_1_paper.md:128-132); the
driver's contract is `python bench.py --gpus N`, so for N > 1 outside torchrun bench.py must start the N ranks itself."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_plan_single_process_and_torchrun_ranks_run_in_place():
    assert bench.launch_plan(None, {}, 0, "bench.py", []) is None
    assert bench.launch_plan(1, {}, 8, "bench.py", []) is None
    assert bench.launch_plan(8, {"WORLD_SIZE": "8"}, 8, "bench.py", []) is None          # the driver's torchrun form: a rank
    assert bench.launch_plan(None, {"WORLD_SIZE": "2"}, 1, "bench.py", []) is None       # rehearsal without --gpus


def test_plan_spawns_one_rank_per_gpu_with_the_same_flags():
    cmd = bench.launch_plan(4, {}, 8, "/x/bench.py", ["--gpus", "4", "--steps", "20", "--warmup", "5"], port=29512)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29512"
    assert cmd[-7:] == ["/x/bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"]


def test_plan_refuses_what_it_cannot_run():
    with pytest.raises(SystemExit) as e:
        bench.launch_plan(8, {}, 1, "bench.py", [])                                      # RCCL: one process per GPU
    assert e.value.code == 2
    with pytest.raises(SystemExit) as e:
        bench.launch_plan(4, {"WORLD_SIZE": "8"}, 8, "bench.py", [])                     # torchrun world != --gpus
    assert e.value.code == 2
    with pytest.raises(SystemExit):
        bench.launch_plan(0, {}, 8, "bench.py", [])
    # the gloo rehearsal shares the visible devices among the ranks
    assert bench.launch_plan(2, {"EAGLE_BENCH_BACKEND": "gloo"}, 1, "bench.py", []) is not None


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="the no-GPU refusals")
def test_parent_refuses_without_gpus_and_forwards_a_failing_childs_status():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"EAGLE_BENCH_BACKEND": "nccl"})
    assert r.returncode == 2 and "visible GPUs" in r.stderr and r.stdout.strip() == ""
    # gloo rehearsal on a box without a GPU: both ranks start, refuse ("needs an MI355X"), and the parent reports failure
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-train", "--no-batched"], {"EAGLE_BENCH_BACKEND": "gloo"})
    assert r.returncode != 0 and "needs an MI355X" in r.stderr and r.stdout.strip() == ""
