"""Pin the oracle: oracle.draft_model must reproduce, bit for bit, what the reference
(eagle/llama2.py, run by tests/golden/make_golden.py) produced on the same inputs."""
import os

import pytest
import torch
from safetensors.torch import load_file

from oracle import draft_model as dm
from oracle import loss as ol
from oracle.tree import generate_tree_buffers

TINY = dm.DraftDims(hidden=64, inter=128, heads=4, kv_heads=4, vocab=100)


def _w(fx, dtype):
    return {k[2:]: v.to(dtype) for k, v in fx.items() if k.startswith("w.")}


@pytest.mark.parametrize("tag,dtype", [("fp32", torch.float32), ("bf16", torch.bfloat16)])
def test_g1_tiny_bit_exact(golden_dir, tag, dtype):
    fx = load_file(os.path.join(golden_dir, f"g1_tiny_{tag}.safetensors"))
    w = _w(fx, dtype)
    with torch.no_grad():
        out = dm.forward(w, TINY, fx["a.hidden"], fx["a.ids"])
        assert torch.equal(out, fx["a.out"])
        o, kv = dm.forward(w, TINY, fx["a.hidden"], fx["a.ids"], use_cache=True)
        assert torch.equal(o, fx["b.out"]) and torch.equal(kv[0][0], fx["b.k"]) and torch.equal(kv[0][1], fx["b.v"])
        o2, kv2 = dm.forward(w, TINY, fx["c.hidden"], fx["c.ids"], past_key_values=kv, use_cache=True)
        assert torch.equal(o2, fx["c.out"]) and torch.equal(kv2[0][0], fx["c.k"]) and torch.equal(kv2[0][1], fx["c.v"])
        o3, kv3 = dm.forward(w, TINY, fx["d.hidden"], fx["d.ids"], past_key_values=kv2, use_cache=True)
        assert torch.equal(o3, fx["d.out"]) and torch.equal(kv3[0][0], fx["d.k"])
        tb = generate_tree_buffers()
        past = kv
        for d in range(2):
            od, past = dm.forward(w, TINY, fx[f"e{d}.hidden"], fx[f"e{d}.ids"], past_key_values=past,
                                  position_ids=fx[f"e{d}.pos"], use_cache=True, tree_mask=tb.step_attn_mask[d])
            assert torch.equal(od, fx[f"e{d}.out"])
            assert torch.equal(past[0][0], fx[f"e{d}.k"]) and torch.equal(past[0][1], fx[f"e{d}.v"])
        of = dm.forward(w, TINY, fx["f.hidden"], fx["f.ids"], attention_mask=fx["f.mask"].bool())
        assert torch.equal(of, fx["f.out"])


@pytest.mark.parametrize("tag,dtype", [("bf16", torch.bfloat16), ("fp32", torch.float32)])
def test_g2_real_dims_bit_exact(golden_dir, tag, dtype):
    fx = load_file(os.path.join(golden_dir, f"g2_real_{tag}.safetensors"))
    dims = dm.DraftDims()  # resources/eagle_config.json: D=4096 I=11008 H=32 V=32000
    w = {k: v.to(dtype) for k, v in dm.init_weights(dims, seed=1234).items()}
    with torch.no_grad():
        o, kv = dm.forward(w, dims, fx["hidden"], fx["ids"], use_cache=True)
        assert torch.equal(o, fx["out"]) and torch.equal(kv[0][0], fx["k"]) and torch.equal(kv[0][1], fx["v"])
        tb = generate_tree_buffers()
        o2, _ = dm.forward(w, dims, fx["t.hidden"], fx["t.ids"], past_key_values=kv, position_ids=fx["t.pos"],
                           use_cache=True, tree_mask=tb.step_attn_mask[0])
        assert torch.equal(o2, fx["t.out"])


def g3_big_weight(V, D, dtype):
    """Seed recipe of the q case's lm_head (tests/golden/make_golden.py: 151936 x 32, not stored)."""
    return (torch.randn(V, D, generator=torch.Generator().manual_seed(33)) * 0.3).to(dtype)


@pytest.mark.parametrize("tag,dtype", [("fp32", torch.float32), ("bf16", torch.bfloat16)])
def test_g3_loss_fixture(golden_dir, tag, dtype):
    """The fixture holds what the REFERENCE'S OWN loss statements (eagle/train.py:106-123 == eagle/train_tp.py:88-105,
    ast-extracted and executed by make_golden.py) computed; oracle.loss.eager_loss must match it bit for bit.
    q = the full Qwen3 vocabulary (N=4, V=151936); x (bf16 file) = train.py under accelerate bf16 mixed precision,
    where predict reaches the loss as fp32."""
    fx = load_file(os.path.join(golden_dir, f"g3_loss_{tag}.safetensors"))
    cases = ("s", "m", "q") + (("x",) if tag == "bf16" else ())
    for c in cases:
        W = g3_big_weight(151936, 32, dtype) if c == "q" else fx[f"{c}.W"]
        predict = fx[f"{c}.predict"].clone().requires_grad_(True)
        r = ol.eager_loss(predict, fx[f"{c}.target"], W, fx[f"{c}.mask"])
        r["out_logits"].retain_grad()
        r["loss"].backward()
        assert torch.equal(r["target_logits"], fx[f"{c}.t_logits"]) and torch.equal(r["out_logits"], fx[f"{c}.o_logits"])
        assert torch.equal(r["ploss"].float().reshape(1), fx[f"{c}.ploss"])
        assert torch.equal(r["vloss"].float().reshape(1), fx[f"{c}.vloss"])
        assert torch.equal(r["loss"].float().reshape(1), fx[f"{c}.loss"])
        assert torch.equal(r["correct"].reshape(1), fx[f"{c}.correct"])
        assert torch.equal(r["out_logits"].grad, fx[f"{c}.dlogits"])
        assert torch.equal(predict.grad, fx[f"{c}.dpredict"])
        # the wide-accumulation form (what the fused kernel targets) agrees within bf16 tolerance
        ex = ol.exact_loss_from_logits(fx[f"{c}.t_logits"], fx[f"{c}.o_logits"], fx[f"{c}.predict"],
                                       fx[f"{c}.target"], fx[f"{c}.mask"])
        rtol = 1e-5 if dtype == torch.float32 else 1e-2
        assert abs(float(ex["ploss"]) - float(fx[f"{c}.ploss"])) <= rtol * abs(float(ex["ploss"])) + 1e-6
        assert abs(float(ex["vloss"]) - float(fx[f"{c}.vloss"])) <= rtol * abs(float(ex["vloss"])) + 1e-6
        if dtype == torch.float32:
            assert int(ex["correct"]) == int(fx[f"{c}.correct"])
        dlog, dpv = ol.exact_grads_from_logits(fx[f"{c}.t_logits"], fx[f"{c}.o_logits"], fx[f"{c}.predict"],
                                               fx[f"{c}.target"], fx[f"{c}.mask"], g_p=0.1, g_v=1.0)
        tol = 1e-6 if dtype == torch.float32 else 2e-3
        assert (dlog.float() - fx[f"{c}.dlogits"].float().reshape(dlog.shape)).abs().max() <= tol


def test_make_eagle_input_shift():
    ids = torch.arange(1, 13).view(2, 6)
    lm = torch.tensor([[0, 0, 1, 1, 1, 1], [0, 1, 1, 1, 0, 0]])
    h = torch.arange(2 * 6 * 3, dtype=torch.float32).view(2, 6, 3)
    r = ol.make_eagle_input(ids, lm, h, max_len=5)
    assert r["input_ids"].tolist() == [[2, 3, 4, 5, 0], [8, 9, 10, 11, 0]]
    assert torch.equal(r["target"][:, :-1], h[:, 1:5]) and float(r["target"][:, -1].abs().sum()) == 0
    assert torch.equal(r["loss_mask"], lm[:, :5]) and torch.equal(r["hidden_states"], h[:, :5])
