"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the
header declares, and the Python host mirrors the reference interface and fails loudly without
a GPU (no CPU fallback).  No compute calls here."""
import ctypes
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "eagle_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eagle_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from eagle_amd import _hip

    assert os.path.exists(_hip.LIB_PATH), "run `python -m eagle_amd.build` (or __graft_entry__.build())"
    h = ctypes.CDLL(_hip.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/eagle_hip.h but not exported"
    # the ctypes table binds exactly the declared functions
    assert sorted(_hip.SIGNATURES) == names


def test_export_set_equals_the_header():
    """Nothing is exported that include/eagle_hip.h does not declare (and vice versa)."""
    import subprocess
    from eagle_amd import _hip

    out = subprocess.run(["nm", "-D", "--defined-only", _hip.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in "TW"
                       and ln.split()[-1].startswith("eagle_")})
    assert exported == _declared()


def test_tree_layout_host_call():
    """eagle_tree_layout runs on the host (no launch): sizes of the 25-path tree and rejection of invalid trees."""
    from eagle_amd import ops
    from eagle_amd.tree import mc_sim_7b_63

    _, meta = ops.tree_layout(mc_sim_7b_63, 10, False)
    assert meta == {"n_nodes": 26, "max_depth": 5, "n_steps": 4, "n_fwd": 10, "n_flat_rows": 11, "P": 15, "W": 6,
                    "mask_bytes": 4 * 4 + 4 * 8 + 1 * 9 + 1 * 10, "step_n": [4, 4, 1, 1]}
    _, meta = ops.tree_layout(mc_sim_7b_63, 10, True)
    assert meta["step_n"] == [4, 8, 8, 3, 2] and meta["n_fwd"] == 25 and meta["n_steps"] == 5
    for bad in ([[0], [1, 0]], [[0], [10]], [[0], [0]], [[0, 0]], [[i % 10, i // 10] for i in range(70)]):
        with pytest.raises(ValueError):
            ops.tree_layout(bad, 10, False)


def test_header_cites_reference_lines():
    text = open(os.path.join(ROOT, "include", "eagle_hip.h")).read()
    for cite in ("eagle/llama2.py:515", "eagle/llama2.py:330-335", "eagle/llama2.py:226-273", "eagle/train.py:108-123"):
        assert cite in text


def test_llama2model_mirrors_reference_interface():
    from eagle_amd.config import DraftConfig
    from eagle_amd.llama2 import Llama2Model

    sig = inspect.signature(Llama2Model.__init__)
    assert list(sig.parameters)[1:] == ["config", "load_emb", "path", "bias"]
    fsig = inspect.signature(Llama2Model.forward)
    assert list(fsig.parameters)[1:] == [
        "hidden_states", "input_ids", "attention_mask", "position_ids", "past_key_values", "inputs_embeds",
        "use_cache", "output_attentions", "output_hidden_states", "return_dict", "std"]
    cfg = DraftConfig(hidden_size=64, intermediate_size=128, num_attention_heads=4, vocab_size=100)
    m = Llama2Model(cfg)
    keys = list(m.state_dict().keys())
    assert keys == [
        "embed_tokens.weight",
        "layers.0.self_attn.q_proj.weight", "layers.0.self_attn.k_proj.weight",
        "layers.0.self_attn.v_proj.weight", "layers.0.self_attn.o_proj.weight",
        "layers.0.mlp.gate_proj.weight", "layers.0.mlp.up_proj.weight", "layers.0.mlp.down_proj.weight",
        "layers.0.post_attention_layernorm.weight", "fc.weight", "fc.bias"]
    assert not m.embed_tokens.weight.requires_grad and m.fc.weight.requires_grad
    assert m.gradient_checkpointing is True
    assert "fc.bias" not in Llama2Model(cfg, bias=False).state_dict()
    m.tree_mask = torch.ones(1, 1, 2, 2)
    m.reset()
    assert m.tree_mask is None


def test_no_cpu_fallback():
    from eagle_amd import ops
    from eagle_amd._hip import HipKernelError
    from eagle_amd.config import DraftConfig
    from eagle_amd.llama2 import Llama2Model

    cfg = DraftConfig(hidden_size=64, intermediate_size=128, num_attention_heads=4, vocab_size=100)
    m = Llama2Model(cfg).to(torch.bfloat16).eval()
    with pytest.raises(HipKernelError):
        m(torch.zeros(1, 2, 64, dtype=torch.bfloat16), torch.zeros(1, 2, dtype=torch.long))
    with pytest.raises(HipKernelError):
        ops.topk(torch.zeros(2, 100, dtype=torch.bfloat16), 3)
    with pytest.raises(HipKernelError):
        ops.rmsnorm(torch.zeros(2, 64, dtype=torch.bfloat16), torch.ones(64, dtype=torch.bfloat16), 1e-6)


def test_product_never_imports_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "eagle_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_host_tree_buffers_match_oracle():
    from eagle_amd.tree import TreeBuffers
    from oracle import tree as ot

    for expand in ("wchild", "all"):
        tb, ob = TreeBuffers(expand=expand), ot.generate_tree_buffers(expand=expand)
        assert tb.n_nodes == ob.n_nodes and tb.parent == ob.parent and tb.depth == ob.depth
        assert [s.nodes for s in tb.steps] == ob.step_nodes
        for s, ti, pr, rn, am in zip(tb.steps, ob.step_tree_indices, ob.step_parent_row, ob.step_repeat_nums, ob.step_attn_mask):
            assert torch.equal(s.tree_indices, ti) and torch.equal(s.parent_row, pr) and s.repeat_nums == rn
            assert torch.equal(s.attn_mask, am) and torch.equal(s.attn_mask_u8.float(), am[0, 0])
        assert torch.equal(tb.tree_indices, ob.tree_indices)
        assert torch.equal(tb.retrieve_indices, ob.retrieve_indices)
        assert torch.equal(tb.tree_position_ids, ob.tree_position_ids)
        assert torch.equal(tb.tree_attn_mask, ob.tree_attn_mask)
        assert tb.flat_row == ob.flat_row
    # a second, irregular tree
    paths = [[0], [1], [0, 0], [0, 3], [1, 2], [1, 2, 0], [0, 3, 1], [0, 3, 1, 4]]
    tb, ob = TreeBuffers(paths), ot.generate_tree_buffers(paths)
    assert torch.equal(tb.tree_indices, ob.tree_indices) and torch.equal(tb.retrieve_indices, ob.retrieve_indices)
    assert [s.tree_indices.tolist() for s in tb.steps] == [t.tolist() for t in ob.step_tree_indices]
    with pytest.raises(ValueError):
        TreeBuffers([[0], [1, 0]])
    with pytest.raises(ValueError):
        TreeBuffers([[0], [10]])
