"""Checkpoint layout: what eagle/train_tp.py:140-146 writes is what
eagle_amd.checkpoint writes, key for key, and it reads back bit-exactly.  CPU only (no kernels run)."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
BF = torch.bfloat16

# the state_dict keys of the reference's Llama2Model (printed from the reference).
REFERENCE_KEYS = {
    "embed_tokens.weight", "fc.weight", "fc.bias",
    "layers.0.self_attn.q_proj.weight", "layers.0.self_attn.k_proj.weight", "layers.0.self_attn.v_proj.weight",
    "layers.0.self_attn.o_proj.weight", "layers.0.mlp.gate_proj.weight", "layers.0.mlp.up_proj.weight",
    "layers.0.mlp.down_proj.weight", "layers.0.post_attention_layernorm.weight",
}


def _model(seed):
    from eagle_amd.config import DraftConfig
    from eagle_amd.llama2 import Llama2Model
    torch.manual_seed(seed)
    cfg = DraftConfig(hidden_size=256, intermediate_size=256, num_attention_heads=2, vocab_size=512, max_position_embeddings=64)
    return cfg, Llama2Model(cfg).to(BF)


def test_checkpoint_layout_and_round_trip(tmp_path):
    from safetensors import safe_open
    from eagle_amd.checkpoint import load_checkpoint, save_checkpoint
    cfg, model = _model(0)
    src_cfg = tmp_path / "eagle_config.json"
    src_cfg.write_text(json.dumps({"architectures": ["LlamaForCausalLM"], "hidden_size": 256, "intermediate_size": 256,
                                   "num_attention_heads": 2, "num_hidden_layers": 1, "vocab_size": 512}))
    out = tmp_path / "epoch_0_step_10"
    save_checkpoint(model, str(src_cfg), out)
    assert sorted(os.listdir(out)) == ["config.json", "model.safetensors"]
    written = json.loads((out / "config.json").read_text())
    assert written["architectures"] == ["LlamaForCausalLMEagle"]          # eagle/train_tp.py:144
    assert written["hidden_size"] == 256 and written["vocab_size"] == 512
    with safe_open(str(out / "model.safetensors"), framework="pt") as f:
        assert set(f.keys()) == REFERENCE_KEYS                            # no "model." prefix, embed included, no lm_head
    _, other = _model(1)
    assert not torch.equal(other.fc.weight, model.fc.weight)
    res = load_checkpoint(other, out)
    assert not res.missing_keys and not res.unexpected_keys
    for (k, a), (_, b) in zip(model.state_dict().items(), other.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(FileExistsError):                                  # the reference's mkdir raises as well
        save_checkpoint(model, str(src_cfg), out)


def test_verifier_lm_head_reader(tmp_path):
    """eagle/train.py:215-228: lm_head.weight found through the weight map, columns cut to hidden_size."""
    from safetensors.torch import save_file
    from eagle_amd.checkpoint import load_verifier_lm_head
    from eagle_amd.llama2 import _load_verifier_tensor
    g = torch.Generator().manual_seed(0)
    head, emb = torch.randn(100, 32, generator=g).to(BF), torch.randn(100, 32, generator=g).to(BF)
    save_file({"lm_head.weight": head}, str(tmp_path / "model-00002-of-00002.safetensors"))
    save_file({"model.embed_tokens.weight": emb}, str(tmp_path / "model-00001-of-00002.safetensors"))
    (tmp_path / "config.json").write_text(json.dumps({"hidden_size": 32, "vocab_size": 100}))
    (tmp_path / "model.safetensors.index.json").write_text(json.dumps({"weight_map": {
        "lm_head.weight": "model-00002-of-00002.safetensors", "model.embed_tokens.weight": "model-00001-of-00002.safetensors"}}))
    assert torch.equal(load_verifier_lm_head(tmp_path), head)
    assert torch.equal(_load_verifier_tensor(str(tmp_path), "model.embed_tokens.weight"), emb)


def test_trainer_state_round_trip(tmp_path, monkeypatch):
    import cpu_ops
    cpu_ops.install(monkeypatch.setattr)
    from eagle_amd.checkpoint import load_checkpoint, load_trainer_state, save_checkpoint, save_trainer_state
    from eagle_amd.train import EagleTrainer
    g = torch.Generator().manual_seed(5)
    lm_head = (torch.randn(512, 256, generator=g) * 0.2).to(BF)
    cfg, model = _model(2)
    tr = EagleTrainer(model, lm_head, mode="tp", num_warmup_steps=2, num_training_steps=10)
    tr.m.copy_(torch.randn(tr.m.shape, generator=g))
    tr.v.copy_(torch.rand(tr.v.shape, generator=g))
    tr.step_count, tr.sched_step = 7, 7
    out = tmp_path / "ck"
    save_checkpoint(model, {"hidden_size": 256}, out)
    save_trainer_state(tr, out)
    _, model2 = _model(3)
    tr2 = EagleTrainer(model2, lm_head, mode="tp", num_warmup_steps=2, num_training_steps=10)
    v0 = tr2._pversion
    load_checkpoint(model2, out)
    load_trainer_state(tr2, out)
    assert torch.equal(tr2.flat, tr.flat) and torch.equal(tr2.m, tr.m) and torch.equal(tr2.v, tr.v)
    assert (tr2.step_count, tr2.sched_step) == (7, 7) and tr2._pversion > v0
    # the module's parameters are still views of the flat buffer after load_state_dict
    assert model2.fc.weight.data_ptr() == tr2.flat[tr2.offs["fc.weight"][0]:].data_ptr() if "fc.weight" in tr2.offs else True
