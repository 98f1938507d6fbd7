"""Checkpoint layout of the reference (SURVEY.md §8b "Checkpoint layout", §8f-2).

A draft-head checkpoint is a directory with
  * `model.safetensors` = `Llama2Model.state_dict()` exactly as the reference writes it
    (eagle/train_tp.py:141: no `model.` prefix, frozen `embed_tokens.weight` included, no `lm_head`);
  * `config.json` = the draft config file with `architectures = ["LlamaForCausalLMEagle"]`
    (eagle/train_tp.py:142-146, eagle/train.py:164-168), which is what SGLang's EAGLE worker loads.

`save_trainer_state` / `load_trainer_state` (resume) have no counterpart in the reference
(`accelerator.save_state`, train.py:163, is never read back): they are an extension, one extra file
next to the two above, so the directory stays loadable by SGLang.
"""
from __future__ import annotations

import json
import os
from typing import Union

import torch

EAGLE_ARCHITECTURES = ["LlamaForCausalLMEagle"]


def _config_dict(config) -> dict:
    if isinstance(config, (str, os.PathLike)):
        with open(config, "r") as f:
            return json.load(f)
    if isinstance(config, dict):
        return dict(config)
    return config.to_dict()                      # transformers PretrainedConfig


def save_checkpoint(model, config: Union[str, os.PathLike, dict, object], out_dir: Union[str, os.PathLike]) -> None:
    """eagle/train_tp.py:140-146.  `config` is the draft config path (as `--eagle-config-path`), a dict,
    or a transformers config object."""
    from safetensors.torch import save_file

    os.makedirs(out_dir, exist_ok=False)         # the reference's mkdir(parents=True) raises on an existing directory too
    save_file({k: v.detach().contiguous().cpu() for k, v in model.state_dict().items()},
              os.path.join(out_dir, "model.safetensors"))
    data = _config_dict(config)
    data["architectures"] = list(EAGLE_ARCHITECTURES)
    with open(os.path.join(out_dir, "config.json"), "w") as f:
        json.dump(data, f, ensure_ascii=False, indent=4)


def load_checkpoint(model, ckpt_dir: Union[str, os.PathLike], strict: bool = True):
    """Load `model.safetensors` of a reference-layout checkpoint into `model` (keys as Llama2Model.state_dict())."""
    from safetensors.torch import load_file

    sd = load_file(os.path.join(ckpt_dir, "model.safetensors"), device="cpu")
    return model.load_state_dict(sd, strict=strict)


def load_verifier_lm_head(verifier_path: Union[str, os.PathLike], dtype=torch.bfloat16) -> torch.Tensor:
    """The frozen lm_head weight `[V, D]` of the verifier checkpoint (eagle/train.py:215-228: the shard
    named by `model.safetensors.index.json`, columns cut to `hidden_size`)."""
    from safetensors import safe_open

    with open(os.path.join(verifier_path, "config.json"), "r") as f:
        D = json.load(f)["hidden_size"]
    with open(os.path.join(verifier_path, "model.safetensors.index.json"), "r") as f:
        shard = json.load(f)["weight_map"]["lm_head.weight"]
    with safe_open(os.path.join(verifier_path, shard), framework="pt") as f:
        w = f.get_slice("lm_head.weight")[:, :D]
    return w.to(dtype)


def save_trainer_state(trainer, out_dir: Union[str, os.PathLike]) -> None:
    """Extension (resume): AdamW moments in the trainer's flat layout + the step counters."""
    from safetensors.torch import save_file

    save_file({"exp_avg": trainer.m.detach().cpu(), "exp_avg_sq": trainer.v.detach().cpu()},
              os.path.join(out_dir, "trainer_state.safetensors"),
              metadata={"step_count": str(trainer.step_count), "sched_step": str(trainer.sched_step),
                        "layout": json.dumps({k: list(v) if isinstance(v, (tuple, list)) else v for k, v in trainer.offs.items()})})


def load_trainer_state(trainer, ckpt_dir: Union[str, os.PathLike]) -> None:
    from safetensors import safe_open

    with safe_open(os.path.join(ckpt_dir, "trainer_state.safetensors"), framework="pt", device="cpu") as f:
        meta = f.metadata()
        m, v = f.get_tensor("exp_avg"), f.get_tensor("exp_avg_sq")
    if m.numel() != trainer.m.numel():
        raise ValueError(f"trainer state holds {m.numel()} elements, this trainer {trainer.m.numel()}")
    trainer.m.copy_(m)
    trainer.v.copy_(v)
    trainer.step_count, trainer.sched_step = int(meta["step_count"]), int(meta["sched_step"])
    trainer._pversion += 1          # the weights were (re)loaded too: drop the cached transposed copies
    trainer.model.invalidate_weight_caches()
