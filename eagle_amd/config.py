"""Draft-head configuration: the fields of the HF ``LlamaConfig`` JSON that eagle/llama2.py reads
(resources/eagle_config.json, resources/eagle_config_qwen3_8b.json), with LlamaConfig's defaults
for the ones the shipped files omit (vocab_size 32000, max_position_embeddings 2048,
pad_token_id None, pretraining_tp 1, rope_scaling None).  A ``transformers`` config object works
just as well: ``Llama2Model`` only reads attributes."""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import Optional


@dataclass
class DraftConfig:
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_attention_heads: int = 32
    num_key_value_heads: Optional[int] = None
    num_hidden_layers: int = 1
    vocab_size: int = 32000
    max_position_embeddings: int = 2048
    rms_norm_eps: float = 1e-6
    hidden_act: str = "silu"
    pad_token_id: Optional[int] = None
    pretraining_tp: int = 1
    rope_scaling: Optional[dict] = None
    rope_theta: float = 10000.0
    raw: dict = field(default_factory=dict, repr=False)

    def __post_init__(self):
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads

    @classmethod
    def from_json(cls, path: str) -> "DraftConfig":
        with open(path) as f:
            raw = json.load(f)
        known = {k: raw[k] for k in cls.__dataclass_fields__ if k in raw and k != "raw"}
        return cls(raw=raw, **known)


LLAMA2_7B_DRAFT = dict(hidden_size=4096, intermediate_size=11008, num_attention_heads=32, vocab_size=32000)
QWEN3_8B_DRAFT = dict(hidden_size=4096, intermediate_size=12288, num_attention_heads=32, vocab_size=151936)
