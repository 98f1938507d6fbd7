"""Host side of the training input path on either side of the hot loop.

``Collator``            eagle/train.py:242-253 without the tokenizer: the reference tokenises chat messages with the
                        verifier's chat template (needs tokenizer files; out of scope) and pads the batch with id 0
                        (`pad_token_id = 0`, `padding=True`); here the features are already tokenised
                        ({"input_ids", "loss_mask"} per conversation, e.g. from `apply_chat_template(...,
                        return_assistant_tokens_mask=True)` run offline) and only the padding semantics remain.
``ChatCollator``        eagle/train.py:242-253 with a tokenizer object: chat template + assistant mask + padding, the reference's call.
``make_eagle_batch``    eagle/train.py:256-269 (`_make_eagle_input`) given the verifier's last hidden state: truncation
                        to max_model_len, uniform noise (train.py:272-275) drawn on the device, and the shift kernel
                        (eagle_make_eagle_input_bf16).  Returns the reference's dict keys; the result feeds
                        EagleTrainer.step / Llama2Model.forward unchanged.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import ops


class Collator:
    """Pads already-tokenised conversations to the longest of the batch: ids with ``pad_token_id`` (0 as in the
    reference), the assistant mask with 0.  ``padding_side`` follows the tokenizer in the reference ("right" for the
    Qwen and the fast Llama tokenizers it is used with)."""

    def __init__(self, pad_token_id: int = 0, padding_side: str = "right") -> None:
        if padding_side not in ("right", "left"):
            raise ValueError("padding_side must be 'right' or 'left'")
        self.pad_token_id, self.padding_side = pad_token_id, padding_side

    def __call__(self, features: Sequence[Dict[str, Sequence[int]]]) -> Dict[str, torch.Tensor]:
        if not features:
            raise ValueError("empty batch")
        L = max(len(f["input_ids"]) for f in features)
        ids = torch.full((len(features), L), self.pad_token_id, dtype=torch.long)
        mask = torch.zeros((len(features), L), dtype=torch.long)
        for i, f in enumerate(features):
            n = len(f["input_ids"])
            if len(f["loss_mask"]) != n:
                raise ValueError(f"feature {i}: input_ids and loss_mask differ in length")
            sl = slice(0, n) if self.padding_side == "right" else slice(L - n, L)
            ids[i, sl] = torch.as_tensor(f["input_ids"], dtype=torch.long)
            mask[i, sl] = torch.as_tensor(f["loss_mask"], dtype=torch.long)
        return {"input_ids": ids, "loss_mask": mask}


class ChatCollator:
    """eagle/train.py:242-253 WITH the tokenizer: conversations ({"messages": [...]} features) go through the verifier's chat
    template with the assistant-token mask as the loss mask, padded to the longest of the batch with id 0.

    The tokenizer is the caller's: ``ChatCollator.from_pretrained(model_path)`` loads it as the reference does
    (``AutoTokenizer.from_pretrained(path, use_fast=True)``; needs the tokenizer files, which are not in this container), and the
    constructor takes any object with ``apply_chat_template`` (a ``PreTrainedTokenizerFast`` built in memory in the tests).  As in
    the reference ``pad_token`` becomes "[PAD]" with id 0.  The chat template must mark assistant turns with
    ``{% generation %}`` for the mask to be non-trivial (transformers' rule, not this class's).  Output keys and dtypes are the
    reference's: ``input_ids``, ``loss_mask`` int64 [B, L] on the host; ``make_eagle_batch`` / ``make_eagle_draft_input`` take it from there."""

    def __init__(self, tokenizer) -> None:
        if not hasattr(tokenizer, "apply_chat_template"):
            raise TypeError("ChatCollator needs a tokenizer with apply_chat_template (transformers PreTrainedTokenizerFast)")
        self._tokenizer = tokenizer
        self._tokenizer.pad_token = "[PAD]"                       # train.py:245-246
        self._tokenizer.pad_token_id = 0

    @classmethod
    def from_pretrained(cls, model_path) -> "ChatCollator":
        import transformers
        return cls(transformers.AutoTokenizer.from_pretrained(str(model_path), use_fast=True))   # train.py:244

    def __call__(self, features: Sequence[Dict]) -> Dict[str, torch.Tensor]:
        if not features:
            raise ValueError("empty batch")
        result = self._tokenizer.apply_chat_template([m["messages"] for m in features], tokenize=True, add_generation_prompt=False,
                                                     return_dict=True, return_assistant_tokens_mask=True, return_tensors="pt",
                                                     padding=True)                                # train.py:249
        ids = torch.as_tensor(result["input_ids"], dtype=torch.long)
        mask = torch.as_tensor(result["assistant_masks"], dtype=torch.long)
        if ids.shape != mask.shape:
            raise ValueError("the tokenizer returned an assistant mask of another shape than input_ids")
        return {"input_ids": ids, "loss_mask": mask}


def make_eagle_batch(batch: Dict[str, torch.Tensor], verifier_hidden: torch.Tensor, max_model_len: int,
                     transform_uniform_low: float = 0.0, transform_uniform_high: float = 0.0,
                     generator: Optional[torch.Generator] = None, noise: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """batch = Collator output; verifier_hidden [B, L, D] bf16 on the GPU = the verifier's last hidden state over
    batch["input_ids"][:, :max_model_len] (the verifier forward is the caller's).  hidden_states = verifier_hidden +
    U[low, high) noise; target[:, :-1] = hidden_states[:, 1:], last row zero (so the target is noisy too);
    input_ids shifted left with a trailing 0; loss_mask truncated, not shifted.  ``noise`` (already drawn) overrides
    the device draw, for reproducible tests.  The draw here is torch's (``torch.rand`` on the device under the caller's
    ``generator``, as the reference's torch.rand_like); ``make_eagle_draft_input`` below draws Philox noise in-kernel."""
    dev = verifier_hidden.device
    ids = batch["input_ids"][:, :max_model_len].to(dev)
    lm = batch["loss_mask"][:, :max_model_len].to(dev)
    h = verifier_hidden[:, :max_model_len]
    if h.shape[:2] != ids.shape:
        raise ValueError(f"verifier_hidden {tuple(h.shape[:2])} does not match input_ids {tuple(ids.shape)}")
    if noise is None and transform_uniform_high != transform_uniform_low:
        noise = torch.rand(h.shape, generator=generator, device=dev, dtype=torch.float32)
        noise = (noise * (transform_uniform_high - transform_uniform_low) + transform_uniform_low).to(h.dtype)
    elif noise is not None:
        noise = noise[:, :max_model_len].to(device=dev, dtype=h.dtype)
    ho, tg, io = ops.make_eagle_input(h, ids, noise)
    return {"input_ids": io, "hidden_states": ho, "target": tg, "loss_mask": lm}


_noise_calls = 0     # Philox stream of the next make_eagle_draft_input call that is given no ``offset``


def noise_state() -> int:
    """The process-wide noise stream counter (save it with the trainer state; ``set_noise_state`` restores it on resume)."""
    return _noise_calls


def set_noise_state(n: int) -> None:
    global _noise_calls
    _noise_calls = int(n)


def make_eagle_draft_input(batch: Dict[str, torch.Tensor], verifier_hidden: torch.Tensor, embed_weight: torch.Tensor,
                           max_model_len: int, transform_uniform_low: float = 0.0, transform_uniform_high: float = 0.0,
                           seed: int = 0, offset: Optional[int] = None, want_noise: bool = False) -> Dict[str, torch.Tensor]:
    """``make_eagle_batch`` fused with the draft head's embedding gather + concat: one kernel reads the verifier's hidden
    state once, draws the uniform noise on the fly (Philox4x32-10 keyed by ``seed``, stream ``offset``), shifts, and
    writes the fc input ``x0`` [B*L, 2D] and the (noisy) ``target`` directly -- no noise tensor, no hidden_states copy,
    no separate embed_concat pass (eagle/train.py:256-275 + eagle/llama2.py:515, :537-542).  The result feeds
    ``EagleTrainer.step`` (which takes ``x0`` in place of hidden_states / input_ids for the forward).

    The reference draws FRESH noise per call (torch.rand_like, eagle/train.py:272-275).  So does this: with ``offset``
    left at None every call takes the next stream of a process-wide counter (``noise_state`` / ``set_noise_state`` for
    checkpoint / resume; use a per-rank ``seed`` under data parallelism); an explicit ``offset`` reproduces one draw
    (tests).  The stream used is returned as ``noise_offset``."""
    global _noise_calls
    if offset is None:
        offset = _noise_calls
        _noise_calls += 1
    dev = verifier_hidden.device
    ids = batch["input_ids"].to(dev)
    L = min(max_model_len, ids.shape[1], verifier_hidden.shape[1])
    lm = batch["loss_mask"][:, :L].to(dev)
    if verifier_hidden.shape[0] != ids.shape[0] or verifier_hidden.shape[1] < L:
        raise ValueError(f"verifier_hidden {tuple(verifier_hidden.shape[:2])} does not cover input_ids {tuple(ids.shape)}")
    x0, tg, io, noise = ops.eagle_input_embed_concat(verifier_hidden[:, :L], ids, embed_weight, transform_uniform_low,
                                                     transform_uniform_high, seed, offset, want_noise)
    out = {"input_ids": io, "x0": x0, "target": tg, "loss_mask": lm, "hidden_states": x0.view(ids.shape[0], L, -1)[:, :, x0.shape[1] // 2:]}
    out["noise_offset"] = int(offset)
    if want_noise:
        out["noise"] = noise
    return out


def _make_eagle_input(batch: Dict[str, torch.Tensor], verifier_model, max_model_len: int, transform_uniform_low: float,
                      transformer_uniform_high: float, device, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
    """eagle/train.py:256-269 with its own signature, verifier forward included: truncate to ``max_model_len``, run
    ``verifier_model(input_ids, output_hidden_states=True, use_cache=False)`` under no_grad, take the last hidden state, add the
    uniform noise, shift.  ``verifier_model`` is the caller's (any module whose output has ``.hidden_states``, e.g. a transformers
    causal LM); everything after its forward runs in this library's kernels (``make_eagle_batch``).  Returns the reference's keys."""
    ids = batch["input_ids"].to(device)[:, :max_model_len]
    with torch.no_grad():
        hidden = verifier_model(ids, output_hidden_states=True, use_cache=False).hidden_states[-1]
    if hidden.dtype != torch.bfloat16:
        hidden = hidden.to(torch.bfloat16)
    return make_eagle_batch(batch, hidden.contiguous(), max_model_len, transform_uniform_low, transformer_uniform_high, generator=generator)

