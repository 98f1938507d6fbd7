"""Collator padding semantics (eagle/train.py:242-253 without the tokenizer) and, on the GPU, make_eagle_batch
(eagle/train.py:256-269) against the oracle's restatement, including truncation to max_model_len."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd.data import Collator  # noqa: E402
from oracle import loss as ol  # noqa: E402


def test_collator_pads_right_with_id_zero_and_mask_zero():
    feats = [{"input_ids": [5, 6, 7], "loss_mask": [0, 1, 1]}, {"input_ids": [9], "loss_mask": [1]},
             {"input_ids": [1, 2, 3, 4, 5], "loss_mask": [0, 0, 1, 1, 1]}]
    out = Collator()(feats)
    assert out["input_ids"].tolist() == [[5, 6, 7, 0, 0], [9, 0, 0, 0, 0], [1, 2, 3, 4, 5]]
    assert out["loss_mask"].tolist() == [[0, 1, 1, 0, 0], [1, 0, 0, 0, 0], [0, 0, 1, 1, 1]]
    assert out["input_ids"].dtype == torch.long and out["loss_mask"].dtype == torch.long
    left = Collator(padding_side="left")(feats)
    assert left["input_ids"][1].tolist() == [0, 0, 0, 0, 9] and left["loss_mask"][0].tolist() == [0, 0, 0, 1, 1]


def test_collator_rejects_bad_features():
    with pytest.raises(ValueError):
        Collator()([])
    with pytest.raises(ValueError):
        Collator()([{"input_ids": [1, 2], "loss_mask": [1]}])
    with pytest.raises(ValueError):
        Collator(padding_side="middle")


@pytest.mark.gpu
@pytest.mark.parametrize("max_len", [16, 7])
def test_make_eagle_batch_matches_oracle(max_len):
    from eagle_amd.data import make_eagle_batch
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    feats = [{"input_ids": torch.randint(1, 900, (n,), generator=g).tolist(), "loss_mask": torch.randint(0, 2, (n,), generator=g).tolist()}
             for n in (11, 4, 9)]
    batch = Collator()(feats)
    B, L = batch["input_ids"].shape
    D = 64
    Lt = min(L, max_len)
    h = torch.randn(B, Lt, D, generator=g).to(torch.bfloat16)
    nz = ((torch.rand(B, Lt, D, generator=g) - 0.5) * 0.2).to(torch.bfloat16)
    ref = ol.make_eagle_input(batch["input_ids"], batch["loss_mask"], h, max_len, noise=nz)
    got = make_eagle_batch(batch, h.to(dev), max_len, noise=nz)
    for k in ("input_ids", "hidden_states", "target", "loss_mask"):
        assert torch.equal(got[k].cpu(), ref[k]), k
    # device-drawn noise: inside [low, high), target = shifted noisy hidden, last row zero
    got = make_eagle_batch(batch, h.to(dev), max_len, -0.1, 0.1, generator=torch.Generator(device=dev).manual_seed(1))
    d = (got["hidden_states"].float().cpu() - h.float())
    assert float(d.abs().max()) <= 0.1 + 2 ** -6 and float(d.abs().max()) > 0.01
    assert torch.equal(got["target"][:, :-1], got["hidden_states"][:, 1:]) and not got["target"][:, -1].any()
