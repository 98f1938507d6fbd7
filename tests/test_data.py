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


def _toy_tokenizer():
    """A fast tokenizer built in memory (no files): word-level vocabulary, a chat template with {% generation %} marks."""
    import transformers
    from tokenizers import Tokenizer, models, pre_tokenizers
    vocab = {"[PAD]": 0, "<unk>": 1, "<|user|>": 2, "<|assistant|>": 3, "<|end|>": 4}
    for w in "hello world how are you fine thanks what is the answer forty two".split():
        vocab[w] = len(vocab)
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    fast = transformers.PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>")
    fast.chat_template = ("{% for m in messages %}{% if m['role'] == 'user' %}<|user|> {{ m['content'] }} <|end|> {% else %}"
                          "<|assistant|> {% generation %}{{ m['content'] }} <|end|> {% endgeneration %}{% endif %}{% endfor %}")
    return fast


def test_chat_collator_is_the_reference_call():
    """eagle/train.py:242-253: pad token "[PAD]" / id 0, apply_chat_template(..., return_assistant_tokens_mask=True, padding=True),
    keys input_ids / loss_mask.  Checked against the same transformers call made directly, on an in-memory tokenizer."""
    from eagle_amd.data import ChatCollator
    feats = [{"messages": [{"role": "user", "content": "hello how are you"}, {"role": "assistant", "content": "fine thanks"}]},
             {"messages": [{"role": "user", "content": "what is the answer"}, {"role": "assistant", "content": "forty two"},
                           {"role": "user", "content": "hello"}, {"role": "assistant", "content": "world"}]}]
    col = ChatCollator(_toy_tokenizer())
    out = col(feats)
    ref_tok = _toy_tokenizer()
    ref_tok.pad_token = "[PAD]"
    ref_tok.pad_token_id = 0
    ref = ref_tok.apply_chat_template([f["messages"] for f in feats], tokenize=True, add_generation_prompt=False, return_dict=True,
                                      return_assistant_tokens_mask=True, return_tensors="pt", padding=True)
    assert set(out) == {"input_ids", "loss_mask"} and out["input_ids"].dtype == torch.long and out["loss_mask"].dtype == torch.long
    assert torch.equal(out["input_ids"], ref["input_ids"]) and torch.equal(out["loss_mask"], torch.as_tensor(ref["assistant_masks"]))
    assert out["input_ids"].shape == (2, 16) and out["input_ids"][0, 10:].tolist() == [0] * 6      # right-padded with id 0
    assert out["input_ids"][0, :10].tolist() == [2, 5, 7, 8, 9, 4, 3, 10, 11, 4]                 # <|user|> hello how are you <|end|> <|assistant|> fine thanks <|end|>
    assert out["loss_mask"][:, :7].sum() == 0 and out["loss_mask"][0, 7:10].tolist() == [1, 1, 1]  # the first assistant turn, not its header
    with pytest.raises(ValueError):
        col([])
    with pytest.raises(TypeError):
        ChatCollator(object())


@pytest.mark.gpu
def test_make_eagle_input_runs_the_verifier_forward_like_the_reference():
    """eagle/train.py:256-269 end to end: a (tiny, randomly initialised, in-memory) transformers Llama as the verifier, the
    reference's statements written out on the CPU side, this library's kernels on the device side."""
    import transformers
    from eagle_amd.data import _make_eagle_input
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = transformers.LlamaConfig(vocab_size=128, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                                   num_key_value_heads=4, max_position_embeddings=64)
    verifier = transformers.LlamaForCausalLM(cfg).to(torch.bfloat16).to(dev).eval()
    g = torch.Generator().manual_seed(5)
    batch = Collator()([{"input_ids": torch.randint(1, 128, (n,), generator=g).tolist(), "loss_mask": torch.randint(0, 2, (n,), generator=g).tolist()}
                        for n in (13, 20, 7)])
    max_len = 16
    got = _make_eagle_input(batch, verifier, max_len, 0.0, 0.0, dev)
    # the reference's statements (train.py:257-268) on the same verifier, noise off
    ids = batch["input_ids"].to(dev)[:, :max_len]
    with torch.no_grad():
        h = verifier(ids, output_hidden_states=True, use_cache=False).hidden_states[-1]
    target = h.new_zeros(h.shape)
    target[:, :-1, :] = h[:, 1:, :]
    shifted = torch.cat((ids[:, 1:], torch.zeros(ids.size(0), 1, dtype=ids.dtype, device=dev)), dim=1)
    assert torch.equal(got["hidden_states"], h) and torch.equal(got["target"], target)
    assert torch.equal(got["input_ids"], shifted) and torch.equal(got["loss_mask"].cpu(), batch["loss_mask"][:, :max_len])
    noisy = _make_eagle_input(batch, verifier, max_len, -0.1, 0.1, dev, generator=torch.Generator(device=dev).manual_seed(2))
    d = (noisy["hidden_states"].float() - h.float()).abs()
    assert 0.01 < float(d.max()) <= 0.1 + float(h.float().abs().max()) * 2 ** -7
    assert torch.equal(noisy["target"][:, :-1], noisy["hidden_states"][:, 1:]) and not noisy["target"][:, -1].any()
