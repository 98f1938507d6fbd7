"""Generate the golden fixtures in this directory by running the REFERENCE on CPU.

Run once in the build container (the reference is mounted read-only at /root/reference and is
not available on the GPU box):

    python tests/golden/make_golden.py

It imports ``eagle.llama2.Llama2Model`` from /root/reference (with the one patch SURVEY.md 0.4
documents: ``config.rope_scaling = None`` because transformers 5.x fills the field and
eagle/llama2.py:174-178 then raises KeyError), feeds it seeded inputs and stores inputs and
outputs as safetensors.  Nothing from the reference's source text is stored: only tensors.

Fixtures
--------
g1_tiny_{fp32,bf16}.safetensors   D=64 I=128 H=4 V=100: weights, inputs, eval forward, cached
                                  forward, q=1 cached step, tree-masked step with explicit
                                  position_ids (B=1), padded batch (B=2) forward.
g2_real_{fp32,bf16}.safetensors   resources/eagle_config.json dims (D=4096 I=11008 H=32 V=32000),
                                  B=1 L=16; weights are NOT stored: they come from the seed recipe
                                  ``oracle.draft_model.init_weights(dims, seed=1234)``.
g3_loss_{fp32,bf16}.safetensors   loss block, produced by EXECUTING THE REFERENCE'S OWN STATEMENTS: eagle/train.py and
                                  eagle/train_tp.py cannot be imported (top-level ``import clearml``, not installed)
                                  and the loss lines sit inline in ``coach()``, so the statements that follow
                                  ``predict = model(...)`` up to ``loss = arguments.v_w * vloss + ...``
                                  (train.py:106-123, train_tp.py:88-105; the line ranges are asserted) are taken
                                  out of the parsed file with ``ast``, compiled and run in a namespace holding
                                  ``torch``, ``arguments``, ``lm_head`` (nn.Linear, bias=False), ``criterion``
                                  (SmoothL1Loss(reduction="none"), train.py:65), ``batch`` and ``predict``; autograd
                                  gives dlogits / dpredict.  Both scripts' blocks are run and must agree bit for bit.
                                  Cases: s (N=2x4, V=1000), m (1x6, V=4099), q (1x4, V=151936; W from a seed recipe,
                                  not stored); the bf16 file also holds x = train.py under accelerate bf16 mixed
                                  precision (predict fp32, lm_head / target bf16).
"""
import ast
import os
import sys
import types

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import transformers  # noqa: E402
from eagle.llama2 import Llama2Model  # noqa: E402  (the reference)

from oracle.draft_model import DraftDims, init_weights  # noqa: E402
from oracle.tree import generate_tree_buffers  # noqa: E402


def ref_model(dims: DraftDims, weights, dtype):
    cfg = transformers.AutoConfig.from_pretrained("/root/reference/resources/eagle_config.json")
    cfg.rope_scaling = None
    cfg.hidden_size, cfg.intermediate_size = dims.hidden, dims.inter
    cfg.num_attention_heads, cfg.num_key_value_heads = dims.heads, dims.kv_heads
    cfg.vocab_size = dims.vocab
    m = Llama2Model(cfg)
    m.load_state_dict(weights, strict=True)
    return m.to(dtype).eval()


def c(t):
    return t.detach().contiguous().clone()


@torch.no_grad()
def g1(dtype, name):
    dims = DraftDims(hidden=64, inter=128, heads=4, kv_heads=4, vocab=100)
    w = init_weights(dims, seed=0, std=0.2)
    m = ref_model(dims, w, dtype)
    g = torch.Generator().manual_seed(1)
    out = {f"w.{k}": c(v) for k, v in w.items()}
    # (a) plain eval forward, B=1, L=7
    h = torch.randn(1, 7, 64, generator=g).to(dtype)
    ids = torch.randint(0, 100, (1, 7), generator=g)
    out["a.hidden"], out["a.ids"] = c(h), c(ids)
    out["a.out"] = c(m(h, ids))
    # (b) use_cache forward then a 3-row continuation and a q=1 step
    o, kv = m(h, ids, use_cache=True)
    out["b.out"], out["b.k"], out["b.v"] = c(o), c(kv[0][0]), c(kv[0][1])
    h2 = torch.randn(1, 3, 64, generator=g).to(dtype)
    ids2 = torch.randint(0, 100, (1, 3), generator=g)
    o2, kv2 = m(h2, ids2, past_key_values=kv, use_cache=True)
    out["c.hidden"], out["c.ids"] = c(h2), c(ids2)
    out["c.out"], out["c.k"], out["c.v"] = c(o2), c(kv2[0][0]), c(kv2[0][1])
    h3 = torch.randn(1, 1, 64, generator=g).to(dtype)
    ids3 = torch.randint(0, 100, (1, 1), generator=g)
    o3, kv3 = m(h3, ids3, past_key_values=kv2, use_cache=True)
    out["d.hidden"], out["d.ids"] = c(h3), c(ids3)
    out["d.out"], out["d.k"] = c(o3), c(kv3[0][0])
    # (e) two tree-masked steps chained after (b)'s cache, with the 25-path tree's draft buffers
    tb = generate_tree_buffers()
    past, L = kv, 7
    for d in range(2):
        n = len(tb.step_nodes[d])
        hd_ = torch.randn(1, n, 64, generator=g).to(dtype)
        idd = torch.randint(0, 100, (1, n), generator=g)
        pos = L + d + tb.step_position_ids[d]
        m.tree_mask = tb.step_attn_mask[d]
        od, past = m(hd_, idd, past_key_values=past, position_ids=pos, use_cache=True)
        out[f"e{d}.hidden"], out[f"e{d}.ids"], out[f"e{d}.pos"] = c(hd_), c(idd), c(pos)
        out[f"e{d}.out"], out[f"e{d}.k"], out[f"e{d}.v"] = c(od), c(past[0][0]), c(past[0][1])
    m.reset()
    # (f) right-padded batch of two with an explicit attention_mask
    hb = torch.randn(2, 5, 64, generator=g).to(dtype)
    idb = torch.randint(0, 100, (2, 5), generator=g)
    am = torch.tensor([[1, 1, 1, 1, 1], [1, 1, 1, 0, 0]], dtype=torch.bool)
    out["f.hidden"], out["f.ids"], out["f.mask"] = c(hb), c(idb), c(am.to(torch.uint8))
    out["f.out"] = c(m(hb, idb, attention_mask=am))
    save_file(out, os.path.join(HERE, name))


@torch.no_grad()
def g2(dtype, name):
    dims = DraftDims.from_json("/root/reference/resources/eagle_config.json")
    w = init_weights(dims, seed=1234)
    m = ref_model(dims, w, dtype)
    g = torch.Generator().manual_seed(2)
    h = torch.randn(1, 16, dims.hidden, generator=g).to(dtype)
    ids = torch.randint(0, dims.vocab, (1, 16), generator=g)
    o, kv = m(h, ids, use_cache=True)
    out = {"hidden": c(h), "ids": c(ids), "out": c(o), "k": c(kv[0][0]), "v": c(kv[0][1])}
    h2 = torch.randn(1, 4, dims.hidden, generator=g).to(dtype)
    ids2 = torch.randint(0, dims.vocab, (1, 4), generator=g)
    tb = generate_tree_buffers()
    m.tree_mask = tb.step_attn_mask[0]
    pos = 16 + tb.step_position_ids[0]
    o2, _ = m(h2, ids2, past_key_values=kv, position_ids=pos, use_cache=True)
    out.update({"t.hidden": c(h2), "t.ids": c(ids2), "t.pos": c(pos), "t.out": c(o2)})
    save_file(out, os.path.join(HERE, name))


def ref_loss_block(path, first, last):
    """The reference's loss statements as a compiled code object: everything after ``predict = model(...)`` up to and
    including ``loss = arguments.v_w * vloss + arguments.p_w * ploss`` inside ``coach()``."""
    tree = ast.parse(open(path).read(), filename=path)

    def is_assign_to(st, name):
        return isinstance(st, ast.Assign) and len(st.targets) == 1 and isinstance(st.targets[0], ast.Name) \
            and st.targets[0].id == name

    for node in ast.walk(tree):
        body = getattr(node, "body", None)
        if not isinstance(body, list):
            continue
        for i, st in enumerate(body):
            if is_assign_to(st, "predict"):
                block = []
                for st2 in body[i + 1:]:
                    block.append(st2)
                    if is_assign_to(st2, "loss"):
                        break
                assert (block[0].lineno, block[-1].end_lineno) == (first, last), (path, block[0].lineno, block[-1].end_lineno)
                mod = ast.Module(body=block, type_ignores=[])
                return compile(mod, path, "exec")
    raise RuntimeError(f"loss block not found in {path}")


REF_BLOCKS = None


def run_ref_loss(predict, target, W, mask, v_w=1.0, p_w=0.1):
    """Execute the reference's loss lines (both scripts) on the given tensors; returns the namespace of train.py's run."""
    global REF_BLOCKS
    if REF_BLOCKS is None:
        REF_BLOCKS = [ref_loss_block("/root/reference/eagle/train.py", 106, 123),
                      ref_loss_block("/root/reference/eagle/train_tp.py", 88, 105)]
    runs = []
    for code in REF_BLOCKS:
        p = predict.detach().clone().requires_grad_(True)
        lm_head = torch.nn.Linear(W.shape[1], W.shape[0], bias=False, dtype=W.dtype)   # train.py:215-228
        lm_head.weight.data = W.clone()
        lm_head.weight.requires_grad_(False)
        ns = {
            "torch": torch,
            "arguments": types.SimpleNamespace(verifier_model_lm_head_dtype=str(W.dtype).split(".")[1], v_w=v_w, p_w=p_w),
            "lm_head": lm_head,
            "criterion": torch.nn.SmoothL1Loss(reduction="none"),                        # train.py:65
            "batch": {"target": target, "loss_mask": mask},
            "predict": p,
            "step_correctly_predicted_tokens_count": 0,
        }
        exec(code, ns)
        ns["out_head"].retain_grad()
        ns["loss"].backward()
        runs.append(ns)
    a, b = runs
    for k in ("target_head", "out_head", "ploss", "vloss", "loss"):
        assert torch.equal(a[k], b[k]), k
    assert a["step_correctly_predicted_tokens_count"] == b["step_correctly_predicted_tokens_count"]
    assert torch.equal(a["out_head"].grad, b["out_head"].grad) and torch.equal(a["predict"].grad, b["predict"].grad)
    return a


def g3_big_weight(V, D, dtype):
    """Seed recipe of the q case's lm_head (151936 x 32: not stored in the fixture)."""
    return (torch.randn(V, D, generator=torch.Generator().manual_seed(33)) * 0.3).to(dtype)


def g3(dtype, name):
    g = torch.Generator().manual_seed(3)
    out = {}
    cases = {"s": (2, 4, 32, 1000, dtype), "m": (1, 6, 64, 4099, dtype), "q": (1, 4, 32, 151936, dtype)}
    if dtype == torch.bfloat16:
        cases["x"] = (2, 4, 32, 1000, torch.float32)        # accelerate mixed precision: predict arrives as fp32
    for tag, (B, L, D, V, pdtype) in cases.items():
        W = g3_big_weight(V, D, dtype) if tag == "q" else (torch.randn(V, D, generator=g) * 0.3).to(dtype)
        predict = torch.randn(B, L, D, generator=g).to(dtype).to(pdtype)
        target = (predict.detach().float() + 0.7 * torch.randn(B, L, D, generator=g)).to(dtype)
        mask = (torch.rand(B, L, generator=g) < 0.7).long()
        if tag == "q":
            mask[0, 0] = 1
        r = run_ref_loss(predict, target, W, mask, v_w=1.0, p_w=0.1)
        if tag != "q":
            out[f"{tag}.W"] = c(W)
        out.update({
            f"{tag}.predict": c(predict), f"{tag}.target": c(target), f"{tag}.mask": c(mask),
            f"{tag}.t_logits": c(r["target_head"]), f"{tag}.o_logits": c(r["out_head"]),
            f"{tag}.ploss": c(r["ploss"].float().reshape(1)), f"{tag}.vloss": c(r["vloss"].float().reshape(1)),
            f"{tag}.loss": c(r["loss"].float().reshape(1)),
            f"{tag}.correct": torch.tensor([int(r["step_correctly_predicted_tokens_count"])], dtype=torch.int64),
            f"{tag}.dlogits": c(r["out_head"].grad), f"{tag}.dpredict": c(r["predict"].grad),
        })
    save_file(out, os.path.join(HERE, name))


if __name__ == "__main__":
    torch.set_num_threads(8)
    g1(torch.float32, "g1_tiny_fp32.safetensors")
    g1(torch.bfloat16, "g1_tiny_bf16.safetensors")
    g3(torch.float32, "g3_loss_fp32.safetensors")
    g3(torch.bfloat16, "g3_loss_bf16.safetensors")
    g2(torch.float32, "g2_real_fp32.safetensors")
    g2(torch.bfloat16, "g2_real_bf16.safetensors")
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".safetensors"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
