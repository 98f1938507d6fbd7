"""Oracle (TEST INFRASTRUCTURE ONLY): static draft tree, expansion loop, greedy acceptance,
KV / hidden compaction.  **parity unpinned**.

None of this exists in /root/reference: ``Llama2Model.init_tree`` (eagle/llama2.py:453-455)
names ``mc_sim_7b_63`` and ``generate_tree_buffers`` without defining them and inference is
delegated to SGLang (benchmark/benchmark.py:48-56).  This module restates the published
upstream EAGLE-1 algorithm those names refer to (SafeAILab/EAGLE v1: choices.py, cnets.py
``topK_genrate`` / ``generate_tree_buffers``, utils.py ``generate_tree_buffers`` /
``evaluate_posterior`` / ``update_inference_inputs``), as frozen in DESIGN.md.  Its pins are
the hand-derived known answers in tests/test_tree_oracle.py.

Conventions frozen here
-----------------------
* Node ids: root = 0; the tree paths sorted by (len, path) are nodes 1..n.
* top-k tie-break: value descending, then index ascending (a stable descending sort).
  ``torch.topk`` is NOT used: on CPU it does not break ties by lowest index.
* ``expand="wchild"`` (upstream behaviour, default): only nodes that have children are run
  through the draft head (a leaf's own top-k is never consumed).  ``expand="all"`` runs every
  node (SURVEY row a20's (4,8,8,3,2) row counts for the 25-path tree).
* flat top-k table: row 0 = top-k after the catch-up forward, then one row per forwarded node
  in node-id order.  A node's draft token = table[row(parent)][choice].
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

TOPK = 10  # eagle/llama2.py:11 (module-level top_k = 10)

# The 25-path static tree (+ root = 26 nodes); depth counts [4, 8, 8, 3, 2].
MC_SIM_7B_63: List[List[int]] = [
    [0], [1], [2], [3],
    [0, 0], [0, 1], [0, 2], [1, 0], [1, 1], [2, 0], [2, 1], [3, 0],
    [0, 0, 0], [0, 0, 1], [0, 0, 2], [0, 1, 0], [0, 1, 1], [0, 2, 0], [0, 2, 1], [1, 0, 0],
    [0, 0, 0, 0], [0, 0, 0, 1], [0, 0, 0, 2],
    [0, 0, 0, 0, 0], [0, 0, 0, 0, 1],
]


@dataclass
class TreeBuffers:
    topk: int
    n_nodes: int                      # including root
    parent: List[int]                 # [n_nodes], parent[0] = -1
    depth: List[int]                  # [n_nodes], root 0
    choice: List[int]                 # [n_nodes], root -1
    # ---- draft side: one entry per expansion step (depth d = 1..n_steps) ----
    step_nodes: List[List[int]] = field(default_factory=list)     # node ids forwarded at step d
    step_tree_indices: List[torch.Tensor] = field(default_factory=list)  # into flat top-k of previous step
    step_parent_row: List[torch.Tensor] = field(default_factory=list)    # row of previous out_hidden to repeat
    step_repeat_nums: List[List[int]] = field(default_factory=list)
    step_position_ids: List[torch.Tensor] = field(default_factory=list)  # zeros; caller adds len_posi + d - 1
    step_attn_mask: List[torch.Tensor] = field(default_factory=list)     # [1,1,n_d,sum n_<=d]
    # ---- verify side ----
    tree_attn_mask: Optional[torch.Tensor] = None     # [1,1,n,n]
    tree_indices: Optional[torch.Tensor] = None       # [n] into cat(root_token, flat_topk)
    tree_position_ids: Optional[torch.Tensor] = None  # [n] depth
    retrieve_indices: Optional[torch.Tensor] = None   # [n_paths, max_depth+1], -1 padded
    flat_row: List[int] = field(default_factory=list)  # flat top-k row produced BY each node (-1: not forwarded)


def generate_tree_buffers(paths: Sequence[Sequence[int]] = MC_SIM_7B_63, topk: int = TOPK,
                          expand: str = "wchild") -> TreeBuffers:
    paths = sorted([list(p) for p in paths], key=lambda p: (len(p), p))
    ids = {(): 0}
    for i, p in enumerate(paths):
        ids[tuple(p)] = i + 1
    n = len(paths) + 1
    parent, depth, choice = [-1] * n, [0] * n, [-1] * n
    children: List[List[int]] = [[] for _ in range(n)]
    for p in paths:
        i = ids[tuple(p)]
        if tuple(p[:-1]) not in ids:
            raise ValueError(f"tree path {p} has no parent in the tree")
        if not 0 <= p[-1] < topk:
            raise ValueError(f"choice {p[-1]} outside top-{topk}")
        parent[i], depth[i], choice[i] = ids[tuple(p[:-1])], len(p), p[-1]
        children[parent[i]].append(i)
    max_depth = max(depth)
    if expand == "wchild":
        fwd = [i for i in range(1, n) if children[i]]
        n_steps = max_depth - 1
    elif expand == "all":
        fwd = list(range(1, n))
        n_steps = max_depth
    else:
        raise ValueError(expand)
    tb = TreeBuffers(topk=topk, n_nodes=n, parent=parent, depth=depth, choice=choice)
    tb.flat_row = [-1] * n
    tb.flat_row[0] = 0
    for r, i in enumerate(fwd):
        tb.flat_row[i] = r + 1

    def ancestors(i):  # node ids from depth 1 down to i
        chain = []
        while i > 0:
            chain.append(i)
            i = parent[i]
        return chain[::-1]

    # draft side
    seen: List[int] = []  # forwarded nodes of depth <= d, in KV order
    prev: List[int] = [0]
    for d in range(1, n_steps + 1):
        cur = [i for i in fwd if depth[i] == d]
        if not cur:
            raise ValueError(f"no node to expand at depth {d}")
        for i in cur:
            if parent[i] not in prev:
                raise ValueError("a forwarded node's parent was not forwarded")
        prow = [prev.index(parent[i]) for i in cur]
        tb.step_nodes.append(cur)
        tb.step_parent_row.append(torch.tensor(prow, dtype=torch.long))
        tb.step_tree_indices.append(torch.tensor([choice[i] + topk * r for i, r in zip(cur, prow)], dtype=torch.long))
        rep: List[int] = []
        for j, r in enumerate(prow):
            if j == 0 or r != prow[j - 1]:
                rep.append(1)
            else:
                rep[-1] += 1
        tb.step_repeat_nums.append(rep)
        tb.step_position_ids.append(torch.zeros(len(cur), dtype=torch.long))
        seen = seen + cur
        m = torch.zeros(len(cur), len(seen))
        for r, i in enumerate(cur):
            for a in ancestors(i):
                m[r, seen.index(a)] = 1
        tb.step_attn_mask.append(m[None, None])
        prev = cur

    # verify side
    tm = torch.eye(n)
    tm[:, 0] = 1
    for i in range(1, n):
        for a in ancestors(i):
            tm[i, a] = 1
    tb.tree_attn_mask = tm[None, None]
    ti = torch.zeros(n, dtype=torch.long)
    for i in range(1, n):
        row = tb.flat_row[parent[i]]
        if row < 0:
            raise ValueError("node whose parent is never forwarded")
        ti[i] = 1 + topk * row + choice[i]
    tb.tree_indices = ti
    tb.tree_position_ids = torch.tensor(depth, dtype=torch.long)
    covered = set()
    rows: List[List[int]] = []
    for i in range(n - 1, 0, -1):  # deepest / last first; every leaf yields one root->leaf path
        if i in covered:
            continue
        chain = ancestors(i)
        covered.update(chain)
        rows.append([0] + chain)
    width = max_depth + 1
    big = n + 5
    rows = [r + [-1] * (width - len(r)) for r in rows]
    rows.sort(key=lambda r: [x if x >= 0 else big for x in r])
    tb.retrieve_indices = torch.tensor(rows, dtype=torch.long)
    return tb


# --------------------------------------------------------------------------------------
def stable_topk(x: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(values, indices) of the k largest per row; ties -> lowest index first.  -0.0 == +0.0;
    NaN sorts as largest (torch.sort semantics)."""
    s = torch.sort(x, dim=-1, descending=True, stable=True)
    return s.values[..., :k], s.indices[..., :k]


def first_argmax(x: torch.Tensor) -> torch.Tensor:
    return stable_topk(x, 1)[1][..., 0]


DraftFn = Callable[..., Tuple[torch.Tensor, tuple]]


def topk_generate(draft_fn: DraftFn, lm_head_w: torch.Tensor, tb: TreeBuffers,
                  hidden_states: torch.Tensor, input_ids: torch.Tensor, stable_kv=None, head_fn=None,
                  forced_ids: Optional[torch.Tensor] = None):
    """The expansion loop (upstream cnets.py ``topK_genrate``, greedy branch).

    hidden_states [1, n_new, D]: target hidden states of the tokens not yet in ``stable_kv``.
    input_ids [1, ctx+1]: full token sequence INCLUDING the freshly sampled root token; the
      first token is dropped (the draft head pairs hidden[i] with token[i+1]).
    draft_fn(hidden, ids, past_key_values=, position_ids=, tree_mask=, use_cache=True)
      -> (out_hidden, kv).
    head_fn(hidden [rows, D]) -> logits [rows, V]; default: the frozen lm_head as a bias-free
      Linear (eagle/train.py:215-228).
    forced_ids [1+n_fwd, k] (tests only): the tokens fed to the next step are taken from this table instead of
      from the loop's own top-k, so that a second implementation whose logits differ by a rounding can be followed
      row by row without the two expansions drifting apart; the returned ids / values / logits are still the
      loop's own.
    Returns (flat_topk_ids [1+n_fwd, k], flat_topk_vals, new_stable_kv, debug dict).
    """
    if head_fn is None:
        def head_fn(h):
            return torch.nn.functional.linear(h, lm_head_w)
    ids = input_ids[:, 1:]
    len_posi = ids.shape[1]
    kv_len = 0 if stable_kv is None else stable_kv[0][0].shape[2]
    out_hidden, kv = draft_fn(hidden_states, ids[:, kv_len:], past_key_values=stable_kv,
                              position_ids=None, tree_mask=None, use_cache=True)
    new_stable = kv
    logits = head_fn(out_hidden[:, -1])  # [1,V]
    all_ids, all_vals, step_inputs, all_logits = [], [], [], []
    row = 0
    for d in range(len(tb.step_nodes)):
        vals, idx = stable_topk(logits, tb.topk)
        all_ids.append(idx)
        all_vals.append(vals)
        all_logits.append(logits)
        feed = idx if forced_ids is None else forced_ids[row: row + idx.shape[0]].to(idx.dtype)
        row += idx.shape[0]
        sel = feed.reshape(-1)[tb.step_tree_indices[d]]
        src = out_hidden[:, -1:] if d == 0 else out_hidden
        hid = src[:, tb.step_parent_row[d]]
        pos = len_posi + tb.step_position_ids[d]
        step_inputs.append({"input_ids": sel.clone(), "hidden": hid.clone(), "position_ids": pos.clone()})
        out_hidden, kv = draft_fn(hid, sel[None, :], past_key_values=kv, position_ids=pos,
                                  tree_mask=tb.step_attn_mask[d], use_cache=True)
        len_posi += 1
        logits = head_fn(out_hidden[0])
    vals, idx = stable_topk(logits, tb.topk)
    all_ids.append(idx)
    all_vals.append(vals)
    all_logits.append(logits)
    return torch.cat(all_ids), torch.cat(all_vals), new_stable, {"steps": step_inputs, "logits": torch.cat(all_logits)}


def tree_candidates(root_token: int, flat_topk_ids: torch.Tensor, tb: TreeBuffers):
    """candidates[n] = token of tree node n; cart[p, j] = token at step j of path p (-1 pad).
    (upstream utils.py ``generate_candidates``, greedy branch)"""
    flat = torch.cat([torch.tensor([root_token], dtype=torch.long), flat_topk_ids.reshape(-1).long()])
    tree_tok = flat[tb.tree_indices]
    ext = torch.cat([tree_tok, torch.tensor([-1], dtype=torch.long)])
    cart = ext[tb.retrieve_indices]
    return tree_tok, cart


def greedy_accept(cart: torch.Tensor, target_argmax: torch.Tensor, tb: TreeBuffers):
    """SURVEY a22.  target_argmax [n_nodes] = first-argmax of the verifier's logits at each tree
    node.  Returns (best_path, accept_length, next_token, per-path accept lengths).

    match[p,j] = cart[p,j+1] == target_argmax[retrieve[p,j]]; accept = prefix-product sum;
    best = first path with the maximum (0 when nothing is accepted)."""
    ri = tb.retrieve_indices
    tgt = target_argmax[ri.clamp(min=0)]  # padded slots are never reached: their candidate is -1
    match = (cart[:, 1:] == tgt[:, :-1]).int()
    acc = torch.cumprod(match, dim=1).sum(dim=1)
    accept_length = int(acc.max())
    best = 0 if accept_length == 0 else int((acc == accept_length).nonzero()[0])  # first max
    next_token = int(target_argmax[ri[best, accept_length]])
    return best, accept_length, next_token, acc


def compact(kv_layers: List[torch.Tensor], hidden_tree: torch.Tensor, tb: TreeBuffers,
            best: int, accept_length: int, prev_len: int):
    """SURVEY a23.  kv_layers: tensors [..., max_len, hd] holding the verifier's cache with the
    tree's n_nodes rows written at [prev_len, prev_len+n_nodes).  Copies the accepted rows to
    [prev_len, prev_len+accept_length+1) in place; returns (new_len, accepted hidden
    [1, accept_length+1, D])."""
    sel = tb.retrieve_indices[best, : accept_length + 1]
    for data in kv_layers:
        rows = data[..., (sel + prev_len), :].clone()
        data[..., prev_len: prev_len + rows.shape[-2], :] = rows
    return prev_len + accept_length + 1, hidden_tree[:, sel]
