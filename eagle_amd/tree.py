"""Static draft tree: buffers for tree expansion and greedy verification (host-side index logic).

``Llama2Model.init_tree`` in the reference names ``mc_sim_7b_63`` and ``generate_tree_buffers``
without defining them (eagle/llama2.py:453-455).  This module provides both, following the
upstream EAGLE-1 conventions frozen in DESIGN.md:

* nodes are the tree paths sorted by (length, path); node 0 is the root;
* top-k order everywhere is value-descending / index-ascending;
* ``expand="wchild"`` (default, upstream): only nodes that have children are run through the
  draft head; ``expand="all"`` runs every node;
* the flat top-k table has row 0 = catch-up forward, then one row per forwarded node.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

TOPK = 10  # eagle/llama2.py:11

mc_sim_7b_63 = [[0], [1], [2], [3], [0, 0], [0, 1], [0, 2], [1, 0], [1, 1], [2, 0], [2, 1], [3, 0],
                [0, 0, 0], [0, 0, 1], [0, 0, 2], [0, 1, 0], [0, 1, 1], [0, 2, 0], [0, 2, 1], [1, 0, 0],
                [0, 0, 0, 0], [0, 0, 0, 1], [0, 0, 0, 2], [0, 0, 0, 0, 0], [0, 0, 0, 0, 1]]


@dataclass
class DraftStep:
    nodes: List[int]
    tree_indices: torch.Tensor   # int64 [n_d]: index into the previous step's flattened top-k
    parent_row: torch.Tensor     # int64 [n_d]: row of the previous step's out_hidden to reuse
    repeat_nums: List[int]
    position_ids: torch.Tensor   # int64 [n_d] zeros (caller adds len_posi)
    attn_mask: torch.Tensor      # float [1,1,n_d,T_d]  (reference tree_mask format)
    attn_mask_u8: torch.Tensor   # uint8 [n_d,T_d]      (kernel format)
    row_base: int                # first flat top-k row produced by this step's nodes


class TreeBuffers:
    """All index buffers of one static tree, resident on ``device``."""

    def __init__(self, paths: Sequence[Sequence[int]] = mc_sim_7b_63, topk: int = TOPK,
                 expand: str = "wchild", device="cpu"):
        if expand not in ("wchild", "all"):
            raise ValueError(f"expand must be 'wchild' or 'all', got {expand!r}")
        order = sorted((tuple(p) for p in paths), key=lambda p: (len(p), p))
        if len(set(order)) != len(order):
            raise ValueError("duplicate tree path")
        node_of: Dict[Tuple[int, ...], int] = {(): 0}
        for p in order:
            node_of[p] = len(node_of)
        n = len(node_of)
        par = [-1] * n
        dep = [0] * n
        cho = [-1] * n
        nkids = [0] * n
        for p, i in node_of.items():
            if not p:
                continue
            if p[:-1] not in node_of:
                raise ValueError(f"tree path {list(p)} has no parent in the tree")
            if not 0 <= p[-1] < topk:
                raise ValueError(f"choice {p[-1]} outside top-{topk}")
            par[i], dep[i], cho[i] = node_of[p[:-1]], len(p), p[-1]
            nkids[par[i]] += 1
        self.topk, self.n_nodes, self.expand = topk, n, expand
        self.parent, self.depth, self.choice = par, dep, cho
        self.max_depth = max(dep)
        runs = (lambda i: nkids[i] > 0) if expand == "wchild" else (lambda i: True)
        forwarded = [i for i in range(1, n) if runs(i)]
        # flat top-k row produced by each forwarded node (root's catch-up forward = row 0)
        self.flat_row = [-1] * n
        self.flat_row[0] = 0
        for r, i in enumerate(forwarded):
            self.flat_row[i] = 1 + r
        self.n_flat_rows = 1 + len(forwarded)

        def chain(i: int) -> List[int]:
            out = []
            while i > 0:
                out.append(i)
                i = par[i]
            return out[::-1]

        # ---- draft side ----
        self.steps: List[DraftStep] = []
        n_steps = self.max_depth - 1 if expand == "wchild" else self.max_depth
        kv_order: List[int] = []
        prev_nodes = [0]
        row_base = 1
        for d in range(1, n_steps + 1):
            cur = [i for i in forwarded if dep[i] == d]
            if not cur:
                raise ValueError(f"no node to expand at depth {d}")
            rank = {node: r for r, node in enumerate(prev_nodes)}
            if any(par[i] not in rank for i in cur):
                raise ValueError("a forwarded node's parent was not forwarded")
            prow = [rank[par[i]] for i in cur]
            reps: List[int] = []
            for j, r in enumerate(prow):
                if j and r == prow[j - 1]:
                    reps[-1] += 1
                else:
                    reps.append(1)
            kv_order += cur
            col = {node: c for c, node in enumerate(kv_order)}
            m = torch.zeros(len(cur), len(kv_order), dtype=torch.uint8)
            for r, i in enumerate(cur):
                m[r, [col[a] for a in chain(i)]] = 1
            self.steps.append(DraftStep(
                nodes=cur,
                tree_indices=torch.tensor([cho[i] + topk * r for i, r in zip(cur, prow)], dtype=torch.int64, device=device),
                parent_row=torch.tensor(prow, dtype=torch.int64, device=device),
                repeat_nums=reps,
                position_ids=torch.zeros(len(cur), dtype=torch.int64, device=device),
                attn_mask=m.to(torch.float32)[None, None].to(device),
                attn_mask_u8=m.contiguous().to(device),
                row_base=row_base,
            ))
            row_base += len(cur)
            prev_nodes = cur

        # ---- verify side ----
        tm = torch.zeros(n, n)
        tm[:, 0] = 1
        for i in range(n):
            tm[i, i] = 1
            if i:
                tm[i, chain(i)] = 1
        ti = [0] * n
        for i in range(1, n):
            if self.flat_row[par[i]] < 0:
                raise ValueError("node whose parent is never forwarded")
            ti[i] = 1 + topk * self.flat_row[par[i]] + cho[i]
        leaves = [i for i in range(1, n) if nkids[i] == 0]
        width = self.max_depth + 1
        rows = [([0] + chain(i) + [-1] * width)[:width] for i in leaves]
        big = n + 5
        rows.sort(key=lambda r: [x if x >= 0 else big for x in r])
        self.tree_attn_mask = tm[None, None].to(device)
        self.tree_indices = torch.tensor(ti, dtype=torch.int64, device=device)
        self.tree_position_ids = torch.tensor(dep, dtype=torch.int64, device=device)
        self.retrieve_indices = torch.tensor(rows, dtype=torch.int64, device=device)

    @classmethod
    def build_on_device(cls, paths: Sequence[Sequence[int]] = mc_sim_7b_63, topk: int = TOPK, expand: str = "wchild",
                        device="cuda") -> "TreeBuffers":
        """The same buffers through the C ABI (eagle_tree_layout + eagle_tree_build): the index logic runs in the
        library (host analysis for the sizes, one kernel for the contents), which is what a non-Python host binds.
        The drafters use this path; ``TreeBuffers(...)`` above is the host-only equivalent (CPU tests, tooling)."""
        from . import ops
        if expand not in ("wchild", "all"):
            raise ValueError(f"expand must be 'wchild' or 'all', got {expand!r}")
        meta, o = ops.tree_build(paths, topk, expand == "all", device)
        self = cls.__new__(cls)
        n = meta["n_nodes"]
        self.topk, self.n_nodes, self.expand = topk, n, expand
        self.parent = o["parent"].cpu().tolist()
        self.depth = o["tree_position_ids"].cpu().tolist()
        ti = o["tree_indices"].cpu().tolist()
        self.choice = [-1] + [(ti[i] - 1) % topk for i in range(1, n)]
        self.max_depth = meta["max_depth"]
        self.flat_row = o["flat_row"].cpu().tolist()
        self.n_flat_rows = meta["n_flat_rows"]
        fwd = [i for i in range(1, n) if self.flat_row[i] > 0]
        reps = o["step_repeat_nums"].cpu().tolist()
        self.steps = []
        first, moff, cols, row_base = 0, 0, 0, 1
        for nd in meta["step_n"]:
            cols += nd
            m = o["step_attn_mask"][moff:moff + nd * cols].view(nd, cols)
            self.steps.append(DraftStep(
                nodes=fwd[first:first + nd], tree_indices=o["step_tree_indices"][first:first + nd],
                parent_row=o["step_parent_row"][first:first + nd], repeat_nums=[r for r in reps[first:first + nd] if r > 0],
                position_ids=torch.zeros(nd, dtype=torch.int64, device=device), attn_mask=m.to(torch.float32)[None, None],
                attn_mask_u8=m, row_base=row_base))
            first, moff, row_base = first + nd, moff + nd * cols, row_base + nd
        self.tree_attn_mask = o["tree_attn_mask"].to(torch.float32)[None, None]
        self.tree_indices, self.tree_position_ids = o["tree_indices"], o["tree_position_ids"]
        self.retrieve_indices = o["retrieve_indices"]
        return self

    # dict view with the upstream key names (what ``model.tree_buffer[...]`` users index)
    def as_dict(self) -> dict:
        return {
            "tree_indices": [s.tree_indices for s in self.steps],
            "attn_mask": [s.attn_mask for s in self.steps],
            "position_ids": [s.position_ids for s in self.steps],
            "repeat_nums": [s.repeat_nums for s in self.steps],
        }


def generate_tree_buffers(tree_choices=mc_sim_7b_63, device="cuda", topk: int = TOPK,
                          expand: str = "wchild") -> TreeBuffers:
    """The name eagle/llama2.py:455 calls.  Device buffers are built by the library (C ABI); a CPU ``device`` gets the
    host-only construction (there is no CPU compute path, but index buffers are plain integers)."""
    if torch.device(device).type == "cuda":
        return TreeBuffers.build_on_device(tree_choices, topk=topk, expand=expand, device=device)
    return TreeBuffers(tree_choices, topk=topk, expand=expand, device=device)
