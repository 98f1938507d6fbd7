"""Oracle (TEST INFRASTRUCTURE ONLY): dynamic draft tree -- cumulative-score frontier selection,
pool re-ranking into the verified tree, greedy acceptance by tree walk.  **parity unpinned**.

Not in /root/reference: ``benchmark/benchmark.py:48-56`` passes ``speculative_num_steps``,
``speculative_eagle_topk`` and ``speculative_num_draft_tokens`` to ``sglang.Engine`` (SGLang
v0.4.6.post5, ``benchmark/README.md:22``), which is neither vendored nor installed.  This module
restates the published algorithm SGLang's EAGLE worker runs for those flags (EAGLE-2:
``select_top_k_tokens``, ``organize_draft_results``, ``build_tree``, greedy verification), with the
conventions frozen in DESIGN.md section 6b.  Its pins are the hand-derived known answers in
tests/test_dyn_tree_oracle.py.

Conventions
-----------
* k candidates per expanded node, S steps, T verified nodes including the root.
* pool: every candidate in production order -- the k children of the root (step 0), then per step
  i >= 1 the k children of each of the k frontier nodes, frontier row major: index
  ``k + (i-1) k^2 + j k + c``.  ``pool_parent`` = pool index of the parent, -1 for root children.
* score = product of softmax probabilities along the path, fp32.
* every ordering is (score descending, pool / candidate index ascending).  torch.topk is not used.
* frontier after a step = its k best candidates in rank order; the last step's candidates are never
  forwarded.
* verified tree = the T-1 best pool entries, KEPT IN POOL ORDER (a parent precedes its children),
  as nodes 1..T-1; node 0 is the root token; depth = step + 1.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch


def rank_order(scores: torch.Tensor) -> List[int]:
    """Indices sorted by (score desc, index asc)."""
    s = scores.reshape(-1).tolist()
    return sorted(range(len(s)), key=lambda i: (-s[i], i))


def select_frontier(cand_scores: torch.Tensor, k: int) -> List[int]:
    """The k best of the R*k candidates of one step, in rank order (SGLang: fast_topk over the
    flattened expand_scores)."""
    return rank_order(cand_scores)[:k]


def step_masks(prev_mask: torch.Tensor, parents: Sequence[int], step: int, k: int) -> torch.Tensor:
    """Attention mask rows of the new frontier over the (step+1)*k tree rows appended so far:
    the parent's row (width step*k) | the node's own column step*k + rank."""
    m = torch.zeros(k, (step + 1) * k, dtype=torch.uint8)
    for r, j in enumerate(parents):
        if step > 0:
            m[r, : step * k] = prev_mask[j]
        m[r, step * k + r] = 1
    return m


def finalize(pool_tok: torch.Tensor, pool_score: torch.Tensor, pool_parent: torch.Tensor, root_token: int, k: int, S: int,
             T: int):
    """(tree_tok [T], tree_parent [T], tree_pos [T], tree_mask [T,T], sel [T-1]) -- SGLang's
    organize_draft_results + tree construction."""
    P = k + (S - 1) * k * k
    assert pool_tok.numel() == P and T - 1 <= P
    sel = sorted(rank_order(pool_score)[: T - 1])
    node_of = {p: n + 1 for n, p in enumerate(sel)}
    tok = torch.zeros(T, dtype=torch.long)
    par = torch.zeros(T, dtype=torch.int32)
    pos = torch.zeros(T, dtype=torch.long)
    tok[0], par[0] = root_token, -1
    for p in sel:
        n = node_of[p]
        pp = int(pool_parent[p])
        assert pp < 0 or pp in node_of, "a selected node's parent must be selected"
        tok[n] = pool_tok[p]
        par[n] = 0 if pp < 0 else node_of[pp]
        pos[n] = 1 if p < k else 2 + (p - k) // (k * k)
    mask = torch.zeros(T, T, dtype=torch.uint8)
    for n in range(T):
        mask[n, n] = 1
        mask[n, 0] = 1
        a = int(par[n])
        while a > 0:
            mask[n, a] = 1
            a = int(par[a])
    return tok, par, pos, mask, torch.tensor(sel, dtype=torch.int32)


def accept(tree_tok: torch.Tensor, tree_parent: torch.Tensor, target_argmax: torch.Tensor, W: int) -> Tuple[List[int], int, int]:
    """Greedy acceptance: from the root follow the child whose token equals the verifier's argmax at
    the current node (lowest node id on a duplicate) until none matches or the path holds W nodes.
    Returns (path node ids, accept_length, next_token)."""
    T = tree_tok.numel()
    path, cur = [0], 0
    while len(path) < W:
        want = int(target_argmax[cur])
        nxt = [n for n in range(1, T) if int(tree_parent[n]) == cur and int(tree_tok[n]) == want]
        if not nxt:
            break
        cur = nxt[0]
        path.append(cur)
    return path, len(path) - 1, int(target_argmax[cur])


def expand_reference(step_topk: Sequence[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]], k: int, S: int):
    """Pool bookkeeping from per-step (vals [R,k] float, ids [R,k], lse [R]) tuples that were produced
    ALONG THE FRONTIER THIS FUNCTION SELECTS (teacher forcing is the caller's business: tests feed the
    device's own per-step top-k).  Returns (pool_tok, pool_score, pool_parent, frontiers) where
    frontiers[i] = (tokens, scores, pool indices, parent rows, mask) after step i < S-1."""
    P = k + (S - 1) * k * k
    pool_tok = torch.zeros(P, dtype=torch.long)
    pool_score = torch.zeros(P, dtype=torch.float32)
    pool_parent = torch.full((P,), -1, dtype=torch.int32)
    frontiers = []
    fscore = fpool = mask = None
    for i, (vals, ids, lse) in enumerate(step_topk):
        R = 1 if i == 0 else k
        p = torch.exp(vals.float().reshape(R, k) - lse.float().reshape(R, 1))
        sc = p if i == 0 else fscore.reshape(R, 1) * p
        base = 0 if i == 0 else k + (i - 1) * k * k
        pool_tok[base: base + R * k] = ids.reshape(-1)
        pool_score[base: base + R * k] = sc.reshape(-1)
        if i > 0:
            pool_parent[base: base + R * k] = fpool.reshape(R, 1).expand(R, k).reshape(-1)
        if i == S - 1:
            break
        pick = select_frontier(sc, k)
        parents = [c // k for c in pick]
        mask = step_masks(mask, parents, i, k)
        fscore = sc.reshape(-1)[pick]
        fpool = torch.tensor([base + c for c in pick], dtype=torch.int32)
        frontiers.append((ids.reshape(-1)[pick].clone(), fscore.clone(), fpool.clone(), parents, mask.clone()))
    return pool_tok, pool_score, pool_parent, frontiers
