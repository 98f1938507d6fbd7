"""CPU pins of oracle/dyn_tree.py: hand-derived known answers for the dynamic-tree conventions frozen in
DESIGN.md 6b (the algorithm is not in /root/reference: parity unpinned)."""
import math

import torch

from oracle import dyn_tree as od


def test_rank_order_breaks_ties_by_lowest_index():
    assert od.rank_order(torch.tensor([0.5, 0.9, 0.5, 0.9, 0.1])) == [1, 3, 0, 2, 4]
    assert od.select_frontier(torch.tensor([[0.2, 0.2], [0.2, 0.2]]), 2) == [0, 1]


def test_expand_pool_layout_scores_and_masks():
    k, S = 2, 3
    ln = math.log
    # step 0: root children with p = 0.6, 0.3
    s0 = (torch.tensor([[ln(0.6), ln(0.3)]]), torch.tensor([[11, 12]]), torch.tensor([0.0]))
    # step 1: children of frontier rows (11: p .5/.4, 12: p .9/.05) -> scores .30 .24 | .27 .015 -> frontier = [.30 (row0,c0), .27 (row1,c0)]
    s1 = (torch.tensor([[ln(0.5), ln(0.4)], [ln(0.9), ln(0.05)]]), torch.tensor([[21, 22], [23, 24]]), torch.zeros(2))
    s2 = (torch.tensor([[ln(0.5), ln(0.1)], [ln(0.8), ln(0.1)]]), torch.tensor([[31, 32], [33, 34]]), torch.zeros(2))
    tok, sc, par, fr = od.expand_reference([s0, s1, s2], k, S)
    assert tok.tolist() == [11, 12, 21, 22, 23, 24, 31, 32, 33, 34]
    assert par.tolist() == [-1, -1, 0, 0, 1, 1, 2, 2, 4, 4]          # step-2 parents: pool entries 2 (tok 21) and 4 (tok 23)
    assert torch.allclose(sc, torch.tensor([.6, .3, .30, .24, .27, .015, .15, .03, .216, .027]), atol=1e-6)
    f0, f1 = fr
    assert f0[0].tolist() == [11, 12] and f0[3] == [0, 0] and f0[4].tolist() == [[1, 0], [0, 1]]
    assert f1[0].tolist() == [21, 23] and f1[2].tolist() == [2, 4] and f1[3] == [0, 1]
    assert f1[4].tolist() == [[1, 0, 1, 0], [0, 1, 0, 1]]            # each node sees its depth-1 ancestor and itself

    tree_tok, tree_par, pos, mask, sel = od.finalize(tok, sc, par, root_token=7, k=k, S=S, T=6)
    # best 5 scores: .6 (0), .3 (1), .30 (2) [tie with index 1 -> index order], .27 (4), .24 (3) -> pool order 0,1,2,3,4
    assert sel.tolist() == [0, 1, 2, 3, 4]
    assert tree_tok.tolist() == [7, 11, 12, 21, 22, 23]
    assert tree_par.tolist() == [-1, 0, 0, 1, 1, 2]
    assert pos.tolist() == [0, 1, 1, 2, 2, 2]
    assert mask[5].tolist() == [1, 0, 1, 0, 0, 1] and mask[3].tolist() == [1, 1, 0, 1, 0, 0]


def test_finalize_keeps_parents_of_selected_nodes_and_depths():
    k, S, T = 2, 3, 4
    tok = torch.arange(10) + 100
    sc = torch.tensor([.9, .05, .8, .01, .04, .03, .7, .6, .02, .01])
    par = torch.tensor([-1, -1, 0, 0, 1, 1, 2, 2, 4, 4], dtype=torch.int32)
    tree_tok, tree_par, pos, mask, sel = od.finalize(tok, sc, par, 1, k, S, T)
    assert sel.tolist() == [0, 2, 6]                                  # a chain root -> 100 -> 102 -> 106
    assert tree_par.tolist() == [-1, 0, 1, 2] and pos.tolist() == [0, 1, 2, 3]
    assert mask.tolist() == [[1, 0, 0, 0], [1, 1, 0, 0], [1, 1, 1, 0], [1, 1, 1, 1]]


def test_accept_walks_the_tree():
    tree_tok = torch.tensor([7, 11, 12, 21, 22, 23])
    tree_par = torch.tensor([-1, 0, 0, 1, 1, 2], dtype=torch.int32)
    # verifier: after root -> 12, after node 2 (tok 12) -> 23, after node 5 -> 99
    tgt = torch.tensor([12, 0, 23, 0, 0, 99])
    assert od.accept(tree_tok, tree_par, tgt, W=4) == ([0, 2, 5], 2, 99)
    assert od.accept(tree_tok, tree_par, torch.tensor([50, 0, 0, 0, 0, 0]), W=4) == ([0], 0, 50)
    assert od.accept(tree_tok, tree_par, tgt, W=2) == ([0, 2], 1, 23)   # path capped at W nodes
