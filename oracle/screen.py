"""TEST INFRASTRUCTURE ONLY -- makes a full-size synthetic batch *decidable* for gradient parity.

Why.  Two points of the reference's backward (torch autograd through model/gcn.py:54-76) are discontinuous in the
forward values: the LeakyReLU derivative (1 above zero, 0.01 at or below: gcn.py:21,59,63 and the readout's, :41)
and the arg-max of `global_max_pool` (gcn.py:65).  Two correct fp32 evaluations of the forward differ by ~1e-7, so an
element whose pre-activation lies within ~1e-7 of zero, or a (graph, feature) whose two largest node values lie within
~1e-7 of each other, may legitimately take either branch -- and ONE such element moves a weight gradient by ~1e-3 of
its scale (a whole node's term in a random-walk sum over 1e5 nodes).  At 4096 graphs x 30 atoms x 64 features x 2
layers about one such element is EXPECTED per batch, so an unscreened full-size gradient comparison fails (or passes)
by luck of the seed.

What.  `ambiguous_graphs` evaluates the oracle's forward in fp64 and flags the graphs that own such an element (margin
`abs_kink` / `rel_tie`, ten times the fp32 rounding noise); `make_decidable` re-draws the node features of exactly those
graphs (same sizes, same bonds, same targets) until none is left.  Shapes and statistics of the batch do not change:
about 1-4 % of the graphs get new N(0, 1) features.  Exact ties (bit-identical rows: chemically equivalent atoms) are NOT
ambiguous -- both sides split them evenly (torch `amax` semantics) -- and are left alone.

Used by: tests/test_gpu_fullsize.py and bench.py's `parity_gate` (as the checker, never as the thing measured).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from . import gcn_oracle as O


def ambiguous_graphs(params: Dict[str, torch.Tensor], x: torch.Tensor, edge_index: torch.Tensor, batch: torch.Tensor,
                     num_graphs: int, abs_kink: float = 2e-6, rel_tie: float = 2e-6) -> torch.Tensor:
    """-> bool [num_graphs]: graph owns an activation within `abs_kink` of the LeakyReLU kink (conv layers, readout) or a
    max-pool near-tie (0 < gap < rel_tie * max(|max|, 1e-3)).  fp64 evaluation of gcn_oracle's forward."""
    p = {k: v.detach().double() for k, v in params.items()}
    n_conv, n_read = O.infer_depths(p)
    bad = torch.zeros(num_graphs, dtype=torch.bool)
    h = x.double()
    for wk, bk in O.conv_param_names(n_conv):
        pre = O.gcn_conv(h, edge_index, p[wk], p[bk])
        near = (pre.abs() < abs_kink).any(dim=1)
        bad[batch[near]] = True
        h = F.leaky_relu(pre, O.LEAKY_SLOPE)
    idx = batch.unsqueeze(1).expand_as(h)
    top = h.new_full((num_graphs, h.shape[1]), float("-inf")).scatter_reduce(0, idx, h, reduce="amax", include_self=True)
    below = torch.where(h < top[batch], h, torch.full_like(h, float("-inf")))
    second = h.new_full((num_graphs, h.shape[1]), float("-inf")).scatter_reduce(0, idx, below, reduce="amax", include_self=True)
    gap = top - second                                             # inf where the graph has one distinct value / is empty
    bad |= ((gap > 0) & (gap < rel_tie * top.abs().clamp_min(1e-3))).any(dim=1)
    z = torch.cat([O.global_max_pool(h, batch, num_graphs), O.global_mean_pool(h, batch, num_graphs)], dim=1)
    rn = O.readout_param_names(n_read)
    for i, (wk, bk) in enumerate(rn):
        z = F.linear(z, p[wk], p[bk])
        if i < len(rn) - 1:
            bad |= (z.abs() < abs_kink).any(dim=1)
            z = F.leaky_relu(z, O.LEAKY_SLOPE)
    return bad


def make_decidable(params: Dict[str, torch.Tensor], x: torch.Tensor, edge_index: torch.Tensor, batch: torch.Tensor,
                   num_graphs: int, seed: int = 0, max_rounds: int = 12, **margins):
    """-> (x', redrawn): `x` with the node features of every ambiguous graph re-drawn ~ N(0, 1) (generator seeded with
    `seed`), repeated until no graph is ambiguous; `redrawn` = how many graphs were touched.  Raises if `max_rounds`
    do not suffice (a margin far too wide for the batch)."""
    x = x.clone()
    g = torch.Generator().manual_seed(int(seed))
    touched = torch.zeros(num_graphs, dtype=torch.bool)
    for _ in range(max_rounds):
        bad = ambiguous_graphs(params, x, edge_index, batch, num_graphs, **margins)
        if not bool(bad.any()):
            return x, int(touched.sum())
        touched |= bad
        rows = bad[batch]
        x[rows] = torch.randn(int(rows.sum()), x.shape[1], generator=g, dtype=x.dtype)
    raise RuntimeError("make_decidable: ambiguous graphs left after %d rounds" % max_rounds)
