"""Explain-mode hooks of the conv stack.

The reference explains its model with `torch_geometric.explain.Explainer(GNNExplainer(), node_mask_type='attributes',
edge_mask_type='object')` on `GCN_explain` (scripts_experiments/explain_gnn.py:39-50, utils/other_utils.py:58-60).
PyG implements the edge mask by setting three attributes on every `MessagePassing` layer (`explain`, `_edge_mask`,
`_apply_sigmoid`: torch_geometric.explain.algorithm.utils.set_masks / clear_masks) and multiplying each message by
the mask inside `propagate` -- AFTER gcn_norm, in every conv layer, self loops keep mask 1.  The node mask is a plain
elementwise product on `x` done by the caller.

This module provides the same two functions for `hcatgnet_amd.GCNConv`; the forward then runs the any-shape HIP
kernels with the mask as per-edge multiplier and autograd delivers d out / d mask (csrc/layer.hip:
k_edge_weight_grad) and d out / d x.  The optimisation loop of GNNExplainer itself (and the plotting around it) is
outside this package: any torch optimiser over (node_mask, edge_mask) works on these gradients.
"""
from __future__ import annotations

import torch

from .gcn import GCNConv


def set_masks(model: torch.nn.Module, mask: torch.Tensor, edge_index: torch.Tensor = None, apply_sigmoid: bool = True):
    """Attach `mask` ([E], the batch's edge order) to every conv layer of `model` (PyG signature; `edge_index` is
    accepted for compatibility and only used to check the length)."""
    if edge_index is not None and mask.numel() != edge_index.shape[1]:
        raise ValueError(f"edge mask has {mask.numel()} entries for {edge_index.shape[1]} edges")
    for module in model.modules():
        if isinstance(module, GCNConv):
            module.explain = True
            module._edge_mask = mask
            module._apply_sigmoid = apply_sigmoid


def clear_masks(model: torch.nn.Module):
    for module in model.modules():
        if isinstance(module, GCNConv):
            module.explain = False
            module._edge_mask = None
            module._apply_sigmoid = True
