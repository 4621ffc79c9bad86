"""GCN model with the reference's `nn.Module` surface, computing on MI355X through the C ABI.

Mirrors reference model/gcn.py:9-76 (`GCN`) and :79-140 (`GCN_explain`): same constructor
(`opt`, `n_node_features`), same sub-module / parameter names (so the reference's 360
`model_params.pth` state-dicts load unchanged: conv1.lin.weight, conv1.bias, conv_layers.N.*,
readout.N.0.*, readout.L.*), same forward contract:
    model(batch)                      -> out [B, n_classes]
    model(batch, True)                -> (out, graph_emb [B, 2*embedding_dim])
    model(x, edge_index, edge_attr, batch)            (north_star tensor style; edge_attr ignored,
                                                       exactly as the reference ignores it, gcn.py:56)
    model(x=, edge_index=, batch_index=, edge_weight=) (explain style, gcn.py:124)
What differs is only WHERE the arithmetic runs: `GCNConv` / `global_*_pool` / readout `Linear`s
dispatch to libhcatgnet_hip.so instead of torch_geometric + torch ops.
"""
from __future__ import annotations

import argparse
import math
from typing import Optional

import torch
import torch.nn as nn

from . import functional as HF
from .batch import Batch
from .networks import BaseNetwork
from .plan import BatchPlan


class GCNConv(nn.Module):
    """Parameter container + layer call with PyG `GCNConv`'s names: `lin.weight` [out, in] (no
    bias, glorot-uniform) and a separate `bias` [out] (zeros)."""

    def __init__(self, in_channels: int, out_channels: int, improved: bool = False):
        super().__init__()
        self.in_channels, self.out_channels, self.improved = in_channels, out_channels, improved
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.empty(out_channels))
        # explain-mode hooks with PyG MessagePassing's attribute names (set by hcatgnet_amd.explain.set_masks):
        # when `explain` is on, every message is multiplied by `_edge_mask` (after a sigmoid if `_apply_sigmoid`)
        self.explain, self._edge_mask, self._apply_sigmoid = False, None, True
        self.family = "auto"      # "mid": keep a 128-wide layer on the one-graph-per-workgroup kernels (tests / A-B runs)
        self.reset_parameters()

    def reset_parameters(self):
        a = math.sqrt(6.0 / (self.in_channels + self.out_channels))  # glorot
        with torch.no_grad():
            self.lin.weight.uniform_(-a, a)
            self.bias.zero_()

    def forward(self, x, plan: BatchPlan, use_edge_weight: bool = False, apply_act: bool = False,
                fused: bool = True, pool: bool = False):
        """`pool=True` (last conv): returns the pooled graph embedding [B, 2*out] instead of the
        node embeddings.  `fused=True` uses the one-launch-per-layer small-graph kernels when the
        plan / shapes allow, else the any-shape kernels."""
        if self.explain and self._edge_mask is not None:
            m = self._edge_mask.sigmoid() if self._apply_sigmoid else self._edge_mask
            h = HF.gcn_layer(x, self.lin.weight, self.bias, plan, use_edge_weight, apply_act, edge_mult=m)
            return HF.graph_pool(h, plan) if pool else h
        gpt = 0 if (use_edge_weight or not fused) else HF.fused_graphs_per_tile(plan, self.in_channels, self.out_channels)
        if gpt > 0:
            return HF.fused_gcn_layer(x, self.lin.weight, self.bias, plan, gpt, apply_act, pool=pool)
        if fused and not use_edge_weight and self.family != "mid" and HF.tall_supported(plan, self.in_channels, self.out_channels):
            return HF.tall_gcn_layer(x, self.lin.weight, self.bias, plan, apply_act, pool=pool)  # wide layer, large graphs
        if fused and not use_edge_weight and HF.mid_supported(plan, self.in_channels, self.out_channels):
            return HF.mid_gcn_layer(x, self.lin.weight, self.bias, plan, apply_act, pool=pool)   # one graph per workgroup
        h = HF.gcn_layer(x, self.lin.weight, self.bias, plan, use_edge_weight, apply_act)
        return HF.graph_pool(h, plan) if pool else h


class _ReadoutLinear(nn.Linear):
    def forward(self, x, apply_act: bool = False):  # noqa: D401
        return HF.linear(x, self.weight, self.bias, apply_act)


class GCN(BaseNetwork):
    def __init__(self, opt: argparse.Namespace, n_node_features: int):
        super().__init__(opt=opt, n_node_features=n_node_features)
        self._name = "GCN"
        self.improved = opt.improved

        self.conv1 = GCNConv(self.n_node_features, self.embedding_dim, improved=self.improved)
        self.relu1 = nn.LeakyReLU()
        self.conv_layers = nn.ModuleList([])
        for _ in range(self.n_convolutions - 1):
            self.conv_layers.append(GCNConv(self.embedding_dim, self.embedding_dim, self.improved))

        graph_embedding = self.embedding_dim * 2
        self.readout = nn.ModuleList([])
        for _ in range(self.readout_layers - 1):
            reduced_dim = int(graph_embedding / 2)
            self.readout.append(nn.Sequential(_ReadoutLinear(graph_embedding, reduced_dim), nn.LeakyReLU()))
            graph_embedding = reduced_dim
        self.readout.append(_ReadoutLinear(graph_embedding, self._n_classes))

        self._make_loss(opt.problem_type)
        self._make_optimizer(opt.optimizer, opt.lr)
        self._make_scheduler(scheduler=opt.scheduler, step_size=opt.step_size, gamma=opt.gamma, min_lr=opt.min_lr)
        self.plan_mode = getattr(opt, "plan_mode", "auto")
        self.use_fused = bool(getattr(opt, "use_fused", True))
        self.single_node = bool(getattr(opt, "single_autograd_node", True))   # whole fused model as one Function

    # ------------------------------------------------------------------ argument handling
    def _plan_for(self, graph, x, edge_index, batch, edge_weight) -> BatchPlan:
        if graph is not None and edge_weight is None:
            cached = getattr(graph, "_hcg_plan", None)
            if cached is not None and cached.N == x.shape[0]:
                return cached
        num_graphs = getattr(graph, "num_graphs", None) if graph is not None else None
        trusted = bool(getattr(graph, "edges_grouped", False)) if graph is not None else False
        plan = BatchPlan.build(edge_index, batch, x.shape[0], num_graphs=num_graphs, edge_weight=edge_weight,
                               improved=self.improved, mode="blocked" if trusted else self.plan_mode,
                               validate=not trusted, max_nodes=getattr(graph, "max_nodes", None),
                               max_edges=getattr(graph, "max_edges", None))
        if graph is not None and edge_weight is None:
            try:
                graph._hcg_plan = plan
            except Exception:
                pass
        return plan

    def _run(self, x, plan: BatchPlan, use_edge_weight: bool, return_graph_embedding: bool):
        last = self.n_convolutions - 1
        fused = self.use_fused
        masked = any(getattr(c, "explain", False) and c._edge_mask is not None for c in [self.conv1] + list(self.conv_layers))
        if fused and not use_edge_weight and self.single_node and not masked:
            # whole model as one autograd node when every piece has a fused kernel (small graphs, D = 64)
            convs = [self.conv1] + list(self.conv_layers)
            gpts = [HF.fused_graphs_per_tile(plan, c.in_channels, c.out_channels) for c in convs]
            if (all(g > 0 for g in gpts) and self.readout_layers == 2 and x.is_cuda
                    and HF.readout2_supported(self.embedding_dim, self._n_classes)):
                cp = [t for c in convs for t in (c.lin.weight, c.bias)]
                l0, l1 = self.readout[0][0], self.readout[1]
                z, graph_emb = HF.fused_model(plan, gpts, x, cp, [l0.weight, l0.bias, l1.weight, l1.bias])
                return (z, graph_emb) if return_graph_embedding else z
        h = self.conv1(x, plan, use_edge_weight=use_edge_weight, apply_act=True, fused=fused, pool=last == 0)
        for i in range(self.n_convolutions - 1):                                       # gcn.py:61-63
            h = self.conv_layers[i](h, plan, use_edge_weight=False, apply_act=True, fused=fused, pool=i + 1 == last)
        graph_emb = h                                                                  # gcn.py:65-66 (pooled in-layer)
        z = graph_emb
        if (fused and self.readout_layers == 2 and graph_emb.is_cuda
                and HF.readout2_supported(self.embedding_dim, self._n_classes)):
            l0, l1 = self.readout[0][0], self.readout[1]                               # one launch for the head
            z = HF.readout2(graph_emb, l0.weight, l0.bias, l1.weight, l1.bias)
        else:
            for i in range(self.readout_layers):                                       # gcn.py:70-71
                layer = self.readout[i]
                z = layer[0](z, apply_act=True) if isinstance(layer, nn.Sequential) else layer(z)
        if return_graph_embedding:
            return z, graph_emb
        return z

    def forward(self, *args, **kwargs):
        graph = None
        if args and not torch.is_tensor(args[0]) and args[0] is not None:
            # object style: forward(reaction_graph, return_graph_embedding=False)   (gcn.py:54)
            graph = args[0]
            ret = args[1] if len(args) > 1 else kwargs.pop("return_graph_embedding", False)
            x, edge_index, batch = graph.x, graph.edge_index, getattr(graph, "batch", None)
            edge_weight = None                                                       # gcn.py:56
            plan = kwargs.pop("plan", None)
        else:
            # tensor style: forward(x, edge_index, edge_attr, batch) / explain keywords (gcn.py:124)
            names = ("x", "edge_index", "edge_attr", "batch")
            vals = dict(zip(names, args))
            vals.update({k: kwargs.pop(k) for k in list(kwargs) if k in names})
            x, edge_index = vals.get("x"), vals.get("edge_index")
            batch = vals.get("batch")
            if batch is None:
                batch = kwargs.pop("batch_index", None)
            edge_weight = kwargs.pop("edge_weight", None)
            ret = kwargs.pop("return_graph_embedding", False)
            plan = kwargs.pop("plan", None)
        if kwargs:
            raise TypeError(f"unexpected arguments: {sorted(kwargs)}")
        if x is None or edge_index is None:
            raise TypeError("forward needs x and edge_index")
        if plan is None:
            plan = self._plan_for(graph, x, edge_index, batch, edge_weight)
        return self._run(x, plan, edge_weight is not None, bool(ret))


class GCN_explain(GCN):
    """Tensor-style twin used by the explain scripts (reference model/gcn.py:79-140):
    forward(x, edge_index, batch_index, edge_weight) -> out (no embedding return)."""

    def forward(self, x=None, edge_index=None, batch_index=None, edge_weight=None):
        return super().forward(x=x, edge_index=edge_index, batch_index=batch_index, edge_weight=edge_weight)
