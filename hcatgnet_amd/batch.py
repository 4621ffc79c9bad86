"""Minimal graph containers + collation (torch_geometric is not a dependency).

`Data` mirrors the fields the reference's datasets store per reaction graph (data/rhcaa.py:78-92:
x, edge_index, edge_attr, y, idx, fold, ...), `Batch` mirrors what the training / predict loops read
from a PyG batch (utils/utils_model.py:61-68, :94: .x .edge_index .batch .y .idx .num_graphs .to()).
`collate` reproduces PyG's Batch collation rule: concatenate x, concatenate
edge_index + cumulative node offset, batch = repeat_interleave(arange(B), n_g).
Host-side metadata (num_graphs, max nodes / edges per graph) rides along so the GPU plan can be
built without a device->host sync.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch


class Data:
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, idx=None, **extra):
        self.x, self.edge_index, self.edge_attr, self.y, self.idx = x, edge_index, edge_attr, y, idx
        for k, v in extra.items():
            setattr(self, k, v)

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])


_TENSOR_FIELDS = ("x", "edge_index", "edge_attr", "y", "idx", "batch", "ptr", "edge_ptr")


class Batch:
    """A batch of graphs; duck-types the PyG `Batch` attributes the reference loops use."""

    def __init__(self, x, edge_index, batch, num_graphs: int, edge_attr=None, y=None, idx=None, ptr=None,
                 edge_ptr=None, max_nodes: Optional[int] = None, max_edges: Optional[int] = None,
                 edges_grouped: bool = False, n_small: Optional[int] = None):
        self.x, self.edge_index, self.batch = x, edge_index, batch
        self.edge_attr, self.y, self.idx, self.ptr, self.edge_ptr = edge_attr, y, idx, ptr, edge_ptr
        self.num_graphs = int(num_graphs)
        self.max_nodes, self.max_edges = max_nodes, max_edges
        self.edges_grouped = edges_grouped      # True when produced by `collate` (per-graph edge blocks)
        # size-grouped batches (`collate(..., group_by_size=True)`): the first `n_small` graphs have <= SMALL_NODES nodes,
        # the others more -- the fused trainer then runs each group through its own kernel family
        self.n_small = n_small
        self._hcg_plan = None                   # BatchPlan cache (built on first forward)

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    def to(self, device, non_blocking: bool = False) -> "Batch":
        want = torch.device(device)
        if all(t.device.type == want.type and (want.index is None or t.device.index == want.index)
               for t in (getattr(self, f, None) for f in _TENSOR_FIELDS) if torch.is_tensor(t)):
            return self                          # already there (a DeviceLoader's batches): nothing to move, nothing to rebuild
        kw = {}
        for f in _TENSOR_FIELDS:
            t = getattr(self, f, None)
            kw[f] = t.to(device, non_blocking=non_blocking) if torch.is_tensor(t) else t
        out = Batch(kw["x"], kw["edge_index"], kw["batch"], self.num_graphs, kw["edge_attr"], kw["y"], kw["idx"],
                    kw["ptr"], kw["edge_ptr"], self.max_nodes, self.max_edges, self.edges_grouped, self.n_small)
        if self._hcg_plan is not None and self.x.device == out.x.device:
            out._hcg_plan = self._hcg_plan
        return out


SMALL_NODES = 32      # graphs up to here run in the small-graph tiles (csrc/fused.hip)


def collate(graphs: Sequence[Data], group_by_size: bool = False) -> Batch:
    """PyG-equivalent collation of a list of graphs into one block-diagonal batch.
    `group_by_size`: the graphs with <= SMALL_NODES nodes first, the larger ones behind them (a stable partition; the
    order of the graphs inside a batch is arbitrary -- the reference shuffles it every epoch, call_methods.py:46 -- and
    y / idx move with their graphs).  `Batch.n_small` then says where the second group starts."""
    if len(graphs) == 0:
        raise ValueError("cannot collate an empty list of graphs")
    n_small = None
    if group_by_size:
        graphs = [g for g in graphs if g.num_nodes <= SMALL_NODES] + [g for g in graphs if g.num_nodes > SMALL_NODES]
        n_small = sum(1 for g in graphs if g.num_nodes <= SMALL_NODES)
    n = [g.num_nodes for g in graphs]
    e = [g.num_edges for g in graphs]
    dev = graphs[0].x.device
    ptr = torch.zeros(len(graphs) + 1, dtype=torch.int64)
    ptr[1:] = torch.tensor(n, dtype=torch.int64).cumsum(0)
    eptr = torch.zeros(len(graphs) + 1, dtype=torch.int64)
    eptr[1:] = torch.tensor(e, dtype=torch.int64).cumsum(0)
    x = torch.cat([g.x for g in graphs], 0)
    offs = torch.repeat_interleave(ptr[:-1], torch.tensor(e, dtype=torch.int64)).to(dev)
    edge_index = torch.cat([g.edge_index for g in graphs], 1) + offs.unsqueeze(0)
    batch = torch.repeat_interleave(torch.arange(len(graphs), dtype=torch.int64), torch.tensor(n, dtype=torch.int64)).to(dev)
    edge_attr = torch.cat([g.edge_attr for g in graphs], 0) if all(g.edge_attr is not None for g in graphs) else None

    def cat_scalar(name):
        vals = [getattr(g, name, None) for g in graphs]
        if any(v is None for v in vals):
            return None
        return torch.cat([torch.as_tensor(v).reshape(-1) for v in vals], 0)

    return Batch(x, edge_index, batch, len(graphs), edge_attr, cat_scalar("y"), cat_scalar("idx"), ptr.to(dev),
                 eptr.to(dev), max(n), max(e), edges_grouped=True, n_small=n_small)


class DataLoader:
    """Tiny stand-in for `torch_geometric.loader.DataLoader(dataset, batch_size, shuffle)`
    (reference call_methods.py:41-46): yields collated `Batch` objects."""

    def __init__(self, dataset: Sequence[Data], batch_size: int = 1, shuffle: bool = False,
                 generator: Optional[torch.Generator] = None):
        self.dataset, self.batch_size, self.shuffle, self.generator = list(dataset), int(batch_size), shuffle, generator

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = torch.randperm(len(self.dataset), generator=self.generator).tolist() if self.shuffle \
            else list(range(len(self.dataset)))
        for i in range(0, len(order), self.batch_size):
            yield collate([self.dataset[j] for j in order[i:i + self.batch_size]])
