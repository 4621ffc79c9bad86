"""Synthetic catalyst-sized molecular graphs for the throughput harness.

Per graph: n_g atoms, a random spanning tree plus `extra` ring-closure bonds (no self loops, no
duplicate bonds, atom degree <= max_degree); every bond is stored in both directions,
INTERLEAVED `[i, j], [j, i]` in bond-list order exactly as the reference's featuriser emits them
(data/rhcaa.py:156-166), and per-graph blocks are concatenated with cumulative node offsets as
PyG collation does (data/rhcaa.py:66-67) -- i.e. NOT sorted by target node.
x ~ N(0, 1) fp32, y ~ N(0, 10^2) fp32.  Seed: options/base_options.py:372 (20232023) + 1000*rank.

CONFIGS names the BASELINE.json configurations:
  C1  one 30-atom graph (32 bonds -> 64 directed edges), 64-d
  C2/C3/C4  4096 graphs x 30 atoms x 64 directed edges x 64-d (per GPU)
  C5  1024 graphs x 200 atoms x 424 directed edges x 128-d, max degree <= 6
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from .batch import Batch

BASE_SEED = 20232023

CONFIGS = {
    "C1": dict(num_graphs=1, nodes=30, extra_bonds=3, max_degree=4, feat=64, hidden=64),
    "C2": dict(num_graphs=4096, nodes=30, extra_bonds=3, max_degree=4, feat=64, hidden=64),
    "C5": dict(num_graphs=1024, nodes=200, extra_bonds=13, max_degree=6, feat=128, hidden=128),
}
# the reference's own regime (SURVEY 8 "Real data"): 56-184 atoms (mean 87), F = 25 (diene) -- not a BASELINE config
CONFIGS["REAL"] = dict(num_graphs=4096, nodes=87, nodes_jitter=30, extra_bonds=4, max_degree=4, feat=25, hidden=64)
CONFIGS["REAL40"] = dict(CONFIGS["REAL"], num_graphs=40)      # the reference's batch_size (options/base_options.py:269-274)
# development: every graph in the one-graph-per-wave size class (37-63 atoms), the C3 widths -- not a BASELINE config
CONFIGS["W50"] = dict(num_graphs=4096, nodes=50, nodes_jitter=13, extra_bonds=3, max_degree=4, feat=64, hidden=64)
CONFIGS["C3"] = CONFIGS["C2"]
CONFIGS["C4"] = CONFIGS["C2"]


def _random_bonds(rng: np.random.Generator, n: np.ndarray, extra: int, max_degree: int):
    """Vectorised over graphs.  Returns (bi, bj) int arrays [B, nmax-1+extra] (-1 = absent)."""
    B, nmax = len(n), int(n.max())
    deg = np.zeros((B, nmax), np.int32)
    adj = np.zeros((B, nmax, nmax), bool)
    nb = nmax - 1 + extra
    bi = np.full((B, nb), -1, np.int64)
    bj = np.full((B, nb), -1, np.int64)
    rows = np.arange(B)
    for k in range(1, nmax):  # spanning tree: attach atom k to a random earlier atom with room
        live = k < n
        score = rng.random((B, k))
        score[deg[:, :k] >= max_degree] = -1.0
        p = score.argmax(1)
        ok = live & (score[rows, p] >= 0)
        r = rows[ok]
        bi[r, k - 1], bj[r, k - 1] = p[ok], k
        deg[r, p[ok]] += 1
        deg[r, k] += 1
        adj[r, p[ok], k] = adj[r, k, p[ok]] = True
    valid = np.arange(nmax)[None, :] < n[:, None]
    for t in range(extra):  # ring closures
        score = rng.random((B, nmax, nmax))
        room = (deg < max_degree) & valid
        mask = room[:, :, None] & room[:, None, :] & ~adj & ~np.eye(nmax, dtype=bool)[None]
        score[~mask] = -1.0
        flat = score.reshape(B, -1).argmax(1)
        i, j = flat // nmax, flat % nmax
        ok = score.reshape(B, -1)[rows, flat] >= 0
        r = rows[ok]
        bi[r, nmax - 1 + t], bj[r, nmax - 1 + t] = i[ok], j[ok]
        deg[r, i[ok]] += 1
        deg[r, j[ok]] += 1
        adj[r, i[ok], j[ok]] = adj[r, j[ok], i[ok]] = True
    return bi, bj


@dataclass
class SynthBatch:
    x: torch.Tensor            # [N, F] fp32
    edge_index: torch.Tensor   # [2, E] int64, reference (unsorted, direction-interleaved) order
    batch: torch.Tensor        # [N] int64
    y: torch.Tensor            # [B] fp32
    num_graphs: int
    max_nodes: int
    max_edges: int
    n_small: Optional[int] = None     # size-grouped batch: the first n_small graphs have <= 32 nodes (batch.collate)

    def as_batch(self, device=None) -> Batch:
        b = Batch(self.x, self.edge_index, self.batch, self.num_graphs, y=self.y, max_nodes=self.max_nodes,
                  max_edges=self.max_edges, edges_grouped=True, n_small=self.n_small)
        return b.to(device) if device is not None else b

    def as_graph_list(self):
        """The batch split back into per-graph `Data` objects (local node ids, y, idx = position): the form a
        dataset holds them in (reference data/rhcaa.py:78-92), e.g. to fill a `DeviceGraphStore`."""
        from .batch import Data
        counts = torch.bincount(self.batch, minlength=self.num_graphs)
        ptr = torch.zeros(self.num_graphs + 1, dtype=torch.int64)
        ptr[1:] = counts.cumsum(0)
        eg = self.batch[self.edge_index[0]]
        ecount = torch.bincount(eg, minlength=self.num_graphs)
        eptr = torch.zeros(self.num_graphs + 1, dtype=torch.int64)
        eptr[1:] = ecount.cumsum(0)
        out = []
        for g in range(self.num_graphs):
            a, b, ea, eb = int(ptr[g]), int(ptr[g + 1]), int(eptr[g]), int(eptr[g + 1])
            out.append(Data(x=self.x[a:b].clone(), edge_index=(self.edge_index[:, ea:eb] - a).clone(),
                            y=self.y[g:g + 1].clone(), idx=g))
        return out


def make_batch(num_graphs: int, nodes: int, extra_bonds: int, max_degree: int, feat: int, seed: int = BASE_SEED,
               rank: int = 0, nodes_jitter: int = 0, group_by_size: bool = False, **_unused) -> SynthBatch:
    """`nodes_jitter` j > 0 draws n_g ~ U{nodes-j .. nodes+j} (the SURVEY's variable-size variant).
    `group_by_size`: the graphs with <= 32 nodes first (what `batch.collate(..., group_by_size=True)` emits)."""
    rng = np.random.default_rng(seed + 1000 * rank)
    n = np.full(num_graphs, nodes, np.int64)
    if nodes_jitter:
        n = rng.integers(nodes - nodes_jitter, nodes + nodes_jitter + 1, num_graphs)
    bi, bj = _random_bonds(rng, n, extra_bonds, max_degree)
    n_small = None
    if group_by_size:
        order = np.argsort(n > 32, kind="stable")
        n, bi, bj = n[order], bi[order], bj[order]
        n_small = int((n <= 32).sum())
    ptr = np.zeros(num_graphs + 1, np.int64)
    ptr[1:] = np.cumsum(n)
    present = bi >= 0
    gi = np.broadcast_to(np.arange(num_graphs)[:, None], bi.shape)[present]
    i = bi[present] + ptr[gi]
    j = bj[present] + ptr[gi]
    src = np.stack([i, j], 1).reshape(-1)   # [i0, j0, i1, j1, ...]
    dst = np.stack([j, i], 1).reshape(-1)   # [j0, i0, j1, i1, ...]
    N = int(ptr[-1])
    x = rng.standard_normal((N, feat), dtype=np.float32)
    y = (10.0 * rng.standard_normal(num_graphs)).astype(np.float32)
    batch = np.repeat(np.arange(num_graphs), n)
    edges_per_graph = 2 * present.sum(1)
    return SynthBatch(torch.from_numpy(x), torch.from_numpy(np.stack([src, dst])), torch.from_numpy(batch),
                      torch.from_numpy(y), num_graphs, int(n.max()), int(edges_per_graph.max()), n_small)


def make_config(name: str, rank: int = 0, num_graphs: Optional[int] = None, **kw) -> SynthBatch:
    cfg = dict(CONFIGS[name])
    if num_graphs is not None:
        cfg["num_graphs"] = num_graphs
    cfg.update(kw)
    return make_batch(rank=rank, **cfg)
