"""DeviceGraphStore / DeviceLoader: the dataset resident in HBM, batches collated ON the GPU.

Replaces, for this path, the reference's per-graph `torch.load` (data/datasets.py:74-78) and PyG
`DataLoader(dataset, batch_size, shuffle)` collate (call_methods.py:41-46): one gather launch per batch
(csrc/collate.hip), and because the host knows every graph's size the batch plan (graph_ptr / edge_ptr)
comes for free -- the batch object carries a ready `BatchPlan`, so the fused path launches no plan kernel.
"""
from __future__ import annotations

from typing import Iterable, Optional, Sequence

import ctypes

import numpy as np
import torch

from . import _lib
from .batch import Batch, Data
from .plan import BatchPlan, _shared_status


class DeviceGraphStore:
    def __init__(self, graphs: Sequence[Data], device="cuda"):
        if len(graphs) == 0:
            raise ValueError("empty dataset")
        dev = torch.device(device)
        n = np.array([g.num_nodes for g in graphs], np.int64)
        e = np.array([g.num_edges for g in graphs], np.int64)
        self.num_graphs, self.F = len(graphs), int(graphs[0].x.shape[1])
        self.n_host, self.e_host = n, e
        self.node_ptr_host = np.concatenate([[0], np.cumsum(n)])
        self.edge_ptr_host = np.concatenate([[0], np.cumsum(e)])
        x = torch.cat([g.x.to(torch.float32) for g in graphs], 0)
        ei = torch.cat([g.edge_index for g in graphs], 1) if int(e.sum()) else torch.zeros(2, 0, dtype=torch.int64)
        for g in graphs:
            if g.num_edges and (int(g.edge_index.min()) < 0 or int(g.edge_index.max()) >= g.num_nodes):
                raise ValueError("edge_index of a graph refers to a node outside the graph")
        self.x_all = x.contiguous().to(dev)
        self.src_all = ei[0].to(torch.int32).contiguous().to(dev)
        self.dst_all = ei[1].to(torch.int32).contiguous().to(dev)
        self.node_ptr_all = torch.from_numpy(self.node_ptr_host).to(dev)
        self.edge_ptr_all = torch.from_numpy(self.edge_ptr_host).to(dev)
        ys = [getattr(g, "y", None) for g in graphs]
        self.y_all = torch.cat([torch.as_tensor(v, dtype=torch.float32).reshape(-1)[:1] for v in ys]).to(dev) \
            if all(v is not None for v in ys) else None
        ids = [getattr(g, "idx", None) for g in graphs]
        self.idx_all = torch.tensor([int(v) for v in ids], dtype=torch.int64, device=dev) if all(v is not None for v in ids) else None
        self.device = dev

    def __len__(self):
        return self.num_graphs

    def collate_args(self) -> "_lib.CollateArgs":
        """The dataset half of `hcg_collate`'s argument struct (made once; callers fill the slot half)."""
        a = getattr(self, "_collate_args", None)
        if a is None:
            p = _lib.ptr
            a = _lib.CollateArgs()
            a.x_all, a.src_all, a.dst_all, a.node_ptr_all, a.edge_ptr_all = (p(self.x_all), p(self.src_all), p(self.dst_all),
                                                                            p(self.node_ptr_all), p(self.edge_ptr_all))
            a.y_all, a.idx_all, a.F, a.nslots = p(self.y_all), p(self.idx_all), self.F, 1
            self._collate_args = a
        return a

    def collate(self, graph_ids: Iterable[int], _uploaded=None) -> Batch:
        """Batch of the given graphs (host list / array of indices), gathered on the device.
        `_uploaded` (DeviceLoader): device views (ids, graph_ptr, edge_ptr) of this batch inside ONE upload for the whole
        epoch -- otherwise the three index arrays of the batch go up here, one small copy each."""
        lib = _lib.load()
        ids = np.asarray(list(graph_ids) if not isinstance(graph_ids, np.ndarray) else graph_ids, np.int64)
        if ids.size == 0:
            raise ValueError("cannot collate an empty selection")
        if ids.min() < 0 or ids.max() >= self.num_graphs:
            raise IndexError("graph index out of range")
        B = int(ids.size)
        n, e = self.n_host[ids], self.e_host[ids]
        gp = np.zeros(B + 1, np.int32); gp[1:] = np.cumsum(n)
        ep = np.zeros(B + 1, np.int32); ep[1:] = np.cumsum(e)
        N, E = int(gp[-1]), int(ep[-1])
        dev = self.device
        if _uploaded is not None:
            ids_d, gp_d, ep_d = _uploaded
        else:
            ids_d = torch.from_numpy(ids).to(dev, non_blocking=True)
            gp_d = torch.from_numpy(gp).to(dev, non_blocking=True)
            ep_d = torch.from_numpy(ep).to(dev, non_blocking=True)
        x = torch.empty(N, self.F, dtype=torch.float32, device=dev)
        ei = torch.empty(2, E, dtype=torch.int64, device=dev)
        bvec = torch.empty(N, dtype=torch.int64, device=dev)
        y = torch.empty(B, dtype=torch.float32, device=dev) if self.y_all is not None else None
        idx = torch.empty(B, dtype=torch.int64, device=dev) if self.idx_all is not None else None
        p = _lib.ptr
        a = self.collate_args()
        sl = a.slot
        sl.ids, sl.graph_ptr, sl.edge_ptr, sl.x_out, sl.edge_index_out, sl.batch_out = p(ids_d), p(gp_d), p(ep_d), p(x), p(ei), p(bvec)
        sl.y_out, sl.idx_out, sl.B, sl.N_out, sl.E_out = p(y), p(idx), B, N, E
        _lib.check(lib.hcg_collate(ctypes.byref(a), _lib.stream_ptr()), "hcg_collate")
        batch = Batch(x, ei, bvec, B, y=y, idx=idx, max_nodes=int(n.max()), max_edges=int(e.max()), edges_grouped=True)
        # the plan of the fused path is exactly (graph_ptr, edge_ptr): attach it, nothing left to launch
        plan = BatchPlan()
        plan.N, plan.E, plan.B, plan.fill, plan.mode = N, E, B, 1.0, "blocked"
        plan.edge_index, plan.batch, plan.edge_weight = ei, bvec, None
        plan.graph_ptr, plan.edge_ptr = gp_d, ep_d
        plan.max_nodes, plan.max_edges, plan.validated, plan.has_csr = batch.max_nodes, batch.max_edges, True, False
        plan.shared_status, plan.status, plan.want_eid = True, _shared_status(dev), False
        plan.rowptr = plan.col = plan.eid = plan.rowptr_t = plan.col_t = plan.eid_t = None
        plan.dinv = plan.ew_csr = plan.ew_csc = plan.dinv_unw = None
        batch._hcg_plan = plan
        return batch


class DeviceLoader:
    """`DataLoader(dataset, batch_size, shuffle)` over a DeviceGraphStore: the permutation is drawn on the
    host (a few KB), everything else happens on the GPU."""

    def __init__(self, store: DeviceGraphStore, batch_size: int = 1, shuffle: bool = False, seed: Optional[int] = None,
                 drop_last: bool = False):
        self.store, self.batch_size, self.shuffle, self.drop_last = store, int(batch_size), shuffle, drop_last
        self.rng = np.random.default_rng(seed)
        self.dataset = store       # `len(loader.dataset)` is what the reference's loops divide by (utils_model.py:70)

    def __len__(self):
        n = len(self.store)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        st = self.store
        order = self.rng.permutation(len(st)) if self.shuffle else np.arange(len(st))
        starts = [i for i in range(0, len(order), self.batch_size)
                  if not (self.drop_last and len(order) - i < self.batch_size)]
        # the epoch's index arrays in ONE upload: [ids of every batch | graph_ptr of every batch | edge_ptr of every batch]
        # as int64 words (graph_ptr / edge_ptr are int32: two per word, every batch's block starts on a word) -- a batch
        # then costs its gather launch and no host-to-device copy of its own (three ~10 us copies per step otherwise)
        G = len(order)
        words = lambda B: (B + 2) // 2                       # int64 words that hold B + 1 int32
        tot = sum(words(min(self.batch_size, G - i)) for i in starts)
        buf = np.zeros(G + 2 * tot, np.int64)
        buf[:G] = order
        ptr32 = buf[G:].view(np.int32)
        spans, off = [], 0
        for i in starts:
            ids = order[i:i + self.batch_size]
            B = ids.size
            ptr32[off + 1:off + B + 1] = np.cumsum(st.n_host[ids])
            ptr32[2 * tot + off + 1:2 * tot + off + B + 1] = np.cumsum(st.e_host[ids])
            spans.append((i, B, off))
            off += 2 * words(B)
        dbuf = torch.from_numpy(buf).to(st.device)
        d32 = dbuf[G:].view(torch.int32)
        for i, B, off in spans:
            yield st.collate(order[i:i + B], _uploaded=(dbuf[i:i + B], d32[off:off + B + 1],
                                                         d32[2 * tot + off:2 * tot + off + B + 1]))
