"""BatchPlan: the per-batch index structures of the GCN path, built once per batch on the GPU.

Replaces what PyG recomputes inside every `GCNConv.forward` (gcn_norm: self loops, degree,
deg^-1/2; reference call sites model/gcn.py:58,62) and the `batch`-vector scatter indices of the
pools (model/gcn.py:65-66): CSR by target node + its transpose, `dinv`, `graph_ptr`.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


class BatchPlan:
    __slots__ = ("N", "E", "B", "mode", "fill", "graph_ptr", "edge_ptr", "rowptr", "col", "eid", "rowptr_t",
                 "col_t", "eid_t", "dinv", "ew_csr", "ew_csc", "dinv_unw", "status", "max_nodes", "max_edges", "validated")

    def check_status(self):
        """Synchronising read of the device-side status word; raises on any violation."""
        word = int(self.status[0].item())
        self.validated = True
        if word:
            raise ValueError("hcatgnet_amd: invalid batch: " + _lib.describe_status(word))
        return word

    @staticmethod
    def build(edge_index: torch.Tensor, batch: Optional[torch.Tensor], num_nodes: int,
              num_graphs: Optional[int] = None, edge_weight: Optional[torch.Tensor] = None,
              improved: bool = False, mode: str = "auto", validate: bool = True,
              max_nodes: Optional[int] = None, max_edges: Optional[int] = None) -> "BatchPlan":
        """mode: 'blocked' (edges grouped by graph, the order PyG-style collation emits),
        'general' (any order; device radix sort) or 'auto' (blocked, re-planned as general when the
        device reports ungrouped edges -- needs `validate=True`, i.e. one 4-byte D2H sync).
        `validate=False` + an explicit mode enqueues the build without any host sync."""
        lib = _lib.load()
        _lib.require_gpu(edge_index, batch, edge_weight)
        dev = edge_index.device
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError("edge_index must be int64 [2, E]")
        edge_index = edge_index.contiguous()
        N, E = int(num_nodes), int(edge_index.shape[1])
        if batch is None:
            batch = torch.zeros(N, dtype=torch.int64, device=dev)
            num_graphs = 1 if num_graphs is None else num_graphs
        if batch.dtype != torch.int64 or batch.numel() != N:
            raise ValueError("batch must be int64 [N]")
        batch = batch.contiguous()
        if num_graphs is None:  # same sync PyG's global pools do (`int(batch.max()) + 1`)
            num_graphs = int(batch[-1].item()) + 1 if N > 0 else 0
        B = int(num_graphs)
        if edge_weight is not None:
            if edge_weight.numel() != E:
                raise ValueError("edge_weight must have E entries")
            edge_weight = edge_weight.detach().to(torch.float32).contiguous()
            fill = 2.0 if improved else 1.0
        else:
            fill = 1.0  # `improved` is a no-op when edge_weight is None
        if mode not in ("auto", "blocked", "general"):
            raise ValueError(f"unknown plan mode {mode!r}")
        if mode == "auto" and not validate:
            mode = "general"
        want = "blocked" if mode in ("auto", "blocked") else "general"

        p = BatchPlan()
        p.N, p.E, p.B, p.fill = N, E, B, fill
        p.max_nodes, p.max_edges, p.validated = max_nodes, max_edges, False
        i32 = dict(dtype=torch.int32, device=dev)
        p.graph_ptr = torch.empty(B + 1, **i32)
        p.edge_ptr = torch.empty(B + 1, **i32)
        p.rowptr = torch.empty(N + 1, **i32)
        p.rowptr_t = torch.empty(N + 1, **i32)
        p.col = torch.empty(max(E, 1), **i32)
        p.col_t = torch.empty(max(E, 1), **i32)
        p.dinv = torch.empty(max(N, 1), dtype=torch.float32, device=dev)
        p.status = torch.empty(4, **i32)
        if edge_weight is not None:
            p.eid = torch.empty(max(E, 1), **i32)
            p.eid_t = torch.empty(max(E, 1), **i32)
            p.ew_csr = torch.empty(max(E, 1), dtype=torch.float32, device=dev)
            p.ew_csc = torch.empty(max(E, 1), dtype=torch.float32, device=dev)
            p.dinv_unw = torch.empty(max(N, 1), dtype=torch.float32, device=dev)
        else:
            p.eid = p.eid_t = p.ew_csr = p.ew_csc = p.dinv_unw = None

        def run(which: str):
            m = _lib.HCG_PLAN_BLOCKED if which == "blocked" else _lib.HCG_PLAN_GENERAL
            wsb = lib.hcg_plan_workspace_bytes(N, E, B, m)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            rc = lib.hcg_plan_build(_lib.ptr(edge_index), _lib.ptr(batch), _lib.ptr(edge_weight), N, E, B, fill, m,
                                    _lib.ptr(p.graph_ptr), _lib.ptr(p.edge_ptr), _lib.ptr(p.rowptr), _lib.ptr(p.col),
                                    _lib.ptr(p.eid), _lib.ptr(p.rowptr_t), _lib.ptr(p.col_t), _lib.ptr(p.eid_t),
                                    _lib.ptr(p.dinv), _lib.ptr(p.ew_csr), _lib.ptr(p.ew_csc), _lib.ptr(p.dinv_unw),
                                    _lib.ptr(p.status),
                                    _lib.ptr(ws), wsb, _lib.stream_ptr())
            _lib.check(rc, "hcg_plan_build")
            p.mode = which

        run(want)
        if validate:
            word = int(p.status[0].item())
            if word & _lib.STATUS_EDGE_UNGROUPED and mode == "auto" and not (word & ~_lib.STATUS_EDGE_UNGROUPED):
                run("general")
                word = int(p.status[0].item())
            p.validated = True
            if word:
                raise ValueError("hcatgnet_amd: invalid batch: " + _lib.describe_status(word))
            if p.max_nodes is None and B > 0:   # index bookkeeping for kernel selection (we are synchronising anyway)
                p.max_nodes = int((p.graph_ptr[1:] - p.graph_ptr[:-1]).max().item())
        return p
