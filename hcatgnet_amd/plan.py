"""BatchPlan: the per-batch index structures of the GCN path, built once per batch on the GPU.

Replaces what PyG recomputes inside every `GCNConv.forward` (gcn_norm: self loops, degree,
deg^-1/2; reference call sites model/gcn.py:58,62) and the `batch`-vector scatter indices of the
pools (model/gcn.py:65-66).  Two levels:
  * pointers  graph_ptr / edge_ptr (+ device-side validation) -- ONE launch; all the fused
              small-graph kernels need (they rebuild gcn_norm on chip from the raw edges);
  * CSR       rowptr/col (+ transpose) and dinv for the any-shape kernels, built lazily by
              `ensure_csr()` the first time a layer needs them.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


_SHARED_STATUS = {}


def _shared_status(dev: torch.device) -> torch.Tensor:
    """One once-zeroed status buffer per device, shared by the plans that are built WITHOUT validation
    (trusted batches): their builds skip the per-step memset; flags from any of them accumulate until
    `check_status()` reads and clears the buffer."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    t = _SHARED_STATUS.get(key)
    if t is None:
        t = torch.zeros(4, dtype=torch.int32, device=dev)
        _SHARED_STATUS[key] = t
    return t


class BatchPlan:
    __slots__ = ("N", "E", "B", "mode", "fill", "edge_index", "batch", "edge_weight", "graph_ptr", "edge_ptr",
                 "rowptr", "col", "eid", "rowptr_t", "col_t", "eid_t", "dinv", "ew_csr", "ew_csc", "dinv_unw",
                 "status", "max_nodes", "max_edges", "validated", "has_csr", "shared_status", "want_eid")

    def check_status(self):
        """Synchronising read of the device-side status word; raises on any violation.  (For plans built
        without validation the word is shared per device and cleared by this call.)"""
        word = int(self.status[0].item())
        if self.shared_status and word:
            self.status.zero_()
        self.validated = True
        if word:
            raise ValueError("hcatgnet_amd: invalid batch: " + _lib.describe_status(word))
        return word

    # ------------------------------------------------------------------ device build
    def _run(self, which: str, csr: bool):
        lib = _lib.load()
        dev = self.edge_index.device
        N, E, B = self.N, self.E, self.B
        i32 = dict(dtype=torch.int32, device=dev)
        if csr and self.rowptr is None:
            self.rowptr = torch.empty(N + 1, **i32)
            self.rowptr_t = torch.empty(N + 1, **i32)
            self.col = torch.empty(max(E, 1), **i32)
            self.col_t = torch.empty(max(E, 1), **i32)
            self.dinv = torch.empty(max(N, 1), dtype=torch.float32, device=dev)
            if self.edge_weight is not None or getattr(self, "want_eid", False):
                self.eid = torch.empty(max(E, 1), **i32)
                self.eid_t = torch.empty(max(E, 1), **i32)
            if self.edge_weight is not None:
                self.ew_csr = torch.empty(max(E, 1), dtype=torch.float32, device=dev)
                self.ew_csc = torch.empty(max(E, 1), dtype=torch.float32, device=dev)
                self.dinv_unw = torch.empty(max(N, 1), dtype=torch.float32, device=dev)
        m = _lib.HCG_PLAN_BLOCKED if which == "blocked" else _lib.HCG_PLAN_GENERAL
        if not csr:
            m |= _lib.HCG_PLAN_PTRS_ONLY
        if self.shared_status:
            m |= _lib.HCG_PLAN_KEEP_STATUS
        wsb = lib.hcg_general_workspace_bytes(_lib.HCG_WS_PLAN, N, E, B, m)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        p = _lib.ptr
        rc = lib.hcg_plan_build(p(self.edge_index), p(self.batch), p(self.edge_weight), N, E, B, self.fill, m,
                                p(self.graph_ptr), p(self.edge_ptr), p(self.rowptr), p(self.col), p(self.eid),
                                p(self.rowptr_t), p(self.col_t), p(self.eid_t), p(self.dinv), p(self.ew_csr),
                                p(self.ew_csc), p(self.dinv_unw), p(self.status), p(ws), wsb, _lib.stream_ptr())
        _lib.check(rc, "hcg_plan_build")
        self.mode = which
        self.has_csr = csr

    def rebuild(self):
        """Re-derive this plan from the CURRENT contents of its edge_index / batch tensors, into the same buffers (one
        launch for a pointers-only plan).  For pipelined loaders: the next batch's plan can be built on a forked stream
        -- or a forked branch of a captured hipGraph (`FusedTrainStep.capture(..., prefetch=next_plan.rebuild)`) -- beside
        the current step, and a batch object that carries it (`batch._hcg_plan = plan`) then starts without a plan launch."""
        self._run(self.mode, csr=self.has_csr)
        return self

    def ensure_eid(self):
        """CSR / CSC plus `eid` / `eid_t` (position of every CSR / CSC entry in the caller's edge_index): what the
        explain path needs to carry a per-edge mask into CSR order and its gradient back."""
        if self.eid is None:
            self.want_eid = True
            if self.status is None:
                self.status = _shared_status(self.edge_index.device)
            self.rowptr = None          # re-run the CSR build with the id arrays attached
            self.has_csr = False
            self._run(self.mode, csr=True)
        return self

    def ensure_csr(self):
        """Build rowptr/col/dinv (and the transpose) if this plan so far only holds the pointers."""
        if not self.has_csr:
            if self.status is None:
                self.status = _shared_status(self.edge_index.device)
            self._run(self.mode, csr=True)
        return self

    @staticmethod
    def build(edge_index: torch.Tensor, batch: Optional[torch.Tensor], num_nodes: int,
              num_graphs: Optional[int] = None, edge_weight: Optional[torch.Tensor] = None,
              improved: bool = False, mode: str = "auto", validate: bool = True,
              max_nodes: Optional[int] = None, max_edges: Optional[int] = None, csr: Optional[bool] = None) -> "BatchPlan":
        """mode: 'blocked' (edges grouped by graph, the order PyG-style collation emits),
        'general' (any order; device radix sort) or 'auto' (blocked, re-planned as general when the
        device reports ungrouped edges -- needs `validate=True`, i.e. one 4-byte D2H sync).
        `validate=False` + an explicit mode enqueues the build without any host sync.
        csr: None = lazily (blocked plans start with the pointers only), True = build it now."""
        _lib.load()
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
        p.edge_index, p.batch, p.edge_weight = edge_index, batch, edge_weight
        p.max_nodes, p.max_edges, p.validated, p.has_csr = max_nodes, max_edges, False, False
        p.want_eid = False
        i32 = dict(dtype=torch.int32, device=dev)
        p.graph_ptr = torch.empty(B + 1, **i32)
        p.edge_ptr = torch.empty(B + 1, **i32)
        p.shared_status = not validate
        p.status = _shared_status(dev) if p.shared_status else torch.empty(4, **i32)
        p.rowptr = p.col = p.eid = p.rowptr_t = p.col_t = p.eid_t = None
        p.dinv = p.ew_csr = p.ew_csc = p.dinv_unw = None

        def want_csr(which):
            if csr is not None:
                return bool(csr) or which == "general"
            return which == "general" or edge_weight is not None

        p._run(want, want_csr(want))
        if validate:
            word = int(p.status[0].item())
            if word & _lib.STATUS_EDGE_UNGROUPED and mode == "auto" and not (word & ~_lib.STATUS_EDGE_UNGROUPED):
                p._run("general", True)
                word = int(p.status[0].item())
            p.validated = True
            if word:
                raise ValueError("hcatgnet_amd: invalid batch: " + _lib.describe_status(word))
            if p.max_nodes is None and B > 0:   # index bookkeeping for kernel selection (we are synchronising anyway)
                p.max_nodes = int((p.graph_ptr[1:] - p.graph_ptr[:-1]).max().item())
            if p.max_edges is None and B > 0 and p.mode == "blocked":
                p.max_edges = int((p.edge_ptr[1:] - p.edge_ptr[:-1]).max().item())
        return p
