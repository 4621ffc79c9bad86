"""Batch-of-graphs data parallelism: one process per GPU, weights replicated, ONE exchange per
training step -- an all-reduce of one flat fp32 gradient buffer (16 641 floats = 66.6 KB at
F_in = D = 64) over RCCL/xGMI (`torch.distributed` backend "nccl" is RCCL on ROCm).

The reference has no distributed code at all (single process, scripts_experiments/train_GNN.py:29);
this wrapper is build-defined.  Graphs never exchange messages across ranks (block-
diagonal adjacency), so nothing else crosses GPUs: no graph partitioning, no halo, no activation
collective.  Gradient semantics, `combine=`:
  "mean"  every rank computes sqrt(MSE) over ITS graphs; the per-rank gradients are averaged (DDP
          convention) = the single-process gradient of the MEAN OF PER-RANK RMSEs;
  "sse"   the fused head leaves the gradients of SSE / 2 and [SSE, count] behind the flat buffer; ONE
          all-reduce(sum) of the n + 2 floats, then one scale 1 / (count * sqrt(SSE / count)) = the
          gradient of sqrt(MSE) over the CONCATENATED batch of all ranks -- exactly what the reference's
          single-device step (utils/utils_model.py:64-65) computes on that batch; ranks may hold different
          numbers of graphs.  Fused trainer only (`attach` / `make_train_step`).
Both are tested against the CPU oracle with two ranks (tests/test_gpu_ddp.py).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


class DataParallelGCN(nn.Module):
    def __init__(self, module: nn.Module, process_group=None, force_collective: bool = False, combine: str = "mean"):
        super().__init__()
        if combine not in ("mean", "sse"):
            raise ValueError(f"combine must be 'mean' or 'sse', got {combine!r}")
        self.combine = combine
        self.module = module
        self.process_group = process_group
        self.force_collective = force_collective   # run the collectives even at world size 1 (tests)
        self._params: List[nn.Parameter] = [p for p in module.parameters() if p.requires_grad]
        self._numel = sum(p.numel() for p in self._params)
        self._flat: Optional[torch.Tensor] = None       # the buffer of the last reduction (own copy or in-place view)
        self._own_flat: Optional[torch.Tensor] = None   # copy-path buffer (never aliases a gradient)
        self._inplace = False
        # ReduceOp.AVG exists for the NCCL/RCCL backend only (gloo: SUM + divide)
        self._avg_ok = dist.is_available() and dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.broadcast_parameters()

    # attributes the reference's loops read from the model (utils/utils_model.py:62-66)
    @property
    def optimizer(self):
        return self.module.optimizer

    @property
    def loss(self):
        return self.module.loss

    @property
    def scheduler(self):
        return self.module.scheduler

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def world_size(self) -> int:
        return dist.get_world_size(self.process_group) if dist.is_available() and dist.is_initialized() else 1

    def broadcast_parameters(self, src: int = 0):
        """Replicate rank-`src` weights (one flat broadcast)."""
        if self.world_size() == 1 and not self.force_collective:
            return
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1) for p in self._params])
            dist.broadcast(flat, src=src, group=self.process_group)
            off = 0
            for p in self._params:
                p.copy_(flat[off:off + p.numel()].view_as(p))
                off += p.numel()

    def flat_gradient(self, grads=None) -> torch.Tensor:
        """`grads`: explicit gradient tensors in parameter order (e.g. the static tensors a captured
        hipGraph writes into); default = each parameter's `.grad`."""
        if grads is None:
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self._params]
        # fast path: the fused backward already wrote every gradient into one flat buffer in parameter order
        g0 = grads[0]
        if g0.is_contiguous() and g0.dtype == torch.float32:
            base, off, ok = g0.data_ptr(), 0, True
            for g in grads:
                if g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() != base + 4 * off:
                    ok = False
                    break
                off += g.numel()
            if ok and off == self._numel:
                try:
                    flat = torch.as_strided(g0, (self._numel,), (1,))      # same storage, no copy
                    self._flat, self._inplace = flat, True
                    return flat
                except RuntimeError:
                    pass                                                    # storage smaller than expected: copy path
        grads = [g.reshape(-1) for g in grads]
        self._inplace = False
        if self._own_flat is None or self._own_flat.device != grads[0].device:
            self._own_flat = torch.empty(self._numel, dtype=torch.float32, device=grads[0].device)
        torch.cat(grads, out=self._own_flat)
        self._flat = self._own_flat
        return self._flat

    def reduce_flat(self, flat: torch.Tensor, average: bool = True) -> torch.Tensor:
        """All-reduce a flat gradient buffer in place (the `grad_sync` hook of `train.FusedTrainStep`: the fused
        backward wrote every gradient into `flat`, the parameters' `.grad` are views of it)."""
        ws = self.world_size()
        if ws > 1 or self.force_collective:
            if average and self._avg_ok:
                dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.process_group)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.process_group)
                if average and ws > 1:
                    flat.div_(ws)
        self._flat, self._inplace = flat, True
        return flat

    def reduce_flat_sum(self, flat_ext: torch.Tensor) -> torch.Tensor:
        """The `grad_sync` hook of the "sse" form: all-reduce(SUM) of [gradients | SSE | count], in place, no division
        (the one scale that follows is computed from the summed SSE and count)."""
        return self.reduce_flat(flat_ext, average=False)

    def attach(self, step):
        """Make a `train.FusedTrainStep` on this module data parallel with this wrapper's `combine` semantics."""
        if step.model is not self.module:
            raise ValueError("the trainer belongs to another module")
        if self.combine == "sse" and not step.rmse:
            raise ValueError("combine='sse' needs rmse=True")
        step.combine = self.combine
        step.grad_sync = self.reduce_flat_sum if self.combine == "sse" else self.reduce_flat
        return step

    def make_train_step(self, **kw):
        from .train import FusedTrainStep
        return self.attach(FusedTrainStep(self.module, **kw))

    def reduce_gradients(self, average: bool = True, grads=None) -> torch.Tensor:
        """All-reduce(sum) the flat gradient, divide by world size, re-attach the views as .grad."""
        flat = self.flat_gradient(grads)
        ws = self.world_size()
        if ws > 1 or self.force_collective:
            if average and self._avg_ok:
                dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.process_group)   # one collective, no div kernel
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.process_group)
                if average and ws > 1:
                    flat.div_(ws)
        if not self._inplace:   # (in-place: the gradients the caller already holds ARE views of `flat`)
            off = 0
            for p in self._params:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        return flat
