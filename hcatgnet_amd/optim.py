"""FusedAdam: `torch.optim.Adam` semantics (the reference builds `Adam(self.parameters(), lr, eps=1e-9)`,
model/networks.py:38) with the update done by ONE HIP launch when the gradients sit in one flat buffer
(which the fused backward guarantees), one gather + one launch otherwise.

It IS a `torch.optim.Optimizer` (param_groups, state_dict, zero_grad, lr schedulers -- e.g. the reference's
`ReduceLROnPlateau` -- all work): `lr` is read from `param_groups` at every step.  Parameters and both
moment buffers are re-based onto flat storages the first time `step()` sees them on the GPU; the
`nn.Parameter` objects stay the same (only `.data` is re-pointed), so `state_dict()/load_state_dict()` of
the module are unaffected.  Restrictions (checked): float32, amsgrad / weight_decay / maximize off.
"""
from __future__ import annotations

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._flat = {}      # group index -> dict(param, exp_avg, exp_avg_sq flat tensors)
        self.capturable = False

    # ------------------------------------------------------------------ hipGraph-capturable mode
    def enable_capturable(self):
        """Keep the step count and the learning rate in device memory (hcg_adam_step_dev), so that `step()` can be
        captured into a hipGraph and replayed: from here on the device counter is the authority (`steps_done()`),
        `sync_lr()` pushes a learning rate a scheduler changed.  Needs flat gradients (the fused backward's)."""
        self.capturable = True
        for gi, group in enumerate(self.param_groups):
            fl = self._flat.get(gi)
            if fl is not None:
                self._make_dev_state(fl, group)

    def _make_dev_state(self, fl, group):
        """step_dev = [Adam step count, exchange stamp].  Word 1 counts the steps this optimiser OBJECT has started and is
        carried over every re-base / `load_state_dict` (`_keep_stamp`): the one-shot exchange (xgmi.py) stamps its granules
        with it, and a stamp must never repeat although the Adam step count may go back to a checkpoint's."""
        if "step_dev" not in fl:
            dev = fl["p"].device
            stamps = getattr(self, "_stamps", {})
            fl["step_dev"] = torch.tensor([fl["step"], stamps.get(fl.get("gi", 0), 0)], dtype=torch.int32, device=dev)
            fl["lr_dev"] = torch.tensor([float(group["lr"])], dtype=torch.float32, device=dev)
            fl["lr_host"] = float(group["lr"])

    def sync_lr(self):
        """Push param_groups' learning rates to the device words the captured update reads (host compare only
        when nothing changed)."""
        for gi, group in enumerate(self.param_groups):
            fl = self._flat.get(gi)
            if fl is not None and "lr_dev" in fl and fl["lr_host"] != float(group["lr"]):
                fl["lr_host"] = float(group["lr"])
                fl["lr_dev"].fill_(fl["lr_host"])

    def steps_done(self, gi: int = 0) -> int:
        fl = self._flat.get(gi)
        if fl is None:
            return 0
        if self.capturable and "step_dev" in fl:
            return int(fl["step_dev"][0].item())     # synchronises
        return fl["step"]

    def load_state_dict(self, state_dict):
        """Restored moments / step counts are copied into fresh flat buffers right away (torch's `load_state_dict` may
        alias the tensors of the dict it is given: a source optimizer that keeps stepping must not leak into this one)."""
        for gi in list(self._flat):
            self._keep_stamp(gi)
        super().load_state_dict(state_dict)
        self._flat = {}
        with torch.no_grad():
            for gi, group in enumerate(self.param_groups):
                ps = [p for p in group["params"] if p.requires_grad]
                if ps and all(p.is_cuda for p in ps):
                    fl = self._rebase(gi, group)
                    if self.capturable:
                        self._make_dev_state(fl, group)

    def state_dict(self):
        for gi, group in enumerate(self.param_groups):
            fl = self._flat.get(gi)
            if fl is not None and self.capturable:
                n = self.steps_done(gi)
                fl["step"] = n
                for p in fl["params"]:
                    self.state[p]["step"] = torch.tensor(float(n))
        return super().state_dict()

    def _keep_stamp(self, gi):
        """Remember the exchange stamp of a flat state that is about to be replaced (synchronises; re-bases are rare)."""
        old = self._flat.get(gi)
        if old is not None and "step_dev" in old:
            if not hasattr(self, "_stamps"):
                self._stamps = {}
            self._stamps[gi] = max(self._stamps.get(gi, 0), int(old["step_dev"][1].item()))

    def _rebase(self, gi, group):
        """Move the group's parameters and moments onto flat buffers (in parameter order)."""
        self._keep_stamp(gi)
        ps = [p for p in group["params"] if p.requires_grad]
        dev = ps[0].device
        n = sum(p.numel() for p in ps)
        flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in ps:
                if p.dtype != torch.float32:
                    raise _lib.HcgError("FusedAdam handles float32 parameters only")
                k = p.numel()
                flat_p[off:off + k].copy_(p.reshape(-1))
                st = self.state[p]
                if "exp_avg" in st:            # keep moments restored by load_state_dict
                    flat_m[off:off + k].copy_(st["exp_avg"].reshape(-1))
                    flat_v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                p.data = flat_p[off:off + k].view(p.shape)
                st["exp_avg"] = flat_m[off:off + k].view(p.shape)
                st["exp_avg_sq"] = flat_v[off:off + k].view(p.shape)
                st.setdefault("step", torch.tensor(0.0))
                off += k
        self._flat[gi] = dict(params=ps, p=flat_p, m=flat_m, v=flat_v, n=n, gi=gi, step=int(ps and self.state[ps[0]]["step"]) if ps else 0)
        return self._flat[gi]

    def fused_update_ready(self, flat_grad: torch.Tensor):
        """Device word the head must advance (`step_counter` of hcg_head_fwd_bwd / hcg_fused_forward) if the NEXT update can be
        fused into the step's last launch (`step_with_reduction`), else None -- nothing is launched or changed here."""
        if not self.capturable or len(self.param_groups) != 1:
            return None
        group = self.param_groups[0]
        ps = [p for p in group["params"] if p.requires_grad]
        if not ps or not all(p.is_cuda for p in ps):
            return None
        fl = self._flat.get(0)
        if fl is None or fl["params"] != ps or fl["p"].device != ps[0].device or ps[0].data_ptr() != fl["p"].data_ptr():
            with torch.no_grad():
                fl = self._rebase(0, group)
        grads = [p.grad for p in ps]
        if any(g is None for g in grads) or not self._grads_flat(grads):
            return None
        if grads[0].data_ptr() != flat_grad.data_ptr() or flat_grad.numel() != fl["n"]:
            return None
        self._make_dev_state(fl, group)
        self._ready = (flat_grad.data_ptr(), fl)       # `step_with_reduction` of the same step need not ask again
        return fl["step_dev"]

    def step_with_reduction(self, jobs_addr: int, njobs: int, flat_grad: torch.Tensor, next_plan=None, exchange=None,
                            flat_ext=None, mode: str = "mean", loss_buf=None, loss_mode: int = _lib.HCG_LOSS_RMSE,
                            loss_count: float = 0.0) -> bool:
        """The step's last launch (hcg_step_tail): the backward's slab reductions, the loss with its deferred scale
        (`loss_mode`, `loss_count` = B * C; used when a job carries the head's SSE partials), this optimiser's update, and
        optionally the data-parallel exchange and the NEXT batch's plan.  `jobs_addr` = host address of the hcg_reduce_job
        array whose segments write `flat_grad` (the buffer the parameters' `.grad` are views of, in parameter order).  The
        step word returned by `fused_update_ready` must have been advanced earlier in this step (the head does).  Returns
        False -- nothing launched -- when the preconditions do not hold; the caller then issues reduction and update
        separately.  `next_plan`: a pointers-only blocked `BatchPlan`."""
        ready, self._ready = getattr(self, "_ready", None), None
        if (ready is not None and ready[0] == flat_grad.data_ptr() and self._flat.get(0) is ready[1]
                and self.capturable and len(self.param_groups) == 1):
            # `fused_update_ready(flat_grad)` answered earlier in this very step (the head launch in between advanced the
            # step word): same parameters, same flat buffers -- the walk over them is not repeated
            group, fl = self.param_groups[0], ready[1]
        else:
            if not self.capturable or len(self.param_groups) != 1:
                return False
            group = self.param_groups[0]
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps or not all(p.is_cuda for p in ps):
                return False
            fl = self._flat.get(0)
            if fl is None or fl["params"] != ps or fl["p"].device != ps[0].device or ps[0].data_ptr() != fl["p"].data_ptr():
                with torch.no_grad():
                    fl = self._rebase(0, group)
            grads = [p.grad for p in ps]
            if any(g is None for g in grads) or not self._grads_flat(grads):
                return False
            if grads[0].data_ptr() != flat_grad.data_ptr() or flat_grad.numel() != fl["n"]:
                return False
            self._make_dev_state(fl, group)
        (b1, b2), eps, lr = group["betas"], float(group["eps"]), float(group["lr"])
        if fl["lr_host"] != lr:
            fl["lr_host"] = lr
            fl["lr_dev"].fill_(lr)
        if next_plan is not None:
            np_ = next_plan
            if np_.mode != "blocked" or np_.has_csr or not np_.shared_status:
                raise _lib.HcgError("next_plan must be a pointers-only blocked plan built with validate=False")
        if exchange is not None:      # data parallel: the one-shot xGMI exchange sits between the reduction and the update
            if flat_ext is None or flat_ext.data_ptr() != flat_grad.data_ptr() or flat_ext.numel() != fl["n"] + 2 or loss_buf is None:
                raise _lib.HcgError("one-shot exchange: needs the extended flat buffer [gradients | SSE | count] and the loss buffer")
            exchange.launch(jobs_addr, njobs, flat_ext, fl, b1, b2, eps, mode, loss_buf, next_plan, loss_count=loss_count,
                            loss_mode=loss_mode)
            return True
        _lib.step_tail(jobs_addr, njobs, loss=loss_buf, loss_mode=loss_mode, loss_count=loss_count,
                       adam=dict(grad_flat=flat_grad, param=fl["p"], exp_avg=fl["m"], exp_avg_sq=fl["v"], n=fl["n"],
                                 lr_dev=fl["lr_dev"], step_dev=fl["step_dev"], beta1=b1, beta2=b2, eps=eps),
                       next_plan=next_plan)
        return True

    def step_sse(self, flat_ext: torch.Tensor, loss_buf: torch.Tensor):
        """Data-parallel "sse" form (train.FusedTrainStep combine="sse"): `flat_ext` = [summed SSE/2-gradients | SSE |
        count], the parameters' `.grad` being views of its first n floats.  ONE launch scales the gradients in place to
        those of sqrt(MSE) over all ranks' graphs, stores that loss in `loss_buf[0:2]` and applies the update."""
        if not self.capturable or len(self.param_groups) != 1:
            raise _lib.HcgError("FusedAdam.step_sse needs the capturable mode and one parameter group")
        group = self.param_groups[0]
        ps = [p for p in group["params"] if p.requires_grad]
        _lib.require_gpu(*ps)
        fl = self._flat.get(0)
        if fl is None or fl["params"] != ps or fl["p"].device != ps[0].device or ps[0].data_ptr() != fl["p"].data_ptr():
            with torch.no_grad():
                fl = self._rebase(0, group)
        if flat_ext.numel() != fl["n"] + 2 or flat_ext.dtype != torch.float32 or not flat_ext.is_contiguous():
            raise _lib.HcgError("FusedAdam.step_sse: the flat buffer must hold n gradients + [SSE, count]")
        self._make_dev_state(fl, group)
        (b1, b2), eps, lr = group["betas"], float(group["eps"]), float(group["lr"])
        if fl["lr_host"] != lr:
            fl["lr_host"] = lr
            fl["lr_dev"].fill_(lr)
        lib = _lib.load()
        _lib.check(lib.hcg_adam_step_dev_sse(fl["p"].data_ptr(), flat_ext.data_ptr(), fl["m"].data_ptr(), fl["v"].data_ptr(),
                                             fl["n"], fl["lr_dev"].data_ptr(), b1, b2, eps, fl["step_dev"].data_ptr(),
                                             loss_buf.data_ptr(), _lib.stream_ptr()), "hcg_adam_step_dev_sse")

    @staticmethod
    def _grads_flat(grads) -> bool:
        g0, off = grads[0], 0
        if not (g0.is_contiguous() and g0.dtype == torch.float32):
            return False
        base = g0.data_ptr()
        for g in grads:
            if g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() != base + 4 * off:
                return False
            off += g.numel()
        return True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                continue
            _lib.require_gpu(*ps)
            fl = self._flat.get(gi)
            if fl is None or fl["params"] != ps or fl["p"].device != ps[0].device or ps[0].data_ptr() != fl["p"].data_ptr():
                fl = self._rebase(gi, group)
            grads = [p.grad for p in ps]
            if any(g is None for g in grads):
                raise _lib.HcgError("FusedAdam.step(): a parameter has no gradient")
            (b1, b2), eps, lr = group["betas"], float(group["eps"]), float(group["lr"])
            stream = _lib.stream_ptr()
            if self.capturable:
                self._make_dev_state(fl, group)
                gflat = grads[0]
                if not self._grads_flat(grads):      # e.g. the autograd path's per-tensor gradients: one gather first
                    if any(g.dtype != torch.float32 for g in grads):
                        raise _lib.HcgError("FusedAdam handles float32 gradients only")
                    gflat = torch.cat([g.reshape(-1) for g in grads])
                if fl["lr_host"] != lr:
                    fl["lr_host"] = lr
                    fl["lr_dev"].fill_(lr)
                _lib.check(lib.hcg_adam_step_dev(fl["p"].data_ptr(), gflat.data_ptr(), fl["m"].data_ptr(),
                                                 fl["v"].data_ptr(), fl["n"], fl["lr_dev"].data_ptr(), b1, b2, eps,
                                                 fl["step_dev"].data_ptr(), stream), "hcg_adam_step_dev")
                continue
            fl["step"] += 1
            step = fl["step"]
            # one launch when the gradients are one flat buffer in parameter order (fused backward / DP wrapper)
            g0 = grads[0]
            flat_ok = self._grads_flat(grads)
            if flat_ok:
                _lib.check(lib.hcg_adam_step(fl["p"].data_ptr(), g0.data_ptr(), fl["m"].data_ptr(), fl["v"].data_ptr(),
                                             fl["n"], lr, b1, b2, eps, step, stream), "hcg_adam_step")
            else:
                # per-tensor gradients (autograd path): ONE gather into a flat buffer, then the same single launch
                if any(g.dtype != torch.float32 for g in grads):
                    raise _lib.HcgError("FusedAdam handles float32 gradients only")
                gflat = torch.cat([g.reshape(-1) for g in grads])
                _lib.check(lib.hcg_adam_step(fl["p"].data_ptr(), gflat.data_ptr(), fl["m"].data_ptr(), fl["v"].data_ptr(),
                                             fl["n"], lr, b1, b2, eps, step, stream), "hcg_adam_step")
            for p in ps:
                self.state[p]["step"] = torch.tensor(float(step))
        return loss
