"""How a training loop over resident batches is LAUNCHED, and -- data parallel -- which exchange form it uses.

`Rotation`: NB batches that live in HBM, visited round-robin, each with its own `FusedTrainStep` (= its own step buffers),
every step deriving the NEXT batch's plan inside its last launch.  Three launch forms of the same launches:
  "eager"   the host issues every launch (4 per step on the small-graph tiles)
  "graph"   one hipGraph per step
  "window"  whole rotations (NB consecutive steps) as ONE hipGraph launch + per-step graphs for the remainder
`Rotation.pick` measures them on the machine it runs on and keeps the fastest (on a quiet host eager launches run ahead of
the GPU; on a jittery one the graphs win; a window saves the ~3.7 us bubble between two graph launches).

Data parallel (`set_exchange_form`): "rccl" = all-reduce of the flat gradient between the captured backward and the update
launch; "rccl-captured" = the collective and the update recorded INTO the step's graph; "oneshot" = `xgmi.OneShotExchange`
inside the step's last launch.  All ranks must call these functions with the same arguments (they issue collectives).

The reference has no counterpart (one eager PyTorch loop on one device, utils/utils_model.py:55-70,
scripts_experiments/train_GNN.py:29): this is build-defined launch plumbing, moved out of bench.py.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

from . import _lib
from .train import FusedTrainStep, StepWindow

EXCHANGE_FORMS = ("rccl", "rccl-captured", "oneshot")


def all_ranks_agree(ok: bool, device) -> bool:
    """MIN over the ranks of a local verdict (one collective; identity without a process group): whatever fails on one
    rank is dropped on every rank, so that the ranks keep issuing the same collectives."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bool(ok)
    v = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
    dist.all_reduce(v, op=dist.ReduceOp.MIN)
    return bool(int(v.item()))


class Rotation:
    def __init__(self, trainers: List[FusedTrainStep], fresh: List[Callable], planned: List[Callable], plans: list,
                 plan_overlap: str = "fused"):
        """`fresh[i]()` -> a Batch of resident batch i without a plan (its plan launch runs in front of the forward);
        `planned[i]()` -> the same batch carrying the persistent `plans[i]`; `plan_overlap`: "fused" (the previous step's
        last launch derives plans[i]), "fork" (a forked branch of the previous step's graph), "none"."""
        if not (len(trainers) == len(fresh) == len(planned) == len(plans)) or not trainers:
            raise ValueError("Rotation: one trainer, one fresh / planned batch factory and one plan per resident batch")
        self.trainers, self.fresh, self.planned, self.plans = trainers, fresh, planned, plans
        self.NB, self.plan_overlap = len(trainers), plan_overlap
        self.graphs = False
        self.window: Optional[StepWindow] = None
        self.tail_window: Optional[StepWindow] = None
        self.tail_n = 0
        self.errors = {}

    # ---- forms
    def eager(self, i: int):
        return self.trainers[i](self.fresh[i]())

    def capture(self) -> bool:
        """One hipGraph per step.  -> success on EVERY rank (a failure anywhere drops the graphs everywhere)."""
        self.drop_graphs()
        err = None
        try:
            for i, tr in enumerate(self.trainers):
                nxt = self.plans[(i + 1) % self.NB]
                if self.plan_overlap == "none":
                    tr.capture(self.fresh[i])
                elif self.plan_overlap == "fork":
                    tr.capture(self.planned[i], prefetch=nxt.rebuild)
                else:
                    tr.capture(self.planned[i], next_plan=nxt)
        except Exception as exc:      # noqa: BLE001  (reported, never hidden: the eager form stands)
            err = f"{type(exc).__name__}: {exc}"
        self.graphs = all_ranks_agree(err is None, self._device())
        if not self.graphs:
            self.errors["graph"] = err or "capture failed on another rank"
            for tr in self.trainers:
                tr._graph = None
        return self.graphs

    def build_windows(self, k_steps: int) -> bool:
        """Whole rotations as one graph (+ one shorter window for the remainder K mod NB, issued first: a short graph reaches
        the GPU sooner after a synchronize and hides the launch of the long one behind it)."""
        self.window = self.tail_window = None
        self.tail_n = 0
        ok_local = self.graphs and self.plan_overlap == "fused" and self.NB > 1 and all(
            t.grad_sync is None or t.capture_exchange for t in self.trainers)
        err = None
        if ok_local:
            try:
                self.window = StepWindow(self.trainers, self.planned)
                tail_n = k_steps % self.NB
                if tail_n >= 2:
                    self.tail_window = StepWindow(self.trainers[self.NB - tail_n:], self.planned[self.NB - tail_n:])
                    self.tail_n = tail_n
            except Exception as exc:  # noqa: BLE001
                err = f"{type(exc).__name__}: {exc}"
                ok_local = False
        ok = all_ranks_agree(ok_local, self._device())
        if not ok:
            if err or ok_local:
                self.errors["window"] = err or "window capture failed on another rank"
            self.window = self.tail_window = None
            self.tail_n = 0
        return ok

    def drop_graphs(self):
        self.graphs, self.window, self.tail_window, self.tail_n = False, None, None, 0
        for tr in self.trainers:
            tr._graph = None

    def _device(self):
        return self.plans[0].graph_ptr.device

    # ---- exactly k steps from index `start`
    def run(self, form: str, start: int, k: int):
        NB = self.NB
        if form == "eager":
            for j in range(k):
                self.eager((start + j) % NB)
        elif form == "graph":
            for j in range(k):
                self.trainers[(start + j) % NB].replay()
        elif form == "window":
            j = 0
            while j < k:
                i = (start + j) % NB
                if i == 0 and k - j >= NB:
                    self.window.replay()
                    j += NB
                elif self.tail_window is not None and i == NB - self.tail_n and k - j >= self.tail_n:
                    self.tail_window.replay()
                    j += self.tail_n
                else:
                    self.trainers[i].replay()
                    j += 1
        else:
            raise ValueError(form)

    def forms(self):
        return ["eager"] + (["graph"] if self.graphs else []) + (["window"] if self.window is not None else [])

    def pick(self, timed: Callable, k: int):
        """`timed(k, runner)` -> seconds (collective-consistent: MAX over ranks).  Measures every available form over k steps
        (k rounded to whole rotations) and returns (fastest form, {form: ms per step})."""
        kp = max(k, 4 * self.NB) // self.NB * self.NB
        ms = {}
        for form in self.forms():
            timed(min(kp, 2 * self.NB), lambda s, n, f=form: self.run(f, s, n))          # warm
            ms[form] = timed(kp, lambda s, n, f=form: self.run(f, s, n)) / kp * 1e3
        best = min(ms, key=ms.get)
        return best, ms

    def losses_finite(self) -> bool:
        """Synchronising: every trainer's last loss is finite (a timed-out exchange leaves NaN)."""
        ok = True
        for tr in self.trainers:
            cap = tr._bufs.get("cap")
            if cap is not None:
                ok = ok and bool(torch.isfinite(cap["loss"]).all().item())
        return ok


def set_exchange_form(trainers: List[FusedTrainStep], dp, form: str, xchg=None):
    """Route the data-parallel trainers (already attached to `dp`) through one exchange form.  Captured graphs are dropped:
    capture again afterwards."""
    if form not in EXCHANGE_FORMS:
        raise ValueError(f"exchange form {form!r} not in {EXCHANGE_FORMS}")
    for tr in trainers:
        dp.attach(tr)                       # (re-)sets grad_sync = the collective, combine
        tr.exchange = None
        tr.exchange_fallback_sync = None
        tr.capture_exchange = form == "rccl-captured"
        tr._graph = None
        if form == "oneshot":
            if xchg is None:
                raise _lib.HcgError("exchange form 'oneshot' needs a OneShotExchange that passed its self test")
            xchg.attach(tr)
