"""One-shot gradient exchange over xGMI for the data-parallel step.

RCCL's all-reduce of the 66 KB flat gradient is latency-bound (a ring is 2 x 7 hops on a point-to-point fabric) and sits,
with its own launch and the separate update launch, on the critical path of a ~0.11 ms step.  On MI355X every GPU has a
direct link to each of its 7 peers, so the exchange can be ONE hop: each rank writes its partial gradient straight into
every peer's inbox and sums the 8 contributions it received -- fused INTO the slab reduction + Adam launch
(`hcg_step_tail` with an inbox, csrc/reduce.hip), so the data-parallel step keeps the launch count of the single-GPU step
and is captured as one hipGraph.

    xchg = OneShotExchange(n_params)            # after torch.distributed is up: inboxes allocated, mapped across processes
    ok = xchg.self_test()                       # a few exchanges of known data against torch.distributed's all-reduce
    step = dp.make_train_step(); xchg.attach(step)   # only if ok: else the RCCL path stays

The reference has no distributed code (scripts_experiments/train_GNN.py:29); this is build-defined.  The kernel's polls
are bounded: a peer that never writes gives NaN gradients and `check()` raises -- never a hang.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch
import torch.distributed as dist

from . import _lib


def inbox_slot(parity: int, world: int, writer: int, n_ext: int, elem: int) -> int:
    """Granule index, inside ANY rank's inbox, of element `elem` written by rank `writer` in a step of parity `parity`
    (csrc/reduce.hip: xchg_publish / xchg_gather use the same formula; tests/test_xchg_model.py walks it for world 8)."""
    return (parity * world + writer) * n_ext + elem


class OneShotExchange:
    def __init__(self, n: int, process_group=None, device: Optional[torch.device] = None):
        """Collective: every rank of the group constructs one.  `self.ok` is the ALL-RANK verdict of the set-up (allocation,
        IPC export, mapping of every peer's inbox): whatever fails on whichever rank, every rank issues the same
        collectives and ends with the same `ok` -- a rank must never be left waiting inside one."""
        if not (dist.is_available() and dist.is_initialized()):
            raise _lib.HcgError("OneShotExchange needs an initialised torch.distributed process group")
        lib = _lib.load()
        self.lib, self.group, self.n = lib, process_group, int(n)
        self.rank, self.world = dist.get_rank(process_group), dist.get_world_size(process_group)
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.inbox, self.bytes, self._opened = None, 0, []
        self.peers = (ctypes.c_void_p * max(self.world, 1))()
        self.err = torch.zeros(4, dtype=torch.int32, device=self.device)
        good, handle = self.world <= _lib.HCG_XCHG_MAX_WORLD, b""
        if good:
            try:
                self.bytes = int(lib.hcg_xchg_inbox_bytes(self.n, self.world))
                ptr = ctypes.c_void_p()
                _lib.check(lib.hcg_xchg_alloc(self.bytes, ctypes.byref(ptr)), "hcg_xchg_alloc")
                self.inbox = ptr.value
                hbuf = ctypes.create_string_buffer(64)
                _lib.check(lib.hcg_xchg_ipc_export(self.inbox, hbuf), "hcg_xchg_ipc_export")
                handle = hbuf.raw
            except Exception:                                       # noqa: BLE001
                good = False
        handles: List[Optional[bytes]] = [None] * self.world
        dist.all_gather_object(handles, handle if good else b"", group=process_group)
        if good and all(h is not None and len(h) == 64 for h in handles):
            try:
                for q in range(self.world):
                    if q == self.rank:
                        self.peers[q] = self.inbox
                        continue
                    pp = ctypes.c_void_p()
                    _lib.check(lib.hcg_xchg_ipc_open(handles[q], ctypes.byref(pp)), "hcg_xchg_ipc_open")
                    self.peers[q] = pp.value
                    self._opened.append(pp.value)
            except Exception:                                       # noqa: BLE001
                good = False
        else:
            good = False
        verdict = torch.tensor([1 if good else 0], device=self.device, dtype=torch.int32)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=process_group)   # (also: every inbox is mapped everywhere from here on)
        self.ok = bool(int(verdict.item()))

    # ------------------------------------------------------------------ the fused launch (called by FusedAdam)
    def tail_args(self, mode: str) -> dict:
        """The `xchg` part of `_lib.step_tail`: the exchange rides between the slab reduction and the update."""
        return dict(inbox=self.inbox, peers_host=ctypes.addressof(self.peers), rank=self.rank, world=self.world,
                    mode=_lib.HCG_XCHG_SSE if mode == "sse" else _lib.HCG_XCHG_MEAN, err=self.err)

    def launch(self, jobs_addr, njobs, flat_ext, fl, b1, b2, eps, mode, loss_buf, next_plan=None, loss_count=0.0,
               loss_mode=_lib.HCG_LOSS_SSE):
        """ONE launch: slab reductions, (loss scale,) exchange, Adam update, optionally the next batch's plan."""
        _lib.step_tail(jobs_addr, njobs, loss=loss_buf, loss_mode=loss_mode, loss_count=loss_count,
                       adam=dict(grad_flat=flat_ext, param=fl["p"], exp_avg=fl["m"], exp_avg_sq=fl["v"], n=fl["n"],
                                 lr_dev=fl["lr_dev"], step_dev=fl["step_dev"], beta1=b1, beta2=b2, eps=eps),
                       next_plan=next_plan, xchg=self.tail_args(mode))

    def attach(self, step):
        """Route a `train.FusedTrainStep` (already made data parallel by `DataParallelGCN.attach`) through this exchange:
        its `grad_sync` hook is dropped -- the exchange happens inside the step's last launch."""
        if not self.ok:
            raise _lib.HcgError("this exchange failed its set-up or self test: keep the RCCL path")
        n = sum(q.numel() for q in step.model.parameters() if q.requires_grad)
        if n != self.n:
            raise ValueError(f"exchange built for {self.n} gradients, the model has {n}")
        if not (step.optimizer_step and hasattr(step.model.optimizer, "step_with_reduction")):
            raise ValueError("the one-shot exchange rides in the fused update: it needs optimizer_step=True and FusedAdam")
        if step.grad_sync is None:
            raise ValueError("attach the trainer to a DataParallelGCN first (`dp.attach(step)` / `dp.make_train_step()`): its "
                             "collective stays behind as the fallback of the steps the fused update cannot carry")
        step.exchange = self
        # steps whose last launch cannot carry the exchange (a readout the one-launch head does not cover, an optimiser whose
        # state is not one flat group) must still exchange: they take the collective this hook performs (such a
        # step used to run with NO exchange at all and the replicas drifted apart silently)
        step.exchange_fallback_sync = step.grad_sync
        step.grad_sync = None
        return step

    def check(self):
        """Synchronising read of the error word: raises if a poll timed out (that step's gradients are NaN)."""
        if int(self.err[0].item()) & _lib.HCG_XCHG_ERR_TIMEOUT:
            self.err.zero_()
            raise _lib.HcgError("one-shot exchange: a peer's gradients never arrived (poll timed out)")

    # ------------------------------------------------------------------ self test against torch.distributed
    def self_test(self, rounds: int = 3) -> bool:
        """A few exchanges of rank-dependent data through a throw-away parameter set, checked against the process group's
        own all-reduce on EVERY rank (the verdict is all-reduced too: all ranks agree).  False = do not use this exchange."""
        from .optim import FusedAdam
        ok = self.ok
        dev, n = self.device, self.n
        try:
            p = torch.nn.Parameter(torch.zeros(n, device=dev))
            opt = FusedAdam([p], lr=0.0)
            opt.enable_capturable()
            flat_ext = torch.zeros(n + 2, device=dev)
            p.grad = flat_ext[:n]
            if opt.fused_update_ready(flat_ext[:n]) is None:
                raise _lib.HcgError("self test: the fused update is not available")
            fl = opt._flat[0]
            loss = torch.zeros(2, device=dev)
            # one reduction job over ONE slab = the data itself
            jb = _lib.job_bytes()
            job = ctypes.create_string_buffer(jb)
            slab = torch.empty(n, device=dev)

            assert ctypes.sizeof(_lib.ReduceJob) == jb
            j = _lib.ReduceJob.from_buffer(job)
            j.slabs, j.sse_part, j.nslabs, j.slab_floats, j.nseg = slab.data_ptr(), None, 1, n, 1
            j.seg[0] = _lib.ReduceSeg(0, n, n, n, flat_ext.data_ptr())
            g = torch.Generator(device="cpu").manual_seed(1234 + self.rank)
        except Exception:                                           # noqa: BLE001  (any failure = fall back to RCCL)
            ok = False
        # the rounds: every rank issues the SAME collectives whatever fails locally (a rank that raised must not leave
        # the others waiting inside an all-reduce)
        for r in range(rounds):
            ref = torch.zeros(self.n + 2, device=self.device)
            if ok:
                try:
                    data = torch.randn(n, generator=g).to(dev)
                    slab.copy_(data)
                    flat_ext[n] = float(self.rank + 1) * (r + 1)        # "SSE"
                    flat_ext[n + 1] = 10.0 * (self.rank + 1)            # "count"
                    fl["step_dev"] += 1                                 # what the head kernel does in a real step (count + stamp)
                    ref = torch.cat([data, flat_ext[n:].clone()])
                    self.launch(ctypes.addressof(job), 1, flat_ext, fl, 0.9, 0.999, 1e-9, "sse", loss)
                    torch.cuda.synchronize()
                except Exception:                                       # noqa: BLE001
                    ok = False
            dist.all_reduce(ref, group=self.group)
            if ok:
                sse, cnt = float(ref[n]), float(ref[n + 1])
                want = ref[:n] / (cnt * (sse / cnt) ** 0.5)
                got = flat_ext[:n]
                err = float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))
                if (not (err <= 1e-5) or int(self.err[0].item()) != 0
                        or not (abs(float(loss[0]) - (sse / cnt) ** 0.5) <= 1e-5 * (sse / cnt) ** 0.5)):
                    ok = False
        verdict = torch.tensor([1 if ok else 0], device=self.device, dtype=torch.int32)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=self.group)
        self.err.zero_()
        # the throw-away optimiser's step stamps are gone: the inbox must be clean for the real one
        self.reset()
        self.ok = bool(int(verdict.item()))
        return self.ok

    def reset(self):
        """Zero this rank's inbox (between optimiser OBJECTS: a stamp counts the steps one FusedAdam has started -- `step_dev[1]`,
        carried over its re-bases and `load_state_dict`, so reloading a checkpoint mid-run needs no reset).  Collective: all ranks call it."""
        torch.cuda.synchronize()
        dist.barrier(group=self.group)
        if self.inbox:
            self.lib.hcg_xchg_zero(self.inbox, self.bytes)
        dist.barrier(group=self.group)

    def close(self):
        for pp in self._opened:
            self.lib.hcg_xchg_ipc_close(pp)
        self._opened = []
        if self.inbox:
            self.lib.hcg_xchg_free(self.inbox)
            self.inbox = None


# ---------------------------------------------------------------------------------------------------------------------
# the probe: both data-parallel graph forms tried once in a SACRIFICIAL process group (`python -m hcatgnet_amd.xgmi`, one per rank)
# ---------------------------------------------------------------------------------------------------------------------
def probe_main(argv=None) -> int:
    """What a caller runs in a child process per rank BEFORE its own process touches the GPU (`bench.py: isolated_probe`).
    Phase 1, the one-shot exchange: process group, set-up, self test and a free-running soak of real steps.  Phase 2, the
    RCCL form with the collective recorded into the step's hipGraph (`FusedTrainStep.capture_exchange`): capture, 32
    replays, then a two-step `StepWindow` replayed 8 times; finite loss and bitwise-equal weights on every rank.  Verdicts go to the JSON file named by HCG_PROBE_OUT as
    soon as a phase ends ({"oneshot": bool, "captured": bool}); exit code 0 = both passed.  What a `try` cannot catch in the
    caller's own process -- a GPU memory fault on a peer mapping aborts the process, a wedged collective never returns --
    ends this child instead, and the caller keeps the plain RCCL form.  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env."""
    import argparse
    import json
    import os
    ap = argparse.ArgumentParser(prog="python -m hcatgnet_amd.xgmi")
    ap.add_argument("--soak-steps", type=int, default=64)
    ap.add_argument("--graphs", type=int, default=512)
    ap.add_argument("--combine", default="sse", choices=("sse", "mean"))
    ap.add_argument("--one-device", action="store_true", help="every rank on device 0 over gloo (one-GPU rehearsal)")
    a = ap.parse_args(argv)
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    out_path, verdicts = os.environ.get("HCG_PROBE_OUT"), {}

    def record(key, ok):
        verdicts[key] = bool(ok)
        print(f"[xgmi probe] rank {rank}: {key} {'PASS' if ok else 'FAIL'}", flush=True)
        if out_path:
            with open(out_path + ".tmp", "w") as f:
                json.dump(verdicts, f)
            os.replace(out_path + ".tmp", out_path)

    if a.one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if a.one_device:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)
    world = dist.get_world_size()
    import hcatgnet_amd as H
    from . import synth
    from .ddp import DataParallelGCN
    cfg = synth.CONFIGS["C3"]
    opt = H.default_options(embedding_dim=cfg["hidden"])
    sb = synth.make_config("C3", rank=rank, num_graphs=a.graphs)
    x, ei, bvec, y = sb.x.to(dev), sb.edge_index.to(dev), sb.batch.to(dev), sb.y.to(dev)
    fresh = lambda: H.Batch(x, ei, bvec, sb.num_graphs, y=y, max_nodes=sb.max_nodes, max_edges=sb.max_edges,
                            edges_grouped=True, n_small=sb.n_small)

    def all_agree(good):
        v = torch.tensor([1 if good else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        return bool(int(v.item()))

    # ---- phase 1: the one-shot exchange
    model = H.make_network("GCN", opt, cfg["feat"]).to(dev)
    dp = DataParallelGCN(model, combine=a.combine)
    xchg = OneShotExchange(sum(p.numel() for p in model.parameters()))
    ok = xchg.ok and xchg.self_test()
    if ok and a.soak_steps > 0:
        step = xchg.attach(dp.make_train_step())
        last = None
        for _ in range(a.soak_steps):
            last = step(fresh())
        torch.cuda.synchronize()
        ok = all_agree(int(xchg.err[0].item()) == 0 and bool(torch.isfinite(last).item()))
    record("oneshot", ok)
    xchg.close()

    # ---- phase 2: the RCCL collective recorded into the step's graph (gloo collectives run on the host: nothing to record)
    cap = False
    if not a.one_device:
        model2 = H.make_network("GCN", opt, cfg["feat"]).to(dev)
        dp2 = DataParallelGCN(model2, combine=a.combine, force_collective=world == 1)
        st = dp2.make_train_step()
        st.capture_exchange = True
        good = True
        try:
            st.capture(fresh)
            last = None
            for _ in range(32):
                last = st.replay()
            torch.cuda.synchronize()
            good = bool(torch.isfinite(last).item())
            # ... and several steps with their collectives in ONE graph (what bench.py replays: train.StepWindow)
            from .train import StepWindow
            st2 = dp2.make_train_step()
            st2.capture_exchange = True
            win = StepWindow([st, st2], [fresh, fresh])
            for _ in range(8):
                last = win.replay()[-1]
            torch.cuda.synchronize()
            good = good and bool(torch.isfinite(last).item())
        except Exception as exc:                       # noqa: BLE001  (every rank still issues the collectives below)
            print(f"[xgmi probe] rank {rank}: captured form raised {type(exc).__name__}: {exc}", flush=True)
            good = False
        chk = torch.cat([p.detach().reshape(-1) for p in model2.parameters()]).double().sum().reshape(1)
        hi, lo = chk.clone(), chk.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        cap = all_agree(good and bool((hi == lo).item()) and bool(torch.isfinite(chk).item()))
    record("captured", cap)
    dist.destroy_process_group()
    return 0 if (ok and (cap or a.one_device)) else 3


if __name__ == "__main__":
    raise SystemExit(probe_main())
