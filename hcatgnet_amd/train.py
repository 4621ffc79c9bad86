"""The reference's per-batch loops on MI355X (reference utils/utils_model.py:55-111).

`train_network` / `eval_network` / `predict_network` keep the reference's names, arguments and return
values; what changes is how a step is issued:

  reference step (utils/utils_model.py:60-68)          here (`FusedTrainStep`)
  -------------------------------------------          ---------------------------------------------------
  optimizer.zero_grad()                                 -- (every gradient is overwritten, never accumulated)
  out = model(batch)                                    hcg_fused_stack2_fwd / hcg_mid_layer_fwd x n_conv   (+ plan: 1)
  loss = sqrt(MSELoss(out, y.unsqueeze(1)))             hcg_head_fwd_bwd                1 launch: readout fwd,
  loss.backward()                                           loss, readout bwd (grid barrier inside)
                                                        hcg_fused_layer_bwd / hcg_mid_layer_bwd x n_conv
                                                        hcg_reduce_slabs                1 launch -> ONE flat gradient
  [data parallel]                                       RCCL all-reduce of that buffer, in place
  optimizer.step()                                      hcg_adam_step(_dev)             1 launch
  loss.item()                                           -- (the loss stays on the device; ONE sync per epoch)

No autograd graph is built: the step is a fixed sequence of C-ABI calls, which is also what makes it capturable
into a hipGraph (`FusedTrainStep.capture`).  Graphs up to 32 nodes run through the small-graph tiles (csrc/fused.hip),
graphs up to 192 nodes -- the reference's own reaction graphs -- through the one-graph-per-workgroup kernels
(csrc/mid.hip).  Models / batches neither covers (widths other than 64, explicit edge weights, larger graphs) take
the autograd path with the same arithmetic contract.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch
import torch.distributed

from . import _lib
from . import functional as HF

class FusedTrainStep:
    """One training step of the reference's loop as 6 enqueued launches (small-graph tiles; n_conv + 4 in general), no autograd, no host sync.

        step = FusedTrainStep(model)            # model: hcatgnet_amd.GCN on the GPU
        loss = step(batch)                      # 0-d device tensor: sqrt(MSE) of this batch, weights already updated

    `rmse=True` is the reference's `torch.sqrt(model.loss(...))`; `optimizer_step=False` stops after the backward
    (gradients in `model.parameters()[i].grad`, views of one flat buffer); `grad_sync` is called with the flat gradient
    between backward and optimiser (data parallel: `DataParallelGCN.attach(step)` sets it).

    `combine` (data parallel, needs rmse): "mean" = every rank's own sqrt(MSE), gradients averaged (DDP convention);
    "sse" = the head leaves the gradients of SSE / 2 plus [SSE, count] behind the flat buffer, `grad_sync` SUMS all of it
    over the ranks and ONE scale 1 / (count * sqrt(SSE / count)) gives the gradient of sqrt(MSE) over the concatenated
    batch of all ranks -- what the reference's step computes on one device (utils/utils_model.py:64-65); the loss returned
    is then that global loss.

    Every instance owns its exchange words for the head kernel's grid-wide sum (`check_health()` reports a timed-out
    exchange): two trainers may run on two streams at once.
    """

    # development switches for A/B measurements (tools/ab_env.sh sets them from the environment; never set in product
    # code): POOLBITS = False stores the pooled layer's activations as the plain forms do, PREMASK = False leaves every
    # activation derivative to the layer that owns it
    POOLBITS = True
    PREMASK = True
    OVERLAP_GROUPS = True          # captured size-grouped steps: the two kernel families as two branches of the hipGraph

    def __init__(self, model, rmse: bool = True, optimizer_step: bool = True, grad_sync=None, combine: str = "mean"):
        if combine not in ("mean", "sse"):
            raise ValueError(f"combine must be 'mean' or 'sse', got {combine!r}")
        if combine == "sse" and not rmse:
            raise ValueError("combine='sse' reproduces sqrt(MSE) over the concatenated batch: it needs rmse=True")
        self.model, self.rmse, self.optimizer_step, self.grad_sync = model, rmse, optimizer_step, grad_sync
        self.combine = combine
        if optimizer_step and hasattr(model.optimizer, "enable_capturable"):
            model.optimizer.enable_capturable()     # step count / lr in device memory: same launches eager and captured
        self._bufs = {}
        self._graph = None
        self._sync = {}
        self._capturing_split = False
        self._side_streams = {}
        # pipelined loaders: a pointers-only `BatchPlan` of the NEXT batch (built once with validate=False); the step's last
        # launch (slab reduction + Adam) re-derives its graph_ptr / edge_ptr from the tensors' current contents, so the next
        # step -- on a batch object that carries that plan (`batch._hcg_plan = plan`) -- starts without a plan launch
        self.next_plan = None
        # data parallel: a `xgmi.OneShotExchange` (set by its `attach`): the gradient exchange then happens INSIDE the slab
        # reduction + Adam launch instead of as an RCCL collective between two launches
        self.exchange = None
        self._last_carried = False              # the last step's update launch carried reduction + (exchange) + Adam
        self.exchange_fallback_sync = None      # the collective hook `OneShotExchange.attach` took out of `grad_sync`
        # one-device rehearsals only (two ranks sharing a GPU): called right before the launch that carries the exchange.
        # A rank's polling launch fills every CU, and the OTHER process's conv kernels (488 of a SIMD's 512 VGPRs per
        # workgroup) cannot be placed beside it, so on one device the ranks must meet (synchronize + barrier) before that
        # launch; on one GPU per rank nothing of another process ever runs on the device and this stays None
        self.pre_exchange_hook = None
        # data parallel, RCCL form: True = `capture()` / `StepWindow` record the collective and the update launch too (RCCL
        # collectives survive hipGraph capture: tools/exp_rccl_capture.py), so the whole data-parallel step is ONE graph.
        # Default False: the graph ends after the slab reduction, `replay()` issues collective + update eagerly behind it
        self.capture_exchange = False
        self._pcache = None

    def _trainable(self):
        """The model's parameters in `model.parameters()` order, from a cache: the module walk of `nn.Module.parameters()`
        costs ~20 us and a step needs the list four times (a quarter of the host time of a step at the reference's batch
        size 40).  The cache is checked on every use -- each (module, name) slot still holds the same Parameter object and
        the model's direct children are the same modules -- and rebuilt when not."""
        model, c = self.model, self._pcache
        key = tuple(map(id, model._modules.values()))
        if c is None or c[0] != key or any(m._parameters.get(n) is not q for m, n, q in c[1]):
            entries, seen = [], set()
            for m in model.modules():
                for n, q in m._parameters.items():
                    if q is not None and id(q) not in seen:
                        seen.add(id(q))
                        entries.append((m, n, q))
            c = self._pcache = (key, entries, [q for _, _, q in entries])
        return c[2]

    def reason(self, batch=None) -> Optional[str]:
        """`unsupported_reason(self.model, batch)` with the cached parameter list."""
        return self.unsupported_reason(self.model, batch, self._trainable())

    def _sync_words(self, dev: torch.device) -> torch.Tensor:
        """The HCG_HEAD_SYNC_WORDS exchange words of hcg_head_fwd_bwd: once-zeroed, owned by THIS trainer (launches that
        share a set must be stream-ordered; a trainer issues its launches on one stream at a time)."""
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        t = self._sync.get(key)
        if t is None:
            t = self._sync[key] = torch.zeros(_lib.HCG_HEAD_SYNC_WORDS, dtype=torch.int32, device=dev)
        return t

    def check_health(self):
        """Synchronising read of the head kernel's error word: raises if a grid-wide exchange timed out (the losses of
        that step are NaN).  The words are re-zeroed so the trainer can be used again."""
        for t in self._sync.values():
            if int(t[1].item()) & _lib.HCG_HEAD_ERR_TIMEOUT:
                t.zero_()
                raise _lib.HcgError("hcg_head_fwd_bwd: the grid-wide exchange timed out (workgroups not co-resident, or "
                                    "launches of two streams sharing one set of sync words); losses of that step are NaN")

    # ------------------------------------------------------------------ support check (host only)
    @staticmethod
    def unsupported_reason(model, batch=None, _params=None) -> Optional[str]:
        if getattr(model, "readout_layers", None) != 2:
            return "readout depth other than 2"
        if not bool(getattr(model, "use_fused", True)):
            return "fused kernels disabled on the model"
        if model.n_convolutions + 1 > 4:
            return "more than 3 conv layers"
        lib = _lib.load()
        # (heads the one-launch kernel does not cover -- widths other than 64 / 128 -- run as five launches of the any-shape
        #  kernels inside the same no-autograd step)
        if type(model.loss).__name__ != "MSELoss":
            return "loss other than MSE"
        if not all(q.requires_grad for q in (model.parameters() if _params is None else _params)):
            return "frozen parameters (the fused backward writes every gradient)"
        if batch is not None:
            if getattr(batch, "y", None) is None:
                return "batch has no targets"
            mx = getattr(batch, "max_nodes", None)
            if mx is None or not getattr(batch, "edges_grouped", False):
                return "batch lacks collate metadata (max_nodes / grouped edges)"
            me = getattr(batch, "max_edges", None)
            convs = [model.conv1] + list(model.conv_layers)
            for c in convs:
                if lib.hcg_fused_graphs_per_tile(c.in_channels, c.out_channels, mx) <= 0 and not (
                        me is not None and lib.hcg_mid_supported(c.in_channels, c.out_channels, mx, me)):
                    return "graph / layer shape outside the fused kernels (small-graph tiles and one-graph-per-workgroup)"
        return None

    def _side_stream(self, dev):
        st = self._side_streams.get(dev)
        if st is None:
            st = self._side_streams[dev] = torch.cuda.Stream(device=dev)
        return st

    # ------------------------------------------------------------------ size-grouped batches: two kernel families per layer
    def _size_groups(self, batch, plan, convs, D, C, n_conv):
        """-> n_small when the batch is size-grouped (`collate(..., group_by_size=True)`: the first n_small graphs have <= 32
        nodes, the others 33 .. 64) and both groups are non-empty: the small graphs then run in the small-graph tiles, the
        larger ones one graph per wave, instead of everything on the slower family."""
        ns = getattr(batch, "n_small", None)
        lib = _lib.load()
        if (ns is None or not (0 < ns < plan.B) or n_conv != 2 or D != 64 or not lib.hcg_head_supported(D, C)
                or plan.max_nodes is None or plan.max_nodes <= 32 or plan.max_edges is None):
            return None
        for c in convs:
            if lib.hcg_fused_graphs_per_tile(c.in_channels, c.out_channels, 32) <= 0:
                return None
            if not HF.mid_supported(plan, c.in_channels, c.out_channels):
                return None
        return int(ns)

    def _routed_step(self, batch, plan, bufs, x, y2, convs, l0, l1, n_small, _forward_only):
        """The step of `__call__` for a size-grouped batch: graphs [0, n_small) through csrc/fused.hip (one graph per
        32-row tile), graphs [n_small, B) through the hcg_mid_* entry points (one graph per wave, csrc/wave.hip) -- every
        launch gets the sub-range of graph_ptr / edge_ptr / emb / demb it owns; node rows are absolute, so x, the
        activations and dx need no offsets.  Both groups' slabs of a layer sit back to back and are ONE reduction job."""
        lib, model = _lib.load(), self.model
        p = _lib.ptr
        N, F, B, D, C = x.shape[0], x.shape[1], plan.B, model.embedding_dim, model._n_classes
        dev, stream, slope = x.device, _lib.stream_ptr(), HF.LEAKY_SLOPE
        mxn, mxe = plan.max_nodes, plan.max_edges
        Bs, Bb = n_small, B - n_small
        # The two groups' launches touch disjoint rows, graphs and slabs.  While the step is being CAPTURED the larger graphs'
        # launches go to a second stream, forked before each conv phase and joined behind it = two branches of the hipGraph:
        # each family's last workgroups fill the CUs the other has already left (neither fills the chip: ~1.4 tiles per wave /
        # ~0.6 graphs per wave slot at 4096 graphs).  Measured on the ragged bench: replay 0.1975 -> 0.188 ms/step; the eager
        # step is host-bound and the four extra event calls cost it 0.196 -> 0.215, so eager steps stay on one stream.
        main_s = torch.cuda.current_stream(dev)
        side_s = self._side_stream(dev) if (self.OVERLAP_GROUPS and torch.cuda.is_current_stream_capturing()) else None
        stream_b = ctypes.c_void_p(side_s.cuda_stream) if side_s is not None else stream

        def fork():
            if side_s is not None:
                side_s.wait_stream(main_s)

        def join():
            if side_s is not None:
                main_s.wait_stream(side_s)
        gp, ep = plan.graph_ptr, plan.edge_ptr
        gp_b, ep_b = gp.data_ptr() + 4 * Bs, ep.data_ptr() + 4 * Bs
        acts, emb, demb = bufs["acts"], bufs["emb"], bufs["demb"]
        emb_b, demb_b = emb.data_ptr() + 4 * 2 * D * Bs, demb.data_ptr() + 4 * 2 * D * Bs
        W = [HF._f32c(c.lin.weight) for c in convs]
        bs = [HF._f32c(c.bias) for c in convs]
        gpt = int(lib.hcg_fused_graphs_per_tile(F, D, 32))
        # ---- forward: tiles (both layers + pooling in one launch, pooled layer kept on chip as 2 bits / element), then waves
        nb = lib.hcg_fused_poolbits_bytes(Bs, gpt)
        poolbits = bufs["ws"].get("poolbits_r")
        if poolbits is None or poolbits.numel() < nb:
            poolbits = bufs["ws"]["poolbits_r"] = torch.empty(int(nb * 1.25), dtype=torch.uint8, device=dev)
        fork()
        _lib.check(lib.hcg_fused_stack2_fwd_train(p(x), p(W[0]), p(bs[0]), p(W[1]), p(bs[1]), p(plan.edge_index), plan.E, p(gp), p(ep),
                                                  N, Bs, F, D, gpt, slope, 1, p(acts[0]), p(emb), p(poolbits), p(plan.status), stream),
                   "hcg_fused_stack2_fwd_train")
        _lib.check(lib.hcg_mid_layer_fwd(p(x), p(W[0]), p(bs[0]), p(plan.edge_index), plan.E, gp_b, ep_b, N, Bb, F, D, mxn, mxe,
                                         slope, 1, p(acts[0]), None, p(plan.status), stream_b), "hcg_mid_layer_fwd")
        _lib.check(lib.hcg_mid_layer_fwd(p(acts[0]), p(W[1]), p(bs[1]), p(plan.edge_index), plan.E, gp_b, ep_b, N, Bb, D, D, mxn, mxe,
                                         slope, 1, p(acts[1]), emb_b, p(plan.status), stream_b), "hcg_mid_layer_fwd")
        join()
        # ---- head over all graphs
        opt = model.optimizer
        step_word, flat, g = None, None, None
        if not _forward_only:
            params = self._trainable()
            flat = self._flat_grads(params, dev)
            # address of every parameter's slice of the flat gradient buffer (cached per buffer: eight tensor slices per
            # step otherwise)
            ga = getattr(self, "_gaddr", None)
            if ga is None or ga[0] != flat.data_ptr() or ga[1] is not params:
                addr, off = {}, 0
                for q in params:
                    addr[id(q)] = flat.data_ptr() + 4 * off
                    off += q.numel()
                ga = self._gaddr = (flat.data_ptr(), params, addr)
            g = (lambda prm, _a=ga[2]: _a[id(prm)])
            if (self.optimizer_step and self.grad_sync is None and not self._capturing_split
                    and (self.combine == "mean" or self.exchange is not None) and hasattr(opt, "fused_update_ready")):
                step_word = opt.fused_update_ready(flat)
        sse = self.combine == "sse" and not _forward_only
        mode = _lib.HCG_HEAD_SSE if sse else int(self.rmse)
        tail = self._flat_ext[flat.numel():] if sse else None
        _lib.check(lib.hcg_head_fwd_bwd_ex(p(emb), p(y2), p(HF._f32c(l0.weight)), p(HF._f32c(l0.bias)), p(HF._f32c(l1.weight)),
                                           p(HF._f32c(l1.bias)), B, D, C, slope, mode, p(bufs["z"]), p(bufs["out"]), p(bufs["loss"]),
                                           p(demb), p(bufs["ws_head"]), bufs["ws_head_bytes"], p(self._sync_words(dev)),
                                           p(step_word), p(tail), stream), "hcg_head_fwd_bwd_ex")
        self.last_out = bufs["out"]
        if _forward_only:
            return bufs["loss"][0]
        jb = lib.hcg_reduce_job_bytes()
        jobs = ctypes.create_string_buffer(jb * 8)
        tmp = ctypes.create_string_buffer(jb)
        jaddr, taddr = ctypes.addressof(jobs), ctypes.addressof(tmp)
        _lib.check(lib.hcg_head_reduce_job(p(bufs["ws_head"]), bufs["ws_head_bytes"], B, C, g(l0.weight), g(l0.bias),
                                           g(l1.weight), g(l1.bias), jaddr), "hcg_head_reduce_job")
        njobs = 1
        # ---- conv stack backward, last layer first; each layer: tiles on the small graphs, waves on the others
        premask = bool(self.PREMASK)
        dx = bufs["dacts"][0]
        fork()
        for l in (1, 0):
            inp = x if l == 0 else acts[0]
            Fl = inp.shape[1]
            ws_a = lib.hcg_fused_workspace_bytes(Bs, Fl, D, gpt)
            ws_b = lib.hcg_mid_workspace_bytes(Bb, Fl, D, mxn, mxe)
            off_b = ws_a - 256                                  # = the tile launch's slabs, exactly (the query pads by 256)
            ws = bufs["ws"].get(("r", l))
            if ws is None or ws.numel() < off_b + ws_b:
                ws = bufs["ws"][("r", l)] = torch.empty(int((off_b + ws_b) * 1.25), dtype=torch.uint8, device=dev)
            wsb_ptr = ws.data_ptr() + off_b
            if l == 1:
                flags = 1 | (2 if premask else 0)
                _lib.check(lib.hcg_fused_layer_bwd_poolbits(p(demb), p(poolbits), p(inp), p(W[l]), p(plan.edge_index), plan.E, p(gp),
                                                            p(ep), N, Bs, Fl, D, gpt, slope, flags, p(dx), p(plan.status), p(ws),
                                                            off_b, stream), "hcg_fused_layer_bwd_poolbits")
                _lib.check(lib.hcg_mid_layer_bwd(None, demb_b, emb_b, p(acts[1]), p(inp), p(W[l]), p(plan.edge_index), plan.E, gp_b,
                                                 ep_b, N, Bb, Fl, D, mxn, mxe, slope, flags, p(dx), p(plan.status), wsb_ptr, ws_b,
                                                 stream_b), "hcg_mid_layer_bwd")
            else:
                act = 0 if premask else 1
                a_out = p(acts[0]) if act else None
                _lib.check(lib.hcg_fused_layer_bwd(p(dx), None, None, a_out, p(inp), p(W[l]), p(plan.edge_index), plan.E, p(gp), p(ep),
                                                   N, Bs, Fl, D, gpt, slope, act, None, p(plan.status), p(ws), off_b, stream),
                           "hcg_fused_layer_bwd")
                _lib.check(lib.hcg_mid_layer_bwd(p(dx), None, None, a_out, p(inp), p(W[l]), p(plan.edge_index), plan.E, gp_b, ep_b, N,
                                                 Bb, Fl, D, mxn, mxe, slope, act, None, p(plan.status), wsb_ptr, ws_b, stream_b),
                           "hcg_mid_layer_bwd")
            _lib.check(lib.hcg_fused_reduce_job(p(ws), off_b, N, Bs, Fl, D, gpt, g(convs[l].lin.weight), g(convs[l].bias),
                                                jaddr + njobs * jb), "hcg_fused_reduce_job")
            _lib.check(lib.hcg_mid_reduce_job(wsb_ptr, ws_b, Bb, Fl, D, mxn, mxe, 0, g(convs[l].lin.weight), g(convs[l].bias), taddr),
                       "hcg_mid_reduce_job")
            _lib.check(lib.hcg_reduce_job_append(jaddr + njobs * jb, taddr), "hcg_reduce_job_append")
            njobs += 1
        join()
        self._last_carried = step_word is not None
        if step_word is not None:
            if self.exchange is not None and self.pre_exchange_hook is not None:
                self.pre_exchange_hook()
            if not opt.step_with_reduction(jaddr, njobs, flat, next_plan=self.next_plan, exchange=self.exchange,
                                           flat_ext=self._flat_ext, mode=self.combine, loss_buf=bufs["loss"]):
                raise _lib.HcgError("optimizer state changed between head launch and update")
        else:
            _lib.check(lib.hcg_reduce_slabs(jaddr, njobs, stream), "hcg_reduce_slabs")
            if self.next_plan is not None:
                self.next_plan.rebuild()                   # (no fused update to ride in: its own launch)
            if not self._capturing_split:
                self._exchange_and_update(bufs["loss"])
        return bufs["loss"][0]

    # ------------------------------------------------------------------ the step
    def _buffers(self, key, N, B, F, D, C, n_conv, dev):
        """Step buffers: allocated for the largest (N, B) seen so far and handed out as views -- the variable-size
        batches of a shuffled epoch then reuse one allocation instead of ~12 `torch.empty` per step."""
        lib = _lib.load()
        cap = self._bufs.get("cap")
        sig = (F, D, C, n_conv, dev)
        if cap is None or cap["sig"] != sig or cap["N"] < N or cap["B"] < B:
            capN = max(N, int(cap["N"] * 1.25) if cap and cap["sig"] == sig else 0)
            capB = max(B, cap["B"] if cap and cap["sig"] == sig else 0)
            f32 = dict(dtype=torch.float32, device=dev)
            hb = lib.hcg_head_workspace_bytes_d(capB, D if lib.hcg_head_supported(D, C) else 64)
            cap = {"sig": sig, "N": capN, "B": capB,
                   "acts": [torch.empty(capN, D, **f32) for _ in range(n_conv)],
                   "dacts": [torch.empty(capN, D, **f32) for _ in range(n_conv - 1)],
                   "emb": torch.empty(capB, 2 * D, **f32), "demb": torch.empty(capB, 2 * D, **f32),
                   "z": torch.empty(capB, D, **f32), "out": torch.empty(capB, C, **f32), "loss": torch.empty(2, **f32),
                   "ws_head": torch.empty(hb, dtype=torch.uint8, device=dev), "ws": {}}
            self._bufs = {"cap": cap}
            self._graph = None                          # a captured graph holds the old buffers' addresses
        b = self._bufs.get(key)
        if b is None:
            b = {"acts": [t[:N] for t in cap["acts"]], "dacts": [t[:N] for t in cap["dacts"]], "emb": cap["emb"][:B],
                 "demb": cap["demb"][:B], "z": cap["z"][:B], "out": cap["out"][:B], "loss": cap["loss"],
                 "ws_head": cap["ws_head"], "ws_head_bytes": lib.hcg_head_workspace_bytes_d(B, D if lib.hcg_head_supported(D, C) else 64), "ws": cap["ws"]}
            self._bufs = {"cap": cap, key: b}          # views of the current shape (one live shape at a time)
        return b

    def _tall_ws(self, bufs, l, N, B, F, D, dev):
        """Workspace of layer l's wide-layer kernels (H / dH round trip + gradient slabs): forward and backward share it."""
        wsb = _lib.load().hcg_tall_workspace_bytes(N, B, F, D)
        ws = bufs["ws"].get(("tall", l))
        if ws is None or ws.numel() < wsb:
            ws = bufs["ws"][("tall", l)] = torch.empty(int(wsb * 1.25), dtype=torch.uint8, device=dev)
        return ws, wsb

    def _head_buffers(self, bufs, B, D, C, dev):
        """Scratch of the any-shape head (five launches): allocated once per capacity."""
        lib = _lib.load()
        hb = bufs["ws"].get("head_generic")
        if hb is None or hb["B"] < B:
            f32 = dict(dtype=torch.float32, device=dev)
            u8 = dict(dtype=torch.uint8, device=dev)
            hb = {"B": B, "dout": torch.empty(B, C, **f32), "dz": torch.empty(B, D, **f32),
                  "dz_ws1": torch.empty(B, C, **f32), "dz_ws0": torch.empty(B, D, **f32),
                  "ws1": torch.empty(max(lib.hcg_linear_workspace_bytes(B, D, C), 256), **u8),
                  "ws0": torch.empty(max(lib.hcg_linear_workspace_bytes(B, 2 * D, D), 256), **u8)}
            bufs["ws"]["head_generic"] = hb
        return hb

    def _flat_grads(self, params, dev):
        """ONE flat gradient buffer in parameter order (+ two floats behind it: [SSE, count] of the data-parallel "sse"
        form); the parameters' `.grad` are (re-)attached as views of it."""
        n = sum(p.numel() for p in params)
        flat = getattr(self, "_flat", None)
        if flat is None or flat.numel() != n or flat.device != dev:
            self._flat_ext = torch.zeros(n + 2, dtype=torch.float32, device=dev)
            flat = self._flat_ext[:n]
            self._flat = flat
            self._graph = None
            off = 0
            for p in params:                      # .grad = view of the flat buffer, parameter order
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        else:
            off = 0
            for p in params:
                g = p.grad
                if g is None or g.data_ptr() != flat.data_ptr() + 4 * off:
                    p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        return flat

    def evaluate(self, batch, _checked: bool = False):
        """Forward + loss only (the reference's `eval_network` body, utils/utils_model.py:75-78): plan, conv stack, head
        -- 3 launches; the head kernel's backward half runs too (a few us) but nothing is reduced or updated."""
        return self(batch, _forward_only=True, _checked=_checked)

    def __call__(self, batch, _forward_only: bool = False, _checked: bool = False):
        model = self.model
        if not _checked:                 # (the epoch loops below have just asked `reason(batch)` themselves)
            why = self.reason(batch)
            if why is not None:
                raise _lib.HcgError(f"FusedTrainStep does not cover this model/batch: {why}")
        lib = _lib.load()
        x, y = batch.x, batch.y
        _lib.require_gpu(x, y, batch.edge_index)
        x = HF._f32c(x)
        plan = model._plan_for(batch, x, batch.edge_index, batch.batch, None)
        convs = [model.conv1] + list(model.conv_layers)
        l0, l1 = model.readout[0][0], model.readout[1]
        N, F, B, D, C = x.shape[0], x.shape[1], plan.B, model.embedding_dim, model._n_classes
        n_conv, dev, stream, slope = len(convs), x.device, _lib.stream_ptr(), HF.LEAKY_SLOPE
        y2 = HF._f32c(y).reshape(B, -1)
        if y2.shape[1] != C:
            raise ValueError(f"targets have {y2.shape[1]} columns, the model predicts {C}")
        # kernel family per layer: gpt > 0 = small-graph tiles (csrc/fused.hip), 0 = one graph per workgroup (csrc/mid.hip)
        gpts = [HF.fused_graphs_per_tile(plan, c.in_channels, c.out_channels) for c in convs]
        for c, gpt in zip(convs, gpts):
            if gpt <= 0 and not HF.mid_supported(plan, c.in_channels, c.out_channels):
                raise _lib.HcgError("FusedTrainStep: graph / layer shape outside the fused kernels")
        # 128-wide layers over large graphs: dense row-streaming transform + per-graph segmented sum (csrc/tall.hip)
        tall = [gpt <= 0 and getattr(c, "family", "auto") != "mid" and HF.tall_supported(plan, c.in_channels, c.out_channels)
                for c, gpt in zip(convs, gpts)]
        mxn, mxe = plan.max_nodes, plan.max_edges
        bufs = self._buffers((N, B, F, plan.E), N, B, F, D, C, n_conv, dev)
        n_small = self._size_groups(batch, plan, convs, D, C, n_conv)
        if n_small is not None:
            return self._routed_step(batch, plan, bufs, x, y2, convs, l0, l1, n_small, _forward_only)
        acts, emb = bufs["acts"], bufs["emb"]
        p = _lib.ptr
        W = [HF._f32c(c.lin.weight) for c in convs]
        bs = [HF._f32c(c.bias) for c in convs]
        # ---- forward (conv stack + pooling).  Small-graph tiles: the pooled layer's activations stay on chip, two bits
        #      per element (sign, is-the-column-max) are all its backward needs of them
        poolbits = None
        if gpts[-1] > 0 and self.POOLBITS:
            nb = lib.hcg_fused_poolbits_bytes(B, gpts[-1])
            poolbits = bufs["ws"].get("poolbits")
            if poolbits is None or poolbits.numel() < nb:
                poolbits = bufs["ws"]["poolbits"] = torch.empty(int(nb * 1.25), dtype=torch.uint8, device=dev)
        if n_conv == 2 and gpts[0] == gpts[1] and gpts[0] > 0 and poolbits is None:
            rc = lib.hcg_fused_stack2_fwd(p(x), p(W[0]), p(bs[0]), p(W[1]), p(bs[1]), p(plan.edge_index), plan.E,
                                          p(plan.graph_ptr), p(plan.edge_ptr), N, B, F, D, gpts[0], slope, 1, p(acts[0]),
                                          p(acts[1]), p(emb), p(plan.status), stream)
            _lib.check(rc, "hcg_fused_stack2_fwd")
        elif n_conv == 2 and gpts[0] == gpts[1] and gpts[0] > 0:
            rc = lib.hcg_fused_stack2_fwd_train(p(x), p(W[0]), p(bs[0]), p(W[1]), p(bs[1]), p(plan.edge_index), plan.E,
                                                p(plan.graph_ptr), p(plan.edge_ptr), N, B, F, D, gpts[0], slope, 1,
                                                p(acts[0]), p(emb), p(poolbits), p(plan.status), stream)
            _lib.check(rc, "hcg_fused_stack2_fwd_train")
        else:
            h = x
            for l in range(n_conv):
                pe = p(emb) if l == n_conv - 1 else None
                if gpts[l] > 0 and pe is not None and poolbits is not None:
                    rc = lib.hcg_fused_layer_fwd_train(p(h), p(W[l]), p(bs[l]), p(plan.edge_index), plan.E, p(plan.graph_ptr),
                                                       p(plan.edge_ptr), N, B, h.shape[1], D, gpts[l], slope, 1, pe,
                                                       p(poolbits), p(plan.status), stream)
                    _lib.check(rc, "hcg_fused_layer_fwd_train")
                elif gpts[l] > 0:
                    rc = lib.hcg_fused_layer_fwd(p(h), p(W[l]), p(bs[l]), p(plan.edge_index), plan.E, p(plan.graph_ptr),
                                                 p(plan.edge_ptr), N, B, h.shape[1], D, gpts[l], slope, 1, p(acts[l]), pe,
                                                 p(plan.status), stream)
                    _lib.check(rc, "hcg_fused_layer_fwd")
                elif tall[l]:
                    ws, wsb = self._tall_ws(bufs, l, N, B, h.shape[1], D, dev)
                    rc = lib.hcg_tall_layer_fwd(p(h), p(W[l]), p(bs[l]), p(plan.edge_index), plan.E, p(plan.graph_ptr),
                                                p(plan.edge_ptr), N, B, h.shape[1], D, mxn, mxe, slope, 1, p(acts[l]), pe,
                                                p(plan.status), p(ws), wsb, stream)
                    _lib.check(rc, "hcg_tall_layer_fwd")
                else:
                    rc = lib.hcg_mid_layer_fwd(p(h), p(W[l]), p(bs[l]), p(plan.edge_index), plan.E, p(plan.graph_ptr),
                                               p(plan.edge_ptr), N, B, h.shape[1], D, mxn, mxe, slope, 1, p(acts[l]), pe,
                                               p(plan.status), stream)
                    _lib.check(rc, "hcg_mid_layer_fwd")
                h = acts[l]
        # ---- head: readout forward, loss, readout backward
        opt = model.optimizer
        step_word, flat, g = None, None, None
        if not _forward_only:
            params = self._trainable()
            flat = self._flat_grads(params, dev)
            ga = getattr(self, "_gaddr", None)         # (see _routed_step)
            if ga is None or ga[0] != flat.data_ptr() or ga[1] is not params:
                addr, off = {}, 0
                for q in params:
                    addr[id(q)] = flat.data_ptr() + 4 * off
                    off += q.numel()
                ga = self._gaddr = (flat.data_ptr(), params, addr)
            g = (lambda prm, _a=ga[2]: _a[id(prm)])
            # without an exchange between backward and update, the slab reduction applies Adam itself; the head kernel
            # advances the step number that launch reads
            if (self.optimizer_step and self.grad_sync is None and not self._capturing_split
                    and (self.combine == "mean" or self.exchange is not None) and hasattr(opt, "fused_update_ready")):
                step_word = opt.fused_update_ready(flat)
        sse = self.combine == "sse" and not _forward_only
        mode = _lib.HCG_HEAD_SSE if sse else int(self.rmse)
        tail = self._flat_ext[flat.numel():] if sse else None
        fused_head = bool(lib.hcg_head_supported(D, C))
        W0, b0, W1, b1 = HF._f32c(l0.weight), HF._f32c(l0.bias), HF._f32c(l1.weight), HF._f32c(l1.bias)
        jb = lib.hcg_reduce_job_bytes()
        jobs = ctypes.create_string_buffer(jb * 8)
        jaddr = ctypes.addressof(jobs)
        njobs = 0
        if fused_head:
            rc = lib.hcg_head_fwd_bwd_ex(p(emb), p(y2), p(W0), p(b0), p(W1), p(b1), B, D, C, slope, mode, p(bufs["z"]),
                                         p(bufs["out"]), p(bufs["loss"]), p(bufs["demb"]), p(bufs["ws_head"]),
                                         bufs["ws_head_bytes"], p(self._sync_words(dev)), p(step_word), p(tail), stream)
            _lib.check(rc, "hcg_head_fwd_bwd_ex")
            if _forward_only:
                self.last_out = bufs["out"]
                return bufs["loss"][0]
            _lib.check(lib.hcg_head_reduce_job_d(p(bufs["ws_head"]), bufs["ws_head_bytes"], B, D, C, g(l0.weight), g(l0.bias),
                                                 g(l1.weight), g(l1.bias), jaddr), "hcg_head_reduce_job_d")
            njobs = 1
        else:
            # any-shape head (widths other than 64 / 128): Linear + LeakyReLU, Linear, loss with its gradient, two Linear backwards
            # that write straight into the flat gradient buffer (reference model/gcn.py:70-71, utils/utils_model.py:64-65)
            hb = self._head_buffers(bufs, B, D, C, dev)
            z, out = bufs["z"], bufs["out"]
            _lib.check(lib.hcg_linear_fwd(p(emb), p(W0), p(b0), p(z), B, 2 * D, D, _lib.HCG_ACT_LEAKY, slope, stream), "hcg_linear_fwd")
            _lib.check(lib.hcg_linear_fwd(p(z), p(W1), p(b1), p(out), B, D, C, _lib.HCG_ACT_NONE, slope, stream), "hcg_linear_fwd")
            _lib.check(lib.hcg_loss_fwd_bwd(p(out), p(y2), B * C, mode, p(bufs["loss"]), p(hb["dout"]), p(tail), stream),
                       "hcg_loss_fwd_bwd")
            if _forward_only:
                self.last_out = bufs["out"]
                return bufs["loss"][0]
            _lib.check(lib.hcg_linear_bwd(p(hb["dout"]), p(out), p(z), p(W1), p(hb["dz"]), g(l1.weight), g(l1.bias), p(hb["dz_ws1"]),
                                          B, D, C, _lib.HCG_ACT_NONE, slope, p(hb["ws1"]), hb["ws1"].numel(), stream), "hcg_linear_bwd")
            _lib.check(lib.hcg_linear_bwd(p(hb["dz"]), p(z), p(emb), p(W0), p(bufs["demb"]), g(l0.weight), g(l0.bias), p(hb["dz_ws0"]),
                                          B, 2 * D, D, _lib.HCG_ACT_LEAKY, slope, p(hb["ws0"]), hb["ws0"].numel(), stream), "hcg_linear_bwd")
            step_word = None          # the head's gradients are not slabs: reduction and update stay two launches
        # ---- conv stack backward, last layer first
        # a fused-tile layer can hand its dx down already multiplied by the activation derivative of the layer below
        # (it holds those rows anyway, for dW); that layer then never reads its own output: one tensor less per step
        dh, premasked = None, False
        for l in reversed(range(n_conv)):
            inp = x if l == 0 else acts[l - 1]
            Fl = inp.shape[1]
            dx = bufs["dacts"][l - 1] if l > 0 else None
            small = gpts[l] > 0
            wsb = (lib.hcg_fused_workspace_bytes(B, Fl, D, gpts[l]) if small else
                   (0 if tall[l] else lib.hcg_mid_workspace_bytes(B, Fl, D, mxn, mxe)))
            ws = bufs["ws"].get(l)
            if (ws is None or ws.numel() < wsb) and wsb > 0:
                ws = torch.empty(int(wsb * 1.25), dtype=torch.uint8, device=dev)
                bufs["ws"][l] = ws
            last = l == n_conv - 1
            up = (None if last else p(dh), p(bufs["demb"]) if last else None, p(emb) if last else None)
            if small and last and poolbits is not None:
                premasked = self.PREMASK and l > 0
                rc = lib.hcg_fused_layer_bwd_poolbits(p(bufs["demb"]), p(poolbits), p(inp), p(W[l]), p(plan.edge_index), plan.E,
                                                      p(plan.graph_ptr), p(plan.edge_ptr), N, B, Fl, D, gpts[l], slope,
                                                      1 | (2 if premasked else 0), p(dx), p(plan.status), p(ws), wsb, stream)
                _lib.check(rc, "hcg_fused_layer_bwd_poolbits")
                _lib.check(lib.hcg_fused_reduce_job(p(ws), wsb, N, B, Fl, D, gpts[l], g(convs[l].lin.weight),
                                                    g(convs[l].bias), jaddr + njobs * jb), "hcg_fused_reduce_job")
            elif small:
                act = 0 if premasked else 1
                premasked = self.PREMASK and l > 0
                rc = lib.hcg_fused_layer_bwd(*up, p(acts[l]) if (act or last) else None, p(inp), p(W[l]), p(plan.edge_index),
                                             plan.E, p(plan.graph_ptr), p(plan.edge_ptr), N, B, Fl, D, gpts[l], slope,
                                             act | (2 if premasked else 0), p(dx), p(plan.status), p(ws), wsb, stream)
                _lib.check(rc, "hcg_fused_layer_bwd")
                _lib.check(lib.hcg_fused_reduce_job(p(ws), wsb, N, B, Fl, D, gpts[l], g(convs[l].lin.weight),
                                                    g(convs[l].bias), jaddr + njobs * jb), "hcg_fused_reduce_job")
            elif tall[l]:
                # (never premasks: the layer below reads its own output row-contiguous instead -- the premask would be 4-byte
                #  strided loads in the dense kernel's epilogue for the same bytes)
                act = 0 if premasked else 1
                premasked = False
                tws, twsb = self._tall_ws(bufs, l, N, B, Fl, D, dev)
                rc = lib.hcg_tall_layer_bwd(*up, p(acts[l]) if (act or last) else None, p(inp), p(W[l]), p(plan.edge_index),
                                            plan.E, p(plan.graph_ptr), p(plan.edge_ptr), N, B, Fl, D, mxn, mxe, slope, act,
                                            p(dx), p(plan.status), p(tws), twsb, stream)
                _lib.check(rc, "hcg_tall_layer_bwd")
                _lib.check(lib.hcg_tall_reduce_jobs(p(tws), twsb, N, B, Fl, D, g(convs[l].lin.weight), g(convs[l].bias),
                                                    jaddr + njobs * jb), "hcg_tall_reduce_jobs")
                njobs += 1                          # (two jobs: dW, db)
            else:
                act = 0 if premasked else 1
                premasked = self.PREMASK and l > 0
                rc = lib.hcg_mid_layer_bwd(*up, p(acts[l]) if (act or last) else None, p(inp), p(W[l]), p(plan.edge_index),
                                           plan.E, p(plan.graph_ptr), p(plan.edge_ptr), N, B, Fl, D, mxn, mxe, slope,
                                           act | (2 if premasked else 0), p(dx), p(plan.status), p(ws), wsb, stream)
                _lib.check(rc, "hcg_mid_layer_bwd")
                for half in range(D // 64):      # one slab set (= one job) per 64-column half
                    if half > 0:
                        njobs += 1
                    _lib.check(lib.hcg_mid_reduce_job(p(ws), wsb, B, Fl, D, mxn, mxe, half, g(convs[l].lin.weight),
                                                      g(convs[l].bias), jaddr + njobs * jb), "hcg_mid_reduce_job")
            njobs += 1
            dh = dx
        # ---- slab reduction -> flat gradient, exchange, update.  Without an exchange in between, reduction and Adam
        #      are one launch (the update reads each gradient element as it is produced)
        self._last_carried = step_word is not None
        if step_word is not None:
            if self.exchange is not None and self.pre_exchange_hook is not None:
                self.pre_exchange_hook()
            if not opt.step_with_reduction(jaddr, njobs, flat, next_plan=self.next_plan, exchange=self.exchange,
                                           flat_ext=self._flat_ext, mode=self.combine, loss_buf=bufs["loss"]):    # (same preconditions as fused_update_ready)
                raise _lib.HcgError("optimizer state changed between head launch and update")
        else:
            _lib.check(lib.hcg_reduce_slabs(jaddr, njobs, stream), "hcg_reduce_slabs")
            if self.next_plan is not None:
                self.next_plan.rebuild()                   # (no fused update to ride in: its own launch)
            if not self._capturing_split:
                self._exchange_and_update(bufs["loss"])
        self.last_out = bufs["out"]
        return bufs["loss"][0]

    def _exchange_and_update(self, loss_buf):
        """Behind the slab reduction: gradient exchange (data parallel), the "sse" scale, the optimiser."""
        lib, opt, flat = _lib.load(), self.model.optimizer, self._flat
        sync = self.grad_sync
        if sync is None and self.exchange is not None:
            # a one-shot exchange is attached but this step's update is not the fused launch that carries it (any-shape head,
            # optimiser state not one flat group): the ranks exchange through the collective instead -- never not at all
            sync = self.exchange_fallback_sync
            if sync is None:
                raise _lib.HcgError("a one-shot exchange is attached, this step cannot carry it and no collective fallback is "
                                    "set: the replicas would diverge")
        if self.combine == "sse":
            ext = self._flat_ext
            if sync is not None:
                sync(ext)                                    # SUM of [gradients | SSE | count] over the ranks
            if self.optimizer_step and hasattr(opt, "step_sse"):
                self._flat_grads(self._trainable(), flat.device)
                opt.step_sse(ext, loss_buf)                  # scale + update, one launch
            else:
                _lib.check(lib.hcg_sse_finalize(_lib.ptr(ext), flat.numel(), _lib.ptr(loss_buf), _lib.stream_ptr()),
                           "hcg_sse_finalize")
                if self.optimizer_step:
                    opt.step()
            return
        if sync is not None:
            sync(flat)
        if self.optimizer_step:
            # the update reads the parameters' `.grad`: they must be views of THIS trainer's buffer (another trainer on
            # the same model may have re-pointed them since)
            self._flat_grads(self._trainable(), flat.device)
            opt.step()

    # ------------------------------------------------------------------ hipGraph
    def capture(self, batch, prefetch=None, next_plan=None):
        """Capture the step on `batch`'s tensors into a hipGraph; `replay()` re-runs it on whatever those tensors
        hold then (copy the next batch into them, or re-collate in place).  `batch` may be a callable returning the
        batch: whatever it enqueues (a device collate, the plan build of a fresh `Batch`) is captured too.  The optimiser switches to its
        device-side step counter / learning rate (`FusedAdam.enable_capturable`); the gradient exchange
        (`grad_sync`) is NOT captured: with one, the graph ends after the slab reduction and `replay()` issues the
        collective and the (single-launch) update eagerly behind it.
        `prefetch`: a callable whose launches are captured on a FORKED branch of the graph (forks at the start of the
        step, joins at its end): work for the NEXT step that does not depend on this one -- the next batch's plan build
        (`BatchPlan.rebuild`), a device collate -- runs beside this step instead of in front of the next.
        `next_plan`: sets `self.next_plan` (see `__init__`) for the captured step."""
        if next_plan is not None:
            self.next_plan = next_plan
        opt = self.model.optimizer
        if self.optimizer_step:
            if not hasattr(opt, "enable_capturable"):
                raise _lib.HcgError("capture() with optimizer_step needs hcatgnet_amd.optim.FusedAdam")
            opt.enable_capturable()
        sync, do_opt = self.grad_sync, self.optimizer_step

        def get():
            if callable(batch):
                return batch()
            # a fixed Batch object: drop its cached plan so that the plan build (graph_ptr / edge_ptr from the int64
            # batch vector and edge_index) is enqueued -- and captured -- every time; replay() then follows whatever
            # graph boundaries the tensors hold.  max_nodes / max_edges of `batch` act as capacities.
            try:
                batch._hcg_plan = None
            except Exception:
                pass
            return batch
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                       # warm-up: every buffer allocated, optimiser state re-based
                self(get())
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        orig_sync = sync
        if sync is None and self.exchange is not None and not self._last_carried:
            sync = self.exchange_fallback_sync         # (this step's update cannot carry the one-shot exchange: collective form)
        split = sync is not None and not self.capture_exchange
        try:
            if split or sync is None:
                self.grad_sync = None
            self._capturing_split = split              # with an exchange, the graph ends after the slab reduction
                                                       # (unless `capture_exchange`: collective + update are recorded too)
            g_main = torch.cuda.CUDAGraph()
            fork = torch.cuda.Stream() if prefetch is not None else None
            # (a live process group has helper threads that query events: their calls must not fail this capture)
            mode = "thread_local" if (torch.distributed.is_available() and torch.distributed.is_initialized()) else "global"
            with torch.cuda.graph(g_main, capture_error_mode=mode):
                if fork is not None:
                    main = torch.cuda.current_stream()
                    fork.wait_stream(main)
                    with torch.cuda.stream(fork):
                        prefetch()
                loss = self(get())
                if fork is not None:
                    main.wait_stream(fork)
        finally:
            self.grad_sync, self._capturing_split = orig_sync, False
        self._graph = (g_main, loss, split, self._graph_fingerprint())
        return self

    def _graph_fingerprint(self):
        """Addresses a captured graph has baked in and that later calls could replace: step buffers, flat gradient,
        the optimiser's flat parameter / moment storages."""
        cap = self._bufs.get("cap")
        fp = [t.data_ptr() for t in cap["acts"] + cap["dacts"]] if cap else []
        if cap:      # kernel workspaces (slabs, pooled bits, the wide-layer kernels' dH buffer): they grow on demand
            fp += [(k if isinstance(k, (str, int)) else str(k), t.data_ptr()) for k, t in sorted(cap["ws"].items(), key=lambda kv: str(kv[0]))
                   if torch.is_tensor(t)]
            fp.append(cap["ws_head"].data_ptr())
        fp.append(self._flat.data_ptr() if getattr(self, "_flat", None) is not None else 0)
        fl = getattr(self.model.optimizer, "_flat", {}).get(0) if self.optimizer_step else None
        if fl is not None:
            fp += [fl["p"].data_ptr(), fl["m"].data_ptr(), fl["v"].data_ptr()]
        fp += [q.data_ptr() for q in self._trainable()]      # (cached walk: this runs on every replay)
        return tuple(fp)

    def replay(self):
        if self._graph is None:
            raise _lib.HcgError("replay(): no captured step (never captured, or its buffers were re-allocated by a larger "
                                "eager batch): call capture() again")
        g_main, loss, split, fp = self._graph
        if fp != self._graph_fingerprint():
            self._graph = None
            raise _lib.HcgError("replay(): parameter / optimiser / step buffers changed since capture() "
                                "(load_state_dict, a larger eager batch): call capture() again")
        if self.optimizer_step:
            self.model.optimizer.sync_lr()
        g_main.replay()
        if split:                                     # exchange between the captured backward and the update:
            self._exchange_and_update(loss_buf=self._bufs["cap"]["loss"])   # one collective, then ONE eager launch
        elif not self.optimizer_step:
            # gradients-only step: the parameters' `.grad` must be THIS trainer's buffer (another trainer on the same model
            # may have re-pointed them since the capture)
            self._flat_grads(self._trainable(), self._flat.device)
        return loss


class StepWindow:
    """Several consecutive training steps as ONE hipGraph: `steps[i]` (a `FusedTrainStep`, all on the same model) run on
    `batches[i]` (a Batch, or a callable returning one), in order, the weights carried from step to step exactly as
    separate launches would.  What a window saves is the bubble between two graph launches (~3.7 us on MI355X / ROCm 7.2:
    C3 0.1152 -> 0.1115 ms/step, the reference's batch size 40 0.0713 -> 0.0675; `tools/exp_multistep_graph.py`) -- for a
    loader whose batches are known ahead (a resident dataset visited in a fixed or pre-drawn order), an epoch is one launch.
    Each step must be capturable on its own first (`FusedTrainStep.capture` has run, or would succeed): same launches, same
    device-side step count / learning rate, the next batch's plan inside each step's last launch if `next_plan` is set.
    A step whose gradient exchange is a separate collective (`grad_sync`) cannot sit inside a window.

        for i, st in enumerate(steps): st.capture(batch_fn[i], next_plan=plans[(i + 1) % n])
        window = StepWindow(steps, batch_fn);  losses = window.replay()      # one launch = n steps"""

    def __init__(self, steps, batches, forward_only: bool = False):
        """`forward_only`: the steps' `evaluate` form (plan, conv stack, head: loss only, nothing reduced or updated)."""
        steps, batches = list(steps), list(batches)
        if not steps or len(steps) != len(batches):
            raise ValueError("StepWindow needs as many batches as steps (at least one)")
        model = steps[0].model
        for st in steps:
            if st.model is not model:
                raise ValueError("the steps of a window train ONE model")
            if st.grad_sync is not None and not st.capture_exchange:
                raise _lib.HcgError("a step with a separate gradient collective (grad_sync) cannot be captured into a window "
                                    "(unless its `capture_exchange` is set: the collective is then recorded with the step)")
            if st.optimizer_step and not hasattr(model.optimizer, "enable_capturable"):
                raise _lib.HcgError("StepWindow with optimizer_step needs hcatgnet_amd.optim.FusedAdam")
        self.forward_only = bool(forward_only)
        if steps[0].optimizer_step and not self.forward_only:
            model.optimizer.enable_capturable()
        self.steps, self.model = steps, model
        get = lambda b: b() if callable(b) else b
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up: every buffer allocated, optimiser state re-based
            for st, b in zip(steps, batches):
                st(get(b), _forward_only=self.forward_only)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        mode = "thread_local" if (torch.distributed.is_available() and torch.distributed.is_initialized()) else "global"
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.losses = [st(get(b), _forward_only=self.forward_only) for st, b in zip(steps, batches)]
        self._fp = [st._graph_fingerprint() for st in steps]

    def replay(self):
        """-> the steps' loss tensors (device scalars, overwritten by the next replay)."""
        if self._fp != [st._graph_fingerprint() for st in self.steps]:
            raise _lib.HcgError("StepWindow.replay(): parameter / optimiser / step buffers changed since the capture: build it again")
        if self.steps[0].optimizer_step and not self.forward_only:
            self.model.optimizer.sync_lr()
        self.graph.replay()
        return self.losses


# ---------------------------------------------------------------------------------------------------------------
# the reference's loops (same names / arguments / return values)
# ---------------------------------------------------------------------------------------------------------------
def _rmse_autograd(model, batch):
    out = model(batch)
    return torch.sqrt(model.loss(out, batch.y.unsqueeze(1)))


def _accumulate(total, loss, num_graphs):
    """total += loss * num_graphs on the device, one launch (the reference does `loss.item() * batch.num_graphs`)."""
    if total is None:
        return loss.detach() * float(num_graphs)
    return total.add_(loss.detach(), alpha=float(num_graphs))


def train_network(model, train_loader, device):
    """reference utils/utils_model.py:55-70: one epoch; returns sum(loss * num_graphs) / len(dataset).
    The per-batch `loss.item()` of the reference is replaced by a device-side accumulation and ONE sync."""
    model.train()
    fused = getattr(model, "_hcg_train_step", None)
    if fused is None:
        fused = FusedTrainStep(model)
        try:
            model._hcg_train_step = fused
        except Exception:
            pass
    total = None
    for batch in train_loader:
        batch = batch.to(device)
        if fused.reason(batch) is None:
            loss = fused(batch, _checked=True)
        else:
            model.optimizer.zero_grad()
            loss = _rmse_autograd(model, batch)
            loss.backward()
            model.optimizer.step()
            loss = loss.detach()
        total = _accumulate(total, loss, batch.num_graphs)
    if total is None:
        return 0.0
    return float(total.item()) / len(train_loader.dataset)


EVAL_WINDOW_MAX_BATCHES = 64


def _eval_window(model, loader, fused):
    """A `store.DeviceLoader` that does not shuffle yields the same batches every epoch (the reference's validation / test
    loaders, call_methods.py:41-46): they are collated once, their `evaluate` steps captured as ONE hipGraph
    (`StepWindow(forward_only=True)`) and an `eval_network` call is one graph launch + one reduction instead of
    (collate + 3 launches) per batch.  -> the epoch's value, or None when this path does not apply (the caller loops)."""
    from .store import DeviceLoader
    if not (isinstance(loader, DeviceLoader) and not loader.shuffle and 0 < len(loader) <= EVAL_WINDOW_MAX_BATCHES):
        return None
    cache = getattr(loader, "_hcg_eval_window", None)
    for attempt in range(2):
        if cache is None or cache["model"] is not model:
            batches = list(loader)
            if not all(fused.reason(b) is None for b in batches):
                return None
            steps = [FusedTrainStep(model, optimizer_step=False) for _ in batches]
            win = StepWindow(steps, batches, forward_only=True)
            counts = torch.tensor([float(b.num_graphs) for b in batches], dtype=torch.float32, device=batches[0].x.device)
            cache = {"model": model, "window": win, "batches": batches, "counts": counts}
            try:
                loader._hcg_eval_window = cache
            except Exception:
                pass
        try:
            losses = cache["window"].replay()
        except _lib.HcgError:                 # parameters re-based since the capture (load_state_dict on new storages ...)
            cache = None
            continue
        total = torch.dot(torch.stack(losses), cache["counts"])
        return float(total.item()) / len(loader.dataset)
    return None


def eval_network(model, loader, device):
    """reference utils/utils_model.py:72-79 (forward + sqrt(MSE) per batch; no parameter update)."""
    model.eval()
    fused = getattr(model, "_hcg_train_step", None)
    if fused is None:
        fused = FusedTrainStep(model)
        try:
            model._hcg_train_step = fused
        except Exception:
            pass
    total = _eval_window(model, loader, fused)
    if total is not None:
        return total
    total = None
    with torch.no_grad():
        for batch in loader:
            batch = batch.to(device)
            if fused.reason(batch) is None:
                loss = fused.evaluate(batch, _checked=True)
            else:
                loss = _rmse_autograd(model, batch)
            total = _accumulate(total, loss, batch.num_graphs)
    if total is None:
        return 0.0
    return float(total.item()) / len(loader.dataset)


def predict_network(model, loader, return_emb: bool = False, device=None):
    """reference utils/utils_model.py:82-111: -> (y_pred, y_true, idx[, embeddings DataFrame]).
    The reference moves the model to the CPU for this; here it stays on the GPU (`device` defaults to the
    model's) and only the results come back.  The embeddings frame has the reference's columns: `0..2D-1`
    (graph_emb = [max, mean]), `ddG_exp`, `ddG_pred`, `index`."""
    import numpy as np
    model.eval()
    if device is None:
        device = next(model.parameters()).device
    y_pred, y_true, idx, embs = [], [], [], []
    with torch.no_grad():
        for batch in loader:
            batch = batch.to(device)
            out, emb = model(batch, True)
            y_pred.append(out.reshape(-1))
            y_true.append(batch.y.reshape(-1))
            idx.append(batch.idx.reshape(-1) if batch.idx is not None else torch.full((batch.num_graphs,), -1, device=out.device))
            if return_emb:
                embs.append(emb)
    y_pred = torch.cat(y_pred).cpu().numpy().ravel()
    y_true = torch.cat(y_true).cpu().numpy().ravel()
    idx = torch.cat(idx).cpu().numpy().ravel()
    if not return_emb:
        return y_pred, y_true, idx
    import pandas as pd
    frame = pd.DataFrame(torch.cat(embs).cpu().numpy())
    frame["ddG_exp"] = y_true
    frame["ddG_pred"] = y_pred
    frame["index"] = idx
    return y_pred, y_true, idx, frame
