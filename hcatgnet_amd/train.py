"""The reference's per-batch loops on MI355X (reference utils/utils_model.py:55-111).

`train_network` / `eval_network` / `predict_network` keep the reference's names, arguments and return
values; what changes is how a step is issued:

  reference step (utils/utils_model.py:60-68)          here (`FusedTrainStep`)
  -------------------------------------------          ---------------------------------------------------
  optimizer.zero_grad()                                 -- (every gradient is overwritten, never accumulated)
  out = model(batch)                                    hcg_fused_forward               1 launch: both conv layers, pooling AND
  loss = sqrt(MSELoss(out, y.unsqueeze(1)))                 the readout head (forward, squared error, unscaled backward)
  loss.backward()                                       hcg_fused_layer_bwd x n_conv    (other graph sizes: hcg_mid_* / hcg_tall_*
                                                            per layer + hcg_head_fwd_bwd)
  [data parallel]                                       [one-shot xGMI exchange inside the last launch, or an RCCL all-reduce]
  optimizer.step()                                      hcg_step_tail                   1 launch: slab reductions -> ONE flat
  loss.item()                                               gradient, loss + its deferred scale, Adam, the next batch's plan
                                                        -- (the loss stays on the device; ONE sync per epoch)

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

class _Ctx:
    """Everything one step needs, resolved once: shapes, weights, buffers, the kernel family of every layer."""
    __slots__ = ("batch", "plan", "x", "y2", "convs", "l0", "l1", "N", "F", "B", "D", "C", "n_conv", "dev", "W", "bs",
                 "gpts", "tall", "bufs", "n_small", "head_fused", "forward_only", "flat", "gaddr", "step_word",
                 "jobs", "jaddr", "jb", "njobs", "loss_mode", "sse_split", "poolbits", "xagg")


class FusedTrainStep:
    """One training step of the reference's loop as FOUR enqueued launches on small-graph tiles (conv stack + pooling +
    readout head, two conv backward launches, the step tail), n_conv + 4 in general -- no autograd, no host sync.

        step = FusedTrainStep(model)            # model: hcatgnet_amd.GCN on the GPU
        loss = step(batch)                      # 0-d device tensor: sqrt(MSE) of this batch, weights already updated

    `rmse=True` is the reference's `torch.sqrt(model.loss(...))`; `optimizer_step=False` stops after the backward
    (gradients in `model.parameters()[i].grad`, views of one flat buffer); `grad_sync` is called with the flat gradient
    between backward and optimiser (data parallel: `DataParallelGCN.attach(step)` sets it).

    The loss scale is DEFERRED: the backward is linear in dloss/dout = scale * (out - y), so the head and every conv
    backward launch run on the unscaled error, the head leaves one partial sum of squared errors per workgroup, and the
    step's last launch (`hcg_step_tail`) derives loss and scale and applies the scale while it reduces the gradient slabs
    -- there is no grid-wide exchange anywhere in the step (csrc/head_tile.h, csrc/reduce.hip).

    `combine` (data parallel, needs rmse): "mean" = every rank's own sqrt(MSE), gradients averaged (DDP convention);
    "sse" = gradients stay those of SSE / 2, [SSE, count] sit behind the flat buffer, `grad_sync` SUMS all of it over the
    ranks and ONE scale 1 / (count * sqrt(SSE / count)) gives the gradient of sqrt(MSE) over the concatenated batch of all
    ranks -- what the reference's step computes on one device (utils/utils_model.py:64-65); the loss returned is then that
    global loss.
    """

    # development switches for A/B measurements (tools/ab_env.sh sets them from the environment; never set in product
    # code): POOLBITS = False stores the pooled layer's activations as the plain forms do, PREMASK = False leaves every
    # activation derivative to the layer that owns it, HEAD_IN_FORWARD = False keeps the head a launch of its own
    POOLBITS = True
    XAGG_MID = True                # ... and behind the one-graph-per-workgroup kernels (batches under functional.TALL_MIN_NODES_D64)
    XAGG = True                    # first layer on the wide-layer route: Ahat x + sign pieces from the forward, one dense backward launch
    TALL_PREMASK = False           # (measured: the dense dx kernel's strided mask loads cost what the layer below saves -- DESIGN 7)
    PREMASK = True
    HEAD_IN_FORWARD = True
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
        self._capturing_split = False
        self._side_streams = {}
        # pipelined loaders: a pointers-only `BatchPlan` of the NEXT batch (built once with validate=False); the step's last
        # launch re-derives its graph_ptr / edge_ptr from the tensors' current contents, so the next step -- on a batch object
        # that carries that plan (`batch._hcg_plan = plan`) -- starts without a plan launch
        self.next_plan = None
        # data parallel: a `xgmi.OneShotExchange` (set by its `attach`): the gradient exchange then happens INSIDE the step's
        # last launch instead of as an RCCL collective between two launches
        self.exchange = None
        self._last_carried = False              # the last step's tail launch carried reduction + (exchange) + Adam
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

    # ------------------------------------------------------------------ buffers
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
            # head slabs: the stand-alone head's (<= one per CU) or the forward tail's (one per workgroup of the tile launch:
            # graphs_per_tile 1 bounds it)
            hb = max(lib.hcg_head_workspace_bytes(capB, D if lib.hcg_head_supported(D, C) else 64),
                     lib.hcg_fused_aux_bytes(_lib.HCG_FUSED_HEAD_WS, capB, 1))
            cap = {"sig": sig, "N": capN, "B": capB,
                   "acts": [torch.empty(capN, D, **f32) for _ in range(n_conv)],
                   "dacts": [torch.empty(capN, D, **f32) for _ in range(n_conv - 1)],
                   "emb": torch.empty(capB, 2 * D, **f32), "demb": torch.empty(capB, 2 * D, **f32),
                   "z": torch.empty(capB, D, **f32), "out": torch.empty(capB, C, **f32), "loss": torch.zeros(2, **f32),
                   "ws_head": torch.empty(hb, dtype=torch.uint8, device=dev), "ws": {}}
            self._bufs = {"cap": cap}
            self._graph = None                          # a captured graph holds the old buffers' addresses
        b = self._bufs.get(key)
        if b is None:
            b = {"acts": [t[:N] for t in cap["acts"]], "dacts": [t[:N] for t in cap["dacts"]], "emb": cap["emb"][:B],
                 "demb": cap["demb"][:B], "z": cap["z"][:B], "out": cap["out"][:B], "loss": cap["loss"],
                 "ws_head": cap["ws_head"], "ws_head_bytes": cap["ws_head"].numel(), "ws": cap["ws"]}
            self._bufs = {"cap": cap, key: b}          # views of the current shape (one live shape at a time)
        return b

    def _tall_ws(self, bufs, l, N, B, F, D, dev):
        """Workspace of layer l's wide-layer kernels (H / dH round trip + gradient slabs): forward and backward share it."""
        wsb = _lib.load().hcg_tall_workspace_bytes(N, B, F, D)
        ws = bufs["ws"].get(("tall", l))
        if ws is None or ws.numel() < wsb:
            ws = bufs["ws"][("tall", l)] = torch.empty(int(wsb * 1.25), dtype=torch.uint8, device=dev)
        return ws, wsb

    def _ws(self, bufs, key, nbytes, dev):
        ws = bufs["ws"].get(key)
        if (ws is None or ws.numel() < nbytes) and nbytes > 0:
            ws = bufs["ws"][key] = torch.empty(int(nbytes * 1.25), dtype=torch.uint8, device=dev)
        return ws

    def _head_buffers(self, bufs, B, D, C, dev):
        """Scratch of the any-shape head (five launches): allocated once per capacity."""
        lib = _lib.load()
        hb = bufs["ws"].get("head_generic")
        if hb is None or hb["B"] < B:
            f32 = dict(dtype=torch.float32, device=dev)
            u8 = dict(dtype=torch.uint8, device=dev)
            hb = {"B": B, "dout": torch.empty(B, C, **f32), "dz": torch.empty(B, D, **f32),
                  "dz_ws1": torch.empty(B, C, **f32), "dz_ws0": torch.empty(B, D, **f32),
                  "ws1": torch.empty(max(lib.hcg_general_workspace_bytes(_lib.HCG_WS_LINEAR, B, D, C, 0), 256), **u8),
                  "ws0": torch.empty(max(lib.hcg_general_workspace_bytes(_lib.HCG_WS_LINEAR, B, 2 * D, D, 0), 256), **u8)}
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

    # ------------------------------------------------------------------ the step, piece by piece
    def _prepare(self, batch, forward_only: bool) -> _Ctx:
        model, lib = self.model, _lib.load()
        c = _Ctx()
        c.batch, c.forward_only = batch, forward_only
        x, y = batch.x, batch.y
        _lib.require_gpu(x, y, batch.edge_index)
        c.x = x = HF._f32c(x)
        c.plan = plan = model._plan_for(batch, x, batch.edge_index, batch.batch, None)
        c.convs = convs = [model.conv1] + list(model.conv_layers)
        c.l0, c.l1 = model.readout[0][0], model.readout[1]
        c.N, c.F, c.B, c.D, c.C = x.shape[0], x.shape[1], plan.B, model.embedding_dim, model._n_classes
        c.n_conv, c.dev = len(convs), x.device
        c.y2 = HF._f32c(y).reshape(c.B, -1)
        if c.y2.shape[1] != c.C:
            raise ValueError(f"targets have {c.y2.shape[1]} columns, the model predicts {c.C}")
        # kernel family per layer: gpt > 0 = small-graph tiles (csrc/fused.hip), 0 = one graph per workgroup (csrc/mid.hip)
        c.gpts = [HF.fused_graphs_per_tile(plan, cv.in_channels, cv.out_channels) for cv in convs]
        for cv, gpt in zip(convs, c.gpts):
            if gpt <= 0 and not HF.mid_supported(plan, cv.in_channels, cv.out_channels):
                raise _lib.HcgError("FusedTrainStep: graph / layer shape outside the fused kernels")
        # 128-wide layers over large graphs: dense row-streaming transform + per-graph segmented sum (csrc/tall.hip)
        c.tall = [gpt <= 0 and getattr(cv, "family", "auto") != "mid" and HF.tall_supported(plan, cv.in_channels, cv.out_channels)
                  for cv, gpt in zip(convs, c.gpts)]
        c.bufs = self._buffers((c.N, c.B, c.F, plan.E), c.N, c.B, c.F, c.D, c.C, c.n_conv, c.dev)
        c.W = [HF._f32c(cv.lin.weight) for cv in convs]
        c.bs = [HF._f32c(cv.bias) for cv in convs]
        c.head_fused = bool(lib.hcg_head_supported(c.D, c.C))
        c.n_small = self._size_groups(batch, plan, convs, c.D, c.C, c.n_conv)
        c.jb = _lib.job_bytes()
        c.jobs = ctypes.create_string_buffer(c.jb * _lib.HCG_REDUCE_MAX_JOBS)
        c.jaddr, c.njobs = ctypes.addressof(c.jobs), 0
        c.flat = c.gaddr = c.step_word = None
        c.poolbits = None
        c.xagg = None
        # how the loss scale reaches the gradients (see the class docstring).  "sse" with a collective between backward and
        # update: the tail leaves the gradients unscaled and [SSE, count] behind the flat buffer; everything else: the tail
        # applies this rank's own scale ("sse" without any exchange IS sqrt(MSE) of the own batch)
        dp_sync = self.grad_sync is not None or self.exchange is not None or self._capturing_split
        c.sse_split = self.combine == "sse" and dp_sync and not forward_only
        c.loss_mode = _lib.HCG_LOSS_SSE if c.sse_split else (_lib.HCG_LOSS_RMSE if self.rmse else _lib.HCG_LOSS_MSE)
        if not forward_only:
            params = self._trainable()
            c.flat = self._flat_grads(params, c.dev)
            # address of every parameter's slice of the flat gradient buffer (cached per buffer: eight tensor slices per
            # step otherwise)
            ga = getattr(self, "_gaddr", None)
            if ga is None or ga[0] != c.flat.data_ptr() or ga[1] is not params:
                addr, off = {}, 0
                for q in params:
                    addr[id(q)] = c.flat.data_ptr() + 4 * off
                    off += q.numel()
                ga = self._gaddr = (c.flat.data_ptr(), params, addr)
            c.gaddr = ga[2]
            # without a collective between backward and update, the step's last launch applies Adam itself; the head
            # advances the step number that launch reads
            opt = model.optimizer
            # (a one-shot exchange rides in that launch only when all its polling workgroups are resident at once: the
            #  64-wide model's slabs need ~540 of ~1800; the 128-wide one's ~2100 -- it keeps the collective, see
            #  hcg_xchg_resident_blocks)
            if (self.optimizer_step and self.grad_sync is None and not self._capturing_split and c.head_fused
                    and hasattr(opt, "fused_update_ready") and (self.exchange is None or c.D == 64)):
                c.step_word = opt.fused_update_ready(c.flat)
        return c

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

    def _g(self, c: _Ctx, prm) -> int:
        return c.gaddr[id(prm)]

    def _job_slot(self, c: _Ctx) -> int:
        return c.jaddr + c.njobs * c.jb

    def _tiles_args(self, c: _Ctx, B, gpt, **extra):
        plan = c.plan
        return dict(x=c.x, W1=c.W[0], b1=c.bs[0], W2=c.W[1], b2=c.bs[1], edge_index=plan.edge_index, E=plan.E,
                    graph_ptr=plan.graph_ptr, edge_ptr=plan.edge_ptr, N=c.N, B=B, F=c.F, D=c.D, graphs_per_tile=gpt,
                    apply_act=1, slope=HF.LEAKY_SLOPE, out1=c.bufs["acts"][0], emb=c.bufs["emb"], status=plan.status, **extra)

    def _head_in_forward(self, c: _Ctx) -> bool:
        """The C3 form: both conv layers on small-graph tiles, pooled layer on chip, one-launch head -> everything up to
        the loss is ONE launch."""
        return (self.HEAD_IN_FORWARD and self.POOLBITS and c.n_small is None and c.n_conv == 2 and c.head_fused
                and c.gpts[0] == c.gpts[1] and c.gpts[0] > 0)

    def _forward_with_head(self, c: _Ctx):
        """conv stack + pooling + readout head (forward, squared error, unscaled readout backward): one launch."""
        lib, bufs, gpt = _lib.load(), c.bufs, c.gpts[0]
        c.poolbits = self._ws(bufs, "poolbits", lib.hcg_fused_aux_bytes(_lib.HCG_FUSED_POOLBITS, c.B, gpt), c.dev)
        l0, l1 = c.l0, c.l1
        _lib.fused_forward(**self._tiles_args(
            c, c.B, gpt, poolbits=c.poolbits, y=c.y2, head_W0=HF._f32c(l0.weight), head_b0=HF._f32c(l0.bias),
            head_W1=HF._f32c(l1.weight), head_b1=HF._f32c(l1.bias), C=c.C, z=bufs["z"], out=bufs["out"], demb=bufs["demb"],
            head_workspace=bufs["ws_head"], head_workspace_bytes=bufs["ws_head_bytes"], step_counter=c.step_word,
            head_flags=_lib.HCG_HEAD_FORWARD_ONLY if c.forward_only else 0))
        g = (lambda q: self._g(c, q)) if not c.forward_only else (lambda q: None)
        _lib.check(lib.hcg_fused_head_reduce_job(_lib.ptr(bufs["ws_head"]), bufs["ws_head_bytes"], c.B, gpt, c.C, g(l0.weight),
                                                 g(l0.bias), g(l1.weight), g(l1.bias), self._job_slot(c)),
                   "hcg_fused_head_reduce_job")
        c.njobs += 1

    def _forward_layers(self, c: _Ctx):
        """The conv stack (+ pooling) of every other shape: one launch per layer (stacked pair: one)."""
        lib, plan, bufs = _lib.load(), c.plan, c.bufs
        p, stream, slope = _lib.ptr, _lib.stream_ptr(), HF.LEAKY_SLOPE
        acts, emb, gpts, n_conv, N, B, D = bufs["acts"], bufs["emb"], c.gpts, c.n_conv, c.N, c.B, c.D
        mxn, mxe = plan.max_nodes, plan.max_edges
        # small-graph tiles: the pooled layer's activations stay on chip, two bits per element (sign, is-the-column-max)
        # are all its backward needs of them
        if gpts[-1] > 0 and self.POOLBITS:
            c.poolbits = self._ws(bufs, "poolbits", lib.hcg_fused_aux_bytes(_lib.HCG_FUSED_POOLBITS, B, gpts[-1]), c.dev)
        if n_conv == 2 and gpts[0] == gpts[1] and gpts[0] > 0:
            _lib.fused_forward(**self._tiles_args(c, B, gpts[0], poolbits=c.poolbits,
                                                  out2=None if c.poolbits is not None else acts[1]))
            return
        h = c.x
        for l in range(n_conv):
            pe = emb if l == n_conv - 1 else None
            Fl = h.shape[1]
            if gpts[l] > 0:
                bits = c.poolbits if pe is not None else None
                _lib.fused_forward(x=h, W1=c.W[l], b1=c.bs[l], edge_index=plan.edge_index, E=plan.E, graph_ptr=plan.graph_ptr,
                                   edge_ptr=plan.edge_ptr, N=N, B=B, F=Fl, D=D, graphs_per_tile=gpts[l], apply_act=1,
                                   slope=slope, out1=None if bits is not None else acts[l], emb=pe, poolbits=bits,
                                   status=plan.status)
            elif c.tall[l]:
                ws, wsb = self._tall_ws(bufs, l, N, B, Fl, D, c.dev)
                bits = None
                if pe is not None and self.POOLBITS and Fl <= (64 if D == 64 else 128):
                    # the pooled layer's activations stay on chip: one byte per (row, 4 columns) -- sign, is-the-column-max --
                    # is all its backward (csrc/tall.hip: k_gseg_bwd) needs of them
                    bits = c.poolbits = self._ws(bufs, "poolbits_tall", N * (D // 4), c.dev)
                xagg = signs = None
                if (l == 0 and n_conv >= 2 and (Fl <= 64 or D == 128) and self.XAGG and not c.forward_only and c.tall[1]):
                    # training form of the FIRST layer: Ahat x [N, 32 | 64 | 128] and the sign pieces of its output leave too; its
                    # whole backward is then ONE dense launch (csrc/tall.hip: k_tall_dw<FIRST>) -- no transpose sum, no dH round trip
                    kp = 32 if Fl <= 32 else (64 if Fl <= 64 else 128)
                    xagg = self._ws(bufs, "xagg", N * kp * 4, c.dev)
                    signs = self._ws(bufs, "signs", N * (D // 8), c.dev)
                    c.xagg = (xagg, signs)
                rc = lib.hcg_tall_layer_fwd(p(h), p(c.W[l]), p(c.bs[l]), p(plan.edge_index), plan.E, p(plan.graph_ptr),
                                            p(plan.edge_ptr), N, B, Fl, D, mxn, mxe, slope, 1, None if bits is not None else p(acts[l]),
                                            p(pe), p(bits), p(xagg), p(signs), p(plan.status), p(ws), wsb, stream)
                _lib.check(rc, "hcg_tall_layer_fwd")
            else:
                xagg = signs = None
                if (l == 0 and n_conv >= 2 and D == 64 and Fl <= 64 and self.XAGG and self.XAGG_MID and not c.forward_only
                        and gpts[1] <= 0 and lib.hcg_tall_supported(Fl, D, mxn, mxe)):
                    kp = 32 if Fl <= 32 else 64            # (as on the wide-layer route: Ahat x + sign pieces for the dense backward)
                    xagg = self._ws(bufs, "xagg", N * kp * 4, c.dev)
                    signs = self._ws(bufs, "signs", N * (D // 8), c.dev)
                    c.xagg = (xagg, signs)
                rc = lib.hcg_mid_layer_fwd(p(h), p(c.W[l]), p(c.bs[l]), p(plan.edge_index), plan.E, p(plan.graph_ptr),
                                           p(plan.edge_ptr), N, B, Fl, D, mxn, mxe, slope, 1, p(acts[l]), p(pe), None, p(xagg), p(signs),
                                           p(plan.status), stream)
                _lib.check(rc, "hcg_mid_layer_fwd")
            h = acts[l]

    def _forward_routed(self, c: _Ctx, fork, join, stream_b):
        """Size-grouped batch: graphs [0, n_small) through the tiles (both layers + pooling in one launch, pooled layer on
        chip), graphs [n_small, B) one graph per wave -- every launch gets the sub-range of graph_ptr / edge_ptr / emb it
        owns; node rows are absolute, so x and the activations need no offsets."""
        lib, plan, bufs = _lib.load(), c.plan, c.bufs
        p, slope = _lib.ptr, HF.LEAKY_SLOPE
        Bs, Bb = c.n_small, c.B - c.n_small
        gp_b, ep_b = plan.graph_ptr.data_ptr() + 4 * Bs, plan.edge_ptr.data_ptr() + 4 * Bs
        acts, emb = bufs["acts"], bufs["emb"]
        emb_b = emb.data_ptr() + 4 * 2 * c.D * Bs
        gpt = int(lib.hcg_fused_graphs_per_tile(c.F, c.D, 32))
        c.poolbits = self._ws(bufs, "poolbits_r", lib.hcg_fused_aux_bytes(_lib.HCG_FUSED_POOLBITS, Bs, gpt), c.dev)
        mxn, mxe = plan.max_nodes, plan.max_edges
        fork()
        _lib.fused_forward(**self._tiles_args(c, Bs, gpt, poolbits=c.poolbits))
        _lib.check(lib.hcg_mid_layer_fwd(p(c.x), p(c.W[0]), p(c.bs[0]), p(plan.edge_index), plan.E, gp_b, ep_b, c.N, Bb, c.F, c.D,
                                         mxn, mxe, slope, 1, p(acts[0]), None, None, None, None, p(plan.status), stream_b), "hcg_mid_layer_fwd")
        _lib.check(lib.hcg_mid_layer_fwd(p(acts[0]), p(c.W[1]), p(c.bs[1]), p(plan.edge_index), plan.E, gp_b, ep_b, c.N, Bb, c.D,
                                         c.D, mxn, mxe, slope, 1, p(acts[1]), emb_b, None, None, None, p(plan.status), stream_b), "hcg_mid_layer_fwd")
        join()
        return gpt

    def _head(self, c: _Ctx):
        """The readout head as a launch of its own (one-launch kernel for D = 64 / 128, C <= 8; five launches of the
        any-shape kernels otherwise) -- reference model/gcn.py:70-71, utils/utils_model.py:64-65."""
        lib, bufs = _lib.load(), c.bufs
        p, stream, slope = _lib.ptr, _lib.stream_ptr(), HF.LEAKY_SLOPE
        B, D, C, l0, l1 = c.B, c.D, c.C, c.l0, c.l1
        W0, b0, W1, b1 = HF._f32c(l0.weight), HF._f32c(l0.bias), HF._f32c(l1.weight), HF._f32c(l1.bias)
        emb, z, out = bufs["emb"], bufs["z"], bufs["out"]
        if c.head_fused:
            rc = lib.hcg_head_fwd_bwd(p(emb), p(c.y2), p(W0), p(b0), p(W1), p(b1), B, D, C, slope,
                                      _lib.HCG_HEAD_FORWARD_ONLY if c.forward_only else 0, p(z), p(out), p(bufs["demb"]),
                                      p(bufs["ws_head"]), bufs["ws_head_bytes"], p(c.step_word), stream)
            _lib.check(rc, "hcg_head_fwd_bwd")
            g = (lambda q: self._g(c, q)) if not c.forward_only else (lambda q: None)
            _lib.check(lib.hcg_head_reduce_job(p(bufs["ws_head"]), bufs["ws_head_bytes"], B, D, C, g(l0.weight), g(l0.bias),
                                               g(l1.weight), g(l1.bias), self._job_slot(c)), "hcg_head_reduce_job")
            c.njobs += 1
            return
        # any-shape head (widths other than 64 / 128, more than 8 classes): Linear + LeakyReLU, Linear, loss with its
        # (scaled) gradient, two Linear backwards that write straight into the flat gradient buffer
        hb = self._head_buffers(bufs, B, D, C, c.dev)
        tail = self._flat_ext[c.flat.numel():] if c.sse_split else None
        _lib.check(lib.hcg_linear_fwd(p(emb), p(W0), p(b0), p(z), B, 2 * D, D, _lib.HCG_ACT_LEAKY, slope, stream), "hcg_linear_fwd")
        _lib.check(lib.hcg_linear_fwd(p(z), p(W1), p(b1), p(out), B, D, C, _lib.HCG_ACT_NONE, slope, stream), "hcg_linear_fwd")
        _lib.check(lib.hcg_loss_fwd_bwd(p(out), p(c.y2), B * C, c.loss_mode, p(bufs["loss"]), p(hb["dout"]), p(tail), stream),
                   "hcg_loss_fwd_bwd")
        if c.forward_only:
            return
        g = lambda q: self._g(c, q)
        _lib.check(lib.hcg_linear_bwd(p(hb["dout"]), p(out), p(z), p(W1), p(hb["dz"]), g(l1.weight), g(l1.bias), p(hb["dz_ws1"]),
                                      B, D, C, _lib.HCG_ACT_NONE, slope, p(hb["ws1"]), hb["ws1"].numel(), stream), "hcg_linear_bwd")
        _lib.check(lib.hcg_linear_bwd(p(hb["dz"]), p(z), p(emb), p(W0), p(bufs["demb"]), g(l0.weight), g(l0.bias), p(hb["dz_ws0"]),
                                      B, 2 * D, D, _lib.HCG_ACT_LEAKY, slope, p(hb["ws0"]), hb["ws0"].numel(), stream), "hcg_linear_bwd")

    def _backward_layers(self, c: _Ctx):
        """Conv stack backward, last layer first.  A fused-tile layer can hand its dx down already multiplied by the
        activation derivative of the layer below (it holds those rows anyway, for dW); that layer then never reads its own
        output: one tensor less per step."""
        lib, plan, bufs = _lib.load(), c.plan, c.bufs
        p, stream, slope = _lib.ptr, _lib.stream_ptr(), HF.LEAKY_SLOPE
        acts, emb, gpts, n_conv, N, B, D = bufs["acts"], bufs["emb"], c.gpts, c.n_conv, c.N, c.B, c.D
        mxn, mxe = plan.max_nodes, plan.max_edges
        g = lambda q: self._g(c, q)
        dh, premasked = None, False
        for l in reversed(range(n_conv)):
            inp = c.x if l == 0 else acts[l - 1]
            Fl = inp.shape[1]
            dx = bufs["dacts"][l - 1] if l > 0 else None
            small = gpts[l] > 0
            wsb = (lib.hcg_fused_workspace_bytes(B, Fl, D, gpts[l]) if small else
                   (0 if c.tall[l] else lib.hcg_mid_workspace_bytes(B, Fl, D, mxn, mxe)))
            ws = self._ws(bufs, l, wsb, c.dev)
            last = l == n_conv - 1
            cv = c.convs[l]
            geo = (p(plan.edge_index), plan.E, p(plan.graph_ptr), p(plan.edge_ptr), N, B, Fl, D)
            if small:
                bits = c.poolbits if last else None
                act = 1 if (last or not premasked) else 0
                premasked = self.PREMASK and l > 0
                a_out = p(acts[l]) if (bits is None and (act or last)) else None
                rc = lib.hcg_fused_layer_bwd(None if last else p(dh), p(bufs["demb"]) if last else None,
                                             p(emb) if (last and bits is None) else None, a_out, p(bits), p(inp), p(c.W[l]), *geo,
                                             gpts[l], slope, act | (2 if premasked else 0), p(dx), p(plan.status), p(ws), wsb, stream)
                _lib.check(rc, "hcg_fused_layer_bwd")
                _lib.check(lib.hcg_fused_reduce_job(p(ws), wsb, N, B, Fl, D, gpts[l], g(cv.lin.weight), g(cv.bias),
                                                    self._job_slot(c)), "hcg_fused_reduce_job")
            else:
                up = (None if last else p(dh), p(bufs["demb"]) if last else None, p(emb) if last else None)
                act = 0 if premasked else 1
                if c.tall[l]:
                    # (TALL_PREMASK: the dense dx kernel multiplies by leaky'(x) in its epilogue -- 4-byte strided loads of the
                    #  rows it holds -- and the layer below reads ONE tensor instead of two)
                    premasked = self.TALL_PREMASK and self.PREMASK and l > 0
                    tws, twsb = self._tall_ws(bufs, l, N, B, Fl, D, c.dev)
                    bits = c.poolbits if last else None          # (the forward's bit form stands in for the layer's output and emb)
                    if bits is not None:
                        up = (None, p(bufs["demb"]), None)
                    first = c.xagg if (l == 0 and not last and dx is None) else None     # (Ahat x, sign pieces) of the forward
                    a_out = p(acts[l]) if ((act or last) and bits is None and first is None) else None
                    rc = lib.hcg_tall_layer_bwd(*up, a_out, p(bits), p(first[0]) if first else None, p(first[1]) if first else None,
                                                None, p(inp), p(c.W[l]), *geo, mxn, mxe, slope, act | (2 if premasked else 0), p(dx),
                                                p(plan.status), p(tws), twsb, stream)
                    _lib.check(rc, "hcg_tall_layer_bwd")
                    _lib.check(lib.hcg_tall_reduce_jobs(p(tws), twsb, N, B, Fl, D, 1 if first else 0, g(cv.lin.weight), g(cv.bias),
                                                        self._job_slot(c)), "hcg_tall_reduce_jobs")
                    c.njobs += 1                          # (two jobs: dW, db)
                elif l == 0 and not last and c.xagg is not None:
                    # the first layer's backward as ONE dense launch over the forward's Ahat x (csrc/tall.hip: k_tall_dw<FIRST>) also
                    # behind the one-graph-per-workgroup kernels; the batch's node count is read on the device (graph_ptr[B]): a
                    # captured epoch's slot has a CAPACITY of rows
                    premasked = False
                    tws, twsb = self._tall_ws(bufs, l, N, B, Fl, D, c.dev)
                    rc = lib.hcg_tall_layer_bwd(p(dh), None, None, None, None, p(c.xagg[0]), p(c.xagg[1]),
                                                plan.graph_ptr.data_ptr() + 4 * B, p(inp), p(c.W[l]), *geo, mxn, mxe, slope, act, None,
                                                p(plan.status), p(tws), twsb, stream)
                    _lib.check(rc, "hcg_tall_layer_bwd")
                    _lib.check(lib.hcg_tall_reduce_jobs(p(tws), twsb, N, B, Fl, D, 1, g(cv.lin.weight), g(cv.bias),
                                                        self._job_slot(c)), "hcg_tall_reduce_jobs")
                    c.njobs += 1                          # (two jobs: dW, db)
                else:
                    premasked = self.PREMASK and l > 0
                    rc = lib.hcg_mid_layer_bwd(*up, p(acts[l]) if (act or last) else None, p(inp), p(c.W[l]), *geo, mxn, mxe,
                                               slope, act | (2 if premasked else 0), p(dx), p(plan.status), p(ws), wsb, stream)
                    _lib.check(rc, "hcg_mid_layer_bwd")
                    for half in range(D // 64):      # one slab set (= one job) per 64-column half
                        if half > 0:
                            c.njobs += 1
                        _lib.check(lib.hcg_mid_reduce_job(p(ws), wsb, B, Fl, D, mxn, mxe, half, g(cv.lin.weight), g(cv.bias),
                                                          self._job_slot(c)), "hcg_mid_reduce_job")
            c.njobs += 1
            dh = dx

    def _backward_routed(self, c: _Ctx, gpt, fork, join, stream_b):
        """Size-grouped batch: each layer's backward = tiles on the small graphs + waves on the others; both groups' slabs of
        a layer sit back to back and are ONE reduction job."""
        lib, plan, bufs = _lib.load(), c.plan, c.bufs
        p, stream, slope = _lib.ptr, _lib.stream_ptr(), HF.LEAKY_SLOPE
        Bs, Bb, N, D = c.n_small, c.B - c.n_small, c.N, c.D
        gp, ep = plan.graph_ptr, plan.edge_ptr
        gp_b, ep_b = gp.data_ptr() + 4 * Bs, ep.data_ptr() + 4 * Bs
        acts, emb, demb = bufs["acts"], bufs["emb"], bufs["demb"]
        emb_b, demb_b = emb.data_ptr() + 4 * 2 * D * Bs, demb.data_ptr() + 4 * 2 * D * Bs
        mxn, mxe = plan.max_nodes, plan.max_edges
        g = lambda q: self._g(c, q)
        tmp = ctypes.create_string_buffer(c.jb)
        taddr = ctypes.addressof(tmp)
        premask = bool(self.PREMASK)
        dx = bufs["dacts"][0]
        fork()
        for l in (1, 0):
            inp = c.x if l == 0 else acts[0]
            Fl = inp.shape[1]
            ws_a = lib.hcg_fused_workspace_bytes(Bs, Fl, D, gpt)
            ws_b = lib.hcg_mid_workspace_bytes(Bb, Fl, D, mxn, mxe)
            off_b = ws_a - 256                                  # = the tile launch's slabs, exactly (the query pads by 256)
            ws = self._ws(bufs, ("r", l), off_b + ws_b, c.dev)
            wsb_ptr = ws.data_ptr() + off_b
            geo_a = (p(plan.edge_index), plan.E, p(gp), p(ep), N, Bs, Fl, D)
            geo_b = (p(plan.edge_index), plan.E, gp_b, ep_b, N, Bb, Fl, D)
            cv = c.convs[l]
            if l == 1:
                flags = 1 | (2 if premask else 0)
                _lib.check(lib.hcg_fused_layer_bwd(None, p(demb), None, None, p(c.poolbits), p(inp), p(c.W[l]), *geo_a, gpt, slope,
                                                   flags, p(dx), p(plan.status), p(ws), off_b, stream), "hcg_fused_layer_bwd")
                _lib.check(lib.hcg_mid_layer_bwd(None, demb_b, emb_b, p(acts[1]), p(inp), p(c.W[l]), *geo_b, mxn, mxe, slope, flags,
                                                 p(dx), p(plan.status), wsb_ptr, ws_b, stream_b), "hcg_mid_layer_bwd")
            else:
                act = 0 if premask else 1
                a_out = p(acts[0]) if act else None
                _lib.check(lib.hcg_fused_layer_bwd(p(dx), None, None, a_out, None, p(inp), p(c.W[l]), *geo_a, gpt, slope, act, None,
                                                   p(plan.status), p(ws), off_b, stream), "hcg_fused_layer_bwd")
                _lib.check(lib.hcg_mid_layer_bwd(p(dx), None, None, a_out, p(inp), p(c.W[l]), *geo_b, mxn, mxe, slope, act, None,
                                                 p(plan.status), wsb_ptr, ws_b, stream_b), "hcg_mid_layer_bwd")
            _lib.check(lib.hcg_fused_reduce_job(p(ws), off_b, N, Bs, Fl, D, gpt, g(cv.lin.weight), g(cv.bias), self._job_slot(c)),
                       "hcg_fused_reduce_job")
            _lib.check(lib.hcg_mid_reduce_job(wsb_ptr, ws_b, Bb, Fl, D, mxn, mxe, 0, g(cv.lin.weight), g(cv.bias), taddr),
                       "hcg_mid_reduce_job")
            _lib.check(lib.hcg_reduce_job_append(self._job_slot(c), taddr), "hcg_reduce_job_append")
            c.njobs += 1
        join()

    def _tail(self, c: _Ctx):
        """The step's last launch: slab reductions -> ONE flat gradient, the loss and its scale, (exchange,) update, the next
        batch's plan.  With a collective between backward and update the launch stops at the gradient."""
        opt, bufs = self.model.optimizer, c.bufs
        count = float(c.B * c.C)
        self._last_carried = c.step_word is not None
        if c.step_word is not None:
            if self.exchange is not None and self.pre_exchange_hook is not None:
                self.pre_exchange_hook()
            if not opt.step_with_reduction(c.jaddr, c.njobs, c.flat, next_plan=self.next_plan, exchange=self.exchange,
                                           flat_ext=self._flat_ext, mode=self.combine, loss_buf=bufs["loss"],
                                           loss_mode=c.loss_mode, loss_count=count):
                raise _lib.HcgError("optimizer state changed between head launch and update")
            return
        sse_tail = self._flat_ext[c.flat.numel():] if (c.sse_split and c.head_fused) else None
        _lib.step_tail(c.jaddr, c.njobs, loss=bufs["loss"], loss_mode=c.loss_mode, loss_count=count, sse_tail=sse_tail)
        if self.next_plan is not None:
            self.next_plan.rebuild()                   # (no fused update to ride in: its own launch)
        if not self._capturing_split:
            self._exchange_and_update(bufs["loss"], c.sse_split)

    def _finish_forward_only(self, c: _Ctx):
        """Forward-only steps (the reference's eval_network body, utils/utils_model.py:75-78): the loss from the head's SSE
        partials, one tiny launch."""
        if c.head_fused:
            _lib.check(_lib.load().hcg_loss_finalize(c.jaddr, float(c.B * c.C), c.loss_mode, _lib.ptr(c.bufs["loss"]), None,
                                                     _lib.stream_ptr()), "hcg_loss_finalize")
        self.last_out = c.bufs["out"]
        return c.bufs["loss"][0]

    def evaluate(self, batch, _checked: bool = False):
        """Forward + loss only (the reference's `eval_network` body, utils/utils_model.py:75-78): nothing is reduced or
        updated."""
        return self(batch, _forward_only=True, _checked=_checked)

    def __call__(self, batch, _forward_only: bool = False, _checked: bool = False):
        if not _checked:                 # (the epoch loops below have just asked `reason(batch)` themselves)
            why = self.reason(batch)
            if why is not None:
                raise _lib.HcgError(f"FusedTrainStep does not cover this model/batch: {why}")
        c = self._prepare(batch, _forward_only)
        if c.n_small is not None:
            return self._routed_step(c)
        if self._head_in_forward(c):
            self._forward_with_head(c)
        else:
            self._forward_layers(c)
            self._head(c)
        if _forward_only:
            return self._finish_forward_only(c)
        self._backward_layers(c)
        self._tail(c)
        self.last_out = c.bufs["out"]
        return c.bufs["loss"][0]

    def _routed_step(self, c: _Ctx):
        """The step for a size-grouped batch.  The two groups' launches touch disjoint rows, graphs and slabs.  While the step
        is being CAPTURED the larger graphs' launches go to a second stream, forked before each conv phase and joined behind
        it = two branches of the hipGraph: each family's last workgroups fill the CUs the other has already left (neither
        fills the chip: ~1.4 tiles per wave / ~0.6 graphs per wave slot at 4096 graphs).  Measured on the ragged bench: replay
        0.1975 -> 0.188 ms/step; the eager step is host-bound and the four extra event calls cost it 0.196 -> 0.215, so eager
        steps stay on one stream."""
        dev = c.dev
        main_s = torch.cuda.current_stream(dev)
        side_s = self._side_stream(dev) if (self.OVERLAP_GROUPS and torch.cuda.is_current_stream_capturing()) else None
        stream_b = ctypes.c_void_p(side_s.cuda_stream) if side_s is not None else _lib.stream_ptr()

        def fork():
            if side_s is not None:
                side_s.wait_stream(main_s)

        def join():
            if side_s is not None:
                main_s.wait_stream(side_s)
        gpt = self._forward_routed(c, fork, join, stream_b)
        self._head(c)                                   # over all graphs
        if c.forward_only:
            return self._finish_forward_only(c)
        self._backward_routed(c, gpt, fork, join, stream_b)
        self._tail(c)
        self.last_out = c.bufs["out"]
        return c.bufs["loss"][0]

    def _exchange_and_update(self, loss_buf, sse_split: Optional[bool] = None):
        """Behind the step tail when a collective sits between backward and update: gradient exchange (data parallel), the
        "sse" scale, the optimiser."""
        lib, opt, flat = _lib.load(), self.model.optimizer, self._flat
        sync = self.grad_sync
        if sync is None and self.exchange is not None:
            # a one-shot exchange is attached but this step's update is not the fused launch that carries it (any-shape head,
            # optimiser state not one flat group): the ranks exchange through the collective instead -- never not at all
            sync = self.exchange_fallback_sync
            if sync is None:
                raise _lib.HcgError("a one-shot exchange is attached, this step cannot carry it and no collective fallback is "
                                    "set: the replicas would diverge")
        if sse_split is None:
            sse_split = self.combine == "sse" and sync is not None
        if sse_split:
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

    def __init__(self, steps, batches, forward_only: bool = False, counts64=None):
        """`forward_only`: the steps' `evaluate` form (plan, conv stack, head: loss only, nothing reduced or updated).
        `counts64` (float64 [len(steps)], graphs per batch): the epoch sum dot(losses, counts) -- `_weighted_loss_sum`, the same
        three operations -- and its copy into a pinned host scalar are recorded behind the last step, so reading an epoch's
        value is one stream synchronisation (`value()`), not four launches and a blocking copy."""
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
        self.value_host = None
        if counts64 is not None:
            self.value_host = torch.zeros(1, dtype=torch.float64).pin_memory()
            _weighted_loss_dev([counts64.new_zeros((), dtype=torch.float32) for _ in steps], counts64)   # (its kernels loaded before the capture)
            self.value_host.copy_(counts64[:1], non_blocking=True)
            torch.cuda.synchronize()
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.losses = [st(get(b), _forward_only=self.forward_only) for st, b in zip(steps, batches)]
            if counts64 is not None:
                self.value_dev = _weighted_loss_dev(self.losses, counts64)
                self.value_host.copy_(self.value_dev.reshape(1), non_blocking=True)
        self._fp = [st._graph_fingerprint() for st in steps]

    def value(self) -> float:
        """dot(losses, counts64) of the last replay (needs `counts64`): waits for the current stream, reads the pinned scalar."""
        torch.cuda.current_stream().synchronize()
        return float(self.value_host[0])

    def replay(self):
        """-> the steps' loss tensors (device scalars, overwritten by the next replay)."""
        if self._fp != [st._graph_fingerprint() for st in self.steps]:
            raise _lib.HcgError("StepWindow.replay(): parameter / optimiser / step buffers changed since the capture: build it again")
        if self.steps[0].optimizer_step and not self.forward_only:
            self.model.optimizer.sync_lr()
        self.graph.replay()
        return self.losses


class EpochWindow:
    """One training epoch of the reference's loop (utils/utils_model.py:55-70: every batch of a shuffled
    `DataLoader(dataset, batch_size, shuffle=True)`, call_methods.py:41-46, through zero_grad / forward / sqrt(MSE) /
    backward / Adam) as ONE hipGraph launch over a `store.DeviceLoader`.

    What makes an epoch capturable although its batches change shape with every shuffle: the NUMBER of batches and their
    graph counts are fixed (batch_size 40, options/base_options.py:269-274: 13 x 40 + 15 for 535 graphs), and every kernel
    of the fused step takes its sizes from graph_ptr / edge_ptr on the device, not from the launch arguments.  So every batch
    slot owns buffers of a fixed CAPACITY (the batch_size largest graphs of the dataset), the collate launch of slot i reads
    its graph ids and prefix sums from a fixed device buffer, and an epoch is: draw the permutation on the host (a few KB,
    the loader's own generator: the same sequence of permutations as iterating the loader), ONE host-to-device copy of the
    epoch's index arrays, ONE graph launch (collate + 4..6 launches per batch, weights carried from batch to batch), one
    reduction of the per-batch losses.  The per-batch loop of round 2 was host-bound at ~140 us per 77 us of kernels.

    Bitwise the per-batch loop on the same permutations (tests/test_gpu_train_step.py).  Applies when every layer of the
    model runs on a per-graph kernel family at capacity shape (not the dense row-streaming kernels of csrc/tall.hip, which
    walk all N rows of a batch); `build` returns None otherwise and the caller keeps its loop."""

    MAX_BATCHES = 64

    @staticmethod
    def build(model, loader):
        try:
            return EpochWindow(model, loader)
        except _lib.HcgError:
            return None

    def __init__(self, model, loader):
        import numpy as np
        from .batch import Batch
        from .plan import BatchPlan, _shared_status
        st, bs = loader.store, loader.batch_size
        G, dev, F = len(st), st.device, st.F
        starts = [i for i in range(0, G, bs) if not (loader.drop_last and G - i < bs)]
        if not starts or len(starts) > self.MAX_BATCHES or st.y_all is None:
            raise _lib.HcgError("EpochWindow: no batches, too many batches, or a dataset without targets")
        self.model, self.loader, self.G = model, loader, G
        self.Bs = [min(bs, G - i) for i in starts]
        self.starts = starts
        n_desc, e_desc = np.sort(st.n_host)[::-1], np.sort(st.e_host)[::-1]
        maxn, maxe = int(n_desc[0]), int(e_desc[0])
        words = lambda B: (B + 2) // 2                       # int64 words that hold B + 1 int32 (as DeviceLoader.__iter__)
        tot = sum(words(B) for B in self.Bs)
        self.tot = tot
        self.host = torch.zeros(G + 2 * tot, dtype=torch.int64).pin_memory()
        self.dbuf = torch.zeros(G + 2 * tot, dtype=torch.int64, device=dev)
        d32 = self.dbuf[G:].view(torch.int32)
        lib, p = _lib.load(), _lib.ptr
        self.batches, self.spans, slots, off = [], [], [], 0
        for i, B in zip(starts, self.Bs):
            Ncap, Ecap = int(n_desc[:B].sum()), max(int(e_desc[:B].sum()), 1)
            x = torch.zeros(Ncap, F, dtype=torch.float32, device=dev)
            ei = torch.zeros(2, Ecap, dtype=torch.int64, device=dev)
            bvec = torch.zeros(Ncap, dtype=torch.int64, device=dev)
            y = torch.zeros(B, dtype=torch.float32, device=dev)
            idx = torch.zeros(B, dtype=torch.int64, device=dev) if st.idx_all is not None else None
            ids_d, gp_d, ep_d = self.dbuf[i:i + B], d32[off:off + B + 1], d32[2 * tot + off:2 * tot + off + B + 1]
            batch = Batch(x, ei, bvec, B, y=y, idx=idx, max_nodes=maxn, max_edges=maxe, edges_grouped=True)
            plan = BatchPlan()
            plan.N, plan.E, plan.B, plan.fill, plan.mode = Ncap, Ecap, B, 1.0, "blocked"
            plan.edge_index, plan.batch, plan.edge_weight = ei, bvec, None
            plan.graph_ptr, plan.edge_ptr = gp_d, ep_d
            plan.max_nodes, plan.max_edges, plan.validated, plan.has_csr = maxn, maxe, True, False
            plan.shared_status, plan.status, plan.want_eid = True, _shared_status(dev), False
            plan.rowptr = plan.col = plan.eid = plan.rowptr_t = plan.col_t = plan.eid_t = None
            plan.dinv = plan.ew_csr = plan.ew_csc = plan.dinv_unw = None
            batch._hcg_plan = plan
            for c in [model.conv1] + list(model.conv_layers):
                if HF.fused_graphs_per_tile(plan, c.in_channels, c.out_channels) <= 0 and HF.tall_supported(plan, c.in_channels, c.out_channels):
                    raise _lib.HcgError("EpochWindow: a layer would run on the dense row-streaming kernels (capacity-padded rows)")
            self.batches.append(batch)
            self.spans.append((i, B, off))

            sl = _lib.CollateSlot()
            sl.ids, sl.graph_ptr, sl.edge_ptr, sl.x_out, sl.edge_index_out, sl.batch_out = p(ids_d), p(gp_d), p(ep_d), p(x), p(ei), p(bvec)
            sl.y_out, sl.idx_out, sl.B, sl.N_out, sl.E_out = p(y), p(idx), B, Ncap, Ecap
            slots.append(sl)
            off += 2 * words(B)
        # ONE collate launch for the whole epoch, in front of its first step (every slot owns its buffers; the launch reads the
        # slot descriptors from a device array): a collate per batch was 14 launches of 40 workgroups, a tenth of the epoch
        arr = (_lib.CollateSlot * len(slots))(*slots)
        self.slots_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        src = st.collate_args()
        self.cargs = _lib.CollateArgs.from_buffer_copy(src)
        self.cargs.nslots, self.cargs.slots_dev, self.cargs.max_B = len(slots), p(self.slots_dev), max(self.Bs)
        if len(slots) == 1:
            self.cargs.slot = slots[0]

        def first_batch():
            _lib.check(lib.hcg_collate(ctypes.byref(self.cargs), _lib.stream_ptr()), "hcg_collate")
            return self.batches[0]
        fns = [first_batch] + [(lambda b=b: b) for b in self.batches[1:]]
        self.steps = [FusedTrainStep(model) for _ in self.Bs]
        why = self.steps[0].reason(self.batches[0])
        if why is not None:
            raise _lib.HcgError(f"EpochWindow: {why}")
        self.counts = torch.tensor([float(B) for B in self.Bs], dtype=torch.float64, device=dev)
        # the capture's warm-up runs every step once: a hidden extra epoch.  It is undone: parameters, moments and the step
        # count go back to what they were, in place (same storages, so the captured addresses stay valid)
        self._layout(np.arange(G))
        opt = model.optimizer
        opt.enable_capturable()
        for q in model.parameters():                    # re-base onto the optimiser's flat storages BEFORE the snapshot
            _lib.require_gpu(q)
        fl = opt._flat.get(0)
        ps = [q for q in opt.param_groups[0]["params"] if q.requires_grad]
        if (fl is None or len(fl["params"]) != len(ps) or any(a is not b for a, b in zip(fl["params"], ps))
                or ps[0].data_ptr() != fl["p"].data_ptr()):
            with torch.no_grad():
                fl = opt._rebase(0, opt.param_groups[0])
        opt._make_dev_state(fl, opt.param_groups[0])
        saved = [fl["p"].clone(), fl["m"].clone(), fl["v"].clone(), fl["step_dev"].clone()]
        self.window = StepWindow(self.steps, fns, counts64=self.counts)
        fl2 = opt._flat.get(0)
        if fl2 is not fl:
            raise _lib.HcgError("EpochWindow: the optimiser re-based its state during the capture")
        torch.cuda.synchronize()
        with torch.no_grad():
            fl["p"].copy_(saved[0]); fl["m"].copy_(saved[1]); fl["v"].copy_(saved[2]); fl["step_dev"][:1].copy_(saved[3][:1])

    def _layout(self, order):
        """The epoch's index arrays [ids of every batch | graph_ptr of every batch | edge_ptr of every batch] -> device."""
        import numpy as np
        st, G, tot = self.loader.store, self.G, self.tot
        buf = self.host.numpy()
        buf[:] = 0
        buf[:G] = order
        ptr32 = buf[G:].view(np.int32)
        for i, B, off in self.spans:
            ids = order[i:i + B]
            ptr32[off + 1:off + B + 1] = np.cumsum(st.n_host[ids])
            ptr32[2 * tot + off + 1:2 * tot + off + B + 1] = np.cumsum(st.e_host[ids])
        self.dbuf.copy_(self.host, non_blocking=True)

    def draw_order(self):
        """The next permutation of the loader's generator (what iterating the loader would have drawn)."""
        import numpy as np
        ld = self.loader
        return ld.rng.permutation(self.G) if ld.shuffle else np.arange(self.G)

    def launch_epoch(self, order=None):
        """Issue the epoch on the current stream (index upload + ONE graph launch), no synchronisation; `finish_epoch` reads
        its value.  One epoch in flight per window (the pinned index buffer is rewritten by the next launch)."""
        if order is None:
            order = self.draw_order()
        self._layout(order)
        return self.window.replay()

    def finish_epoch(self, losses) -> float:
        """-> sum(loss_i * num_graphs_i) / len(dataset), the reference's `train_network` return value (synchronises)."""
        return self.window.value() / self.G        # (the sync also keeps the pinned index buffer from being rewritten too early)

    def run_epoch(self, order=None) -> float:
        return self.finish_epoch(self.launch_epoch(order))


# ---------------------------------------------------------------------------------------------------------------
# the reference's loops (same names / arguments / return values)
# ---------------------------------------------------------------------------------------------------------------
def _rmse_autograd(model, batch):
    out = model(batch)
    return torch.sqrt(model.loss(out, batch.y.unsqueeze(1)))


class _EpochSum:
    """sum(loss_i * num_graphs_i) of an epoch, on the device, in float64 like the reference's Python-float accumulation
    (`loss.item() * batch.num_graphs`, utils/utils_model.py:68): the per-batch losses are kept (one tiny copy each, no
    sync) and meet in ONE dot product at the end of the epoch -- the same operation the one-graph epoch forms
    (`EpochWindow`, `_eval_window`) apply to their loss buffers, so both forms return bitwise the same value."""

    def __init__(self):
        self.vals, self.ns = [], []

    def add(self, loss, num_graphs):
        self.vals.append(loss.detach().clone())
        self.ns.append(float(num_graphs))

    def value(self, denom) -> float:
        if not self.vals:
            return 0.0
        return _weighted_loss_sum(self.vals, torch.tensor(self.ns, dtype=torch.float64, device=self.vals[0].device)) / denom


def _weighted_loss_dev(losses, counts64):
    return torch.dot(torch.stack([v.reshape(()) for v in losses]).double(), counts64)


def _weighted_loss_sum(losses, counts64) -> float:
    """One synchronising read: dot(losses, counts) in float64."""
    return float(_weighted_loss_dev(losses, counts64).item())


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
    val = _epoch_window(model, train_loader)
    if val is not None:
        return val
    total = _EpochSum()
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
        total.add(loss, batch.num_graphs)
    return total.value(len(train_loader.dataset))


EPOCH_WINDOW = True      # train_network over a DeviceLoader: whole epochs as one hipGraph (EpochWindow); False = per-batch loop


def _epoch_window_launch(model, loader):
    """Issue one epoch through the loader's cached `EpochWindow` (built on first use; rebuilt when the model or the
    optimiser's storages changed) on the current stream -> (window, losses) for `window.finish_epoch(losses)`, or None
    when that form does not apply."""
    from .store import DeviceLoader
    if not (EPOCH_WINDOW and isinstance(loader, DeviceLoader) and hasattr(model.optimizer, "enable_capturable")):
        return None
    if dist_initialized_multi():
        return None                          # (data parallel epochs keep the per-batch loop: the exchange form is the caller's)
    order = None
    for attempt in range(2):
        cache = getattr(loader, "_hcg_epoch_window", None)
        key = (id(model), loader.batch_size, loader.drop_last, len(loader.store))
        if cache is None or cache[0] != key:
            win = EpochWindow.build(model, loader)
            cache = (key, win)
            try:
                loader._hcg_epoch_window = cache
            except Exception:
                pass
        if cache[1] is None:
            return None
        try:
            if order is None:
                order = cache[1].draw_order()            # (drawn once: a rebuild must not skip a permutation)
            return cache[1], cache[1].launch_epoch(order)
        except _lib.HcgError:                 # parameters / optimiser re-based since the capture: build again
            try:
                loader._hcg_epoch_window = None
            except Exception:
                return None
    return None


def _epoch_window(model, loader):
    """-> the epoch's value through `_epoch_window_launch`, or None when that form does not apply."""
    got = _epoch_window_launch(model, loader)
    return None if got is None else got[0].finish_epoch(got[1])


def dist_initialized_multi() -> bool:
    return torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1


EVAL_WINDOW_MAX_BATCHES = 64


def _eval_window(model, loader, fused, launch_only: bool = False):
    """(`launch_only`: issue the graph on the current stream and return its window without synchronising: `window.value()`.)
    A `store.DeviceLoader` that does not shuffle yields the same batches every epoch (the reference's validation / test
    loaders, call_methods.py:41-46): they are collated once, their `evaluate` steps captured as ONE hipGraph
    (`StepWindow(forward_only=True)`) and an `eval_network` call is one graph launch + one reduction instead of
    (collate + 3 launches) per batch.  -> the epoch's value, or None when this path does not apply (the caller loops)."""
    from .store import DeviceLoader
    if not (isinstance(loader, DeviceLoader) and not loader.shuffle and 0 < len(loader) <= EVAL_WINDOW_MAX_BATCHES):
        return None
    cache = getattr(loader, "_hcg_eval_window", None)
    key = (loader.batch_size, loader.drop_last, len(loader.store), bool(getattr(model, "use_fused", True)))
    for attempt in range(2):
        if cache is None or cache["model"] is not model or cache.get("key") != key:
            batches = list(loader)
            if not all(fused.reason(b) is None for b in batches):
                return None
            try:
                steps = [FusedTrainStep(model, optimizer_step=False) for _ in batches]
                counts = torch.tensor([float(b.num_graphs) for b in batches], dtype=torch.float64, device=batches[0].x.device)
                win = StepWindow(steps, batches, forward_only=True, counts64=counts)
            except (_lib.HcgError, RuntimeError):        # capture failed (memory, an unsupported launch): the per-batch loop runs
                return None
            cache = {"model": model, "key": key, "window": win, "batches": batches, "counts": counts}
            try:
                loader._hcg_eval_window = cache
            except Exception:
                pass
        try:
            losses = cache["window"].replay()
        except _lib.HcgError:                 # parameters re-based since the capture (load_state_dict on new storages ...)
            cache = None
            continue
        if launch_only:
            return cache["window"]
        return cache["window"].value() / len(loader.dataset)
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
    total = _EpochSum()
    with torch.no_grad():
        for batch in loader:
            batch = batch.to(device)
            if fused.reason(batch) is None:
                loss = fused.evaluate(batch, _checked=True)
            else:
                loss = _rmse_autograd(model, batch)
            total.add(loss, batch.num_graphs)
    return total.value(len(loader.dataset))


def _fused_of(model):
    fused = getattr(model, "_hcg_train_step", None)
    if fused is None:
        fused = FusedTrainStep(model)
        try:
            model._hcg_train_step = fused
        except Exception:
            pass
    return fused


def _run_stream(model, device):
    """The HIP stream of an independent run (one per model, made on first use)."""
    st = getattr(model, "_hcg_run_stream", None)
    if st is None:
        st = torch.cuda.Stream(device=device)
        try:
            model._hcg_run_stream = st
        except Exception:
            pass
    return st


def train_networks(models, train_loaders, device, active=None):
    """One epoch of SEVERAL independent runs at once -> [train_network(model_k, loader_k, device) for every k] (None where
    `active[k]` is false: a run that has stopped early).

    The reference's nested cross-validation trains folds * (folds - 1) = 90 models one after another
    (scripts_experiments/train_GNN.py:48-50), each on ~535 graphs in batches of 40 (options/base_options.py:269-274): a
    batch of 40 graphs occupies 40 of the 256 CUs, so one run cannot fill the GPU whatever its kernels do.  The runs share
    nothing (own model, own optimiser, own loaders), so here every run issues its epoch -- ONE hipGraph (`EpochWindow`) --
    on its own HIP stream and the GPU overlaps them; the host synchronises once per run and epoch, after all of them are in
    flight.  Each run's arithmetic is untouched: bitwise what `train_network` gives that run alone
    (tests/test_gpu_train_step.py::test_concurrent_runs_equal_the_runs_one_by_one).  A run whose epoch does not fit the
    one-graph form (a host loader, a layer on the dense row-streaming kernels) is trained by `train_network` in turn."""
    K = len(models)
    if len(train_loaders) != K or (active is not None and len(active) != K):
        raise ValueError("train_networks: one loader (and one `active` flag) per model")
    on = [bool(a) for a in active] if active is not None else [True] * K
    out, flying = [None] * K, []
    cur = torch.cuda.current_stream(device)
    for k, (m, ld) in enumerate(zip(models, train_loaders)):
        if not on[k]:
            continue
        m.train()
        _fused_of(m)
        st = _run_stream(m, device)
        st.wait_stream(cur)                         # (whatever the caller did to this run's model on its own stream)
        with torch.cuda.stream(st):
            got = _epoch_window_launch(m, ld)
        if got is None:
            out[k] = train_network(m, ld, device)
        else:
            flying.append((k, st, got))
    for k, st, (win, losses) in flying:
        with torch.cuda.stream(st):
            out[k] = win.finish_epoch(losses)
        cur.wait_stream(st)
    return out


def eval_networks(models, loaders, device, active=None):
    """`eval_network` of several independent runs at once (see `train_networks`): every run's evaluation epoch -- one
    forward-only hipGraph over its fixed loader -- on the run's own stream.  -> [eval_network(model_k, loader_k, device)]."""
    K = len(models)
    if len(loaders) != K or (active is not None and len(active) != K):
        raise ValueError("eval_networks: one loader (and one `active` flag) per model")
    on = [bool(a) for a in active] if active is not None else [True] * K
    out, flying = [None] * K, []
    cur = torch.cuda.current_stream(device)
    for k, (m, ld) in enumerate(zip(models, loaders)):
        if not on[k]:
            continue
        m.eval()
        fused = _fused_of(m)
        st = _run_stream(m, device)
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            got = _eval_window(m, ld, fused, launch_only=True)
        if got is None:
            out[k] = eval_network(m, ld, device)
        else:
            flying.append((k, st, got, len(ld.dataset)))
    for k, st, win, n in flying:
        with torch.cuda.stream(st):
            out[k] = win.value() / n
        cur.wait_stream(st)
    return out


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
