"""autograd Functions over the C ABI: GCN layer, graph pooling, dense linear.

These stand in for what the reference reaches through torch_geometric + torch autograd:
`GCNConv.forward` (gcn_norm, Linear, propagate: gather / message / scatter_add, bias) followed by
LeakyReLU (model/gcn.py:58-63), `global_max_pool` / `global_mean_pool` (model/gcn.py:65-66), the
readout Linear(+LeakyReLU) stack (model/gcn.py:70-71) and `loss.backward()` through them
(utils/utils_model.py:65).  All arithmetic runs in libhcatgnet_hip.so; torch only owns memory.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .plan import BatchPlan

LEAKY_SLOPE = 0.01  # nn.LeakyReLU() default (reference model/gcn.py:21, :63)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise _lib.HcgError(f"hcatgnet_amd computes in float32; got {t.dtype}")
    return t.contiguous()


class _GCNLayerFn(torch.autograd.Function):
    """out = leaky_relu( Ahat (x W^T) + b ), Ahat = D^-1/2 (A + fill I) D^-1/2 from the plan.
    `edge_mult` (optional, [E] in the caller's edge order, may require grad): explain-mode multiplier of every
    message, applied AFTER the normalisation like PyG's Explainer edge mask (self loops keep 1)."""

    @staticmethod
    def forward(ctx, x, weight, bias, plan: BatchPlan, use_edge_weight: bool, apply_act: bool, slope: float, edge_mult=None):
        lib = _lib.load()
        _lib.require_gpu(x, weight, bias, edge_mult)
        x, weight, bias = _f32c(x), _f32c(weight), _f32c(bias)
        N, F = x.shape
        D = weight.shape[0]
        if weight.shape[1] != F or N != plan.N:
            raise ValueError(f"shape mismatch: x {tuple(x.shape)}, weight {tuple(weight.shape)}, plan N {plan.N}")
        mult_csr = mult_csc = None
        if edge_mult is not None:
            if use_edge_weight or plan.ew_csr is not None:
                raise _lib.HcgError("an edge mask cannot be combined with explicit edge weights")
            if edge_mult.numel() != plan.E:
                raise ValueError(f"edge mask has {edge_mult.numel()} entries, the batch has {plan.E} edges")
            plan.ensure_eid()
            em = _f32c(edge_mult.detach()).reshape(-1)
            mult_csr = em[plan.eid.long()[:plan.E]].contiguous() if plan.E else em
            mult_csc = em[plan.eid_t.long()[:plan.E]].contiguous() if plan.E else em
        else:
            plan.ensure_csr()
        out = torch.empty(N, D, dtype=torch.float32, device=x.device)
        h_ws = torch.empty(N, D, dtype=torch.float32, device=x.device)
        ew = plan.ew_csr if use_edge_weight else mult_csr
        fill = plan.fill if use_edge_weight else 1.0
        rc = lib.hcg_gcn_layer_fwd(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(plan.rowptr),
                                   _lib.ptr(plan.col), _lib.ptr(ew), _lib.ptr(ctx_dinv(plan, use_edge_weight)),
                                   fill, slope, int(apply_act), _lib.ptr(h_ws), _lib.ptr(out), N, plan.E, F, D,
                                   _lib.stream_ptr())
        _lib.check(rc, "hcg_gcn_layer_fwd")
        ctx.save_for_backward(x, weight, out, *([mult_csc, h_ws] if edge_mult is not None else []))
        ctx.plan, ctx.use_ew, ctx.apply_act, ctx.slope, ctx.masked = plan, use_edge_weight, apply_act, slope, edge_mult is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        saved = ctx.saved_tensors
        x, weight, out = saved[:3]
        plan = ctx.plan
        dout = _f32c(dout)
        N, F = x.shape
        D = weight.shape[0]
        dev = x.device
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(weight)
        db = torch.empty(D, dtype=torch.float32, device=dev)
        dh_ws = torch.empty(max(N, 1), D, dtype=torch.float32, device=dev)
        wsb = lib.hcg_general_workspace_bytes(_lib.HCG_WS_GCN_LAYER_BWD, N, F, D, 0)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        ew = plan.ew_csc if ctx.use_ew else (saved[3] if ctx.masked else None)
        fill = plan.fill if ctx.use_ew else 1.0
        dinv = ctx_dinv(plan, ctx.use_ew)
        rc = lib.hcg_gcn_layer_bwd(_lib.ptr(dout), _lib.ptr(out), _lib.ptr(x), _lib.ptr(weight),
                                   _lib.ptr(plan.rowptr_t), _lib.ptr(plan.col_t), _lib.ptr(ew),
                                   _lib.ptr(dinv), fill, ctx.slope, int(ctx.apply_act),
                                   _lib.ptr(dh_ws), _lib.ptr(dx), _lib.ptr(dW), _lib.ptr(db), N, plan.E, F, D,
                                   _lib.ptr(ws), wsb, _lib.stream_ptr())
        _lib.check(rc, "hcg_gcn_layer_bwd")
        dmult = None
        if ctx.masked and ctx.needs_input_grad[7]:
            h = saved[4]                                     # x W^T of the forward
            dew = torch.zeros(max(plan.E, 1), dtype=torch.float32, device=dev)
            rc = lib.hcg_gcn_edge_weight_grad(_lib.ptr(dout), _lib.ptr(out), _lib.ptr(h), _lib.ptr(plan.rowptr),
                                              _lib.ptr(plan.col), _lib.ptr(dinv), ctx.slope, int(ctx.apply_act),
                                              _lib.ptr(dew), N, plan.E, D, _lib.stream_ptr())
            _lib.check(rc, "hcg_gcn_edge_weight_grad")
            dmult = torch.zeros(plan.E, dtype=torch.float32, device=dev)
            if plan.E:
                dmult[plan.eid.long()[:plan.E]] = dew[:plan.E]    # CSR order -> the caller's edge order (a permutation)
        return dx, dW, db, None, None, None, None, dmult


def ctx_dinv(plan: BatchPlan, use_edge_weight: bool):
    """dinv to use for a layer.  The reference passes `edge_weight` to conv1 only
    (model/gcn.py:58 vs :62); the other layers run unweighted with fill 1 (plan.dinv_unw)."""
    if plan.ew_csr is not None and not use_edge_weight:
        return plan.dinv_unw
    return plan.dinv


class _PoolFn(torch.autograd.Function):
    """emb = cat[global_max_pool(a), global_mean_pool(a)]  ([B, 2D], max first)."""

    @staticmethod
    def forward(ctx, a, plan: BatchPlan):
        lib = _lib.load()
        _lib.require_gpu(a)
        a = _f32c(a)
        N, D = a.shape
        emb = torch.zeros(plan.B, 2 * D, dtype=torch.float32, device=a.device)
        rc = lib.hcg_pool_fwd(_lib.ptr(a), _lib.ptr(plan.graph_ptr), _lib.ptr(emb), N, plan.B, D, _lib.stream_ptr())
        _lib.check(rc, "hcg_pool_fwd")
        ctx.save_for_backward(a, emb)
        ctx.plan = plan
        return emb

    @staticmethod
    def backward(ctx, demb):
        lib = _lib.load()
        a, emb = ctx.saved_tensors
        plan = ctx.plan
        demb = _f32c(demb)
        N, D = a.shape
        da = torch.zeros_like(a)
        rc = lib.hcg_pool_bwd(_lib.ptr(demb), _lib.ptr(a), _lib.ptr(emb), _lib.ptr(plan.graph_ptr), _lib.ptr(da), N,
                              plan.B, D, _lib.stream_ptr())
        _lib.check(rc, "hcg_pool_bwd")
        return da, None


class _LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) with act in {identity, LeakyReLU}."""

    @staticmethod
    def forward(ctx, x, weight, bias, apply_act: bool, slope: float):
        lib = _lib.load()
        _lib.require_gpu(x, weight, bias)
        x, weight = _f32c(x), _f32c(weight)
        bias = _f32c(bias) if bias is not None else None
        M, K = x.shape
        O = weight.shape[0]
        if weight.shape[1] != K:
            raise ValueError(f"shape mismatch: x {tuple(x.shape)} weight {tuple(weight.shape)}")
        y = torch.empty(M, O, dtype=torch.float32, device=x.device)
        rc = lib.hcg_linear_fwd(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(y), M, K, O,
                                _lib.HCG_ACT_LEAKY if apply_act else _lib.HCG_ACT_NONE, slope, _lib.stream_ptr())
        _lib.check(rc, "hcg_linear_fwd")
        ctx.save_for_backward(x, weight, y)
        ctx.has_bias, ctx.apply_act, ctx.slope = bias is not None, apply_act, slope
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, weight, y = ctx.saved_tensors
        dy = _f32c(dy)
        M, K = x.shape
        O = weight.shape[0]
        dev = x.device
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(weight)
        db = torch.empty(O, dtype=torch.float32, device=dev) if ctx.has_bias else None
        dz = torch.empty(max(M, 1), O, dtype=torch.float32, device=dev)
        wsb = lib.hcg_general_workspace_bytes(_lib.HCG_WS_LINEAR, M, K, O, 0)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        rc = lib.hcg_linear_bwd(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(x), _lib.ptr(weight), _lib.ptr(dx), _lib.ptr(dW),
                                _lib.ptr(db), _lib.ptr(dz), M, K, O,
                                _lib.HCG_ACT_LEAKY if ctx.apply_act else _lib.HCG_ACT_NONE, ctx.slope, _lib.ptr(ws), wsb,
                                _lib.stream_ptr())
        _lib.check(rc, "hcg_linear_bwd")
        return dx, dW, db, None, None


class _FusedLayerFn(torch.autograd.Function):
    """One fused launch per layer for batches of small graphs (csrc/fused.hip): returns the node
    embeddings, or -- with `pool=True`, the last conv layer -- the pooled graph embedding
    [max, mean] directly (the node embeddings stay internal, saved for the backward)."""

    @staticmethod
    def forward(ctx, x, weight, bias, plan: BatchPlan, gpt: int, apply_act: bool, slope: float, pool: bool):
        lib = _lib.load()
        _lib.require_gpu(x, weight, bias)
        x, weight, bias = _f32c(x), _f32c(weight), _f32c(bias)
        N, F = x.shape
        D = weight.shape[0]
        if weight.shape[1] != F or N != plan.N:
            raise ValueError(f"shape mismatch: x {tuple(x.shape)}, weight {tuple(weight.shape)}, plan N {plan.N}")
        dev = x.device
        out = torch.empty(N, D, dtype=torch.float32, device=dev)
        emb = torch.empty(plan.B, 2 * D, dtype=torch.float32, device=dev) if pool else None
        _lib.fused_forward(x=x, W1=weight, b1=bias, edge_index=plan.edge_index, E=plan.E, graph_ptr=plan.graph_ptr,
                           edge_ptr=plan.edge_ptr, N=N, B=plan.B, F=F, D=D, graphs_per_tile=gpt, apply_act=int(apply_act),
                           slope=slope, out1=out, emb=emb, status=plan.status)
        ctx.plan, ctx.gpt, ctx.apply_act, ctx.slope, ctx.pool = plan, gpt, apply_act, slope, pool
        if pool:
            ctx.save_for_backward(x, weight, out, emb)
            return emb
        ctx.save_for_backward(x, weight, out)
        return out

    @staticmethod
    def backward(ctx, grad):
        lib = _lib.load()
        plan = ctx.plan
        if ctx.pool:
            x, weight, out, emb = ctx.saved_tensors
            dout, demb = None, _f32c(grad)
        else:
            x, weight, out = ctx.saved_tensors
            emb, dout, demb = None, _f32c(grad), None
        N, F = x.shape
        D = weight.shape[0]
        dev = x.device
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(weight)
        db = torch.empty(D, dtype=torch.float32, device=dev)
        wsb = lib.hcg_fused_workspace_bytes(plan.B, F, D, ctx.gpt)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        rc = lib.hcg_fused_layer_bwd(_lib.ptr(dout), _lib.ptr(demb), _lib.ptr(emb), _lib.ptr(out), None, _lib.ptr(x),
                                     _lib.ptr(weight), _lib.ptr(plan.edge_index), plan.E,
                                     _lib.ptr(plan.graph_ptr), _lib.ptr(plan.edge_ptr), N, plan.B, F, D, ctx.gpt, ctx.slope,
                                     int(ctx.apply_act), _lib.ptr(dx), _lib.ptr(plan.status), _lib.ptr(ws), wsb, _lib.stream_ptr())
        _lib.check(rc, "hcg_fused_layer_bwd")
        job = _lib.ReduceJob()
        _lib.check(lib.hcg_fused_reduce_job(_lib.ptr(ws), wsb, N, plan.B, F, D, ctx.gpt, _lib.ptr(dW), _lib.ptr(db),
                                            ctypes.addressof(job)), "hcg_fused_reduce_job")
        _lib.reduce_jobs(ctypes.addressof(job), 1)
        return dx, dW, db, None, None, None, None, None


class _MidLayerFn(torch.autograd.Function):
    """One fused launch per layer for batches of mid-size graphs, one graph per workgroup (csrc/mid.hip): the size
    range of the reference's own reaction graphs.  Same contract as `_FusedLayerFn`."""

    @staticmethod
    def forward(ctx, x, weight, bias, plan: BatchPlan, apply_act: bool, slope: float, pool: bool):
        lib = _lib.load()
        _lib.require_gpu(x, weight, bias)
        x, weight, bias = _f32c(x), _f32c(weight), _f32c(bias)
        N, F = x.shape
        D = weight.shape[0]
        if weight.shape[1] != F or N != plan.N:
            raise ValueError(f"shape mismatch: x {tuple(x.shape)}, weight {tuple(weight.shape)}, plan N {plan.N}")
        dev = x.device
        out = torch.empty(N, D, dtype=torch.float32, device=dev)
        emb = torch.empty(plan.B, 2 * D, dtype=torch.float32, device=dev) if pool else None
        rc = lib.hcg_mid_layer_fwd(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(plan.edge_index), plan.E,
                                   _lib.ptr(plan.graph_ptr), _lib.ptr(plan.edge_ptr), N, plan.B, F, D, plan.max_nodes,
                                   plan.max_edges, slope, int(apply_act), _lib.ptr(out), _lib.ptr(emb), None, None, None, _lib.ptr(plan.status),
                                   _lib.stream_ptr())
        _lib.check(rc, "hcg_mid_layer_fwd")
        ctx.plan, ctx.apply_act, ctx.slope, ctx.pool = plan, apply_act, slope, pool
        if pool:
            ctx.save_for_backward(x, weight, out, emb)
            return emb
        ctx.save_for_backward(x, weight, out)
        return out

    @staticmethod
    def backward(ctx, grad):
        import ctypes
        lib = _lib.load()
        plan = ctx.plan
        if ctx.pool:
            x, weight, out, emb = ctx.saved_tensors
            dout, demb = None, _f32c(grad)
        else:
            x, weight, out = ctx.saved_tensors
            emb, dout, demb = None, _f32c(grad), None
        N, F = x.shape
        D = weight.shape[0]
        dev = x.device
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(weight)
        db = torch.empty(D, dtype=torch.float32, device=dev)
        wsb = lib.hcg_mid_workspace_bytes(plan.B, F, D, plan.max_nodes, plan.max_edges)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        stream = _lib.stream_ptr()
        rc = lib.hcg_mid_layer_bwd(_lib.ptr(dout), _lib.ptr(demb), _lib.ptr(emb), _lib.ptr(out), _lib.ptr(x), _lib.ptr(weight),
                                   _lib.ptr(plan.edge_index), plan.E, _lib.ptr(plan.graph_ptr), _lib.ptr(plan.edge_ptr), N,
                                   plan.B, F, D, plan.max_nodes, plan.max_edges, ctx.slope, int(ctx.apply_act), _lib.ptr(dx),
                                   _lib.ptr(plan.status), _lib.ptr(ws), wsb, stream)
        _lib.check(rc, "hcg_mid_layer_bwd")
        jb, halves = _lib.job_bytes(), D // 64          # one slab set (= one job) per 64-column half
        jobs = ctypes.create_string_buffer(jb * halves)
        for half in range(halves):
            _lib.check(lib.hcg_mid_reduce_job(_lib.ptr(ws), wsb, plan.B, F, D, plan.max_nodes, plan.max_edges, half,
                                              _lib.ptr(dW), _lib.ptr(db), ctypes.addressof(jobs) + half * jb), "hcg_mid_reduce_job")
        _lib.reduce_jobs(ctypes.addressof(jobs), halves)
        return dx, dW, db, None, None, None, None


def mid_supported(plan: BatchPlan, F: int, D: int) -> bool:
    """True when the one-graph-per-workgroup kernels apply to this plan / layer shape."""
    if (plan.mode != "blocked" or plan.ew_csr is not None or plan.max_nodes is None or plan.max_edges is None
            or plan.B == 0 or plan.N == 0):
        return False
    return bool(_lib.load().hcg_mid_supported(F, D, plan.max_nodes, plan.max_edges))


def mid_gcn_layer(x, weight, bias, plan: BatchPlan, apply_act=True, slope=LEAKY_SLOPE, pool=False):
    return _MidLayerFn.apply(x, weight, bias, plan, apply_act, slope, pool)


class _TallLayerFn(torch.autograd.Function):
    """Layers cut into dense row-streaming parts + per-graph segmented sums (csrc/tall.hip): D = 128 over large graphs
    (forward and backward), D = 64 over graphs of 65 .. 224 nodes (backward; the forward is `hcg_mid_layer_fwd`).
    Same contract as `_MidLayerFn`."""

    @staticmethod
    def forward(ctx, x, weight, bias, plan: BatchPlan, apply_act: bool, slope: float, pool: bool):
        lib = _lib.load()
        _lib.require_gpu(x, weight, bias)
        x, weight, bias = _f32c(x), _f32c(weight), _f32c(bias)
        N, F = x.shape
        D = weight.shape[0]
        if weight.shape[1] != F or N != plan.N:
            raise ValueError(f"shape mismatch: x {tuple(x.shape)}, weight {tuple(weight.shape)}, plan N {plan.N}")
        dev = x.device
        out = torch.empty(N, D, dtype=torch.float32, device=dev)
        emb = torch.empty(plan.B, 2 * D, dtype=torch.float32, device=dev) if pool else None
        wsb = lib.hcg_tall_workspace_bytes(N, plan.B, F, D) if D != 64 else 0     # (64-wide: the forward needs none)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
        rc = lib.hcg_tall_layer_fwd(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(plan.edge_index), plan.E,
                                    _lib.ptr(plan.graph_ptr), _lib.ptr(plan.edge_ptr), N, plan.B, F, D, plan.max_nodes,
                                    plan.max_edges, slope, int(apply_act), _lib.ptr(out), _lib.ptr(emb), None, None, None, _lib.ptr(plan.status),
                                    _lib.ptr(ws), wsb, _lib.stream_ptr())
        _lib.check(rc, "hcg_tall_layer_fwd")
        ctx.plan, ctx.apply_act, ctx.slope, ctx.pool = plan, apply_act, slope, pool
        if pool:
            ctx.save_for_backward(x, weight, out, emb)
            return emb
        ctx.save_for_backward(x, weight, out)
        return out

    @staticmethod
    def backward(ctx, grad):
        import ctypes
        lib = _lib.load()
        plan = ctx.plan
        if ctx.pool:
            x, weight, out, emb = ctx.saved_tensors
            dout, demb = None, _f32c(grad)
        else:
            x, weight, out = ctx.saved_tensors
            emb, dout, demb = None, _f32c(grad), None
        N, F = x.shape
        D = weight.shape[0]
        dev = x.device
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(weight)
        db = torch.empty(D, dtype=torch.float32, device=dev)
        wsb = lib.hcg_tall_workspace_bytes(N, plan.B, F, D)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        stream = _lib.stream_ptr()
        rc = lib.hcg_tall_layer_bwd(_lib.ptr(dout), _lib.ptr(demb), _lib.ptr(emb), _lib.ptr(out), None, None, None, None, _lib.ptr(x), _lib.ptr(weight),
                                    _lib.ptr(plan.edge_index), plan.E, _lib.ptr(plan.graph_ptr), _lib.ptr(plan.edge_ptr), N,
                                    plan.B, F, D, plan.max_nodes, plan.max_edges, ctx.slope, int(ctx.apply_act), _lib.ptr(dx),
                                    _lib.ptr(plan.status), _lib.ptr(ws), wsb, stream)
        _lib.check(rc, "hcg_tall_layer_bwd")
        jb = _lib.job_bytes()
        jobs = ctypes.create_string_buffer(jb * 2)
        _lib.check(lib.hcg_tall_reduce_jobs(_lib.ptr(ws), wsb, N, plan.B, F, D, 0, _lib.ptr(dW), _lib.ptr(db),
                                            ctypes.addressof(jobs)), "hcg_tall_reduce_jobs")
        _lib.reduce_jobs(ctypes.addressof(jobs), 2)
        return dx, dW, db, None, None, None, None


# 64-wide layers: the dense-parts backward of csrc/tall.hip pays off once a batch fills the chip several times over; below
# ~70 k nodes the one-launch-per-layer kernels of csrc/mid.hip are faster (tools/sweep_tall_threshold.sh on the reference's
# graph sizes, end of round 3, ms/step mid vs wide-layer route: 34 k nodes 0.0727 / 0.0766, 45 k 0.0752 / 0.0791, 67 k 0.0968 /
# 0.0971, 89 k 0.1176 / 0.1131, 133 k 0.1587 / 0.1464; the reference's own batch of 40 graphs = 3.5 k nodes: 11 % in round 2)
# (with the first layer's dense backward behind BOTH routes the crossover moved again: 67 k 0.0920 / 0.0959, 89 k 0.1100 / 0.1107,
#  133 k 0.1465 / 0.1458, 177 k 0.1799 / 0.1689 -- profiles/r03_v_sweep_tall_threshold.txt)
TALL_MIN_NODES_D64 = 140000


def tall_supported(plan: BatchPlan, F: int, D: int) -> bool:
    """True when the kernels of csrc/tall.hip apply to this plan / layer shape (D = 128; D = 64 over graphs > 64 nodes)."""
    if (plan.mode != "blocked" or plan.ew_csr is not None or plan.max_nodes is None or plan.max_edges is None
            or plan.B == 0 or plan.N == 0):
        return False
    if D == 64 and plan.N < TALL_MIN_NODES_D64:
        return False
    return bool(_lib.load().hcg_tall_supported(F, D, plan.max_nodes, plan.max_edges))


def tall_gcn_layer(x, weight, bias, plan: BatchPlan, apply_act=True, slope=LEAKY_SLOPE, pool=False):
    return _TallLayerFn.apply(x, weight, bias, plan, apply_act, slope, pool)


class _Readout2Fn(torch.autograd.Function):
    """out = (LeakyReLU(emb W0^T + b0)) W1^T + b1 in one launch (csrc/readout.hip)."""

    @staticmethod
    def forward(ctx, emb, W0, b0, W1, b1, slope: float):
        lib = _lib.load()
        _lib.require_gpu(emb, W0, b0, W1, b1)
        emb, W0, b0, W1, b1 = _f32c(emb), _f32c(W0), _f32c(b0), _f32c(W1), _f32c(b1)
        B, D, C = emb.shape[0], W0.shape[0], W1.shape[0]
        if emb.shape[1] != 2 * D or W0.shape[1] != 2 * D or W1.shape[1] != D:
            raise ValueError("readout shape mismatch")
        z = torch.empty(B, D, dtype=torch.float32, device=emb.device)
        out = torch.empty(B, C, dtype=torch.float32, device=emb.device)
        rc = lib.hcg_readout2_fwd(_lib.ptr(emb), _lib.ptr(W0), _lib.ptr(b0), _lib.ptr(W1), _lib.ptr(b1), B, D, C, slope,
                                  _lib.ptr(z), _lib.ptr(out), _lib.stream_ptr())
        _lib.check(rc, "hcg_readout2_fwd")
        ctx.save_for_backward(emb, z, W0, W1)
        ctx.slope = slope
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        emb, z, W0, W1 = ctx.saved_tensors
        dout = _f32c(dout)
        B, D, C = emb.shape[0], W0.shape[0], W1.shape[0]
        dev = emb.device
        demb = torch.empty_like(emb)
        dW0, dW1 = torch.empty_like(W0), torch.empty_like(W1)
        db0 = torch.empty(D, dtype=torch.float32, device=dev)
        db1 = torch.empty(C, dtype=torch.float32, device=dev)
        wsb = lib.hcg_general_workspace_bytes(_lib.HCG_WS_READOUT2, B, 0, 0, 0)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        rc = lib.hcg_readout2_bwd_partial(_lib.ptr(dout), _lib.ptr(emb), _lib.ptr(z), _lib.ptr(W0), _lib.ptr(W1), B, D, C, ctx.slope,
                                          _lib.ptr(demb), _lib.ptr(ws), wsb, _lib.stream_ptr())
        _lib.check(rc, "hcg_readout2_bwd_partial")
        job = _lib.ReduceJob()
        _lib.check(lib.hcg_readout2_reduce_job(_lib.ptr(ws), wsb, B, C, _lib.ptr(dW0), _lib.ptr(db0), _lib.ptr(dW1), _lib.ptr(db1),
                                               ctypes.addressof(job)), "hcg_readout2_reduce_job")
        _lib.reduce_jobs(ctypes.addressof(job), 1)
        return demb, dW0, db0, dW1, db1, None


class _MSEFn(torch.autograd.Function):
    """mean((input - target)^2) in one launch each way (csrc/loss.hip)."""

    @staticmethod
    def forward(ctx, inp, target):
        lib = _lib.load()
        _lib.require_gpu(inp, target)
        if inp.shape != target.shape:
            raise ValueError(f"MSELoss: input {tuple(inp.shape)} and target {tuple(target.shape)} must have the same shape")
        a, b = _f32c(inp), _f32c(target)
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        _lib.check(lib.hcg_mse_fwd(_lib.ptr(a), _lib.ptr(b), a.numel(), _lib.ptr(loss), _lib.stream_ptr()), "hcg_mse_fwd")
        ctx.save_for_backward(a, b)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, b = ctx.saved_tensors
        g = _f32c(g)
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if da is None and db is None:
            return None, None
        _lib.check(lib.hcg_mse_bwd(_lib.ptr(a), _lib.ptr(b), _lib.ptr(g), a.numel(), _lib.ptr(da), _lib.ptr(db),
                                   _lib.stream_ptr()), "hcg_mse_bwd")
        return da, db


def mse_loss(inp, target):
    return _MSEFn.apply(inp, target)


def readout2_supported(D: int, C: int) -> bool:
    return D == 64 and bool(_lib.load().hcg_head_supported(D, C))


def readout2(emb, W0, b0, W1, b1, slope=LEAKY_SLOPE):
    return _Readout2Fn.apply(emb, W0, b0, W1, b1, slope)


class _FusedModelFn(torch.autograd.Function):
    """The whole small-graph model -- every fused conv layer (the last one with its pooling
    epilogue) and the fused readout head -- as ONE autograd node.  Same kernels as the per-layer
    Functions; what it removes is host time: one `apply`, one backward callback and one set of
    allocations instead of four (the eager step is host-bound: ~0.1 ms of autograd/ctypes per node)."""

    @staticmethod
    def forward(ctx, plan: BatchPlan, gpts, slope: float, x, *params):
        lib = _lib.load()
        _lib.require_gpu(x, *params)
        x = _f32c(x)
        params = [_f32c(p) for p in params]
        n_conv = len(gpts)
        convs, (R0w, R0b, R1w, R1b) = params[:2 * n_conv], params[2 * n_conv:]
        N, dev, stream = x.shape[0], x.device, _lib.stream_ptr()
        if N != plan.N:
            raise ValueError(f"x has {N} rows, plan was built for {plan.N}")
        D = convs[0].shape[0]
        acts, h = [], x
        emb = torch.empty(plan.B, 2 * D, dtype=torch.float32, device=dev)
        if n_conv == 2 and gpts[0] == gpts[1]:
            # the reference's default depth: both conv layers in ONE launch (csrc/fused.hip, STACK2)
            a1 = torch.empty(N, D, dtype=torch.float32, device=dev)
            a2 = torch.empty(N, D, dtype=torch.float32, device=dev)
            _lib.fused_forward(x=x, W1=convs[0], b1=convs[1], W2=convs[2], b2=convs[3], edge_index=plan.edge_index, E=plan.E,
                               graph_ptr=plan.graph_ptr, edge_ptr=plan.edge_ptr, N=N, B=plan.B, F=x.shape[1], D=D,
                               graphs_per_tile=gpts[0], apply_act=1, slope=slope, out1=a1, out2=a2, emb=emb, status=plan.status)
            acts = [a1, a2]
        else:
            for l in range(n_conv):
                W, b = convs[2 * l], convs[2 * l + 1]
                out = torch.empty(N, D, dtype=torch.float32, device=dev)
                _lib.fused_forward(x=h, W1=W, b1=b, edge_index=plan.edge_index, E=plan.E, graph_ptr=plan.graph_ptr,
                                   edge_ptr=plan.edge_ptr, N=N, B=plan.B, F=h.shape[1], D=D, graphs_per_tile=gpts[l], apply_act=1,
                                   slope=slope, out1=out, emb=emb if l == n_conv - 1 else None, status=plan.status)
                acts.append(out)
                h = out
        C = R1w.shape[0]
        z = torch.empty(plan.B, D, dtype=torch.float32, device=dev)
        y = torch.empty(plan.B, C, dtype=torch.float32, device=dev)
        rc = lib.hcg_readout2_fwd(_lib.ptr(emb), _lib.ptr(R0w), _lib.ptr(R0b), _lib.ptr(R1w), _lib.ptr(R1b), plan.B, D, C,
                                  slope, _lib.ptr(z), _lib.ptr(y), stream)
        _lib.check(rc, "hcg_readout2_fwd")
        ctx.save_for_backward(x, emb, z, *acts, *params)
        ctx.plan, ctx.gpts, ctx.slope, ctx.n_conv = plan, list(gpts), slope, n_conv
        ctx.set_materialize_grads(False)   # an unused graph_emb must arrive as None, not as a zero tensor + an add
        return y, emb

    @staticmethod
    def backward(ctx, dy, demb_ext):
        lib = _lib.load()
        plan, gpts, slope, n_conv = ctx.plan, ctx.gpts, ctx.slope, ctx.n_conv
        saved = ctx.saved_tensors
        x, emb, z = saved[0], saved[1], saved[2]
        acts, params = saved[3:3 + n_conv], saved[3 + n_conv:]
        convs, (R0w, R0b, R1w, R1b) = params[:2 * n_conv], params[2 * n_conv:]
        N, dev, stream = x.shape[0], x.device, _lib.stream_ptr()
        D, C, B = convs[0].shape[0], R1w.shape[0], plan.B
        f32 = dict(dtype=torch.float32, device=dev)
        # every backward kernel leaves per-workgroup slabs; ONE batched launch reduces them all at the end
        import ctypes
        jb = _lib.job_bytes()
        batched = n_conv + 1 <= 8
        jobs = ctypes.create_string_buffer(jb * 8) if batched else None
        jaddr = ctypes.addressof(jobs) if batched else 0
        keep = []                                   # workspaces must outlive the batched reduction's enqueue
        # All weight gradients are views of ONE flat buffer laid out in nn.Module parameter order
        # (conv: bias, lin.weight; readout: weight, bias) -- the data-parallel wrapper can then all-reduce it
        # in place, without first concatenating 8 small tensors.
        sizes = []
        for l in range(n_conv):
            sizes += [D, convs[2 * l].numel()]
        sizes += [R0w.numel(), D, R1w.numel(), C]
        flat = torch.empty(sum(sizes), **f32)
        views, off = [], 0
        for n_el in sizes:
            views.append(flat[off:off + n_el])
            off += n_el
        conv_db = [views[2 * l] for l in range(n_conv)]
        conv_dW = [views[2 * l + 1].view_as(convs[2 * l]) for l in range(n_conv)]
        dR0w, dR0b = views[2 * n_conv].view_as(R0w), views[2 * n_conv + 1]
        dR1w, dR1b = views[2 * n_conv + 2].view_as(R1w), views[2 * n_conv + 3]
        # readout head
        dy = _f32c(dy) if dy is not None else torch.zeros(B, C, **f32)
        demb = torch.empty_like(emb)
        wsb = lib.hcg_general_workspace_bytes(_lib.HCG_WS_READOUT2, B, 0, 0, 0)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        keep.append(ws)
        if batched:
            rc = lib.hcg_readout2_bwd_partial(_lib.ptr(dy), _lib.ptr(emb), _lib.ptr(z), _lib.ptr(R0w), _lib.ptr(R1w), B, D, C,
                                              slope, _lib.ptr(demb), _lib.ptr(ws), wsb, stream)
            _lib.check(rc, "hcg_readout2_bwd_partial")
            _lib.check(lib.hcg_readout2_reduce_job(_lib.ptr(ws), wsb, B, C, _lib.ptr(dR0w), _lib.ptr(dR0b), _lib.ptr(dR1w),
                                                   _lib.ptr(dR1b), jaddr), "hcg_readout2_reduce_job")
        else:
            rc = lib.hcg_readout2_bwd_partial(_lib.ptr(dy), _lib.ptr(emb), _lib.ptr(z), _lib.ptr(R0w), _lib.ptr(R1w), B, D, C,
                                              slope, _lib.ptr(demb), _lib.ptr(ws), wsb, stream)
            _lib.check(rc, "hcg_readout2_bwd_partial")
            rjob = _lib.ReduceJob()
            _lib.check(lib.hcg_readout2_reduce_job(_lib.ptr(ws), wsb, B, C, _lib.ptr(dR0w), _lib.ptr(dR0b), _lib.ptr(dR1w),
                                                   _lib.ptr(dR1b), ctypes.addressof(rjob)), "hcg_readout2_reduce_job")
            _lib.reduce_jobs(ctypes.addressof(rjob), 1)
        if demb_ext is not None:          # the caller also used graph_emb downstream
            demb = demb + _f32c(demb_ext)
        # conv stack, last layer first
        grads = [None] * (2 * n_conv)
        dh = None
        njobs = 1
        for l in reversed(range(n_conv)):
            W = convs[2 * l]
            inp = x if l == 0 else acts[l - 1]
            F = inp.shape[1]
            need_dx = l > 0 or ctx.needs_input_grad[3]
            dx = torch.empty_like(inp) if need_dx else None
            dW, db = conv_dW[l], conv_db[l]
            wsb = lib.hcg_fused_workspace_bytes(B, F, D, gpts[l])
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            keep.append(ws)
            last = l == n_conv - 1
            rc = lib.hcg_fused_layer_bwd(None if last else _lib.ptr(dh), _lib.ptr(demb) if last else None,
                                         _lib.ptr(emb) if last else None, _lib.ptr(acts[l]), None, _lib.ptr(inp), _lib.ptr(W),
                                         _lib.ptr(plan.edge_index), plan.E, _lib.ptr(plan.graph_ptr), _lib.ptr(plan.edge_ptr),
                                         N, B, F, D, gpts[l], slope, 1, _lib.ptr(dx), _lib.ptr(plan.status), _lib.ptr(ws), wsb,
                                         stream)
            _lib.check(rc, "hcg_fused_layer_bwd")
            if batched:
                _lib.check(lib.hcg_fused_reduce_job(_lib.ptr(ws), wsb, N, B, F, D, gpts[l], _lib.ptr(dW), _lib.ptr(db),
                                                    jaddr + njobs * jb), "hcg_fused_reduce_job")
                njobs += 1
            else:
                job = _lib.ReduceJob()
                _lib.check(lib.hcg_fused_reduce_job(_lib.ptr(ws), wsb, N, B, F, D, gpts[l], _lib.ptr(dW), _lib.ptr(db),
                                                    ctypes.addressof(job)), "hcg_fused_reduce_job")
                _lib.reduce_jobs(ctypes.addressof(job), 1)
            grads[2 * l], grads[2 * l + 1] = dW, db
            dh = dx
        if batched:
            _lib.reduce_jobs(jaddr, njobs)
        return (None, None, None, dh if ctx.needs_input_grad[3] else None, *grads, dR0w, dR0b, dR1w, dR1b)


def fused_model(plan: BatchPlan, gpts, x, conv_params, readout_params, slope=LEAKY_SLOPE):
    """-> (out [B, C], graph_emb [B, 2D]).  conv_params = [W1, b1, W2, b2, ...], readout_params = [W0, b0, W1, b1]."""
    return _FusedModelFn.apply(plan, tuple(gpts), slope, x, *conv_params, *readout_params)


def fused_graphs_per_tile(plan: BatchPlan, F: int, D: int) -> int:
    """> 0 when the fused small-graph kernels apply to this plan / layer shape."""
    if plan.mode != "blocked" or plan.ew_csr is not None or plan.max_nodes is None or plan.B == 0:
        return 0
    return int(_lib.load().hcg_fused_graphs_per_tile(F, D, plan.max_nodes))


def fused_gcn_layer(x, weight, bias, plan: BatchPlan, gpt: int, apply_act=True, slope=LEAKY_SLOPE, pool=False):
    return _FusedLayerFn.apply(x, weight, bias, plan, gpt, apply_act, slope, pool)


def gcn_layer(x, weight, bias, plan: BatchPlan, use_edge_weight=False, apply_act=True, slope=LEAKY_SLOPE, edge_mult=None):
    return _GCNLayerFn.apply(x, weight, bias, plan, use_edge_weight, apply_act, slope, edge_mult)


def graph_pool(a, plan: BatchPlan):
    return _PoolFn.apply(a, plan)


def linear(x, weight, bias=None, apply_act=False, slope=LEAKY_SLOPE):
    return _LinearFn.apply(x, weight, bias, apply_act, slope)
