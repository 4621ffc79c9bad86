"""GPU parity of the wide-layer kernels (csrc/tall.hip: embedding_dim 128 over large graphs -- BASELINE configs[4]):
dense row-streaming transform + per-graph segmented sum, against the CPU oracle (fp32 and fp64), the
one-graph-per-workgroup kernels and the any-shape path."""
import ctypes

import pytest
import torch

from tests.helpers import rel_inf
from tests.test_gpu_parity import H, oracle, _model_from_params, _rand_params, _step_grads, TOL, TOL_DW  # noqa: F401
from tests.test_gpu_mid import _near_ties

pytestmark = pytest.mark.gpu


def _convs(m):
    return [m.conv1] + list(m.conv_layers)


@pytest.mark.parametrize("nodes,jitter,feat,extra,deg,B,seed,D", [(200, 0, 128, 13, 6, 24, 10, 128), (100, 60, 100, 6, 4, 40, 1, 128),
                                                                  (50, 10, 64, 3, 4, 40, 2, 128), (120, 40, 28, 6, 4, 40, 6, 128),
                                                                  (40, 30, 8, 2, 4, 70, 3, 128),
                                                                  # 64-wide layers (the reference's own regime): the backward only
                                                                  (87, 30, 25, 5, 4, 131, 12, 64), (150, 34, 32, 8, 4, 131, 9, 64),
                                                                  (60, 27, 64, 4, 4, 131, 15, 64), (100, 92, 25, 6, 6, 131, 12, 64)])
def test_tall_layers_vs_oracle_mid_and_general_path(H, oracle, monkeypatch, nodes, jitter, feat, extra, deg, B, seed, D):
    """Forward with the pooled epilogue, both backward variants (pooled gradient + dx, dout without dx), every K padding
    (32 / 64 / 128, incl. widths that are no multiple of 32 or of 4), ragged batches (graphs of 8 .. 200 nodes side by side);
    D = 64: 4-wave workgroups up to 128 nodes, 8-wave ones above."""
    from hcatgnet_amd import functional as HF, synth
    monkeypatch.setattr(HF, "TALL_MIN_NODES_D64", 0)      # (these batches are small: the host would keep 64-wide layers on mid.hip)
    sb = synth.make_batch(num_graphs=B, nodes=nodes, extra_bonds=extra, max_degree=deg, feat=feat, nodes_jitter=jitter, seed=seed)
    params = _rand_params(feat, D, seed=31)
    _, _, acts0 = oracle.gcn_forward(params, sb.x, sb.edge_index, sb.batch, sb.num_graphs, return_intermediates=True)
    assert _near_ties(acts0[-1], sb.batch, sb.num_graphs) == 0, "pick another seed: this batch has a near-tie in the max pooling"
    m = _model_from_params(H, params)
    batch = sb.as_batch("cuda")
    plan = H.BatchPlan.build(batch.edge_index, batch.batch, batch.x.shape[0], num_graphs=sb.num_graphs, mode="blocked",
                             max_nodes=sb.max_nodes, max_edges=sb.max_edges)
    batch._hcg_plan = plan
    assert HF.tall_supported(plan, feat, D) and HF.tall_supported(plan, D, D)
    m.use_fused = True
    out_t, emb_t, g_t = _step_grads(m, batch, batch.y)
    assert plan.check_status() == 0
    for c in _convs(m):
        c.family = "mid"
    out_m, emb_m, g_m = _step_grads(m, batch, batch.y)
    for c in _convs(m):
        c.family = "auto"
    m.use_fused = False
    out_g, emb_g, g_g = _step_grads(m, batch, batch.y)
    o_loss, o_out, o_emb, o_grads = oracle.train_step_grads(params, sb.x, sb.edge_index, sb.batch, sb.y, sb.num_graphs)
    _, _, _, g64 = oracle.train_step_grads(params, sb.x, sb.edge_index, sb.batch, sb.y, sb.num_graphs, dtype=torch.float64)
    assert rel_inf(emb_t, o_emb) <= TOL and rel_inf(out_t, o_out, floor=1.0) <= TOL
    assert rel_inf(emb_t, emb_g) <= 2e-6 and rel_inf(out_t, out_g, floor=1.0) <= 2e-6
    assert rel_inf(emb_t, emb_m) <= 2e-6 and rel_inf(out_t, out_m, floor=1.0) <= 2e-6
    for k, ref in o_grads.items():
        assert rel_inf(g_t[k], g64[k]) <= (TOL_DW if k.endswith("weight") else TOL), k
        assert rel_inf(g_t[k], ref) <= TOL_DW and rel_inf(g_t[k], g_g[k]) <= TOL_DW and rel_inf(g_t[k], g_m[k]) <= TOL_DW, k
    with torch.no_grad():
        h = m.conv1(batch.x, plan, apply_act=True, fused=True)
    assert rel_inf(h, acts0[0]) <= TOL
    # run-to-run: bitwise (sorted rows, fixed tile -> workgroup map, fixed-order slab sums)
    m.use_fused = True
    out_2, emb_2, g_2 = _step_grads(m, batch, batch.y)
    assert torch.equal(out_t, out_2) and torch.equal(emb_t, emb_2) and all(torch.equal(g_t[k], g_2[k]) for k in g_t)


@pytest.mark.parametrize("D,F", [(128, 32), (64, 25)])
def test_tall_input_gradient_and_edge_cases(H, oracle, monkeypatch, D, F):
    """dx of the first layer, multi-edges, explicit self loops, an isolated node, a one-node graph and empty graph slots
    (one of them LAST in the batch) next to a 150-node graph."""
    from hcatgnet_amd import synth, functional as HF
    monkeypatch.setattr(HF, "TALL_MIN_NODES_D64", 0)
    g = torch.Generator().manual_seed(4)
    nbig = 150
    big = synth.make_batch(num_graphs=1, nodes=nbig, extra_bonds=6, max_degree=4, feat=F, seed=2)
    xs = [big.x, torch.randn(1, F, generator=g), torch.randn(5, F, generator=g)]
    e_small = torch.tensor([[0, 1, 1, 2, 2, 2, 3], [1, 0, 2, 1, 2, 1, 3]], dtype=torch.int64)   # multi-edge, two self loops; node 4 isolated
    x = torch.cat(xs)
    ei = torch.cat([big.edge_index, e_small + nbig + 1], 1)
    bv = torch.cat([torch.zeros(nbig, dtype=torch.int64), torch.ones(1, dtype=torch.int64), torch.full((5,), 3, dtype=torch.int64)])
    B = 5                                              # graphs 2 and 4 are empty slots
    y = torch.randn(B, generator=g)
    params = _rand_params(F, D, seed=37)
    m = _model_from_params(H, params)
    xd = x.cuda().requires_grad_(True)
    plan = H.BatchPlan.build(ei.cuda(), bv.cuda(), x.shape[0], num_graphs=B, mode="blocked")
    assert HF.tall_supported(plan, F, D)
    out = m(x=xd, edge_index=ei.cuda(), batch_index=bv.cuda(), plan=plan)
    torch.sqrt(m.loss(out, y.cuda().unsqueeze(1))).backward()
    assert plan.check_status() == 0
    o_loss, o_out, o_emb, o_grads, o_dx = oracle.train_step_grads(params, x, ei, bv, y, B, x_requires_grad=True)
    assert rel_inf(out, o_out, floor=1.0) <= TOL
    assert rel_inf(xd.grad, o_dx) <= TOL
    for k, v in m.named_parameters():
        assert rel_inf(v.grad, o_grads[k]) <= TOL, k


@pytest.mark.parametrize("D,feat", [(128, 128), (64, 64)])
def test_tall_backward_hands_down_a_premasked_dx(H, D, feat):
    """apply_act bit 1 of hcg_tall_layer_bwd: dx leaves multiplied by LeakyReLU'(x); the layer below then runs with bit 0
    clear and out = NULL.  One f32 multiply moved across a launch boundary: bitwise the plain sequence."""
    from hcatgnet_amd import synth, _lib
    from hcatgnet_amd.plan import BatchPlan
    lib, p = _lib.load(), _lib.ptr
    sb = synth.make_config("C2", num_graphs=70, nodes=90, seed=4)
    b = sb.as_batch("cuda")
    plan = BatchPlan.build(b.edge_index, b.batch, b.x.shape[0], num_graphs=b.num_graphs, mode="blocked",
                           max_nodes=sb.max_nodes, max_edges=sb.max_edges)
    N, B, slope, mxn, mxe = plan.N, plan.B, 0.01, sb.max_nodes, sb.max_edges
    assert lib.hcg_tall_supported(feat, D, mxn, mxe)
    gen = torch.Generator().manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=gen).cuda()
    x, out, W, dout = rnd(N, feat), rnd(N, D), rnd(D, feat) * 0.2, rnd(N, D)
    x[::5] = 0.0
    st = _lib.stream_ptr()

    def bwd(dout_, out_, x_, W_, F_, flags, want_dx=True):
        wsb = lib.hcg_tall_workspace_bytes(N, B, F_, D)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        dx = torch.full((N, F_), float("nan"), device="cuda") if want_dx else None
        dW, db = torch.empty(D, F_, device="cuda"), torch.empty(D, device="cuda")
        _lib.check(lib.hcg_tall_layer_bwd(p(dout_), None, None, p(out_), None, None, None, None, p(x_), p(W_), p(plan.edge_index), plan.E, p(plan.graph_ptr),
                                          p(plan.edge_ptr), N, B, F_, D, mxn, mxe, slope, flags, p(dx), p(plan.status), p(ws), wsb,
                                          st), "hcg_tall_layer_bwd")
        jb = _lib.job_bytes()
        jobs = ctypes.create_string_buffer(jb * 2)
        _lib.check(lib.hcg_tall_reduce_jobs(p(ws), wsb, N, B, F_, D, 0, p(dW), p(db), ctypes.addressof(jobs)), "hcg_tall_reduce_jobs")
        _lib.reduce_jobs(ctypes.addressof(jobs), 2)
        return dx, dW, db

    dx_plain, dW_a, db_a = bwd(dout, out, x, W, feat, 1)
    dx_pm, dW_b, db_b = bwd(dout, out, x, W, feat, 1 | 2)
    assert torch.equal(dW_a, dW_b) and torch.equal(db_a, db_b)
    mask = torch.where(x > 0, torch.ones_like(x), torch.full_like(x, slope))
    assert torch.equal(dx_pm, dx_plain * mask)
    # the layer below: premasked upstream gradient + bit 0 clear + out = NULL == plain upstream gradient + bit 0 set
    x0, W0 = rnd(N, 64), rnd(D, 64) * 0.2
    _, dW_c, db_c = bwd(dx_plain, x, x0, W0, 64, 1, want_dx=False)
    _, dW_d, db_d = bwd(dx_pm, None, x0, W0, 64, 0, want_dx=False)
    assert torch.equal(dW_c, dW_d) and torch.equal(db_c, db_d)
    assert plan.check_status() == 0


@pytest.mark.parametrize("D,F,nodes,jitter,B", [(64, 25, 100, 17, 400), (64, 64, 160, 30, 260), (128, 64, 150, 20, 24), (128, 128, 200, 24, 12)])
def test_pooled_layer_bit_form_is_bitwise_the_plain_form(H, monkeypatch, D, F, nodes, jitter, B):
    """Training form of the pooled layer on the wide-layer route (`poolbits` of hcg_tall_layer_fwd / hcg_tall_layer_bwd): its
    activations never leave the chip -- one byte per (row, 4 columns) does (sign, is-the-column-max).  Loss, outputs, pooled
    embedding and EVERY gradient are BITWISE those of the plain form (output + emb read back), with exact max-pool ties in the
    batch (a graph of identical rows: its equal-degree nodes tie).  (Against the oracle: the full-size and fuzz tests run this
    form, it is the default.)"""
    from hcatgnet_amd import synth, functional as HF
    from hcatgnet_amd.train import FusedTrainStep
    monkeypatch.setattr(HF, "TALL_MIN_NODES_D64", 0)      # (these batches are small: the host would keep 64-wide layers on mid.hip)
    sb = synth.make_batch(num_graphs=B, nodes=nodes, nodes_jitter=jitter, feat=F, extra_bonds=4, max_degree=4, seed=77)
    gp = torch.zeros(B + 1, dtype=torch.int64)
    gp[1:] = torch.bincount(sb.batch, minlength=B).cumsum(0)
    sb.x[gp[0]:gp[1]] = sb.x[0]                                # graph 0: identical rows -> equal-degree nodes tie exactly
    params = _rand_params(F, D, seed=5)
    res = []
    for bits in (True, False):
        m = _model_from_params(H, params)
        step = FusedTrainStep(m, optimizer_step=False)
        step.POOLBITS = bits
        batch = sb.as_batch("cuda")
        assert step.reason(batch) is None
        loss = step(batch)
        assert batch._hcg_plan.check_status() == 0
        cap = step._bufs["cap"]
        assert ("poolbits_tall" in cap["ws"]) == bits, "the wide-layer route / its bit form did not run"
        res.append((loss.clone(), step._flat.clone(), cap["emb"][:B].clone(), step.last_out.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert bool(torch.isfinite(res[0][1]).all()) and float(res[0][1].abs().max()) > 0


@pytest.mark.parametrize("D,F,nodes,jitter,B", [(64, 25, 100, 17, 400), (64, 64, 160, 30, 260), (64, 32, 90, 20, 420),
                                                (128, 128, 200, 24, 12), (128, 28, 150, 40, 24), (128, 64, 90, 30, 20)])
def test_first_layer_dense_backward_equals_the_transpose_sum_form(H, oracle, monkeypatch, D, F, nodes, jitter, B):
    """Training form of the FIRST layer on the wide-layer route (`xagg` + `signbits` of hcg_tall_layer_fwd / _bwd): the forward
    also leaves Ahat x and the sign pieces of its output, the backward is ONE dense launch dW = (dA (.) leaky'(A))^T (Ahat x).
    The same sums as the transpose-sum form in another order: every gradient within 1e-5 of that form's (layer-1 weight
    within 1e-4 of the fp64 oracle like every conv weight), everything that does not depend on the order -- loss, outputs, the
    second layer's and the head's gradients -- bitwise for 64-wide layers (128-wide: the training form's forward scales the x
    rows by dinv BEFORE the GEMM instead of behind it, so everything is within 1e-5); exact zeros in the first layer's output
    (an all-zero graph with zero bias: LeakyReLU'(0) = slope) included."""
    from hcatgnet_amd import synth, functional as HF
    from hcatgnet_amd.train import FusedTrainStep
    monkeypatch.setattr(HF, "TALL_MIN_NODES_D64", 0)
    sb = synth.make_batch(num_graphs=B, nodes=nodes, nodes_jitter=jitter, feat=F, extra_bonds=4, max_degree=4, seed=91)
    gp = torch.zeros(B + 1, dtype=torch.int64)
    gp[1:] = torch.bincount(sb.batch, minlength=B).cumsum(0)
    sb.x[gp[1]:gp[2]] = 0.0                                   # graph 1: zero features
    params = _rand_params(F, D, seed=6)
    params["conv1.bias"].zero_()                               # ... and zero bias: its layer-1 outputs are EXACTLY zero
    res = []
    for xagg in (True, False):
        m = _model_from_params(H, params)
        step = FusedTrainStep(m, optimizer_step=False)
        step.XAGG = xagg
        batch = sb.as_batch("cuda")
        assert step.reason(batch) is None
        loss = step(batch)
        assert batch._hcg_plan.check_status() == 0
        cap = step._bufs["cap"]
        assert ("xagg" in cap["ws"]) == xagg, "the wide-layer route / its first-layer form did not run"
        res.append((loss.clone(), step.last_out.clone(), {k: v.grad.detach().clone() for k, v in m.named_parameters()},
                    cap["acts"][0][:sb.x.shape[0]].clone()))
    (la, oa, ga, a1a), (lb, ob, gb, a1b) = res
    if D == 64:
        assert torch.equal(la, lb) and torch.equal(oa, ob) and torch.equal(a1a, a1b)
    else:
        assert abs(float(la) - float(lb)) <= TOL * abs(float(lb)) and rel_inf(oa, ob, floor=1.0) <= TOL and rel_inf(a1a, a1b) <= TOL
    assert bool((a1a[gp[1]:gp[2]] == 0).all())
    for k in ga:
        if k.startswith("conv1.") or D != 64:
            assert rel_inf(ga[k], gb[k]) <= (TOL_DW if k.endswith("lin.weight") and D != 64 else TOL), k
        else:
            assert torch.equal(ga[k], gb[k]), k
    _, _, _, g64 = oracle.train_step_grads(params, sb.x, sb.edge_index, sb.batch, sb.y, B, dtype=torch.float64)
    assert rel_inf(ga["conv1.lin.weight"], g64["conv1.lin.weight"]) <= TOL_DW
    assert rel_inf(ga["conv1.bias"], g64["conv1.bias"]) <= TOL
