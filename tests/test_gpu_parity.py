"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the reference's
golden vectors.  Tolerances (north_star): bit-exact on indexing; <= 1e-5 relative
(||d||_inf / ||ref||_inf per tensor) on fp32 embeddings / outputs / gradients, <= 1e-4 on weight
gradients that sum >1e5 terms (SURVEY 8d states this bound)."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import elementwise_ok, golden_files, load_golden, rel_inf

pytestmark = pytest.mark.gpu

TOL = 1e-5
TOL_DW = 1e-4


@pytest.fixture(scope="module")
def H():
    import hcatgnet_amd
    from hcatgnet_amd import _lib
    _lib.load()
    return hcatgnet_amd


@pytest.fixture(scope="module")
def oracle():
    from oracle import gcn_oracle
    return gcn_oracle


def _model_from_params(H, params, device="cuda"):
    n_conv = 1 + sum(1 for k in params if k.startswith("conv_layers.") and k.endswith("lin.weight"))
    n_read = sum(1 for k in params if k.startswith("readout.") and k.endswith("weight"))
    D, F = params["conv1.lin.weight"].shape
    opt = H.default_options(n_convolutions=n_conv, readout_layers=n_read, embedding_dim=D,
                            n_classes=params[f"readout.{n_read - 1}.weight"].shape[0])
    m = H.make_network("GCN", opt, F)
    m.load_state_dict(params)            # state-dict compatibility with the reference's files
    return m.to(device)


def _rand_params(F, D, n_conv=2, n_read=2, n_classes=1, seed=0):
    g = torch.Generator().manual_seed(seed)
    p = {}
    def glorot(o, i):
        a = (6.0 / (i + o)) ** 0.5
        return (torch.rand(o, i, generator=g) * 2 - 1) * a
    p["conv1.lin.weight"] = glorot(D, F); p["conv1.bias"] = torch.randn(D, generator=g) * 0.1
    for i in range(n_conv - 1):
        p[f"conv_layers.{i}.lin.weight"] = glorot(D, D); p[f"conv_layers.{i}.bias"] = torch.randn(D, generator=g) * 0.1
    dim = 2 * D
    for i in range(n_read - 1):
        p[f"readout.{i}.0.weight"] = glorot(dim // 2, dim); p[f"readout.{i}.0.bias"] = torch.randn(dim // 2, generator=g) * 0.1
        dim //= 2
    p[f"readout.{n_read - 1}.weight"] = glorot(n_classes, dim); p[f"readout.{n_read - 1}.bias"] = torch.randn(n_classes, generator=g) * 0.1
    return p


# ---------------------------------------------------------------------------------- plan / indexing
def _check_plan_against_input(plan, edge_index, batch, B):
    plan.ensure_csr()
    ei = edge_index.cpu().numpy(); N = batch.numel(); E = ei.shape[1]
    if plan.mode == "blocked":   # blocked plans also carry the per-graph edge ranges the fused kernels use
        gid = batch.cpu().numpy()[ei[0]] if E else np.zeros(0, np.int64)
        assert np.array_equal(plan.edge_ptr.cpu().numpy(), np.searchsorted(gid, np.arange(B + 1), side="left"))
    rowptr = plan.rowptr.cpu().numpy(); col = plan.col.cpu().numpy()[:E]
    rowptr_t = plan.rowptr_t.cpu().numpy(); col_t = plan.col_t.cpu().numpy()[:E]
    # bit-exact reconstruction of the (src, dst) multiset, and STABLE order inside every row
    order = np.lexsort((np.arange(E), ei[1]))                       # stable by target
    assert np.array_equal(rowptr, np.concatenate([[0], np.cumsum(np.bincount(ei[1], minlength=N))]))
    selfloop = ei[0] == ei[1]                                        # explicit (i, i): slot kept, marked -1
    assert np.array_equal(col, np.where(selfloop, -1, ei[0])[order])
    order_t = np.lexsort((np.arange(E), ei[0]))
    assert np.array_equal(rowptr_t, np.concatenate([[0], np.cumsum(np.bincount(ei[0], minlength=N))]))
    assert np.array_equal(col_t, np.where(selfloop, -1, ei[1])[order_t])
    gp = plan.graph_ptr.cpu().numpy()
    assert np.array_equal(gp, np.searchsorted(batch.cpu().numpy(), np.arange(B + 1), side="left"))
    deg = 1.0 + np.bincount(ei[1][~selfloop], minlength=N).astype(np.float32)
    assert np.array_equal(plan.dinv.cpu().numpy()[:N], (1.0 / np.sqrt(deg)).astype(np.float32))


@pytest.mark.parametrize("mode", ["blocked", "general"])
def test_plan_roundtrip_bit_exact(H, mode):
    from hcatgnet_amd import synth
    sb = synth.make_config("C2", num_graphs=257, nodes_jitter=6)
    ei, b = sb.edge_index.cuda(), sb.batch.cuda()
    plan = H.BatchPlan.build(ei, b, sb.x.shape[0], num_graphs=sb.num_graphs, mode=mode)
    _check_plan_against_input(plan, sb.edge_index, sb.batch, sb.num_graphs)


def test_plan_general_handles_shuffled_edges_and_auto_falls_back(H):
    from hcatgnet_amd import synth
    sb = synth.make_config("C2", num_graphs=64)
    perm = torch.randperm(sb.edge_index.shape[1], generator=torch.Generator().manual_seed(1))
    ei = sb.edge_index[:, perm].contiguous()
    plan = H.BatchPlan.build(ei.cuda(), sb.batch.cuda(), sb.x.shape[0], mode="auto")
    assert plan.mode == "general"
    _check_plan_against_input(plan, ei, sb.batch, sb.num_graphs)
    with pytest.raises(ValueError):
        H.BatchPlan.build(ei.cuda(), sb.batch.cuda(), sb.x.shape[0], mode="blocked")


def test_plan_flags_bad_indices(H):
    ei = torch.tensor([[0, 1, 5], [1, 0, 2]], dtype=torch.int64).cuda()
    b = torch.zeros(3, dtype=torch.int64).cuda()
    with pytest.raises(ValueError, match="outside"):
        H.BatchPlan.build(ei, b, 3, mode="general")
    with pytest.raises(ValueError, match="sorted"):
        H.BatchPlan.build(torch.tensor([[0], [1]]).cuda(), torch.tensor([1, 0, 1]).cuda(), 3, num_graphs=2, mode="general")


def test_plan_edge_cases(H):
    # graph 0: isolated single node; graph 1: empty slot; graph 2: 3-node path with a duplicate bond; no edges in graph 3
    batch = torch.tensor([0, 2, 2, 2, 3, 3], dtype=torch.int64)
    ei = torch.tensor([[1, 2, 2, 3, 1, 2], [2, 1, 3, 2, 2, 1]], dtype=torch.int64)
    for mode in ("blocked", "general"):
        plan = H.BatchPlan.build(ei.cuda(), batch.cuda(), 6, num_graphs=4, mode=mode)
        _check_plan_against_input(plan, ei, batch, 4)
    # E == 0
    plan = H.BatchPlan.build(torch.zeros(2, 0, dtype=torch.int64).cuda(), batch.cuda(), 6, num_graphs=4, mode="general")
    assert plan.rowptr.cpu().tolist() == [0] * 7


# ---------------------------------------------------------------------------------- forward parity
GOLDEN = golden_files()


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_forward_matches_reference_golden_vectors(H, oracle, path):
    """Reference weights + reference graphs -> the reference's own embeddings.csv numbers."""
    g = load_golden(path)
    m = _model_from_params(H, g["params"])
    with torch.no_grad():
        out, emb = m(H.Batch(g["x"].cuda(), g["edge_index"].cuda(), g["batch"].cuda(), g["num_graphs"]), True)
    # default kernel selection (these 56-184-atom graphs: one graph per workgroup, csrc/mid.hip): the north_star bound
    assert rel_inf(emb, g["ref_emb"]) <= TOL
    # elementwise at the SURVEY 8d bound (1e-5 of each value, floor 1e-3), judged against the fp64 oracle and anchored on
    # the reference's OWN numbers (its embeddings.csv, fp32 arithmetic) measured the same way: a pooled mean that cancels
    # to ~1e-3 carries ~1e-7 of fp32 rounding in ANY summation order (tests/test_gpu_elementwise.py states the gate)
    p64 = {k: v.double() for k, v in g["params"].items()}
    _, emb64 = oracle.gcn_forward(p64, g["x"].double(), g["edge_index"], g["batch"], g["num_graphs"])
    bound = TOL * emb64.abs().clamp_min(1e-3)
    ex_hip = ((emb.double().cpu() - emb64).abs() / bound)
    ex_ref = ((g["ref_emb"].double() - emb64).abs() / bound)
    assert int((ex_hip > 1).sum()) <= int((ex_ref > 1).sum()) + 2
    assert float(ex_hip.max()) <= max(2.0, 2.0 * float(ex_ref.max()))      # (worst element: within 2x the bound)
    assert (out[:, 0].cpu() - g["ref_pred"]).abs().max().item() <= 5e-5
    # any-shape kernels: they add every node's messages in the reference's own edge order, which also holds the much
    # tighter ELEMENTWISE bound (1e-5 of each value, floor 1e-3) on pooled means that cancel to ~1e-3
    m.use_fused = False
    with torch.no_grad():
        out_g, emb_g = m(H.Batch(g["x"].cuda(), g["edge_index"].cuda(), g["batch"].cuda(), g["num_graphs"]), True)
    m.use_fused = True
    assert rel_inf(emb_g, g["ref_emb"]) <= TOL and elementwise_ok(emb_g, g["ref_emb"], rtol=TOL, floor=1e-3)
    assert (out_g[:, 0].cpu() - g["ref_pred"]).abs().max().item() <= 5e-5
    # and node embeddings after every conv vs the oracle
    o_out, o_emb, acts = oracle.gcn_forward(g["params"], g["x"], g["edge_index"], g["batch"], g["num_graphs"],
                                            return_intermediates=True)
    plan = H.BatchPlan.build(g["edge_index"].cuda(), g["batch"].cuda(), g["x"].shape[0], num_graphs=g["num_graphs"])
    h = m.conv1(g["x"].cuda(), plan, apply_act=True)
    assert rel_inf(h, acts[0]) <= TOL
    h = m.conv_layers[0](h, plan, apply_act=True)
    assert rel_inf(h, acts[1]) <= TOL


def _synthetic(name, num_graphs, **kw):
    from hcatgnet_amd import synth
    sb = synth.make_config(name, num_graphs=num_graphs, **kw)
    cfg = synth.CONFIGS[name]
    return sb, cfg


@pytest.mark.parametrize("name,ng,kw", [("C1", 1, {}), ("C2", 64, {}), ("C2", 96, {"nodes_jitter": 6}),
                                        ("C5", 16, {})])
def test_forward_backward_vs_oracle_synthetic(H, oracle, name, ng, kw):
    sb, cfg = _synthetic(name, ng, **kw)
    params = _rand_params(cfg["feat"], cfg["hidden"], seed=3)
    m = _model_from_params(H, params)
    batch = sb.as_batch("cuda")
    out, emb = m(batch, True)
    loss = torch.sqrt(m.loss(out, torch.unsqueeze(batch.y, dim=1)))     # utils/utils_model.py:64
    m.zero_grad()
    loss.backward()
    o_loss, o_out, o_emb, o_grads = oracle.train_step_grads(params, sb.x, sb.edge_index, sb.batch, sb.y, sb.num_graphs)
    assert rel_inf(emb, o_emb) <= TOL and rel_inf(out, o_out, floor=1.0) <= TOL
    assert abs(loss.item() - o_loss.item()) <= TOL * abs(o_loss.item())
    got = {k: v.grad for k, v in m.named_parameters()}
    for k, ref in o_grads.items():
        tol = TOL_DW if k.endswith("weight") else TOL
        assert rel_inf(got[k], ref) <= tol, k
    # fp64 oracle as the tie-breaker for accumulated rounding
    _, _, _, g64 = oracle.train_step_grads(params, sb.x, sb.edge_index, sb.batch, sb.y, sb.num_graphs, dtype=torch.float64)
    for k, ref in g64.items():
        assert rel_inf(got[k], ref) <= TOL_DW, k


def test_input_gradient_matches_oracle(H, oracle):
    sb, cfg = _synthetic("C2", 8)
    params = _rand_params(cfg["feat"], cfg["hidden"], seed=5)
    m = _model_from_params(H, params)
    batch = sb.as_batch("cuda")
    batch.x.requires_grad_(True)
    out = m(batch)
    torch.sqrt(m.loss(out, batch.y.unsqueeze(1))).backward()
    *_, dx = oracle.train_step_grads(params, sb.x, sb.edge_index, sb.batch, sb.y, sb.num_graphs, x_requires_grad=True)
    assert rel_inf(batch.x.grad, dx) <= TOL


def test_general_plan_gives_same_numbers_as_blocked(H):
    sb, cfg = _synthetic("C2", 32)
    params = _rand_params(cfg["feat"], cfg["hidden"], seed=7)
    m = _model_from_params(H, params)
    m.use_fused = False          # same (any-shape) kernels on both plans
    x, ei, b = sb.x.cuda(), sb.edge_index.cuda(), sb.batch.cuda()
    with torch.no_grad():
        o1 = m(x, ei, None, b, plan=H.BatchPlan.build(ei, b, x.shape[0], num_graphs=32, mode="blocked"))
        o2 = m(x, ei, None, b, plan=H.BatchPlan.build(ei, b, x.shape[0], num_graphs=32, mode="general"))
    assert torch.equal(o1, o2)   # both plans are stable sorts -> identical summation order -> bitwise equal


def test_max_pool_ties_split_evenly(H, oracle):
    """Chemically equivalent atoms give bit-identical rows: the max-branch gradient must be split."""
    from hcatgnet_amd import functional as HF
    a = torch.tensor([[1.0, 2.0, -1.0], [1.0, 0.0, -1.0], [0.5, 2.0, -1.0], [3.0, 3.0, 3.0]])
    batch = torch.tensor([0, 0, 0, 1])
    plan = H.BatchPlan.build(torch.zeros(2, 0, dtype=torch.int64).cuda(), batch.cuda(), 4, num_graphs=2, mode="general")
    ag = a.cuda().requires_grad_(True)
    emb = HF.graph_pool(ag, plan)
    w = torch.arange(1, 13, dtype=torch.float32).reshape(2, 6).cuda()
    (emb * w).sum().backward()
    ar = a.clone().requires_grad_(True)
    ref = torch.cat([oracle.global_max_pool(ar, batch, 2), oracle.global_mean_pool(ar, batch, 2)], 1)
    (ref * w.cpu()).sum().backward()
    assert torch.equal(emb.detach().cpu(), ref.detach())
    assert rel_inf(ag.grad, ar.grad) <= 1e-6


def test_edge_cases_forward_backward(H, oracle):
    """isolated node, single-node graph, empty graph slot, duplicate bond."""
    batch = torch.tensor([0, 2, 2, 2, 3, 3], dtype=torch.int64)
    ei = torch.tensor([[1, 2, 2, 3, 1, 2], [2, 1, 3, 2, 2, 1]], dtype=torch.int64)
    x = torch.randn(6, 25, generator=torch.Generator().manual_seed(0))
    y = torch.randn(4, generator=torch.Generator().manual_seed(1))
    params = _rand_params(25, 64, seed=11)
    m = _model_from_params(H, params)
    out, emb = m(x.cuda(), ei.cuda(), None, batch.cuda(), return_graph_embedding=True,
                 plan=H.BatchPlan.build(ei.cuda(), batch.cuda(), 6, num_graphs=4))
    torch.sqrt(m.loss(out, y.cuda().unsqueeze(1))).backward()
    o_loss, o_out, o_emb, o_grads = oracle.train_step_grads(params, x, ei, batch, y, 4)
    assert rel_inf(emb, o_emb) <= TOL and rel_inf(out, o_out, floor=1.0) <= TOL
    assert torch.equal(emb[1].cpu(), torch.zeros(128))          # empty slot -> zeros
    for k, v in m.named_parameters():
        assert rel_inf(v.grad, o_grads[k]) <= TOL, k


@pytest.mark.parametrize("improved", [False, True])
def test_explain_style_edge_weight(H, oracle, improved):
    """GCN_explain.forward(x, edge_index, batch_index, edge_weight) (reference model/gcn.py:124-140):
    weights reach conv1 only; `improved` switches the self-loop fill to 2 when weights are explicit."""
    from hcatgnet_amd.gcn import GCN_explain
    sb, cfg = _synthetic("C2", 4)
    params = _rand_params(cfg["feat"], cfg["hidden"], seed=13)
    opt = H.default_options(improved=improved)
    m = GCN_explain(opt, cfg["feat"]); m.load_state_dict(params); m = m.cuda()
    ew = torch.rand(sb.edge_index.shape[1], generator=torch.Generator().manual_seed(2)) + 0.25
    with torch.no_grad():
        out = m(x=sb.x.cuda(), edge_index=sb.edge_index.cuda(), batch_index=sb.batch.cuda(), edge_weight=ew.cuda())
        ref, _ = oracle.gcn_forward(params, sb.x, sb.edge_index, sb.batch, 4, edge_weight=ew, improved=improved)
    assert rel_inf(out, ref, floor=1.0) <= TOL


def test_odd_widths_general_shapes(H, oracle):
    """F_in = 25 (diene featurisation), D = 48, 3 convs, 3 readout layers, n_classes 2."""
    sb, _ = _synthetic("C2", 8, feat=25)
    params = _rand_params(25, 48, n_conv=3, n_read=3, n_classes=2, seed=17)
    m = _model_from_params(H, params)
    with torch.no_grad():
        out, emb = m(sb.as_batch("cuda"), True)
        ref, remb = oracle.gcn_forward(params, sb.x, sb.edge_index, sb.batch, 8)
    assert rel_inf(emb, remb) <= TOL and rel_inf(out, ref, floor=1.0) <= TOL


# ---------------------------------------------------------------------------------- full-size checks
def test_full_size_c2_vs_oracle_and_determinism(H, oracle):
    """BASELINE configs[1]/[2] at full size: 4096 graphs.  The torch oracle still finishes in < 1 s."""
    sb, cfg = _synthetic("C2", 4096)
    params = _rand_params(cfg["feat"], cfg["hidden"], seed=19)
    m = _model_from_params(H, params)
    batch = sb.as_batch("cuda")

    def step():
        m.zero_grad()
        out, emb = m(batch, True)
        torch.sqrt(m.loss(out, batch.y.unsqueeze(1))).backward()
        return out.detach().clone(), emb.detach().clone(), {k: v.grad.clone() for k, v in m.named_parameters()}

    out, emb, grads = step()
    out2, emb2, grads2 = step()
    assert torch.equal(out, out2) and torch.equal(emb, emb2)                  # no atomics: run-to-run bitwise
    assert all(torch.equal(grads[k], grads2[k]) for k in grads)
    o_loss, o_out, o_emb, o_grads = oracle.train_step_grads(params, sb.x, sb.edge_index, sb.batch, sb.y, sb.num_graphs)
    assert rel_inf(emb, o_emb) <= TOL and rel_inf(out, o_out, floor=1.0) <= TOL
    for k, ref in o_grads.items():
        assert rel_inf(grads[k], ref) <= (TOL_DW if k.endswith("weight") else 5 * TOL), k
    # size-independent property: graphs are independent -> any sub-batch reproduces its slice bitwise
    sub = _synthetic("C2", 4096)[0]
    nsub = 100
    n_nodes = int((sub.batch < nsub).sum()); n_edges = int((sub.edge_index[0] < n_nodes).sum())
    with torch.no_grad():
        o_sub = m(sub.x[:n_nodes].cuda(), sub.edge_index[:, :n_edges].cuda(), None, sub.batch[:n_nodes].cuda())
    assert torch.equal(o_sub, out[:nsub])


def test_cpu_tensors_fail_loudly(H):
    from hcatgnet_amd._lib import HcgError
    sb, cfg = _synthetic("C1", 1)
    m = H.make_network("GCN", H.default_options(), cfg["feat"])
    with pytest.raises(HcgError):
        m(sb.as_batch())


# ---------------------------------------------------------------------------------- fused small-graph kernels
def _step_grads(m, batch, y):
    m.zero_grad()
    out, emb = m(batch, True)
    torch.sqrt(m.loss(out, y.unsqueeze(1))).backward()
    return out.detach(), emb.detach(), {k: v.grad.detach().clone() for k, v in m.named_parameters()}


@pytest.mark.parametrize("nodes,jitter,feat,extra", [(30, 0, 64, 3), (12, 4, 64, 2), (5, 2, 25, 1), (30, 0, 32, 3),
                                                      (20, 6, 40, 3), (2, 0, 7, 0)])
def test_fused_layers_vs_oracle_and_general_path(H, oracle, nodes, jitter, feat, extra):
    """One-launch-per-layer kernels (csrc/fused.hip): every staging variant (vector / scalar,
    K padded to 32 / 64), 1..16 graphs per 32-row tile, against the oracle and the any-shape path."""
    from hcatgnet_amd import functional as HF, synth
    sb = synth.make_batch(num_graphs=203, nodes=nodes, extra_bonds=extra, max_degree=4, feat=feat, nodes_jitter=jitter, seed=5)
    params = _rand_params(feat, 64, seed=23)
    m = _model_from_params(H, params)
    batch = sb.as_batch("cuda")
    plan = H.BatchPlan.build(batch.edge_index, batch.batch, batch.x.shape[0], num_graphs=sb.num_graphs, mode="blocked",
                             max_nodes=sb.max_nodes)
    batch._hcg_plan = plan
    assert HF.fused_graphs_per_tile(plan, feat, 64) == 32 // sb.max_nodes >= 1
    m.use_fused = True
    out_f, emb_f, g_f = _step_grads(m, batch, batch.y)
    assert plan.check_status() == 0
    m.use_fused = False
    out_g, emb_g, g_g = _step_grads(m, batch, batch.y)
    o_loss, o_out, o_emb, o_grads = oracle.train_step_grads(params, sb.x, sb.edge_index, sb.batch, sb.y, sb.num_graphs)
    assert rel_inf(emb_f, o_emb) <= TOL and rel_inf(out_f, o_out, floor=1.0) <= TOL
    assert rel_inf(emb_f, emb_g) <= 2e-6 and rel_inf(out_f, out_g, floor=1.0) <= 2e-6
    for k, ref in o_grads.items():
        assert rel_inf(g_f[k], ref) <= (TOL_DW if k.endswith("weight") else TOL), k
        assert rel_inf(g_f[k], g_g[k]) <= TOL, k
    # node embeddings of the first layer (no pooling epilogue) vs oracle
    _, _, acts = oracle.gcn_forward(params, sb.x, sb.edge_index, sb.batch, sb.num_graphs, return_intermediates=True)
    with torch.no_grad():
        h = m.conv1(batch.x, plan, apply_act=True, fused=True)
    assert rel_inf(h, acts[0]) <= TOL


def test_fused_input_gradient_and_determinism(H, oracle):
    from hcatgnet_amd import synth
    sb = synth.make_config("C2", num_graphs=300)
    params = _rand_params(64, 64, seed=29)
    m = _model_from_params(H, params)
    batch = sb.as_batch("cuda")
    batch.x.requires_grad_(True)
    outs = []
    for _ in range(2):
        m.zero_grad(); batch.x.grad = None
        out = m(batch)
        torch.sqrt(m.loss(out, batch.y.unsqueeze(1))).backward()
        outs.append((out.detach().clone(), batch.x.grad.clone(), {k: v.grad.clone() for k, v in m.named_parameters()}))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert all(torch.equal(outs[0][2][k], outs[1][2][k]) for k in outs[0][2])           # bitwise reproducible
    *_, dx = oracle.train_step_grads(params, sb.x, sb.edge_index, sb.batch, sb.y, sb.num_graphs, x_requires_grad=True)
    assert rel_inf(batch.x.grad, dx) <= TOL


def test_fused_edge_cases(H, oracle):
    """empty graph slots, single-node graphs, an isolated node, duplicate bonds, and a dense
    multigraph whose tile exceeds the LDS neighbour cache (> 256 edges -> global index path)."""
    xs, eis, bs, off = [], [], [], 0
    gen = torch.Generator().manual_seed(3)
    sizes = [1, 0, 3, 20, 1, 0, 0, 7, 2]
    for g, n in enumerate(sizes):
        if n == 0:
            continue
        xs.append(torch.randn(n, 64, generator=gen))
        bs.append(torch.full((n,), g, dtype=torch.int64))
        if n == 20:     # complete digraph: 380 directed edges in one tile
            i, j = torch.meshgrid(torch.arange(n), torch.arange(n), indexing="ij")
            mask = i != j
            eis.append(torch.stack([i[mask], j[mask]]) + off)
        elif n >= 2:
            path = torch.arange(n - 1)
            e = torch.stack([torch.cat([path, path + 1]), torch.cat([path + 1, path])])
            if n == 7:
                e = torch.cat([e[:, :-1], e[:, :2]], 1)   # drop one direction (asymmetric) + duplicate a bond
            eis.append(e + off)
        off += n
    x, ei, b = torch.cat(xs), torch.cat(eis, 1), torch.cat(bs)
    B = len(sizes)
    y = torch.randn(B, generator=gen)
    params = _rand_params(64, 64, seed=31)
    m = _model_from_params(H, params)
    plan = H.BatchPlan.build(ei.cuda(), b.cuda(), x.shape[0], num_graphs=B, mode="blocked", max_nodes=max(sizes))
    from hcatgnet_amd import functional as HF
    assert HF.fused_graphs_per_tile(plan, 64, 64) == 1
    m.zero_grad()
    out, emb = m(x.cuda(), ei.cuda(), None, b.cuda(), return_graph_embedding=True, plan=plan)
    torch.sqrt(m.loss(out, y.cuda().unsqueeze(1))).backward()
    assert plan.check_status() == 0
    o_loss, o_out, o_emb, o_grads = oracle.train_step_grads(params, x, ei, b, y, B)
    assert rel_inf(emb, o_emb) <= TOL and rel_inf(out, o_out, floor=1.0) <= TOL
    assert torch.equal(emb[1].cpu(), torch.zeros(128))
    for k, v in m.named_parameters():
        assert rel_inf(v.grad, o_grads[k]) <= TOL, k


def test_fused_refuses_oversize_tile_without_touching_memory(H):
    """Host metadata lies (max_nodes too small): the kernel must flag SHAPE_LIMIT and skip, not fault."""
    from hcatgnet_amd import synth
    sb = synth.make_batch(num_graphs=16, nodes=40, extra_bonds=2, max_degree=4, feat=64)
    params = _rand_params(64, 64, seed=37)
    m = _model_from_params(H, params)
    plan = H.BatchPlan.build(sb.edge_index.cuda(), sb.batch.cuda(), sb.x.shape[0], num_graphs=16, mode="blocked",
                             max_nodes=30)     # wrong on purpose
    with torch.no_grad():
        m(sb.x.cuda(), sb.edge_index.cuda(), None, sb.batch.cuda(), plan=plan)
    with pytest.raises(ValueError, match="tile"):
        plan.check_status()


@pytest.mark.parametrize("B,C", [(1, 1), (33, 1), (200, 3), (4096, 8)])
def test_fused_readout_head(H, B, C):
    """csrc/readout.hip against torch fp64 autograd of the same two-layer head (model/gcn.py:36-45)."""
    import torch.nn.functional as F
    from hcatgnet_amd import functional as HF
    g = torch.Generator().manual_seed(B + C)
    emb = torch.randn(B, 128, generator=g); W0 = torch.randn(64, 128, generator=g) * 0.1; b0 = torch.randn(64, generator=g) * 0.1
    W1 = torch.randn(C, 64, generator=g) * 0.1; b1 = torch.randn(C, generator=g) * 0.1; go = torch.randn(B, C, generator=g)
    dev = [t.cuda().requires_grad_(True) for t in (emb, W0, b0, W1, b1)]
    out = HF.readout2(*dev)
    out.backward(go.cuda())
    ref = [t.double().requires_grad_(True) for t in (emb, W0, b0, W1, b1)]
    r = F.linear(F.leaky_relu(F.linear(ref[0], ref[1], ref[2]), 0.01), ref[3], ref[4])
    r.backward(go.double())
    assert rel_inf(out, r) <= TOL
    for a, b in zip(dev, ref):
        assert rel_inf(a.grad, b.grad) <= TOL


def test_rccl_gradient_allreduce_world1(H):
    """The RCCL leg of the DP wrapper on the one GPU we have: backend "nccl" (= RCCL), world size 1,
    collectives forced.  (Multi-rank RCCL only runs in the driver's multi-GPU tier; the multi-rank
    logic is covered by tests/test_ddp_gloo.py.)"""
    import socket
    import torch.distributed as dist
    from hcatgnet_amd import synth
    from hcatgnet_amd.ddp import DataParallelGCN
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        sb = synth.make_config("C2", num_graphs=64)
        m = H.make_network("GCN", H.default_options(), 64).cuda()
        dp = DataParallelGCN(m, force_collective=True)
        batch = sb.as_batch("cuda")
        out = dp(batch)
        torch.sqrt(dp.loss(out, batch.y.unsqueeze(1))).backward()
        before = [p.grad.clone() for p in m.parameters()]
        flat = dp.reduce_gradients()
        torch.cuda.synchronize()
        assert flat.numel() == 16641
        for p, g in zip(m.parameters(), before):
            assert torch.equal(p.grad, g)
        dp.optimizer.step()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("combine", ["sse", "mean"])
def test_rccl_exchange_captured_with_the_step_world1(H, combine):
    """`FusedTrainStep.capture_exchange`: the RCCL all-reduce and the update launch are recorded INTO the step's hipGraph
    (backend "nccl", world size 1, collective forced), alone and inside a `StepWindow`: replays == the eager
    data-parallel steps of a twin model, bitwise (same launches in the same order)."""
    import socket
    import torch.distributed as dist
    from hcatgnet_amd import synth
    from hcatgnet_amd.ddp import DataParallelGCN
    from hcatgnet_amd.train import StepWindow
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        sbs = [synth.make_config("C2", num_graphs=128, rank=r) for r in range(2)]
        dev = [(sb, sb.x.cuda(), sb.edge_index.cuda(), sb.batch.cuda(), sb.y.cuda()) for sb in sbs]
        fresh = [(lambda t=t: H.Batch(t[1], t[2], t[3], t[0].num_graphs, y=t[4], max_nodes=t[0].max_nodes,
                                      max_edges=t[0].max_edges, edges_grouped=True)) for t in dev]
        torch.manual_seed(0)
        models = [H.make_network("GCN", H.default_options(), 64).cuda() for _ in range(3)]
        for m in models[1:]:
            m.load_state_dict(models[0].state_dict())
        dps = [DataParallelGCN(m, force_collective=True, combine=combine) for m in models]
        eager = [dps[0].make_train_step() for _ in range(2)]
        single = [dps[1].make_train_step() for _ in range(2)]
        winst = [dps[2].make_train_step() for _ in range(2)]
        for st in single + winst:
            st.capture_exchange = True
        # twin: 2 warm-up steps per capture (single[0] then single[1]) = batches 0 0 1 1, then 0 1 0 1
        for i in (0, 0, 1, 1):
            eager[i](fresh[i]())
        single[0].capture(fresh[0]); single[1].capture(fresh[1])
        le, ls = [], []
        for _ in range(2):
            for i in range(2):
                le.append(float(eager[i](fresh[i]())))
                ls.append(float(single[i].replay()))
        assert ls == le, (ls, le)
        for pa, pb in zip(models[0].parameters(), models[1].parameters()):
            assert torch.equal(pa, pb)
        # window: its warm-up runs batches 0 1 once; a fresh eager twin follows
        models[0].load_state_dict(models[2].state_dict())
        tw_dp = DataParallelGCN(H.make_network("GCN", H.default_options(), 64).cuda(), force_collective=True, combine=combine)
        tw_dp.module.load_state_dict(models[2].state_dict())
        tw = [tw_dp.make_train_step() for _ in range(2)]
        win = StepWindow(winst, fresh)
        for i in range(2):
            tw[i](fresh[i]())
        lw, lt = [], []
        for _ in range(2):
            lt += [float(tw[i](fresh[i]())) for i in range(2)]
            lw += [float(v) for v in win.replay()]
        assert lw == lt, (lw, lt)
        for pa, pb in zip(tw_dp.module.parameters(), models[2].parameters()):
            assert torch.equal(pa, pb)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [1, 7, 4096, 100003])
def test_fused_mse_loss(H, n):
    """hcatgnet_amd.networks.MSELoss == nn.MSELoss() (mean), value and both gradients."""
    from hcatgnet_amd.networks import MSELoss
    g = torch.Generator().manual_seed(n)
    a = torch.randn(n, 1, generator=g); b = torch.randn(n, 1, generator=g) * 3
    ad, bd = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    loss = torch.sqrt(MSELoss()(ad, bd))
    loss.backward()
    ar, br = a.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.sqrt(torch.nn.MSELoss()(ar, br))
    ref.backward()
    assert rel_inf(loss, ref) <= TOL and rel_inf(ad.grad, ar.grad) <= TOL and rel_inf(bd.grad, br.grad) <= TOL
    with pytest.raises(ValueError):
        MSELoss()(ad, bd.reshape(-1))


def test_single_node_model_equals_per_layer_functions(H, oracle):
    """The one-autograd-node form of the fused model launches the same kernels as the per-layer
    Functions: outputs and every gradient must be bitwise equal; graph_emb may carry its own gradient."""
    from hcatgnet_amd import synth
    sb = synth.make_config("C2", num_graphs=130)
    params = _rand_params(64, 64, seed=41)
    m = _model_from_params(H, params)
    batch = sb.as_batch("cuda")
    batch.x.requires_grad_(True)
    res = []
    for single in (True, False):
        m.single_node = single
        m.zero_grad(); batch.x.grad = None
        out, emb = m(batch, True)
        (torch.sqrt(m.loss(out, batch.y.unsqueeze(1))) + 0.1 * emb.square().mean()).backward()
        res.append((out.detach().clone(), emb.detach().clone(), batch.x.grad.clone(),
                    {k: v.grad.clone() for k, v in m.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert all(torch.equal(res[0][3][k], res[1][3][k]) for k in res[0][3])
    # and against the oracle with the same composite loss
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    o_out, o_emb = oracle.gcn_forward(p, sb.x, sb.edge_index, sb.batch, sb.num_graphs)
    (oracle.rmse_loss(o_out, sb.y) + 0.1 * o_emb.square().mean()).backward()
    for k, v in p.items():
        assert rel_inf(res[0][3][k], v.grad) <= (TOL_DW if k.endswith("weight") else TOL), k


def test_fused_backward_writes_one_flat_gradient_buffer(H):
    """The single-node backward lays all 8 gradients out in ONE buffer in parameter order, so the DP
    wrapper all-reduces in place (no concatenation)."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.ddp import DataParallelGCN
    sb = synth.make_config("C2", num_graphs=64)
    m = H.make_network("GCN", H.default_options(), 64).cuda()
    batch = sb.as_batch("cuda")
    torch.sqrt(m.loss(m(batch), batch.y.unsqueeze(1))).backward()
    dp = DataParallelGCN(m)
    grads = [p.grad for p in m.parameters()]
    flat = dp.flat_gradient()
    assert flat.data_ptr() == grads[0].data_ptr() and flat.numel() == 16641
    assert torch.equal(flat, torch.cat([g.reshape(-1) for g in grads]))
    dp.reduce_gradients()                       # world size 1: views re-attached, values unchanged
    assert all(torch.equal(p.grad, g) for p, g in zip(m.parameters(), grads))


def test_fused_adam_matches_torch_adam_and_scheduler(H):
    """FusedAdam == torch.optim.Adam(lr=0.01, eps=1e-9) over several real training steps (flat-gradient
    one-launch path), then with non-flat gradients (per-parameter path), and it honours lr changes made by
    the reference's ReduceLROnPlateau scheduler."""
    from hcatgnet_amd import synth
    sb = synth.make_config("C2", num_graphs=96)
    m = H.make_network("GCN", H.default_options(), 64).cuda()
    ref_params = [p.detach().clone().requires_grad_(True) for p in m.parameters()]
    ref_opt = torch.optim.Adam(ref_params, lr=0.01, eps=1e-9)
    batch = sb.as_batch("cuda")
    for it in range(6):
        m.optimizer.zero_grad()
        loss = torch.sqrt(m.loss(m(batch), batch.y.unsqueeze(1)))
        loss.backward()
        grads = [p.grad.detach().clone() for p in m.parameters()]
        if it >= 3:   # break the flat layout on purpose -> per-parameter launches
            for p, g in zip(m.parameters(), grads):
                p.grad = g.clone()
        m.optimizer.step()
        for rp, g in zip(ref_params, grads):
            rp.grad = g
        ref_opt.step()
        for p, rp in zip(m.parameters(), ref_params):
            assert rel_inf(p, rp) <= 2e-6, it
        if it == 1:   # scheduler writes param_groups[...]['lr']; mirror it on the torch optimiser
            for _ in range(m.scheduler.patience + 2):
                m.scheduler.step(1e9)
            assert m.optimizer.param_groups[0]["lr"] < 0.01
            ref_opt.param_groups[0]["lr"] = m.optimizer.param_groups[0]["lr"]
    # module state dict still has the reference's keys / shapes and the updated values
    sd = m.state_dict()
    assert tuple(sd["conv1.lin.weight"].shape) == (64, 64) and torch.equal(sd["conv1.lin.weight"], m.conv1.lin.weight)


@pytest.mark.parametrize("seed", range(6))
def test_fused_path_random_batches(H, oracle, seed):
    """Randomised sweep of the fused path: random graph sizes (1..32 nodes, empty slots), random
    multigraph edges (duplicates, asymmetric directions, isolated nodes), random F in {7..64}."""
    g = torch.Generator().manual_seed(100 + seed)
    B = int(torch.randint(1, 70, (1,), generator=g))
    F = int(torch.randint(7, 65, (1,), generator=g))
    sizes = torch.randint(0, 33, (B,), generator=g)
    if seed % 2 == 0:
        sizes = sizes.clamp(max=10)                       # several graphs per 32-row tile
    sizes[int(torch.randint(0, B, (1,), generator=g))] = max(1, int(sizes.max()))
    xs, eis, bs, off = [], [], [], 0
    for gi, n in enumerate(sizes.tolist()):
        if n == 0:
            continue
        xs.append(torch.randn(n, F, generator=g))
        bs.append(torch.full((n,), gi, dtype=torch.int64))
        ne = int(torch.randint(0, 3 * n + 1, (1,), generator=g))
        if ne:
            eis.append(torch.randint(0, n, (2, ne), generator=g) + off)     # self loops / duplicates allowed
        off += n
    x, b = torch.cat(xs), torch.cat(bs)
    ei = torch.cat(eis, 1) if eis else torch.zeros(2, 0, dtype=torch.int64)
    y = torch.randn(B, generator=g)
    params = _rand_params(F, 64, seed=200 + seed)
    m = _model_from_params(H, params)
    plan = H.BatchPlan.build(ei.cuda(), b.cuda(), x.shape[0], num_graphs=B, mode="blocked", max_nodes=int(sizes.max()))
    from hcatgnet_amd import functional as HF
    assert HF.fused_graphs_per_tile(plan, F, 64) >= 1
    m.zero_grad()
    out, emb = m(x.cuda(), ei.cuda(), None, b.cuda(), return_graph_embedding=True, plan=plan)
    torch.sqrt(m.loss(out, y.cuda().unsqueeze(1))).backward()
    assert plan.check_status() == 0
    o_loss, o_out, o_emb, o_grads = oracle.train_step_grads(params, x, ei, b, y, B)
    assert rel_inf(emb, o_emb) <= TOL and rel_inf(out, o_out, floor=1.0) <= TOL
    for k, v in m.named_parameters():
        assert rel_inf(v.grad, o_grads[k]) <= TOL, k


def test_explicit_self_loop_edges_collapse_like_pyg(H, oracle):
    """(i, i) edges in the input: PyG's add_remaining_self_loops removes them and gives every node exactly
    one unit self loop -- on the fused path, the blocked and the general any-shape plans."""
    x = torch.randn(7, 64, generator=torch.Generator().manual_seed(0))
    ei = torch.tensor([[0, 1, 1, 2, 2, 2, 3, 5, 6, 6], [1, 0, 1, 2, 1, 2, 3, 6, 5, 6]], dtype=torch.int64)
    b = torch.tensor([0, 0, 0, 1, 2, 2, 2], dtype=torch.int64)
    params = _rand_params(64, 64, seed=43)
    ref, remb = oracle.gcn_forward(params, x, ei, b, 3)
    no_loops = ei[:, ei[0] != ei[1]]
    ref2, _ = oracle.gcn_forward(params, x, no_loops, b, 3)
    assert torch.allclose(ref, ref2)                     # the oracle itself: loops change nothing
    m = _model_from_params(H, params)
    for fused, mode in ((True, "blocked"), (False, "blocked"), (False, "general")):
        m.use_fused = fused
        plan = H.BatchPlan.build(ei.cuda(), b.cuda(), 7, num_graphs=3, mode=mode, max_nodes=3)
        with torch.no_grad():
            out, emb = m(x.cuda(), ei.cuda(), None, b.cuda(), return_graph_embedding=True, plan=plan)
        assert rel_inf(emb, remb) <= TOL and rel_inf(out, ref, floor=1.0) <= TOL, (fused, mode)


def test_weighted_explicit_self_loop_is_flagged(H):
    """edge_weight + an explicit (i, i) edge: PyG would use that edge's weight as the loop weight; this
    build does not implement it and must say so instead of returning different numbers silently."""
    ei = torch.tensor([[0, 1, 1], [1, 0, 1]], dtype=torch.int64).cuda()
    b = torch.zeros(2, dtype=torch.int64).cuda()
    with pytest.raises(ValueError, match="self-loop"):
        H.BatchPlan.build(ei, b, 2, num_graphs=1, edge_weight=torch.ones(3).cuda(), mode="general")
    H.BatchPlan.build(ei, b, 2, num_graphs=1, mode="general")      # unweighted: fine


def test_device_collate_equals_host_collate_and_feeds_the_model(H):
    """f1: DeviceGraphStore.collate (one gather launch, plan attached) == host PyG-rule collate, bitwise;
    the model gives identical numbers on both; DeviceLoader covers the dataset once per epoch."""
    g = torch.Generator().manual_seed(7)
    graphs = []
    for i in range(37):
        n = int(torch.randint(1, 31, (1,), generator=g)); ne = int(torch.randint(0, 3 * n, (1,), generator=g))
        graphs.append(H.Data(x=torch.randn(n, 25, generator=g), edge_index=torch.randint(0, n, (2, ne), generator=g),
                             y=torch.randn(1, generator=g), idx=1000 + i))
    store = H.DeviceGraphStore(graphs, "cuda")
    ids = [5, 0, 36, 17, 17, 3]
    db = store.collate(ids)
    hb = H.collate([graphs[i] for i in ids])
    assert torch.equal(db.x.cpu(), hb.x) and torch.equal(db.edge_index.cpu(), hb.edge_index)
    assert torch.equal(db.batch.cpu(), hb.batch) and torch.equal(db.y.cpu(), hb.y) and torch.equal(db.idx.cpu(), hb.idx)
    assert db.num_graphs == 6 and db.max_nodes == hb.max_nodes and db._hcg_plan is not None
    assert torch.equal(db._hcg_plan.graph_ptr.cpu().long(), hb.ptr) and torch.equal(db._hcg_plan.edge_ptr.cpu().long(), hb.edge_ptr)
    m = H.make_network("GCN", H.default_options(), 25).cuda()
    with torch.no_grad():
        o_dev = m(db)
        o_host = m(hb.to("cuda"))
    assert torch.equal(o_dev, o_host)
    m.use_fused = False                      # the attached pointer-only plan must grow a CSR on demand
    with torch.no_grad():
        o_any = m(store.collate(ids))
    assert rel_inf(o_any, o_dev, floor=1.0) <= 2e-6
    # a loader's batches carry their index arrays as views of ONE upload per epoch: every batch still equals the host
    # collate of the same graphs bitwise, plan pointers included (odd and even batch sizes, the short last batch)
    seen = []
    for bs, drop in ((8, False), (7, False), (5, True)):
        seen = []
        for bt in H.DeviceLoader(store, batch_size=bs, shuffle=True, seed=1, drop_last=drop):
            ids_b = [int(v) - 1000 for v in bt.idx.cpu().tolist()]
            hb = H.collate([graphs[i] for i in ids_b])
            assert torch.equal(bt.x.cpu(), hb.x) and torch.equal(bt.edge_index.cpu(), hb.edge_index)
            assert torch.equal(bt.batch.cpu(), hb.batch) and torch.equal(bt.y.cpu(), hb.y)
            assert torch.equal(bt._hcg_plan.graph_ptr.cpu().long(), hb.ptr)
            assert torch.equal(bt._hcg_plan.edge_ptr.cpu().long(), hb.edge_ptr)
            assert bt.to("cuda") is bt                      # already on the device: nothing is rebuilt
            seen += bt.idx.cpu().tolist()
        if drop:
            assert len(seen) == 35 and len(set(seen)) == 35
        else:
            assert sorted(seen) == [1000 + i for i in range(37)]


@pytest.mark.parametrize("apply_sigmoid", [True, False])
def test_explain_masks_forward_and_mask_gradients(H, oracle, apply_sigmoid):
    """explain-mode hooks.  Edge mask multiplied into every message of every conv layer (PyG Explainer
    semantics, self loops keep 1) + feature mask on x; the outputs and the gradients w.r.t. BOTH masks match oracle
    autograd.  Not pinned by any reference artefact (the reference only calls the third-party Explainer)."""
    from hcatgnet_amd.explain import clear_masks, set_masks
    g = torch.Generator().manual_seed(5)
    from hcatgnet_amd import synth as S
    sb = S.make_batch(num_graphs=3, nodes=57, extra_bonds=4, max_degree=4, feat=25, nodes_jitter=9)
    params = _rand_params(25, 64, seed=23)
    m = _model_from_params(H, params)
    E, N = sb.edge_index.shape[1], sb.x.shape[0]
    em = torch.randn(E, generator=g) if apply_sigmoid else torch.rand(E, generator=g)
    nm = torch.randn(N, 25, generator=g)
    # device
    em_d = em.cuda().requires_grad_(True)
    nm_d = nm.cuda().requires_grad_(True)
    set_masks(m, em_d, sb.edge_index.cuda(), apply_sigmoid=apply_sigmoid)
    out = m(x=sb.x.cuda() * nm_d.sigmoid(), edge_index=sb.edge_index.cuda(), batch_index=sb.batch.cuda())
    out.sum().backward()
    # oracle
    em_o = em.clone().requires_grad_(True)
    nm_o = nm.clone().requires_grad_(True)
    mask = em_o.sigmoid() if apply_sigmoid else em_o
    o_out, _ = oracle.gcn_forward(params, sb.x * nm_o.sigmoid(), sb.edge_index, sb.batch, sb.num_graphs, edge_mask=mask)
    o_out.sum().backward()
    assert rel_inf(out, o_out, floor=1.0) <= TOL
    assert rel_inf(em_d.grad, em_o.grad) <= TOL
    assert rel_inf(nm_d.grad, nm_o.grad) <= TOL
    # the weights still get their gradients on this path, and clearing the masks restores the plain model
    assert all(p.grad is not None for p in m.parameters())
    clear_masks(m)
    with torch.no_grad():
        plain = m(x=sb.x.cuda(), edge_index=sb.edge_index.cuda(), batch_index=sb.batch.cuda())
        o_plain, _ = oracle.gcn_forward(params, sb.x, sb.edge_index, sb.batch, sb.num_graphs)
    assert rel_inf(plain, o_plain, floor=1.0) <= TOL
