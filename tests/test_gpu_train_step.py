"""GPU parity of the fused training step: csrc/head.hip (readout forward + sqrt(MSE) + readout
backward in one launch) and `hcatgnet_amd.train.FusedTrainStep` (the reference's per-batch step,
utils/utils_model.py:60-68, without autograd) against the CPU oracle's autograd + torch.optim.Adam.
Nothing in the reference pins gradients or optimiser trajectories: the oracle is the checker."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import rel_inf
from tests.test_gpu_parity import H, oracle, _model_from_params, _rand_params  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize("B,C,rmse,D", [(1, 1, 1, 64), (31, 1, 1, 64), (32, 3, 1, 64), (4096, 1, 1, 64), (4096, 8, 0, 64), (9000, 2, 1, 64),
                                        (1, 1, 1, 128), (33, 3, 1, 128), (1024, 1, 1, 128), (1024, 8, 0, 128), (9000, 2, 1, 128)])
def test_head_forward_loss_backward_one_launch(H, B, C, rmse, D):
    """hcg_head_fwd_bwd + hcg_step_tail: out / z, the UNSCALED demb, and -- after the tail applied the deferred loss scale --
    loss and weight gradients vs torch fp64 autograd of sqrt(mse_loss(Linear(LeakyReLU(Linear(emb))), y)); B = 9000 makes
    workgroups loop over several tiles; D = 128 is the 8-wave form with W0 read from L2 (BASELINE configs[4]); the
    forward-only form + hcg_loss_finalize gives the same loss."""
    from hcatgnet_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(7 * B + C)
    emb = torch.randn(B, 2 * D, generator=g); W0 = torch.randn(D, 2 * D, generator=g) * 0.1; b0 = torch.randn(D, generator=g) * 0.1
    W1 = torch.randn(C, D, generator=g) * 0.1; b1 = torch.randn(C, generator=g) * 0.1; y = torch.randn(B, C, generator=g) * 3
    d = [t.cuda().contiguous() for t in (emb, y, W0, b0, W1, b1)]
    z = torch.empty(B, D, device="cuda"); out = torch.empty(B, C, device="cuda"); loss = torch.empty(2, device="cuda")
    demb = torch.empty(B, 2 * D, device="cuda")
    wsb = lib.hcg_head_workspace_bytes(B, D)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    grads = [torch.empty_like(t) for t in d[2:]]
    step = torch.zeros(2, dtype=torch.int32, device="cuda")
    p = _lib.ptr
    mode = _lib.HCG_LOSS_RMSE if rmse else _lib.HCG_LOSS_MSE
    for rep in range(2):
        rc = lib.hcg_head_fwd_bwd(p(d[0]), p(d[1]), p(d[2]), p(d[3]), p(d[4]), p(d[5]), B, D, C, 0.01, 0, p(z), p(out),
                                  p(demb), p(ws), wsb, p(step), _lib.stream_ptr())
        _lib.check(rc, "hcg_head_fwd_bwd")
        job = _lib.ReduceJob()
        _lib.check(lib.hcg_head_reduce_job(p(ws), wsb, B, D, C, p(grads[0]), p(grads[1]), p(grads[2]), p(grads[3]),
                                           ctypes.addressof(job)), "hcg_head_reduce_job")
        assert job.sse_part and job.nseg == 4
        _lib.step_tail(ctypes.addressof(job), 1, loss=loss, loss_mode=mode, loss_count=float(B * C))
        torch.cuda.synchronize()
        assert int(step[0]) == rep + 1        # the step number advanced once per launch
    ref = [t.double().requires_grad_(True) for t in (emb, W0, b0, W1, b1)]
    zr = F.leaky_relu(F.linear(ref[0], ref[1], ref[2]), 0.01)
    r = F.linear(zr, ref[3], ref[4])
    mse = F.mse_loss(r, y.double())
    lref = torch.sqrt(mse) if rmse else mse
    lref.backward()
    assert rel_inf(out, r) <= TOL and rel_inf(z, zr) <= TOL
    lref, mse = lref.detach(), mse.detach()
    assert abs(float(loss[0]) - float(lref)) <= TOL * float(lref) and abs(float(loss[1]) - float(mse)) <= TOL * float(mse)
    scale = 1.0 / (B * C * float(torch.sqrt(mse))) if rmse else 2.0 / (B * C)         # dloss/dout = scale * (out - y)
    assert rel_inf(demb.double() * scale, ref[0].grad) <= TOL
    for a, b in zip(grads, ref[1:]):
        assert rel_inf(a, b.grad) <= TOL
    # forward only: z / out / partials, no gradient; the loss alone from the partials
    z2 = torch.zeros_like(z); out2 = torch.zeros_like(out); loss2 = torch.zeros(2, device="cuda")
    rc = lib.hcg_head_fwd_bwd(p(d[0]), p(d[1]), p(d[2]), p(d[3]), p(d[4]), p(d[5]), B, D, C, 0.01, _lib.HCG_HEAD_FORWARD_ONLY,
                              p(z2), p(out2), None, p(ws), wsb, None, _lib.stream_ptr())
    _lib.check(rc, "hcg_head_fwd_bwd")
    job = _lib.ReduceJob()
    _lib.check(lib.hcg_head_reduce_job(p(ws), wsb, B, D, C, None, None, None, None, ctypes.addressof(job)), "hcg_head_reduce_job")
    assert job.nseg == 0
    _lib.check(lib.hcg_loss_finalize(ctypes.addressof(job), float(B * C), mode, p(loss2), None, _lib.stream_ptr()), "hcg_loss_finalize")
    assert torch.equal(z2, z) and torch.equal(out2, out) and torch.equal(loss2, loss)


def _oracle_grads(oracle, m, sb):
    params = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    return oracle.train_step_grads(params, sb.x, sb.edge_index, sb.batch, sb.y, sb.num_graphs)


@pytest.mark.parametrize("cfg,ng,feat,nodes", [("C2", 96, 64, 30), ("C2", 4096, 64, 30), ("C2", 50, 25, 30), ("C2", 64, 25, 87),
                                               ("C2", 40, 64, 150)])
def test_fused_train_step_gradients_match_oracle(H, oracle, cfg, ng, feat, nodes):
    """FusedTrainStep(optimizer_step=False): loss and every weight gradient vs oracle autograd; the gradients sit
    in ONE flat buffer in parameter order.  30-atom graphs run the small-graph tiles, 87 / 150-atom graphs (the
    reference's sizes) the one-graph-per-workgroup kernels."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.train import FusedTrainStep
    sb = synth.make_config(cfg, num_graphs=ng, nodes=nodes, **({} if nodes == 30 else {"seed": 12}))
    if feat != 64:
        sb.x = sb.x[:, :feat].contiguous()
    m = H.make_network("GCN", H.default_options(), feat).cuda()
    with torch.no_grad():
        for prm in m.parameters():          # non-zero biases so that every gradient path matters
            if prm.dim() == 1:
                prm.add_(0.05)
    step = FusedTrainStep(m, optimizer_step=False)
    loss = step(sb.as_batch("cuda"))
    l_ref, out_ref, _, g_ref = _oracle_grads(oracle, m, sb)
    assert abs(float(loss) - float(l_ref)) <= TOL * abs(float(l_ref))
    assert rel_inf(step.last_out, out_ref) <= TOL
    base, off = None, 0
    for name, prm in m.named_parameters():
        tol = 1e-4 if (ng >= 4096 and "conv" in name) else TOL     # sums of > 1e5 terms
        assert rel_inf(prm.grad, g_ref[name]) <= tol, name
        base = prm.grad.data_ptr() if base is None else base
        assert prm.grad.data_ptr() == base + 4 * off, name
        off += prm.numel()


def test_fused_train_step_with_adam_follows_the_reference_loop(H, oracle):
    """Full step incl. the Adam update: (1) each update equals torch.optim.Adam(lr=.01, eps=1e-9) applied to OUR
    gradients (tight), (2) the 6-step loss trajectory follows the oracle's full loop (loose: Adam with eps = 1e-9
    turns rounding-level gradient differences into lr-sized parameter differences), (3) the loss decreases."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.train import FusedTrainStep
    sb = synth.make_config("C2", num_graphs=128)
    m = H.make_network("GCN", H.default_options(), 64).cuda()
    batch = sb.as_batch("cuda")
    p_or, opt_or = oracle.make_train_state({k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    ref_params = [p.detach().clone().requires_grad_(True) for p in m.parameters()]
    ref_opt = torch.optim.Adam(ref_params, lr=0.01, eps=1e-9)
    step = FusedTrainStep(m)
    losses, losses_or = [], []
    for it in range(6):
        losses.append(float(step(batch)))
        for rp, prm in zip(ref_params, m.parameters()):
            rp.grad = prm.grad.detach().clone()
        ref_opt.step()
        for prm, rp in zip(m.parameters(), ref_params):
            assert rel_inf(prm, rp) <= 2e-6, it
        losses_or.append(float(oracle.train_step(p_or, opt_or, sb.x, sb.edge_index, sb.batch, sb.y, sb.num_graphs)))
    assert abs(losses[0] - losses_or[0]) <= TOL * abs(losses_or[0])
    for a, b in zip(losses, losses_or):
        assert abs(a - b) <= 2e-2 * abs(b)
    assert losses[-1] < losses[0]


def test_captured_step_replays_like_eager_steps(H):
    """hipGraph capture of the whole step (plan build, forward, head, backward, reduction, Adam with device-side
    step count): 2 warm-up + 4 replays == 6 eager steps, and a learning-rate change reaches the captured update."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.train import FusedTrainStep
    sb = synth.make_config("C2", num_graphs=256)
    mk = lambda: H.Batch(sb.x.cuda(), sb.edge_index.cuda(), sb.batch.cuda(), sb.num_graphs, y=sb.y.cuda(),
                         max_nodes=sb.max_nodes, max_edges=sb.max_edges, edges_grouped=True)
    torch.manual_seed(0)
    a = H.make_network("GCN", H.default_options(), 64).cuda()
    b = H.make_network("GCN", H.default_options(), 64).cuda()
    b.load_state_dict(a.state_dict())
    ea, eb = FusedTrainStep(a), FusedTrainStep(b)
    x, ei, bv, y = sb.x.cuda(), sb.edge_index.cuda(), sb.batch.cuda(), sb.y.cuda()
    fresh = lambda: H.Batch(x, ei, bv, sb.num_graphs, y=y, max_nodes=sb.max_nodes, max_edges=sb.max_edges, edges_grouped=True)
    la = [float(ea(fresh())) for _ in range(6)]
    eb.capture(fresh)                       # runs 2 eager warm-up steps
    lb = [float(eb.replay()) for _ in range(4)]
    assert b.optimizer.steps_done() == 6
    for u, v in zip(la[2:], lb):
        assert abs(u - v) <= 1e-6 * abs(u)
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert rel_inf(pb, pa) <= 1e-6
    # learning rate: set by a scheduler on the host -> sync_lr() inside replay()
    for opt in (a.optimizer, b.optimizer):
        opt.param_groups[0]["lr"] = 0.001
    l7a, l7b = float(ea(fresh())), float(eb.replay())
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert rel_inf(pb, pa) <= 1e-6
    assert abs(l7a - l7b) <= 1e-6 * abs(l7a)
    assert b.optimizer.state_dict()["state"][0]["step"].item() == 7


def test_step_window_equals_the_same_steps_one_by_one(H):
    """`train.StepWindow`: three consecutive steps on three distinct batches (two kernel families: 30-atom tiles and the
    reference's graph sizes) captured as ONE hipGraph.  Two replays of the window == the same six steps issued eagerly
    on a twin model: losses and weights bitwise (same launches in the same order), step count included; a step with a
    separate gradient collective is refused."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.train import FusedTrainStep, StepWindow
    for cfg, feat, ng in (("C2", 64, 192), ("REAL", 25, 48)):
        sbs = [synth.make_config(cfg, num_graphs=ng, rank=r) for r in range(3)]
        dev = [(sb, sb.x.cuda(), sb.edge_index.cuda(), sb.batch.cuda(), sb.y.cuda()) for sb in sbs]
        fresh = [(lambda t=t: H.Batch(t[1], t[2], t[3], t[0].num_graphs, y=t[4], max_nodes=t[0].max_nodes,
                                      max_edges=t[0].max_edges, edges_grouped=True)) for t in dev]
        torch.manual_seed(0)
        a = H.make_network("GCN", H.default_options(), feat).cuda()
        b = H.make_network("GCN", H.default_options(), feat).cuda()
        b.load_state_dict(a.state_dict())
        sa = [FusedTrainStep(a) for _ in range(3)]
        sb_ = [FusedTrainStep(b) for _ in range(3)]
        win = StepWindow(sb_, fresh)                                    # its warm-up runs the three steps once
        la = [float(sa[i](fresh[i]())) for i in range(3)]               # the twin follows the warm-up (the capture runs nothing)
        lb = []
        for _ in range(2):
            la += [float(sa[i](fresh[i]())) for i in range(3)]
            lb += [float(v) for v in win.replay()]
        assert lb == la[3:], (cfg, lb, la[3:])
        assert b.optimizer.steps_done() == a.optimizer.steps_done() == 9
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.equal(pa, pb), cfg
    from hcatgnet_amd._lib import HcgError
    with pytest.raises(HcgError):
        StepWindow([FusedTrainStep(a, grad_sync=lambda flat: None)], [fresh[0]])


def test_eval_network_window_equals_the_batch_loop(H):
    """`eval_network` over a DeviceLoader that does not shuffle (the reference's validation / test loaders): the batches are
    collated once and their evaluate steps replayed as ONE hipGraph.  Same value as the per-batch loop -- before training,
    after the optimiser has re-based the parameters (the window is rebuilt), and after further epochs (the graph reads the
    live weights); a shuffling loader keeps the loop."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.train import FusedTrainStep, eval_network, train_network
    sb = synth.make_config("REAL", num_graphs=130)
    store = H.DeviceGraphStore(sb.as_graph_list(), device="cuda")
    val = H.DeviceLoader(store, batch_size=40)
    trn = H.DeviceLoader(store, batch_size=40, shuffle=True, seed=3)
    m = H.make_network("GCN", H.default_options(), 25).cuda()

    def loop_value():
        st = FusedTrainStep(m, optimizer_step=False)
        tot = 0.0
        for b in H.DeviceLoader(store, batch_size=40):
            tot += float(st.evaluate(b)) * b.num_graphs
        return tot / len(store)
    for phase in range(3):
        got, want = eval_network(m, val, "cuda"), loop_value()
        assert abs(got - want) <= 1e-6 * abs(want), (phase, got, want)
        assert getattr(val, "_hcg_eval_window", None) is not None
        train_network(m, trn, "cuda")
    assert getattr(trn, "_hcg_eval_window", None) is None
    v = eval_network(m, trn, "cuda")                      # shuffling loader: the loop, no window
    assert getattr(trn, "_hcg_eval_window", None) is None and v > 0


@pytest.mark.parametrize("cfg,feat,G,bs", [("REAL", 25, 135, 40), ("C2", 64, 100, 32)])
def test_epoch_window_equals_the_per_batch_loop_on_the_same_permutations(H, cfg, feat, G, bs):
    """`train_network` over a shuffling DeviceLoader runs whole epochs as ONE hipGraph (train.EpochWindow: capacity-padded
    batch slots, the collate launch inside the graph, one upload of the epoch's index arrays).  Two loaders with the same
    seed draw the same permutations: the window's epoch values and the weights after four epochs are BITWISE those of the
    per-batch loop (EPOCH_WINDOW off) -- including the last, smaller batch -- and building the window leaves the model
    untouched (its warm-up epoch is undone)."""
    from hcatgnet_amd import synth, train
    sb = synth.make_config(cfg, num_graphs=G)
    store = H.DeviceGraphStore(sb.as_graph_list(), device="cuda")
    a = H.make_network("GCN", H.default_options(), feat).cuda()
    b = H.make_network("GCN", H.default_options(), feat).cuda()
    b.load_state_dict(a.state_dict())
    la, lb = H.DeviceLoader(store, batch_size=bs, shuffle=True, seed=11), H.DeviceLoader(store, batch_size=bs, shuffle=True, seed=11)
    before = [q.detach().clone() for q in a.parameters()]
    win = train.EpochWindow.build(a, la)
    assert win is not None, "the epoch window must apply to the reference's own regime"
    la._hcg_epoch_window = ((id(a), la.batch_size, la.drop_last, len(la.store)), win)
    assert all(torch.equal(q, r) for q, r in zip(a.parameters(), before))          # the capture's warm-up epoch was undone
    assert a.optimizer.steps_done() == 0
    va = [train.train_network(a, la, "cuda") for _ in range(4)]
    train.EPOCH_WINDOW = False
    try:
        vb = [train.train_network(b, lb, "cuda") for _ in range(4)]
    finally:
        train.EPOCH_WINDOW = True
    assert va == vb, (va, vb)
    for q, r in zip(a.parameters(), b.parameters()):
        assert torch.equal(q, r)
    assert a.optimizer.steps_done() == b.optimizer.steps_done() == 4 * len(la)
    assert va[-1] < va[0]                                                              # and it learns
    # a scheduler's new learning rate reaches the captured update; reloading the weights rebuilds the window
    for g in a.optimizer.param_groups + b.optimizer.param_groups:
        g["lr"] = 0.003
    b2 = {k: v.clone() for k, v in b.state_dict().items()}
    a.load_state_dict(b2)
    va2 = train.train_network(a, la, "cuda")
    train.EPOCH_WINDOW = False
    try:
        vb2 = train.train_network(b, lb, "cuda")
    finally:
        train.EPOCH_WINDOW = True
    assert va2 == vb2


def test_concurrent_runs_equal_the_runs_one_by_one(H):
    """`train_networks` / `eval_networks`: several independent runs (the reference's nested cross-validation trains 90, one
    after another: scripts_experiments/train_GNN.py:48-50) advance together, every run's epoch ONE hipGraph on the run's own
    HIP stream.  Each run is BITWISE what `train_network` / `eval_network` give it alone: epoch values, evaluation values and
    weights after three epochs -- with different datasets, sizes and seeds per run, one run switched off for an epoch (early
    stopping) and one run over a host loader (no window: trained in turn)."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.train import eval_network, eval_networks, train_network, train_networks
    K = 4
    sizes = [135, 100, 121, 90]

    def make(k):
        sb = synth.make_config("REAL", num_graphs=sizes[k], seed=synth.BASE_SEED + 7 * k)
        graphs = sb.as_graph_list()
        store = H.DeviceGraphStore(graphs, device="cuda")
        torch.manual_seed(100 + k)
        m = H.make_network("GCN", H.default_options(), 25).cuda()
        if k == 3:                                           # a host loader: the per-batch loop, in turn
            trn = H.DataLoader(graphs, batch_size=40, shuffle=False)
        else:
            trn = H.DeviceLoader(store, batch_size=40, shuffle=True, seed=20 + k)
        return m, trn, H.DeviceLoader(store, batch_size=40)
    together, alone = [make(k) for k in range(K)], [make(k) for k in range(K)]
    for (ma, _, _), (mb, _, _) in zip(together, alone):
        mb.load_state_dict(ma.state_dict())
    masks = [[True] * K, [True, False, True, True], [True] * K]
    for active in masks:
        tv = train_networks([r[0] for r in together], [r[1] for r in together], "cuda", active=active)
        ev = eval_networks([r[0] for r in together], [r[2] for r in together], "cuda", active=active)
        for k, (m, trn, val) in enumerate(alone):
            if not active[k]:
                assert tv[k] is None and ev[k] is None
                continue
            assert tv[k] == train_network(m, trn, "cuda"), k
            assert ev[k] == eval_network(m, val, "cuda"), k
    for (ma, _, _), (mb, _, _) in zip(together, alone):
        for q, r in zip(ma.parameters(), mb.parameters()):
            assert torch.equal(q, r)
    assert getattr(together[0][1], "_hcg_epoch_window", (None, None))[1] is not None     # the one-graph form did run
    with pytest.raises(ValueError):
        train_networks([together[0][0]], [], "cuda")


def test_train_network_mirror_runs_an_epoch_and_learns(H):
    """hcatgnet_amd.train.train_network / eval_network / predict_network: the reference's loop signatures
    (utils/utils_model.py:55-111) over a DeviceLoader; fused step for 30-atom graphs, autograd fallback for graphs
    the fused kernels do not cover; one host sync per epoch."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.train import eval_network, predict_network, train_network
    for nodes in (30, 60):
        sb = synth.make_config("C2", num_graphs=160, nodes=nodes)
        store = H.DeviceGraphStore(sb.as_graph_list(), device="cuda")
        loader = H.DeviceLoader(store, batch_size=40, shuffle=True, seed=1)
        m = H.make_network("GCN", H.default_options(), 64).cuda()
        first = train_network(m, loader, "cuda")
        for _ in range(4):
            last = train_network(m, loader, "cuda")
        assert last < first
        ev = eval_network(m, loader, "cuda")
        assert ev > 0 and abs(ev - last) < 0.5 * first
        yp, yt, idx, frame = predict_network(m, H.DeviceLoader(store, batch_size=64), True)
        assert yp.shape == (160,) and yt.shape == (160,) and list(frame.columns[-3:]) == ["ddG_exp", "ddG_pred", "index"]
        assert frame.shape == (160, 128 + 3)


def test_forty_launches_are_bitwise_identical(H):
    """Run-to-run determinism under load: the fused forward / backward on the full C2 batch (4096 tiles over 256 CUs,
    two waves per SIMD) 40 times -- every activation, the loss and every gradient bit-identical.  This is the test
    that caught VALU reads of MFMA accumulators placed at the compiler's minimum distance (a wrong 1x16 block in
    ~25 % of launches; csrc/fused.hip `mfma_results_fence`)."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.train import FusedTrainStep
    sb = synth.make_config("C2", num_graphs=4096)
    m = H.make_network("GCN", H.default_options(), 64).cuda()
    batch = sb.as_batch("cuda")
    step = FusedTrainStep(m, optimizer_step=False)
    ref = None
    for rep in range(40):
        loss = step(batch)
        bufs = [v for k, v in step._bufs.items() if k != "cap"][0]
        cur = [loss.clone(), step._flat.clone(), bufs["acts"][0].clone(), bufs["acts"][1].clone(), bufs["emb"].clone(),
               bufs["demb"].clone(), bufs["dacts"][0].clone(), bufs["out"].clone()]
        if ref is None:
            ref = cur
            continue
        for k, (a, b) in enumerate(zip(ref, cur)):
            assert torch.equal(a, b), (rep, k)


def test_fused_trainer_with_rccl_exchange_world1(H):
    """The multi-GPU step on the one GPU we have: backend "nccl" (= RCCL), world size 1, exchange forced.  Eager
    step = backward -> all-reduce of the flat gradient in place -> Adam; captured step = graph (plan .. slab reduction)
    -> eager all-reduce -> eager single-launch Adam.  Both must equal the plain single-process step."""
    import os
    import torch.distributed as dist
    from hcatgnet_amd import synth
    from hcatgnet_amd.ddp import DataParallelGCN
    from hcatgnet_amd.train import FusedTrainStep
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
        created = True
    try:
        sb = synth.make_config("C2", num_graphs=512)
        x, ei, bv, y = sb.x.cuda(), sb.edge_index.cuda(), sb.batch.cuda(), sb.y.cuda()
        fresh = lambda: H.Batch(x, ei, bv, sb.num_graphs, y=y, max_nodes=sb.max_nodes, max_edges=sb.max_edges, edges_grouped=True)
        models = [H.make_network("GCN", H.default_options(), 64).cuda() for _ in range(3)]
        for m in models[1:]:
            m.load_state_dict(models[0].state_dict())
        plain = FusedTrainStep(models[0])
        dp1 = DataParallelGCN(models[1], force_collective=True)
        eager = FusedTrainStep(models[1], grad_sync=dp1.reduce_flat)
        dp2 = DataParallelGCN(models[2], force_collective=True)
        graphed = FusedTrainStep(models[2], grad_sync=lambda flat: None)
        graphed.capture(fresh)                     # (captured before the hook is live, like bench.py does before RCCL is up)
        graphed.grad_sync = dp2.reduce_flat
        for _ in range(2):
            plain(fresh()); eager(fresh())         # the capture ran two warm-up steps
        la = [float(plain(fresh())) for _ in range(3)]
        lb = [float(eager(fresh())) for _ in range(3)]
        lc = [float(graphed.replay()) for _ in range(3)]
        for u, v, w in zip(la, lb, lc):
            assert abs(u - v) <= 1e-6 * abs(u) and abs(u - w) <= 1e-6 * abs(u)
        for pa, pb, pc in zip(*[m.parameters() for m in models]):
            assert rel_inf(pb, pa) <= 1e-6 and rel_inf(pc, pa) <= 1e-6
    finally:
        if created:
            dist.destroy_process_group()


def test_fused_evaluate_matches_the_autograd_forward_and_leaves_the_model_alone(H):
    """FusedTrainStep.evaluate (eval_network's body): same loss as model(batch) -> sqrt(MSE), no parameter or gradient
    is touched; small-graph and reference-sized batches."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.train import FusedTrainStep
    for nodes in (30, 87):
        sb = synth.make_config("C2", num_graphs=64, nodes=nodes)
        m = H.make_network("GCN", H.default_options(), 64).cuda()
        batch = sb.as_batch("cuda")
        before = [p.detach().clone() for p in m.parameters()]
        step = FusedTrainStep(m)
        loss = step.evaluate(batch)
        with torch.no_grad():
            ref = torch.sqrt(m.loss(m(batch), batch.y.unsqueeze(1)))
        assert abs(float(loss) - float(ref)) <= 1e-6 * abs(float(ref))
        assert all(torch.equal(a, b) for a, b in zip(before, m.parameters()))
        assert all(p.grad is None for p in m.parameters())


def test_fused_adam_state_dict_round_trip_in_capturable_mode(H):
    """Optimizer state saved after k fused steps and loaded into a fresh model continues the trajectory bit for bit
    (moments and the device-side step count survive `state_dict()` / `load_state_dict()`)."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.train import FusedTrainStep
    sb = synth.make_config("C2", num_graphs=128)
    batch = sb.as_batch("cuda")
    a = H.make_network("GCN", H.default_options(), 64).cuda()
    sa = FusedTrainStep(a)
    for _ in range(3):
        sa(batch)
    b = H.make_network("GCN", H.default_options(), 64).cuda()
    b.load_state_dict(a.state_dict())
    b.optimizer.load_state_dict(a.optimizer.state_dict())
    sb_ = FusedTrainStep(b)
    la = [float(sa(batch)) for _ in range(3)]
    lb = [float(sb_(batch)) for _ in range(3)]
    assert la == lb
    assert a.optimizer.steps_done() == b.optimizer.steps_done() == 6
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)


def _fused_reduce(lib, _lib, ws, wsb, N, B, F_, D, gpt, dW, db):
    """hcg_fused_reduce_job + hcg_step_tail (reduction only): the layer's slabs -> dW, db"""
    job = _lib.ReduceJob()
    _lib.check(lib.hcg_fused_reduce_job(_lib.ptr(ws), wsb, N, B, F_, D, gpt, _lib.ptr(dW), _lib.ptr(db), ctypes.addressof(job)),
               "hcg_fused_reduce_job")
    _lib.reduce_jobs(ctypes.addressof(job), 1)


@pytest.mark.parametrize("feat,pooled", [(64, False), (64, True), (25, False)])
def test_fused_backward_hands_down_a_premasked_dx(H, feat, pooled):
    """apply_act bit 1 of hcg_fused_layer_bwd: dx leaves the kernel multiplied by LeakyReLU'(x) and the layer below is
    then called with bit 0 clear and out = NULL.  Both are ONE extra f32 multiply moved across a launch boundary, so
    the results are bitwise those of the plain call sequence."""
    from hcatgnet_amd import synth, _lib
    from hcatgnet_amd import functional as HF
    lib, p = _lib.load(), _lib.ptr
    sb = synth.make_config("C2", num_graphs=300, seed=5)
    from hcatgnet_amd.plan import BatchPlan
    b = sb.as_batch("cuda")
    plan = BatchPlan.build(b.edge_index, b.batch, b.x.shape[0], num_graphs=b.num_graphs, mode="blocked",
                           max_nodes=sb.max_nodes, max_edges=sb.max_edges)
    N, B, D, slope = plan.N, plan.B, 64, 0.01
    gen = torch.Generator().manual_seed(7)
    rnd = lambda *s: torch.randn(*s, generator=gen).cuda()
    x, out, W = rnd(N, feat), rnd(N, D), rnd(D, feat) * 0.2          # x: "previous layer's output", both signs
    x[::7] = 0.0                                                       # exact zeros take the slope side, like the forward
    dout, emb, demb = rnd(N, D), None, None
    if pooled:
        emb = torch.cat([torch.stack([out[plan.graph_ptr[g]:plan.graph_ptr[g + 1]].max(0).values for g in range(B)]),
                         torch.zeros(B, D, device="cuda")], 1).contiguous()
        demb, dout = rnd(B, 2 * D), None
    gpt = HF.fused_graphs_per_tile(plan, feat, D)
    assert gpt > 0
    wsb = lib.hcg_fused_workspace_bytes(B, feat, D, gpt)

    def bwd(dout_, out_, x_, W_, F_, flags, want_dx=True):
        ws = torch.empty(wsb if F_ == feat else lib.hcg_fused_workspace_bytes(B, F_, D, gpt), dtype=torch.uint8, device="cuda")
        dx = torch.full((N, F_), float("nan"), device="cuda") if want_dx else None
        dW, db = torch.empty(D, F_, device="cuda"), torch.empty(D, device="cuda")
        rc = lib.hcg_fused_layer_bwd(p(dout_), p(demb) if dout_ is None else None, p(emb) if dout_ is None else None, p(out_),
                                     None, p(x_), p(W_), p(plan.edge_index), plan.E, p(plan.graph_ptr), p(plan.edge_ptr), N, B, F_, D,
                                     gpt, slope, flags, p(dx), p(plan.status), p(ws), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, "hcg_fused_layer_bwd")
        _fused_reduce(lib, _lib, ws, ws.numel(), N, B, F_, D, gpt, dW, db)
        return dx, dW, db

    dx, dW, db = bwd(dout, out, x, W, feat, 1)
    dxm, dWm, dbm = bwd(dout, out, x, W, feat, 3)
    assert torch.equal(dW, dWm) and torch.equal(db, dbm)
    assert torch.equal(dxm, dx * torch.where(x > 0, 1.0, slope).to(dx.dtype))
    if feat == 64:      # the layer below: premasked gradient, no activation, no `out` -- vs the plain call on the raw gradient
        x0, W0 = rnd(N, 25), rnd(D, 25) * 0.2
        _, dW_a, db_a = bwd(dx, x, x0, W0, 25, 1, want_dx=False)
        _, dW_b, db_b = bwd(dxm, None, x0, W0, 25, 0, want_dx=False)
        assert torch.equal(dW_a, dW_b) and torch.equal(db_a, db_b)
    # misuse is refused, not ignored
    assert lib.hcg_fused_layer_bwd(p(dx), None, None, None, None, p(x), p(W), p(plan.edge_index), plan.E, p(plan.graph_ptr),
                                   p(plan.edge_ptr), N, B, feat, D, gpt, slope, 1, None, p(plan.status),
                                   p(torch.empty(wsb, dtype=torch.uint8, device="cuda")), wsb, _lib.stream_ptr()) != 0


@pytest.mark.parametrize("nodes,feat,ties", [(30, 64, False), (10, 64, False), (30, 25, False), (7, 64, True), (30, 64, True)])
def test_training_forms_keep_the_pooled_layer_on_chip(H, nodes, feat, ties):
    """hcg_fused_forward with poolbits + hcg_fused_layer_bwd with poolbits: the pooled layer's activations are replaced by two bits per
    element.  Same arithmetic per element as the plain forms -> emb, out1, dW, dx bitwise equal; db is summed in another
    (fixed) order.  `ties`: W2 = 0 makes every node of a graph the column maximum (gradient split n ways)."""
    from hcatgnet_amd import synth, _lib
    from hcatgnet_amd import functional as HF
    from hcatgnet_amd.plan import BatchPlan
    lib, p = _lib.load(), _lib.ptr
    sb = synth.make_config("C2", num_graphs=203, nodes=nodes, seed=3)
    b = sb.as_batch("cuda")
    plan = BatchPlan.build(b.edge_index, b.batch, b.x.shape[0], num_graphs=b.num_graphs, mode="blocked",
                           max_nodes=sb.max_nodes, max_edges=sb.max_edges)
    N, B, D, slope = plan.N, plan.B, 64, 0.01
    gen = torch.Generator().manual_seed(11)
    rnd = lambda *s: torch.randn(*s, generator=gen).cuda()
    x = b.x[:, :feat].contiguous()
    W1, b1, W2, b2 = rnd(D, feat) * 0.2, rnd(D) * 0.1, rnd(D, D) * (0.0 if ties else 0.2), rnd(D) * 0.1
    gpt = HF.fused_graphs_per_tile(plan, feat, D)
    assert gpt == 32 // nodes
    st, geo = _lib.stream_ptr(), (p(plan.edge_index), plan.E, p(plan.graph_ptr), p(plan.edge_ptr), N, B)
    new = lambda *s: torch.full(s, float("nan"), device="cuda")
    out1, out2, emb = new(N, D), new(N, D), new(B, 2 * D)
    common = dict(edge_index=plan.edge_index, E=plan.E, graph_ptr=plan.graph_ptr, edge_ptr=plan.edge_ptr, N=N, B=B, D=D,
                  graphs_per_tile=gpt, apply_act=1, slope=slope, status=plan.status)
    _lib.fused_forward(x=x, W1=W1, b1=b1, W2=W2, b2=b2, F=feat, out1=out1, out2=out2, emb=emb, **common)
    bits = torch.zeros(lib.hcg_fused_aux_bytes(_lib.HCG_FUSED_POOLBITS, B, gpt), dtype=torch.uint8, device="cuda")
    out1t, embt = new(N, D), new(B, 2 * D)
    _lib.fused_forward(x=x, W1=W1, b1=b1, W2=W2, b2=b2, F=feat, out1=out1t, emb=embt, poolbits=bits, **common)
    assert torch.equal(out1, out1t) and torch.equal(emb, embt)
    # single pooled layer on its own (the form a stack of != 2 layers ends with): same bits, same emb
    bits1, emb1 = torch.zeros_like(bits), new(B, 2 * D)
    _lib.fused_forward(x=out1, W1=W2, b1=b2, F=D, emb=emb1, poolbits=bits1, **common)
    assert torch.equal(emb1, emb) and torch.equal(bits1, bits)

    demb = rnd(B, 2 * D)
    wsb = lib.hcg_fused_workspace_bytes(B, D, D, gpt)

    def run(*lead):
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        dx, dW, db = new(N, D), new(D, D), new(D)
        _lib.check(lib.hcg_fused_layer_bwd(*lead, p(out1), p(W2), *geo, D, D, gpt, slope, 3, p(dx), p(plan.status), p(ws), wsb, st), "bwd")
        _fused_reduce(lib, _lib, ws, wsb, N, B, D, D, gpt, dW, db)
        return dx, dW, db

    dx, dW, db = run(None, p(demb), p(emb), p(out2), None)
    dxt, dWt, dbt = run(None, p(demb), None, None, p(bits))
    assert torch.equal(dx, dxt) and torch.equal(dW, dWt)
    assert rel_inf(dbt, db) <= 1e-6
    if ties:        # every node shares the max: db of the max half = sum_g demb_max[g] (n_g shares of 1/n_g each)
        lk = torch.where(emb[0, :D] > 0, 1.0, slope)
        want = (demb[:, :D].sum(0) + demb[:, D:].sum(0)) * lk
        assert rel_inf(dbt, want) <= 1e-5
    assert int(plan.status[0]) == 0


@pytest.mark.parametrize("seed,n_conv", [(0, 2), (1, 2), (2, 2), (3, 2), (4, 1), (5, 3), (6, 3), (7, 1)])
def test_fused_train_step_random_batches(H, oracle, seed, n_conv):
    """Randomised sweep of the no-autograd step (on-chip pooled layer, premasked dx, one-launch head): random graph
    sizes (1..32 nodes, empty graph slots, several graphs per tile on even seeds), random multigraph edges (self loops,
    duplicates, one direction only, isolated nodes), random F, 1-3 conv layers, 1-3 targets -- every gradient vs the
    oracle's autograd."""
    from hcatgnet_amd.train import FusedTrainStep
    g = torch.Generator().manual_seed(300 + seed)
    B = int(torch.randint(1, 90, (1,), generator=g))
    feat = int(torch.randint(7, 65, (1,), generator=g))
    C = 1 + seed % 3
    sizes = torch.randint(0, 33, (B,), generator=g)
    if seed % 2 == 0:
        sizes = sizes.clamp(max=10)
    sizes[int(torch.randint(0, B, (1,), generator=g))] = max(1, int(sizes.max()))
    xs, eis, bs, off, max_e = [], [], [], 0, 0
    for gi, n in enumerate(sizes.tolist()):
        if n == 0:
            continue
        xs.append(torch.randn(n, feat, generator=g))
        bs.append(torch.full((n,), gi, dtype=torch.int64))
        ne = int(torch.randint(0, 3 * n + 1, (1,), generator=g))
        if ne:
            eis.append(torch.randint(0, n, (2, ne), generator=g) + off)
        max_e = max(max_e, ne)
        off += n
    x, b = torch.cat(xs), torch.cat(bs)
    ei = torch.cat(eis, 1) if eis else torch.zeros(2, 0, dtype=torch.int64)
    y = torch.randn(B, C, generator=g)
    params = _rand_params(feat, 64, n_conv=n_conv, n_classes=C, seed=400 + seed)
    m = _model_from_params(H, params)
    batch = H.Batch(x.cuda(), ei.cuda(), b.cuda(), B, y=y.cuda(), max_nodes=int(sizes.max()), max_edges=max_e, edges_grouped=True)
    step = FusedTrainStep(m, optimizer_step=False)
    assert step.unsupported_reason(m, batch) is None
    loss = step(batch)
    assert batch._hcg_plan.check_status() == 0
    # oracle autograd on the same restatement the other parity tests use (targets with C columns)
    p64 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    o_out = oracle.gcn_forward(p64, x, ei, b, B)[0]
    o_loss = torch.sqrt(F.mse_loss(o_out, y))
    o_loss.backward()
    assert abs(float(loss) - float(o_loss.detach())) <= TOL * max(1.0, abs(float(o_loss.detach())))
    assert rel_inf(step.last_out, o_out.detach(), floor=1.0) <= TOL
    for k, v in m.named_parameters():
        assert rel_inf(v.grad, p64[k].grad) <= 2 * TOL, k


@pytest.mark.parametrize("combine", ["mean", "sse"])
def test_fused_train_step_embedding_dim_128(H, oracle, combine):
    """BASELINE configs[4] (C5: 200-node graphs, 128-d): the no-autograd step with the any-shape head (five launches) and
    the one-graph-per-workgroup conv kernels, two 64-column halves per layer; loss and every gradient vs the oracle."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.train import FusedTrainStep
    sb = synth.make_config("C5", num_graphs=24)
    m = H.make_network("GCN", H.default_options(embedding_dim=128), 128).cuda()
    with torch.no_grad():
        for prm in m.parameters():
            if prm.dim() == 1:
                prm.add_(0.05)
    assert FusedTrainStep.unsupported_reason(m, sb.as_batch("cuda")) is None
    step = FusedTrainStep(m, optimizer_step=False, combine=combine)
    loss = step(sb.as_batch("cuda"))
    l_ref, out_ref, _, g_ref = _oracle_grads(oracle, m, sb)
    assert abs(float(loss) - float(l_ref)) <= TOL * abs(float(l_ref))
    assert rel_inf(step.last_out, out_ref, floor=1.0) <= TOL
    for name, prm in m.named_parameters():
        assert rel_inf(prm.grad, g_ref[name]) <= TOL, name
    l_eval = step.evaluate(sb.as_batch("cuda"))              # forward + loss only (the reference's eval_network body)
    assert abs(float(l_eval) - float(l_ref)) <= TOL * abs(float(l_ref))
    # with the update, captured: three replays follow three eager steps of a twin
    twin = H.make_network("GCN", H.default_options(embedding_dim=128), 128).cuda()
    twin.load_state_dict(m.state_dict())
    a, b = FusedTrainStep(m, combine=combine), FusedTrainStep(twin, combine=combine)
    batch = sb.as_batch("cuda")
    a.capture(lambda: batch)
    for _ in range(2):
        b(batch)
    for _ in range(3):
        la, lb = float(a.replay()), float(b(batch))
        assert abs(la - lb) <= 1e-5 * abs(lb)


@pytest.mark.parametrize("feat,seed", [(64, 3), (25, 4)])
def test_size_grouped_batch_runs_two_kernel_families(H, oracle, feat, seed):
    """A ragged batch (n_g ~ U{24..36}) collated with group_by_size: graphs <= 32 nodes through the small-graph tiles, the
    others one graph per wave, slabs of both in ONE reduction job per layer.  Loss / outputs / every gradient vs the
    oracle, equal (to rounding) to the ungrouped route of the same batch, and the captured step replays it."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.train import FusedTrainStep
    sb = synth.make_config("C2", num_graphs=300, nodes_jitter=6, group_by_size=True, seed=seed)
    if feat != 64:
        sb.x = sb.x[:, :feat].contiguous()
    assert 0 < sb.n_small < sb.num_graphs and sb.max_nodes > 32
    m = H.make_network("GCN", H.default_options(), feat).cuda()
    with torch.no_grad():
        for prm in m.parameters():
            if prm.dim() == 1:
                prm.add_(0.05)
    step = FusedTrainStep(m, optimizer_step=False)
    batch = sb.as_batch("cuda")
    assert step._size_groups(batch, m._plan_for(batch, batch.x, batch.edge_index, batch.batch, None),
                             [m.conv1] + list(m.conv_layers), 64, 1, 2) == sb.n_small
    loss = float(step(batch))
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    l_ref, out_ref, _, g_ref = _oracle_grads(oracle, m, sb)
    assert abs(loss - float(l_ref)) <= TOL * abs(float(l_ref))
    assert rel_inf(step.last_out, out_ref, floor=1.0) <= TOL
    for name in grads:
        assert rel_inf(grads[name], g_ref[name]) <= TOL, name
    plain = sb.as_batch("cuda")
    plain.n_small = None                                   # the same batch, everything one graph per wave
    loss2 = float(step(plain))
    assert abs(loss2 - loss) <= 2e-6 * abs(loss)
    for name, prm in m.named_parameters():
        assert rel_inf(prm.grad, grads[name]) <= TOL, name
    full = FusedTrainStep(m)
    twin = H.make_network("GCN", H.default_options(), feat).cuda()
    twin.load_state_dict(m.state_dict())
    eager = FusedTrainStep(twin)
    full.capture(lambda: batch)
    for _ in range(2):
        eager(batch)
    for _ in range(3):
        la, lb = float(full.replay()), float(eager(batch))
        assert abs(la - lb) <= 1e-5 * abs(lb)
    # (the captured step runs the two families as two branches of the graph: same launches, same results)
    for pa, pb in zip(m.parameters(), twin.parameters()):
        assert rel_inf(pa, pb) <= 1e-6
