"""Data-parallel step on the GPU: the REAL step -- fused forward / backward kernels, flat gradient buffer,
exchange, update -- split over two ranks must reproduce the single-process gradient:
  combine="mean"  the gradient of the MEAN OF PER-RANK sqrt(MSE)           (DDP convention)
  combine="sse"   the gradient of sqrt(MSE) over the CONCATENATED batch    (the reference's single-device step,
                                                                            utils/utils_model.py:64-65, at batch 2B)
both against the CPU oracle's autograd (nothing in the reference pins gradients: the oracle is the checker).
A one-GPU box cannot run two RCCL ranks on one device, so the two ranks of `test_two_ranks_*` share cuda:0 and
exchange through gloo (the rehearsal `tools/rehearse_multi_rank.sh` uses too); the exchange code above the backend is
the same.  Plus the robustness items of the trainer: capture on a fixed Batch object, two trainers on one model, two
trainers on two streams, stale graphs."""
import os
import socket

import pytest
import torch

from tests.helpers import rel_inf
from tests.test_gpu_parity import H, oracle  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def _halves(sb, H):
    """Split a synthetic batch into two batches of whole graphs (first / second half)."""
    from hcatgnet_amd.batch import collate
    gl = sb.as_graph_list()
    k = len(gl) // 2
    return collate(gl[:k]), collate(gl[k:])


def _oracle_of(oracle, params, b):
    return oracle.train_step_grads(params, b.x, b.edge_index, b.batch, b.y, b.num_graphs, dtype=torch.float64)


def _grads(m):
    return {k: p.grad.detach().clone() for k, p in m.named_parameters()}


def test_sse_form_on_one_rank_equals_the_plain_step(H, oracle):
    from hcatgnet_amd import synth
    from hcatgnet_amd.train import FusedTrainStep
    sb = synth.make_config("C2", num_graphs=192)
    m = H.make_network("GCN", H.default_options(), 64).cuda()
    plain = FusedTrainStep(m, optimizer_step=False)
    sse = FusedTrainStep(m, optimizer_step=False, combine="sse")
    l0 = float(plain(sb.as_batch("cuda"))); g0 = _grads(m)
    l1 = float(sse(sb.as_batch("cuda"))); g1 = _grads(m)
    assert abs(l0 - l1) <= 1e-6 * abs(l0)
    for k in g0:
        assert rel_inf(g1[k], g0[k]) <= 2e-6, k
    # ... and with the update: the same parameters after three steps
    ma = H.make_network("GCN", H.default_options(), 64).cuda()
    mb = H.make_network("GCN", H.default_options(), 64).cuda()
    mb.load_state_dict(ma.state_dict())
    sa, sb2 = FusedTrainStep(ma), FusedTrainStep(mb, combine="sse")
    for _ in range(3):
        la, lb = float(sa(sb.as_batch("cuda"))), float(sb2(sb.as_batch("cuda")))
        assert abs(la - lb) <= 1e-5 * abs(la)
    # (loose: Adam with eps = 1e-9 turns rounding-level gradient differences into lr-sized steps on elements whose
    #  gradient is ~0; the gradient comparison above is the tight one -- Adam is invariant to the gradient's scale)
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        assert rel_inf(pb, pa) <= 2e-3


def test_sse_halves_summed_by_hand_give_the_concatenated_batch_gradient(H, oracle):
    """One process plays both ranks: half A's [gradients | SSE | count] is kept, half B's exchange hook adds it --
    exactly what the all-reduce(sum) leaves on every rank -- and the one scale must give the oracle's gradient of
    sqrt(MSE) over all graphs.  The halves have DIFFERENT numbers of graphs."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.batch import collate
    from hcatgnet_amd.train import FusedTrainStep
    sb = synth.make_config("C2", num_graphs=96)
    gl = sb.as_graph_list()
    a, b = collate(gl[:40]), collate(gl[40:])
    m = H.make_network("GCN", H.default_options(), 64).cuda()
    params = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    kept = {}
    step_a = FusedTrainStep(m, optimizer_step=False, combine="sse", grad_sync=lambda ext: kept.__setitem__("a", ext.clone()))
    step_a(a.to("cuda"))
    step_b = FusedTrainStep(m, optimizer_step=False, combine="sse", grad_sync=lambda ext: ext.add_(kept["a"]))
    loss = float(step_b(b.to("cuda")))
    l_ref, _, _, g_ref = _oracle_of(oracle, params, collate(gl))
    assert abs(loss - float(l_ref)) <= 1e-6 * abs(float(l_ref))
    for k, p in m.named_parameters():
        assert rel_inf(p.grad, g_ref[k]) <= 1e-5, k


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, combine, q):
    """One rank of the two-rank job: its half of the batch through the real fused step, gloo exchange on cuda:0."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import hcatgnet_amd as H
        from hcatgnet_amd import synth
        from hcatgnet_amd.batch import collate
        from hcatgnet_amd.ddp import DataParallelGCN
        sb = synth.make_config("C2", num_graphs=128)
        gl = sb.as_graph_list()
        mine = collate(gl[:64] if rank == 0 else gl[64:]).to("cuda")
        torch.manual_seed(1234 + rank)                      # ranks start from DIFFERENT weights: the wrapper re-syncs them
        m = H.make_network("GCN", H.default_options(global_seed=1234 + rank), 64).cuda()
        dp = DataParallelGCN(m, combine=combine)
        w0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        step = dp.make_train_step(optimizer_step=False)
        loss = float(step(mine))
        grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
        # the same step captured (graph ends before the exchange) must give the same gradients
        cap = dp.make_train_step(optimizer_step=False)
        cap.grad_sync = lambda flat: None
        cap.capture(mine)
        dp.attach(cap)
        loss_c = float(cap.replay())
        grads_c = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
        # and a full data-parallel training step keeps the ranks' weights identical
        tr = dp.make_train_step()
        for _ in range(2):
            tr(mine)
        w_after = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
        npy = lambda d: {k: v.numpy().copy() for k, v in d.items()}     # by value: this process exits before the parent reads
        q.put((rank, npy(w0), loss, npy(grads), loss_c, npy(grads_c), w_after.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("combine", ["mean", "sse"])
def test_two_ranks_real_step_equals_the_single_process_gradient(H, oracle, combine):
    import torch.multiprocessing as mp
    from hcatgnet_amd import synth
    from hcatgnet_amd.batch import collate
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, combine, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    tt = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    (_, w0, l0, g0, lc0, gc0, wa0), (_, w1, l1, g1, lc1, gc1, wa1) = [
        (r[0], tt(r[1]), r[2], tt(r[3]), r[4], tt(r[5]), torch.from_numpy(r[6])) for r in res]
    for k in w0:                                            # rank 0's weights were broadcast
        assert torch.equal(w0[k], w1[k]), k
    sb = synth.make_config("C2", num_graphs=128)
    gl = sb.as_graph_list()
    a, b, whole = collate(gl[:64]), collate(gl[64:]), collate(gl)
    if combine == "sse":        # gradient of sqrt(MSE) over the concatenated batch; every rank reports that loss
        l_ref, _, _, g_ref = _oracle_of(oracle, w0, whole)
        assert abs(l0 - float(l_ref)) <= 1e-6 * abs(float(l_ref)) and abs(l1 - float(l_ref)) <= 1e-6 * abs(float(l_ref))
    else:                       # gradient of the mean of the two ranks' own sqrt(MSE)
        la, _, _, ga = _oracle_of(oracle, w0, a)
        lb, _, _, gb = _oracle_of(oracle, w0, b)
        g_ref = {k: 0.5 * (ga[k] + gb[k]) for k in ga}
        assert abs(l0 - float(la)) <= 1e-6 * abs(float(la)) and abs(l1 - float(lb)) <= 1e-6 * abs(float(lb))
    for k in g_ref:
        # 1e-6 of the gradient's scale for the small tensors; weight gradients sum ~4e3 node terms in fp32 (<= 1e-5)
        tol = 1e-6 if g_ref[k].numel() <= 64 else 1e-5
        for g in (g0, g1, gc0, gc1):
            assert rel_inf(g[k], g_ref[k]) <= tol, (k, rel_inf(g[k], g_ref[k]))
        assert torch.equal(g0[k], g1[k]), k                 # both ranks hold the SAME bits after the exchange
    assert abs(lc0 - l0) <= 1e-6 * abs(l0) and abs(lc1 - l1) <= 1e-6 * abs(l1)
    assert torch.equal(wa0, wa1)                            # two full DP steps: weights stay bitwise in sync


def test_capture_on_a_fixed_batch_object_follows_new_graph_boundaries(H, oracle):
    """capture(batch) with a Batch OBJECT must capture the plan build too.  Capture on one ragged batch, copy a
    differently partitioned batch of the same N / E / B into the tensors, replay: the result must be the second batch's."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.batch import collate
    from hcatgnet_amd.train import FusedTrainStep
    sb = synth.make_config("C2", num_graphs=64, nodes=26, nodes_jitter=5, seed=5)
    gl = sb.as_graph_list()
    first = collate(gl)
    second = collate(gl[::-1])                              # same graphs, reversed order: same N / E / B, other boundaries
    assert first.num_nodes == second.num_nodes and not torch.equal(first.batch, second.batch)
    m = H.make_network("GCN", H.default_options(), 64).cuda()
    params = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    dev = first.to("cuda")
    step = FusedTrainStep(m, optimizer_step=False)
    step.capture(dev)
    l1 = float(step.replay())
    ref1 = _oracle_of(oracle, params, first)
    assert abs(l1 - float(ref1[0])) <= 1e-5 * abs(float(ref1[0]))
    for name in ("x", "edge_index", "batch", "y"):
        getattr(dev, name).copy_(getattr(second, name).to("cuda"))
    l2 = float(step.replay())
    ref2 = _oracle_of(oracle, params, second)
    assert abs(l2 - float(ref2[0])) <= 1e-5 * abs(float(ref2[0]))
    for k, p in m.named_parameters():
        assert rel_inf(p.grad, ref2[3][k]) <= 1e-5, k


def test_two_trainers_on_one_model_interleaved_use_their_own_gradients(H):
    """`replay()` / the eager exchange branch must update from THIS trainer's flat buffer even when another
    trainer on the same model ran last (it re-points the parameters' .grad)."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.train import FusedTrainStep
    sa, sb = synth.make_config("C2", num_graphs=128), synth.make_config("C2", num_graphs=128, seed=77)
    ba, bb = sa.as_batch("cuda"), sb.as_batch("cuda")
    m = H.make_network("GCN", H.default_options(), 64).cuda()
    nothing = lambda flat: None                             # an exchange hook: reduction and update are separate launches
    ta = FusedTrainStep(m, grad_sync=nothing)
    tb = FusedTrainStep(m, grad_sync=nothing)
    ta.capture(lambda: ba)                                  # (each capture runs two warm-up steps on m)
    tb.capture(lambda: bb)
    ref = H.make_network("GCN", H.default_options(), 64).cuda()
    ref.load_state_dict(m.state_dict())
    ref.optimizer.load_state_dict(m.optimizer.state_dict())
    one = FusedTrainStep(ref)
    want = [float(one(b)) for b in (ba, bb, ba, bb)]
    got = [float(ta.replay()), float(tb(bb)), float(ta.replay()), float(tb.replay())]
    for u, v in zip(want, got):
        assert abs(u - v) <= 1e-5 * abs(u), (want, got)
    for pa, pb in zip(m.parameters(), ref.parameters()):
        assert rel_inf(pa, pb) <= 1e-5


def test_two_trainers_on_two_streams_do_not_share_exchange_words(H):
    """every trainer owns its sync words, so two head launches in flight on two streams neither hang
    nor mix their partial sums: the interleaved losses equal the single-stream ones."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.train import FusedTrainStep
    sa, sb = synth.make_config("C2", num_graphs=4096), synth.make_config("C2", num_graphs=4096, seed=3)
    ba, bb = sa.as_batch("cuda"), sb.as_batch("cuda")
    ma = H.make_network("GCN", H.default_options(), 64).cuda()
    mb = H.make_network("GCN", H.default_options(), 64).cuda()
    ta, tb = FusedTrainStep(ma, optimizer_step=False), FusedTrainStep(mb, optimizer_step=False)
    la, lb = float(ta(ba)), float(tb(bb))                   # single stream
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    got = []
    for _ in range(20):
        with torch.cuda.stream(s1):
            xa = ta(ba)
        with torch.cuda.stream(s2):
            xb = tb(bb)
        got.append((xa, xb))
    torch.cuda.synchronize()
    assert ta._bufs["cap"]["ws_head"].data_ptr() != tb._bufs["cap"]["ws_head"].data_ptr()    # nothing shared between trainers
    assert abs(float(got[-1][0]) - la) <= 1e-6 * abs(la) and abs(float(got[-1][1]) - lb) <= 1e-6 * abs(lb)


def test_replay_refuses_a_stale_graph(H):
    from hcatgnet_amd import _lib, synth
    from hcatgnet_amd.train import FusedTrainStep
    small, big = synth.make_config("C2", num_graphs=64), synth.make_config("C2", num_graphs=256)
    m = H.make_network("GCN", H.default_options(), 64).cuda()
    step = FusedTrainStep(m)
    bs = small.as_batch("cuda")
    step.capture(lambda: bs)
    step.replay()
    step(big.as_batch("cuda"))                              # a larger eager batch re-allocates the step buffers
    with pytest.raises(_lib.HcgError):
        step.replay()
    step.capture(lambda: bs)
    step.replay()
    m.optimizer.load_state_dict(m.optimizer.state_dict())   # new flat moment storages
    with pytest.raises(_lib.HcgError):
        step.replay()


def test_plan_of_the_next_batch_on_a_forked_graph_branch(H, oracle):
    """capture(..., prefetch=next_plan.rebuild): the next batch's plan build runs on a forked branch of this step's graph;
    two trainers ping-pong over two batches whose tensors get NEW contents between replays -- every replay must see the
    plan of the data its tensors hold."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.batch import collate
    from hcatgnet_amd.train import FusedTrainStep
    sb = synth.make_config("C2", num_graphs=96, nodes=24, nodes_jitter=8, seed=11)
    gl = sb.as_graph_list()
    a, b = collate(gl[:48]).to("cuda"), collate(gl[48:]).to("cuda")
    m = H.make_network("GCN", H.default_options(), 64).cuda()
    params = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}

    def plan_of(bt):
        return H.BatchPlan.build(bt.edge_index, bt.batch, bt.x.shape[0], num_graphs=bt.num_graphs, mode="blocked", validate=False,
                                 max_nodes=32, max_edges=bt.max_edges + 16)
    pa, pb = plan_of(a), plan_of(b)

    def carrying(bt, pl):
        def get():
            bt._hcg_plan = pl
            return bt
        return get
    ta, tb = FusedTrainStep(m, optimizer_step=False), FusedTrainStep(m, optimizer_step=False)
    ta.capture(carrying(a, pa), prefetch=pb.rebuild)
    tb.capture(carrying(b, pb), prefetch=pa.rebuild)
    for bt, step in ((a, ta), (b, tb)):
        loss = float(step.replay())
        ref = _oracle_of(oracle, params, bt.to("cpu"))
        assert abs(loss - float(ref[0])) <= 1e-5 * abs(float(ref[0]))
    # new contents for batch a (same sizes, other graph boundaries): tb's replay rebuilds a's plan beside its own step
    a2 = collate(gl[:48][::-1])
    for name in ("x", "edge_index", "batch", "y"):
        getattr(a, name).copy_(getattr(a2, name).to("cuda"))
    tb.replay()
    loss = float(ta.replay())
    ref = _oracle_of(oracle, params, a2)
    assert abs(loss - float(ref[0])) <= 1e-5 * abs(float(ref[0]))
    for k, p in m.named_parameters():
        assert rel_inf(p.grad, ref[3][k]) <= 1e-5, k


def test_next_batch_plan_inside_the_last_launch_of_the_step(H, oracle):
    """capture(..., next_plan=plan): the slab reduction + Adam launch of a step also derives graph_ptr / edge_ptr of the NEXT
    batch; the next step starts on a batch that carries that plan and launches no plan kernel.  Two trainers ping-pong over
    two ragged batches; one batch gets new contents (other graph boundaries) between replays."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.batch import collate
    from hcatgnet_amd.train import FusedTrainStep
    sb = synth.make_config("C2", num_graphs=96, nodes=24, nodes_jitter=8, seed=11)
    gl = sb.as_graph_list()
    a, b = collate(gl[:48]).to("cuda"), collate(gl[48:]).to("cuda")
    m = H.make_network("GCN", H.default_options(), 64).cuda()

    def plan_of(bt):
        return H.BatchPlan.build(bt.edge_index, bt.batch, bt.x.shape[0], num_graphs=bt.num_graphs, mode="blocked", validate=False,
                                 max_nodes=32, max_edges=bt.max_edges + 16)
    pa, pb = plan_of(a), plan_of(b)

    def carrying(bt, pl):
        def get():
            bt._hcg_plan = pl
            return bt
        return get
    ta, tb = FusedTrainStep(m), FusedTrainStep(m)
    ta.capture(carrying(a, pa), next_plan=pb)
    tb.capture(carrying(b, pb), next_plan=pa)

    def check(step, bt_cpu):
        params = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        loss = float(step.replay())
        ref = _oracle_of(oracle, params, bt_cpu)
        assert abs(loss - float(ref[0])) <= 1e-5 * abs(float(ref[0]))
    check(ta, collate(gl[:48])); check(tb, collate(gl[48:])); check(ta, collate(gl[:48]))
    a2 = collate(gl[:48][::-1])                             # new contents for batch a: same sizes, other graph boundaries
    for name in ("x", "edge_index", "batch", "y"):
        getattr(a, name).copy_(getattr(a2, name).to("cuda"))
    check(tb, collate(gl[48:]))                             # ... tb's last launch re-derives a's plan
    check(ta, a2)
    assert int(pa.status[0].item()) == 0


def _rank_main_xchg(rank, world, port, q):
    """One rank of a two-rank job that exchanges gradients through `xgmi.OneShotExchange` (both ranks on cuda:0: the peers'
    inboxes are mapped through IPC exactly as across GPUs; what a one-GPU box cannot show is the xGMI hop itself)."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import hcatgnet_amd as H
        from hcatgnet_amd import synth
        from hcatgnet_amd.batch import collate
        from hcatgnet_amd.ddp import DataParallelGCN
        from hcatgnet_amd.xgmi import OneShotExchange
        sb = synth.make_config("C2", num_graphs=128)
        gl = sb.as_graph_list()
        mine = collate(gl[:70] if rank == 0 else gl[70:]).to("cuda")        # unequal halves
        m = H.make_network("GCN", H.default_options(), 64).cuda()
        twin = H.make_network("GCN", H.default_options(), 64).cuda()
        dp, dp_twin = DataParallelGCN(m, combine="sse"), DataParallelGCN(twin, combine="sse")
        w0 = {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}
        xchg = OneShotExchange(sum(p.numel() for p in m.parameters()))
        ok_setup = xchg.ok
        ok_test = xchg.self_test() if ok_setup else False
        res = {"rank": rank, "setup": ok_setup, "selftest": ok_test}
        if ok_test:
            def meet():                                                       # one device, two ranks: see FusedTrainStep.pre_exchange_hook
                torch.cuda.synchronize()
                dist.barrier()
            step = xchg.attach(dp.make_train_step())
            step.pre_exchange_hook = meet
            ref = dp_twin.make_train_step()                                   # the collective path: the comparison
            l1, l2 = float(step(mine)), float(ref(mine))
            res["loss"], res["loss_ref"] = l1, l2
            res["grads"] = {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters()}
            import copy
            ckpt = (copy.deepcopy(m.state_dict()), copy.deepcopy(m.optimizer.state_dict()))
            ckpt_ref = (copy.deepcopy(twin.state_dict()), copy.deepcopy(twin.optimizer.state_dict()))
            for _ in range(5):                                                # five more steps (both step parities, Adam state)
                l1, l2 = float(step(mine)), float(ref(mine))
            res["loss3"], res["loss3_ref"] = l1, l2
            res["w"] = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu().numpy().copy()
            res["w_ref"] = torch.cat([p.detach().reshape(-1) for p in twin.parameters()]).cpu().numpy().copy()
            xchg.check()
            # a checkpoint taken after step 1 is reloaded mid-run: the Adam step count goes BACK to 1 (numbers the inbox has
            # already seen), the exchange stamp does not -- three more steps still follow the collective path
            for mm, ck in ((m, ckpt), (twin, ckpt_ref)):
                mm.load_state_dict(ck[0])
                mm.optimizer.load_state_dict(ck[1])
            assert m.optimizer.steps_done() == 1
            for _ in range(3):
                l1, l2 = float(step(mine)), float(ref(mine))
            res["loss_rl"], res["loss_rl_ref"] = l1, l2
            res["stamp"] = int(m.optimizer._flat[0]["step_dev"][1].item())
            res["w_rl"] = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu().numpy().copy()
            res["w_rl_ref"] = torch.cat([p.detach().reshape(-1) for p in twin.parameters()]).cpu().numpy().copy()
            xchg.check()
        res["w0"] = w0
        q.put(res)
        xchg.close()
    finally:
        dist.destroy_process_group()


def test_one_shot_exchange_two_ranks_equals_the_collective_path(H, oracle):
    """`xgmi.OneShotExchange`: set-up (fine-grained inbox, IPC export / open), self test, then the REAL step with the
    exchange inside its last launch: gradients = the oracle's concatenated-batch gradient, losses and weights follow the
    all-reduce path over six steps, both ranks bitwise equal.  (Two ranks on ONE device must meet before the launch that
    polls -- `pre_exchange_hook` -- so the captured, free-running form is only exercised on one GPU per rank.)"""
    import numpy as np
    import torch.multiprocessing as mp
    from hcatgnet_amd import synth
    from hcatgnet_amd.batch import collate
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main_xchg, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0, r1 = res
    assert r0["setup"] and r1["setup"], "one-shot exchange did not set up (IPC of fine-grained memory)"
    assert r0["selftest"] and r1["selftest"], "one-shot exchange failed its self test"
    sb = synth.make_config("C2", num_graphs=128)
    whole = collate(sb.as_graph_list())
    w0 = {k: torch.from_numpy(v) for k, v in r0["w0"].items()}
    l_ref, _, _, g_ref = _oracle_of(oracle, w0, whole)
    for r in (r0, r1):
        assert abs(r["loss"] - float(l_ref)) <= 1e-6 * abs(float(l_ref)) and abs(r["loss"] - r["loss_ref"]) <= 1e-6 * abs(r["loss_ref"])
        assert abs(r["loss3"] - r["loss3_ref"]) <= 1e-4 * abs(r["loss3_ref"])
        for k in g_ref:
            tol = 1e-6 if g_ref[k].numel() <= 64 else 1e-5
            assert rel_inf(torch.from_numpy(r["grads"][k]), g_ref[k]) <= tol, k
        assert float(np.abs(r["w"] - r["w_ref"]).max()) <= 2e-3 * float(np.abs(r["w_ref"]).max())   # (Adam, eps 1e-9: loose)
    assert np.array_equal(r0["w"], r1["w"])                                    # replicas bitwise in sync
    for r in (r0, r1):                                                         # after the mid-run checkpoint reload
        assert abs(r["loss_rl"] - r["loss_rl_ref"]) <= 1e-4 * abs(r["loss_rl_ref"])
        assert float(np.abs(r["w_rl"] - r["w_rl_ref"]).max()) <= 2e-3 * float(np.abs(r["w_rl_ref"]).max())
        assert r["stamp"] >= 9                                                  # 6 + 3 steps (+ nothing re-based it)
    assert np.array_equal(r0["w_rl"], r1["w_rl"])
    for k in r0["grads"]:
        assert np.array_equal(r0["grads"][k], r1["grads"][k]), k


def _rank_main_xchg_generic_head(rank, world, port, q):
    """Two ranks, a one-shot exchange attached, and a model whose readout the one-launch head does not cover (9 classes >
    RCMAX): the update launch cannot carry the exchange, so the step must fall back to the collective."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import hcatgnet_amd as H
        from hcatgnet_amd import _lib, synth
        from hcatgnet_amd.ddp import DataParallelGCN
        from hcatgnet_amd.xgmi import OneShotExchange
        C = 9
        assert not _lib.load().hcg_head_supported(64, C)
        sb = synth.make_config("C2", num_graphs=40 + 10 * rank, rank=rank)
        mine = sb.as_batch("cuda")
        mine.y = torch.randn(sb.num_graphs, C, generator=torch.Generator().manual_seed(5 + rank)).cuda()
        m = H.make_network("GCN", H.default_options(n_classes=C), 64).cuda()
        twin = H.make_network("GCN", H.default_options(n_classes=C), 64).cuda()
        dp, dp_twin = DataParallelGCN(m, combine="sse"), DataParallelGCN(twin, combine="sse")
        xchg = OneShotExchange(sum(p.numel() for p in m.parameters()))
        res = {"rank": rank, "ok": bool(xchg.ok and xchg.self_test())}
        if res["ok"]:
            step = xchg.attach(dp.make_train_step())
            ref = dp_twin.make_train_step()
            losses = [(float(step(mine)), float(ref(mine))) for _ in range(3)]
            res["carried"] = bool(step._last_carried)
            res["losses"] = losses
            res["w"] = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu().numpy().copy()
            res["w_ref"] = torch.cat([p.detach().reshape(-1) for p in twin.parameters()]).cpu().numpy().copy()
            try:                                     # without a collective to fall back on, the step refuses instead of drifting
                lone = FusedTrainStepLone(m, xchg)
                lone(mine)
                res["lone_raised"] = False
            except _lib.HcgError:
                res["lone_raised"] = True
        q.put(res)
        xchg.close()
    finally:
        dist.destroy_process_group()


def FusedTrainStepLone(model, xchg):
    """A trainer with an exchange but no collective behind it (what `attach` now refuses to build)."""
    from hcatgnet_amd.train import FusedTrainStep
    st = FusedTrainStep(model, combine="sse")
    st.exchange = xchg
    return st


def test_one_shot_exchange_falls_back_to_the_collective_when_the_update_cannot_carry_it(H):
    """a trainer with a one-shot exchange attached whose step has no fused update launch (any-shape
    head) used to run with NO gradient exchange.  Now it takes the collective: both ranks' weights stay bitwise equal and
    equal the plain collective path; a trainer with neither raises."""
    import numpy as np
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main_xchg_generic_head, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    r0, r1 = res
    assert r0["ok"] and r1["ok"], "one-shot exchange did not set up / pass its self test"
    for r in (r0, r1):
        assert r["carried"] is False and r["lone_raised"] is True
        assert np.array_equal(r["w"], r["w_ref"])                 # the same collective + update as the plain data-parallel step
        for a, b in r["losses"]:
            assert a == b
    assert np.array_equal(r0["w"], r1["w"])                       # replicas in sync
    assert r0["losses"][0][0] == r1["losses"][0][0]               # "sse": every rank reports the global loss
