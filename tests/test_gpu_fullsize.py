"""Full-size gradient parity: the no-autograd training step (train.FusedTrainStep, through the C ABI) against the CPU
oracle at the sizes `bench.py` times -- sizes at which EVERY persistent kernel loop iterates:

  C3    4096 x 30 atoms, 64-d      small-graph tiles (fused.hip): 4096 tiles on 2048 waves = 2 per wave
  C5    1024 x 200 atoms, 128-d    k_seg_fwd: 4 graphs per workgroup; k_gseg_bwd >= 2; k_tall_dw: 3200 tiles / 256 = 12.5;
                                   k_tall_mm; k_head<128>
  REAL  4096 x 57-117 atoms, F=25  k_mid_layer_fwd: 8 graphs per workgroup; k_gseg_bwd / k_tall_dw<XVEC=false> / k_tall_mm
  BIARYL 2048 x 57-117 atoms, F=32 the same with 16-byte aligned rows (XVEC)
  RAGGED 4096 x U{24..36} atoms    size-grouped: tiles on the <= 32-node graphs, one graph per wave (wave.hip, ~2 graphs
                                   per wave slot) on the others, ONE reduction job per layer

Compared: loss, outputs, pooled embedding, layer-1 node embeddings (<= 1e-5, north_star) and EVERY gradient -- against the
fp64 oracle (<= 1e-5 biases and readout, <= 1e-4 conv weights: sums of > 1e5 terms) and the fp32 oracle
(<= 1e-4).  The batches are screened with oracle/screen.py so that the comparison is decidable (no activation within
2e-6 of the LeakyReLU kink, no max-pool near-tie): see that file for why a full-size batch needs it.
Reference step: utils/utils_model.py:60-66; model/gcn.py:54-76."""
import pytest
import torch

from tests.helpers import rel_inf
from tests.test_gpu_parity import H, oracle, _model_from_params, _rand_params, TOL, TOL_DW  # noqa: F401

pytestmark = pytest.mark.gpu

CASES = {
    #        config  kwargs                                              F    D    seed
    "C3":     ("C2", dict(),                                             64,  64,  19),
    "C5":     ("C5", dict(),                                             128, 128, 23),
    "REAL":   ("REAL", dict(),                                           25,  64,  29),
    "BIARYL": ("REAL", dict(num_graphs=2048, feat=32),                   32,  64,  31),
    "RAGGED": ("C2", dict(nodes_jitter=6, group_by_size=True),           64,  64,  37),
    # a partial last round of tiles, dealt evenly over the workgroups (fused.hip: TileSeq<DEAL>) with the head in the forward's
    # tail: 2500 tiles on 2048 waves; 7001 graphs of ~10 atoms = 2334 tiles of three graphs, the last one holding ONE graph
    "C3_2500": ("C2", dict(num_graphs=2500),                             64,  64,  41),
    # (trees: a triangle makes structural twins -- two nodes with the same closed neighbourhood have EQUAL outputs in exact
    #  arithmetic, so which of them is "the" maximum is rounding's call and no re-draw of the features makes it decidable)
    "TINY_7001": ("C2", dict(num_graphs=7001, nodes=10, nodes_jitter=0, extra_bonds=0), 64, 64, 43),
}


def _decidable_batch(case):
    from hcatgnet_amd import synth
    from oracle import screen
    name, kw, F, D, seed = CASES[case]
    sb = synth.make_config(name, **kw)
    params = _rand_params(F, D, seed=seed)
    sb.x, redrawn = screen.make_decidable(params, sb.x, sb.edge_index, sb.batch, sb.num_graphs, seed=seed)
    assert redrawn < 0.3 * sb.num_graphs
    return sb, params


@pytest.mark.parametrize("case", list(CASES))
def test_full_size_step_every_gradient_vs_oracle(H, oracle, case):
    from hcatgnet_amd.train import FusedTrainStep
    sb, params = _decidable_batch(case)
    m = _model_from_params(H, params)
    batch = sb.as_batch("cuda")
    step = FusedTrainStep(m, optimizer_step=False)
    assert step.reason(batch) is None
    loss = float(step(batch))
    assert batch._hcg_plan.check_status() == 0
    N, B = sb.x.shape[0], sb.num_graphs
    if case == "RAGGED":
        assert 0 < sb.n_small < B and sb.max_nodes > 32          # both families ran
    if case in ("C3_2500", "TINY_7001"):
        gpt = 32 // sb.max_nodes
        tiles = -(-B // gpt)
        assert tiles > 2048 and tiles % 2048 != 0 and step._head_in_forward(step._prepare(batch, False))
    cap = step._bufs["cap"]
    got = {k: v.grad.detach().clone() for k, v in m.named_parameters()}
    p32 = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    o_out, o_emb, o_acts = oracle.gcn_forward(p32, sb.x, sb.edge_index, sb.batch, B, return_intermediates=True)
    o_loss = oracle.rmse_loss(o_out, sb.y)
    o_loss.backward()
    assert abs(loss - float(o_loss)) <= TOL * abs(float(o_loss))
    assert rel_inf(step.last_out, o_out, floor=1.0) <= TOL
    assert rel_inf(cap["emb"][:B], o_emb) <= TOL
    assert rel_inf(cap["acts"][0][:N], o_acts[0]) <= TOL
    _, _, _, g64 = oracle.train_step_grads(params, sb.x, sb.edge_index, sb.batch, sb.y, B, dtype=torch.float64)
    worst = {}
    for k in got:
        conv_w = k.endswith("lin.weight")
        e64, e32 = rel_inf(got[k], g64[k]), rel_inf(got[k], p32[k].grad)
        worst[k] = (e64, e32)
        assert e64 <= (TOL_DW if conv_w else TOL), (k, worst)
        assert e32 <= TOL_DW, (k, worst)
    # evaluate() = the forward half alone gives the same loss; a second step is bitwise the first
    flat = step._flat.clone()
    assert abs(float(step.evaluate(batch)) - loss) <= 1e-6 * abs(loss)
    assert float(step(batch)) == loss and torch.equal(step._flat, flat)


def test_full_size_window_of_distinct_batches_equals_eager_steps(H):
    """The bench's launch form at full size: 4 distinct C3 batches, each with its own trainer and a persistent plan that
    the PREVIOUS step's last launch re-derives (`next_plan`), all 4 steps ONE hipGraph -- two replays equal 8 eager
    steps of a twin model (fresh batches, plan launch in front of every forward) bitwise: losses and weights."""
    from hcatgnet_amd import synth
    from hcatgnet_amd.train import FusedTrainStep, StepWindow
    nb = 4
    sbs = [synth.make_config("C3", seed=synth.BASE_SEED + 17 * i) for i in range(nb)]
    dev = [(sb.x.cuda(), sb.edge_index.cuda(), sb.batch.cuda(), sb.y.cuda()) for sb in sbs]

    def fresh(i):
        x, ei, bv, y = dev[i]
        return H.Batch(x, ei, bv, sbs[i].num_graphs, y=y, max_nodes=sbs[i].max_nodes, max_edges=sbs[i].max_edges, edges_grouped=True)
    plans = [H.BatchPlan.build(dev[i][1], dev[i][2], dev[i][0].shape[0], num_graphs=sbs[i].num_graphs, mode="blocked", validate=False,
                               max_nodes=sbs[i].max_nodes, max_edges=sbs[i].max_edges) for i in range(nb)]

    def planned(i):
        b = fresh(i)
        b._hcg_plan = plans[i]
        return b
    a = H.make_network("GCN", H.default_options(), 64).cuda()
    b = H.make_network("GCN", H.default_options(), 64).cuda()
    b.load_state_dict(a.state_dict())
    steps = [FusedTrainStep(a) for _ in range(nb)]
    for i, st in enumerate(steps):
        st.next_plan = plans[(i + 1) % nb]
    eager = FusedTrainStep(b)
    win = StepWindow(steps, [lambda i=i: planned(i) for i in range(nb)])     # (its warm-up runs every step once on model a)
    for i in range(nb):
        eager(fresh(i))
    for _ in range(2):
        la = [float(x) for x in win.replay()]
        lb = [float(eager(fresh(i))) for i in range(nb)]
        assert la == lb
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)
    for pl in plans:
        assert pl.check_status() == 0
