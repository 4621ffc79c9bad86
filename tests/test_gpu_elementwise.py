"""Elementwise parity of node embeddings after EVERY conv layer and of the pooled embedding, small-graph tiles
(csrc/fused.hip) and one-graph-per-workgroup kernels (csrc/mid.hip), against the oracle (SURVEY 8d spot-check form:
abs(d) <= 1e-5 * max(abs(ref), 1e-3)).

That bound cannot be met on every element by ANY fp32 evaluation: a node embedding is a ~64-term dot product of O(1)
terms, so a value that cancels to 1e-2 carries ~1e-7 of rounding -- the reference's own fp32 arithmetic (the fp32
oracle, bit-exact with the reference's embeddings.csv) violates it on 0.25-0.45 % of the elements when measured against
the fp64 oracle (tools/measure_elementwise.py).  The gate is therefore stated against the fp64 oracle and
anchored on the reference's own arithmetic, measured the same way on the same batch:
  (1) where abs(ref) >= 0.1 (no cancellation) the bound holds on EVERY element;
  (2) the number of elements outside the bound is no larger than the fp32 oracle's own (+ 5 %), and
  (3) the worst excess over the bound is no larger than 2 x the fp32 oracle's worst excess (an extreme-value statistic of
      ~5e5 elements: measured ratios 0.7 - 1.5);
  (4) no element is off by more than 1e-6 * ||ref||_inf in absolute terms.
I.e. elementwise the HIP kernels are at least as exact as the arithmetic they replace."""
import pytest
import torch

from tests.test_gpu_parity import H, oracle, _model_from_params, _rand_params  # noqa: F401

pytestmark = pytest.mark.gpu

CASES = [("tiles-30", dict(num_graphs=256, nodes=30), 64),
         ("tiles-ragged-24..32", dict(num_graphs=256, nodes=28, nodes_jitter=4), 64),
         ("tiles-F25", dict(num_graphs=128, nodes=20, nodes_jitter=6), 25),
         ("wave-24..36", dict(num_graphs=256, nodes=30, nodes_jitter=6), 64),
         ("wave-50+-14-F32", dict(num_graphs=128, nodes=50, nodes_jitter=14), 32),
         ("mid-87+-30-F25", dict(num_graphs=128, nodes=87, nodes_jitter=30, extra_bonds=4), 25),
         ("mid-200-F64", dict(num_graphs=32, nodes=200, extra_bonds=13, max_degree=6), 64)]


def _excess(a, ref64):
    d = (a.double().cpu() - ref64).abs()
    bound = 1e-5 * ref64.abs().clamp_min(1e-3)
    return d, d / bound


@pytest.mark.parametrize("name,kw,feat", CASES, ids=[c[0] for c in CASES])
def test_elementwise_gate_after_every_conv(H, oracle, name, kw, feat):
    from hcatgnet_amd import synth
    cfg = dict(synth.CONFIGS["C2"]); cfg.update(kw); cfg["feat"] = feat
    sb = synth.make_batch(**cfg)
    params = _rand_params(feat, 64, seed=3)
    m = _model_from_params(H, params)
    p64 = {k: v.double() for k, v in params.items()}
    _, emb64, acts64 = oracle.gcn_forward(p64, sb.x.double(), sb.edge_index, sb.batch, sb.num_graphs, return_intermediates=True)
    _, emb32, acts32 = oracle.gcn_forward(params, sb.x, sb.edge_index, sb.batch, sb.num_graphs, return_intermediates=True)
    b = sb.as_batch("cuda")
    plan = H.BatchPlan.build(b.edge_index, b.batch, b.x.shape[0], num_graphs=sb.num_graphs, mode="blocked",
                             max_nodes=sb.max_nodes, max_edges=sb.max_edges)
    with torch.no_grad():
        h1 = m.conv1(b.x, plan, apply_act=True)                     # default kernel selection (tiles / one graph per workgroup)
        h2 = m.conv_layers[0](h1, plan, apply_act=True)
        _, emb = m(b, True)
    for what, got, ref32, ref64 in (("conv1", h1, acts32[0], acts64[0]), ("conv2", h2, acts32[1], acts64[1]),
                                    ("graph_emb", emb, emb32, emb64)):
        d, ex = _excess(got, ref64)
        d32, ex32 = _excess(ref32, ref64)
        scale = float(ref64.abs().max())
        big = ref64.abs() >= 0.1
        assert bool((ex[big] <= 1.0).all()), (name, what, "bound violated on an un-cancelled element")
        n_out, n_out32 = int((ex > 1.0).sum()), int((ex32 > 1.0).sum())
        assert n_out <= int(1.05 * n_out32) + 2, (name, what, n_out, n_out32)
        assert float(ex.max()) <= max(1.0, 2.0 * float(ex32.max())), (name, what, float(ex.max()), float(ex32.max()))
        assert float(d.max()) <= 1e-6 * scale, (name, what, float(d.max()), scale)
