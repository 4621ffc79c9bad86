#!/usr/bin/env python3
"""Dev tool: elementwise error of node embeddings after each conv (tile and mid kernels) against the fp64 oracle, beside the
fp32 oracle's own error against fp64 -- the data behind the elementwise parity gate of tests/test_gpu_elementwise.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcatgnet_amd as H
from hcatgnet_amd import synth
from oracle import gcn_oracle
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tests.test_gpu_parity import _rand_params, _model_from_params


def stats(a, ref, tag):
    a, ref = a.double().cpu(), ref.double().cpu()
    d = (a - ref).abs()
    bound = 1e-5 * ref.abs().clamp_min(1e-3)
    viol = d > bound
    scale = ref.abs().max()
    print(f"  {tag:28s} max|d| {d.max():.3e}  |ref|inf {scale:.3e}  rel_inf {d.max() / scale:.2e}  viol {int(viol.sum())}/{d.numel()} "
          f"({viol.double().mean() * 100:.4f} %)  max|ref| among viol {ref.abs()[viol].max().item() if viol.any() else 0:.3e}  "
          f"max d/bound {(d / bound).max():.2f}")


for name, kw, feat in [("C2 tiles 30", dict(num_graphs=256, nodes=30), 64), ("ragged 24..32", dict(num_graphs=256, nodes=28, nodes_jitter=4), 64),
                       ("mid 87+-30 F25", dict(num_graphs=128, nodes=87, nodes_jitter=30, extra_bonds=4), 25),
                       ("mid 200 F64", dict(num_graphs=32, nodes=200, extra_bonds=13, max_degree=6), 64)]:
    cfg = dict(synth.CONFIGS["C2"]); cfg.update(kw); cfg["feat"] = feat
    sb = synth.make_batch(**cfg)
    params = _rand_params(feat, 64, seed=3)
    m = _model_from_params(H, params)
    p64 = {k: v.double() for k, v in params.items()}
    _, emb64, acts64 = gcn_oracle.gcn_forward(p64, sb.x.double(), sb.edge_index, sb.batch, sb.num_graphs, return_intermediates=True)
    _, emb32, acts32 = gcn_oracle.gcn_forward(params, sb.x, sb.edge_index, sb.batch, sb.num_graphs, return_intermediates=True)
    b = sb.as_batch("cuda")
    plan = H.BatchPlan.build(b.edge_index, b.batch, b.x.shape[0], num_graphs=sb.num_graphs, mode="blocked", max_nodes=sb.max_nodes,
                             max_edges=sb.max_edges)
    with torch.no_grad():
        h1 = m.conv1(b.x, plan, apply_act=True)
        h2 = m.conv_layers[0](h1, plan, apply_act=True)
        _, emb = m(b, True)
    print(name, "max_nodes", sb.max_nodes)
    stats(h1, acts64[0], "HIP conv1 vs fp64")
    stats(acts32[0], acts64[0], "oracle32 conv1 vs fp64")
    stats(h2, acts64[1], "HIP conv2 vs fp64")
    stats(acts32[1], acts64[1], "oracle32 conv2 vs fp64")
    stats(emb, emb64, "HIP emb vs fp64")
    stats(emb32, emb64, "oracle32 emb vs fp64")
    stats(h1, acts32[0], "HIP conv1 vs oracle32")
    stats(h2, acts32[1], "HIP conv2 vs oracle32")
