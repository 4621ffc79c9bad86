import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcatgnet_amd as H
from hcatgnet_amd import synth, functional as HF
def rel(a, b): return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
sb = synth.make_batch(num_graphs=131, nodes=60, extra_bonds=4, max_degree=4, feat=64, nodes_jitter=27, seed=9)
batch = sb.as_batch("cuda")
plan = H.BatchPlan.build(batch.edge_index, batch.batch, batch.x.shape[0], num_graphs=sb.num_graphs, mode="blocked", max_nodes=sb.max_nodes, max_edges=sb.max_edges)
g = torch.Generator().manual_seed(1)
for bias_scale in (0.0, 0.1):
  for pool in (True, False):
    for act in (True, False):
        W = (torch.randn(64, 64, generator=g) * 0.2).cuda(); b = (torch.randn(64, generator=g) * bias_scale).cuda()
        x = batch.x
        go = torch.randn(sb.num_graphs, 128, generator=g).cuda() if pool else torch.randn(x.shape[0], 64, generator=g).cuda()
        res = []
        for mid in (True, False):
            xx = x.clone().requires_grad_(True); WW = W.clone().requires_grad_(True); bb = b.clone().requires_grad_(True)
            if mid:
                o = HF.mid_gcn_layer(xx, WW, bb, plan, act, pool=pool)
            else:
                o = HF.gcn_layer(xx, WW, bb, plan, False, act)
                if pool: o = HF.graph_pool(o, plan)
            o.backward(go)
            res.append((o.detach(), xx.grad, WW.grad, bb.grad))
        print(f"bias {bias_scale} pool {pool} act {act}: out {rel(res[0][0], res[1][0]):.1e} dx {rel(res[0][1], res[1][1]):.1e} dW {rel(res[0][2], res[1][2]):.1e} db {rel(res[0][3], res[1][3]):.1e}")
