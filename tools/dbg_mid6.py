import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcatgnet_amd as H
from hcatgnet_amd import synth, functional as HF
def rel(a, b): return float((a.detach().cpu() - b.detach().cpu()).abs().max() / b.detach().cpu().abs().max().clamp_min(1e-30))
g = torch.Generator().manual_seed(1)
for (B, nodes, jit, feat) in [(1, 40, 0, 64), (1, 87, 0, 64), (3, 60, 27, 64), (131, 60, 27, 64), (131, 60, 27, 32)]:
    sb = synth.make_batch(num_graphs=B, nodes=nodes, extra_bonds=4, max_degree=4, feat=feat, nodes_jitter=jit, seed=9)
    batch = sb.as_batch("cuda")
    plan = H.BatchPlan.build(batch.edge_index, batch.batch, batch.x.shape[0], num_graphs=sb.num_graphs, mode="blocked", max_nodes=sb.max_nodes, max_edges=sb.max_edges)
    W = (torch.randn(64, feat, generator=g) * 0.2).cuda(); b = (torch.randn(64, generator=g) * 0.1).cuda()
    go = torch.randn(batch.x.shape[0], 64, generator=g).cuda()
    for xg in (False, True):
        res = []
        for mid in (True, False):
            xx = batch.x.clone().requires_grad_(xg); WW = W.clone().requires_grad_(True); bb = b.clone().requires_grad_(True)
            o = HF.mid_gcn_layer(xx, WW, bb, plan, True, pool=False) if mid else HF.gcn_layer(xx, WW, bb, plan, False, True)
            o.backward(go)
            res.append((o.detach(), WW.grad, bb.grad))
        print(B, nodes, jit, feat, "x.grad", xg, f"out {rel(res[0][0], res[1][0]):.1e} dW {rel(res[0][1], res[1][1]):.1e} db {rel(res[0][2], res[1][2]):.1e}")
        if B == 1 and rel(res[0][1], res[1][1]) > 1e-4:
            d = (res[0][1] - res[1][1]).abs()
            bad = (d > 1e-4 * res[1][1].abs().max()).nonzero()
            print("   bad dW entries", bad.shape[0], "rows(d)", bad[:, 0].unique().tolist()[:40], "cols(f)", bad[:, 1].unique().tolist()[:40])
