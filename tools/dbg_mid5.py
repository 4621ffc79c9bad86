import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcatgnet_amd as H
from hcatgnet_amd import synth, functional as HF
from tests.test_gpu_parity import _rand_params
def rel(a, b): return float((a.detach().cpu() - b.detach().cpu()).abs().max() / b.detach().cpu().abs().max().clamp_min(1e-30))
sb = synth.make_batch(num_graphs=131, nodes=60, extra_bonds=4, max_degree=4, feat=64, nodes_jitter=27, seed=9)
params = _rand_params(64, 64, seed=31)
batch = sb.as_batch("cuda")
plan = H.BatchPlan.build(batch.edge_index, batch.batch, batch.x.shape[0], num_graphs=sb.num_graphs, mode="blocked", max_nodes=sb.max_nodes, max_edges=sb.max_edges)
W1 = params["conv1.lin.weight"].cuda(); b1 = params["conv1.bias"].cuda()
W2 = params["conv_layers.0.lin.weight"].cuda(); b2 = params["conv_layers.0.bias"].cuda()
with torch.no_grad():
    a1 = HF.mid_gcn_layer(batch.x, W1, b1, plan, True, pool=False)
go = torch.randn(sb.num_graphs, 128, generator=torch.Generator().manual_seed(3)).cuda()
gp = plan.graph_ptr.cpu()
for name, xin in (("a1", a1), ("randn", torch.randn_like(a1)), ("a1.abs", a1.abs()), ("relu-like", torch.where(torch.randn_like(a1) > 0, torch.randn_like(a1).abs(), -0.01 * torch.randn_like(a1).abs()))):
    res = []
    for mid in (True, False):
        xx = xin.clone().requires_grad_(True); WW = W2.clone().requires_grad_(True); bb = b2.clone().requires_grad_(True)
        o = HF.mid_gcn_layer(xx, WW, bb, plan, True, pool=True) if mid else HF.graph_pool(HF.gcn_layer(xx, WW, bb, plan, False, True), plan)
        o.backward(go)
        res.append((o.detach(), xx.grad, WW.grad, bb.grad))
    print(name, f"out {rel(res[0][0], res[1][0]):.1e} dx {rel(res[0][1], res[1][1]):.1e} dW {rel(res[0][2], res[1][2]):.1e} db {rel(res[0][3], res[1][3]):.1e}")
    d = (res[0][1] - res[1][1]).abs().amax(1).cpu()
    bad = (d > 1e-4 * res[1][1].abs().max().cpu()).nonzero().flatten()
    if bad.numel():
        g = torch.searchsorted(gp, bad, right=True) - 1
        print("   bad dx rows:", bad.numel(), "graphs", g.unique().tolist()[:10], "local rows", (bad - gp[g])[:12].tolist(), "sizes", (gp[g.unique()+1]-gp[g.unique()])[:10].tolist())
