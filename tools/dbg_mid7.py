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
P = [params[k].cuda() for k in ("conv1.lin.weight", "conv1.bias", "conv_layers.0.lin.weight", "conv_layers.0.bias")]
go = torch.randn(sb.num_graphs, 128, generator=torch.Generator().manual_seed(3)).cuda()
def L(kind, x, W, b, pool):
    if kind == "mid":
        return HF.mid_gcn_layer(x, W, b, plan, True, pool=pool)
    o = HF.gcn_layer(x, W, b, plan, False, True)
    return HF.graph_pool(o, plan) if pool else o
def run(k1, k2, detach=False):
    W1, b1, W2, b2 = [t.clone().requires_grad_(True) for t in P]
    a1 = L(k1, batch.x, W1, b1, False)
    if detach:
        a1d = a1.detach().clone().requires_grad_(True)
        emb = L(k2, a1d, W2, b2, True)
        emb.backward(go)
        torch.cuda.synchronize()
        a1.backward(a1d.grad)
    else:
        emb = L(k2, a1, W2, b2, True)
        emb.backward(go)
    return [W1.grad, b1.grad, W2.grad, b2.grad]
ref = run("gen", "gen")
for cfg in [("mid", "mid", False), ("mid", "mid", True), ("gen", "mid", False), ("mid", "gen", False)]:
    g = run(*cfg)
    print(cfg, [f"{rel(a, b):.1e}" for a, b in zip(g, ref)])
print("---- is a1 stable?")
W1, b1, W2, b2 = [t.clone().requires_grad_(True) for t in P]
a1 = L("mid", batch.x, W1, b1, False)
torch.cuda.synchronize()
snap = a1.detach().clone()
a1g = L("gen", batch.x, W1, b1, False).detach()
print("a1 mid vs gen: max abs", float((snap - a1g).abs().max()), "rel_inf", rel(snap, a1g), "nan", int(snap.isnan().sum()))
d = (snap - a1g).abs()
print("elements with |d| > 1e-5:", int((d > 1e-5).sum()), "rows", (d > 1e-5).any(1).nonzero().flatten()[:20].tolist())
emb = L("gen", a1, W2, b2, True)
emb.backward(go)
torch.cuda.synchronize()
print("a1 changed during L2 fwd/bwd + L1 bwd:", float((a1.detach() - snap).abs().max()))
