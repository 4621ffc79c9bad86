import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcatgnet_amd as H
from hcatgnet_amd import synth, functional as HF
from oracle import gcn_oracle as oracle
from tests.test_gpu_parity import _rand_params, _model_from_params
def rel(a, b): return float((a.detach().cpu() - b.detach().cpu()).abs().max() / b.detach().cpu().abs().max().clamp_min(1e-30))
sb = synth.make_batch(num_graphs=131, nodes=60, extra_bonds=4, max_degree=4, feat=64, nodes_jitter=27, seed=9)
params = _rand_params(64, 64, seed=31)
batch = sb.as_batch("cuda")
o_loss, o_out, o_emb, o_grads = oracle.train_step_grads(params, sb.x, sb.edge_index, sb.batch, sb.y, sb.num_graphs)
for xgrad in (False, True):
    m = _model_from_params(H, params)
    x = batch.x.clone().requires_grad_(xgrad)
    b2 = H.Batch(x, batch.edge_index, batch.batch, sb.num_graphs, y=batch.y, max_nodes=sb.max_nodes, max_edges=sb.max_edges, edges_grouped=True)
    out, emb = m(b2, True)
    torch.sqrt(m.loss(out, batch.y.unsqueeze(1))).backward()
    print("x.requires_grad", xgrad, {k: f"{rel(v.grad, o_grads[k]):.1e}" for k, v in m.named_parameters() if "conv" in k})
# layer by layer with the oracle's intermediates
_, _, acts = oracle.gcn_forward(params, sb.x, sb.edge_index, sb.batch, sb.num_graphs, return_intermediates=True)
plan = H.BatchPlan.build(batch.edge_index, batch.batch, batch.x.shape[0], num_graphs=sb.num_graphs, mode="blocked", max_nodes=sb.max_nodes, max_edges=sb.max_edges)
W1 = params["conv1.lin.weight"].cuda().requires_grad_(True); b1 = params["conv1.bias"].cuda().requires_grad_(True)
W2 = params["conv_layers.0.lin.weight"].cuda().requires_grad_(True); b2 = params["conv_layers.0.bias"].cuda().requires_grad_(True)
a1 = HF.mid_gcn_layer(batch.x, W1, b1, plan, True, pool=False)
print("a1 vs oracle", rel(a1, acts[0]))
emb = HF.mid_gcn_layer(a1, W2, b2, plan, True, pool=True)
print("emb vs oracle", rel(emb, o_emb))
go = torch.randn(emb.shape, generator=torch.Generator().manual_seed(3)).cuda()
emb.backward(go)
gm = [W1.grad.clone(), b1.grad.clone(), W2.grad.clone(), b2.grad.clone()]
for t in (W1, b1, W2, b2): t.grad = None
a1g = HF.gcn_layer(batch.x, W1, b1, plan, False, True)
embg = HF.graph_pool(HF.gcn_layer(a1g, W2, b2, plan, False, True), plan)
embg.backward(go)
gg = [W1.grad, b1.grad, W2.grad, b2.grad]
print("two-layer chain, mid vs general:", [f"{rel(a, b):.1e}" for a, b in zip(gm, gg)])
