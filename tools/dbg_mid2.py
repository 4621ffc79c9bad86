import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcatgnet_amd as H
from hcatgnet_amd import synth
from oracle import gcn_oracle as oracle
from tests.test_gpu_parity import _rand_params, _model_from_params
def rel(a, b): return float((a.cpu() - b.cpu()).abs().max() / b.cpu().abs().max().clamp_min(1e-30))
sb = synth.make_batch(num_graphs=131, nodes=60, extra_bonds=4, max_degree=4, feat=64, nodes_jitter=27, seed=9)
params = _rand_params(64, 64, seed=31)
m = _model_from_params(H, params)
batch = sb.as_batch("cuda")
o_loss, o_out, o_emb, o_grads = oracle.train_step_grads(params, sb.x, sb.edge_index, sb.batch, sb.y, sb.num_graphs)
for fused in (True, False, True):
    m.use_fused = fused
    m.zero_grad()
    out, emb = m(batch, True)
    torch.sqrt(m.loss(out, batch.y.unsqueeze(1))).backward()
    print("fused" if fused else "general", {k: f"{rel(v.grad, o_grads[k]):.1e}" for k, v in m.named_parameters()}, "emb", f"{rel(emb, o_emb):.1e}")
