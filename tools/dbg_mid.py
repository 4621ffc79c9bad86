"""Dev probe: per-parameter gradient errors of the mid-size path vs the any-shape path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcatgnet_amd as H
from hcatgnet_amd import synth
def rel(a, b): return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for (B, nodes, jit, feat) in [(1, 60, 0, 64), (1, 33, 0, 64), (1, 96, 0, 64), (7, 60, 27, 64), (131, 60, 27, 64), (131, 60, 27, 32)]:
    sb = synth.make_batch(num_graphs=B, nodes=nodes, extra_bonds=4, max_degree=4, feat=feat, nodes_jitter=jit, seed=9)
    torch.manual_seed(0)
    m = H.make_network("GCN", H.default_options(), feat).cuda()
    batch = sb.as_batch("cuda")
    res = {}
    for fused in (True, False):
        m.use_fused = fused
        m.zero_grad()
        out, emb = m(batch, True)
        torch.sqrt(m.loss(out, batch.y.unsqueeze(1))).backward()
        res[fused] = {k: v.grad.clone() for k, v in m.named_parameters()}
    print(B, nodes, jit, feat, {k: f"{rel(res[True][k], res[False][k]):.1e}" for k in res[True]})
    if B == 1:
        d = (res[True]["conv1.lin.weight"] - res[False]["conv1.lin.weight"]).abs()
        bad = (d > 1e-4 * res[False]["conv1.lin.weight"].abs().max()).nonzero()
        print("   bad entries:", bad.shape[0], "rows", bad[:, 0].unique().tolist()[:20], "cols", bad[:, 1].unique().tolist()[:20])
