#!/usr/bin/env python3
"""Dev tool: time the any-shape path (use_fused=False) and real-graph-sized batches."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hcatgnet_amd as H
from hcatgnet_amd import synth
def run(name, sb, F, fused):
    m = H.make_network("GCN", H.default_options(use_fused=fused), F).cuda()
    b = sb.as_batch("cuda")
    def step():
        m.zero_grad(set_to_none=True); b._hcg_plan = None
        out = m(b); torch.sqrt(m.loss(out, b.y.unsqueeze(1))).backward()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{name:34s} fused={fused!s:5s} {dt*1e3:8.3f} ms/step  {sb.num_graphs/dt:12.0f} graphs/s", flush=True)
run("C3 (30 nodes, 64-d, B=4096)", synth.make_config("C2"), 64, False)
run("C3 (30 nodes, 64-d, B=4096)", synth.make_config("C2"), 64, True)
real = synth.make_batch(num_graphs=4096, nodes=87, extra_bonds=4, max_degree=4, feat=25, nodes_jitter=30)
run("real-sized (57-117 nodes, F=25)", real, 25, True)
run("C5 slice (200 nodes,128-d,B=256)", synth.make_config("C5", num_graphs=256), 128, True)
