#!/usr/bin/env python3
"""Dev tool: where does the HOST time of one eager step go (cProfile over 300 steps)?"""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hcatgnet_amd as H
from hcatgnet_amd import synth
sb = synth.make_config("C2"); m = H.make_network("GCN", H.default_options(), 64).cuda()
x, ei, bv, y = sb.x.cuda(), sb.edge_index.cuda(), sb.batch.cuda(), sb.y.cuda(); y2 = y.unsqueeze(1)
from hcatgnet_amd.train import FusedTrainStep
trainer = FusedTrainStep(m)
def step():      # the product's training step (no autograd)
    trainer(H.Batch(x, ei, bv, sb.num_graphs, y=y, max_nodes=sb.max_nodes, max_edges=sb.max_edges, edges_grouped=True))
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host issue time {1e3*(t1-t0)/300:.3f} ms/step, +drain {1e3*(t2-t0)/300:.3f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
