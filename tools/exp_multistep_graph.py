#!/usr/bin/env python3
"""Dev experiment: does ONE hipGraph that holds several consecutive training steps (distinct batches, weights carried from
step to step) beat one graph per step?  What it would save is the bubble between two graph launches.
usage: python tools/exp_multistep_graph.py [config=C3] [steps_per_graph=16]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcatgnet_amd as H
from hcatgnet_amd import synth
from hcatgnet_amd.train import FusedTrainStep
cfg_name = sys.argv[1] if len(sys.argv) > 1 else "C3"
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
cfg = synth.CONFIGS[cfg_name]
model = H.make_network("GCN", H.default_options(embedding_dim=cfg["hidden"]), cfg["feat"]).to(dev)
class Res:
    def __init__(self, i):
        sb = synth.make_config(cfg_name, rank=i); self.sb = sb
        self.x, self.ei, self.bv, self.y = sb.x.to(dev), sb.edge_index.to(dev), sb.batch.to(dev), sb.y.to(dev)
        self.plan = H.BatchPlan.build(self.ei, self.bv, self.x.shape[0], num_graphs=sb.num_graphs, mode="blocked", validate=False,
                                      max_nodes=sb.max_nodes, max_edges=sb.max_edges)
    def planned(self):
        sb = self.sb
        b = H.Batch(self.x, self.ei, self.bv, sb.num_graphs, y=self.y, max_nodes=sb.max_nodes, max_edges=sb.max_edges,
                    edges_grouped=True, n_small=sb.n_small)
        b._hcg_plan = self.plan
        return b
res = [Res(i) for i in range(NB)]
trainers = [FusedTrainStep(model, optimizer_step=True) for _ in res]
for i, tr in enumerate(trainers):
    tr.capture(res[i].planned, next_plan=res[(i + 1) % NB].plan)
def timed(k, fn):
    torch.cuda.synchronize(); gc.disable(); t0 = time.perf_counter()
    for j in range(k): fn(j)
    torch.cuda.synchronize(); gc.enable()
    return (time.perf_counter() - t0) / k
single = lambda j: trainers[j % NB].replay()
for _ in range(3): timed(NB * 8, single)
# ONE graph over the NB steps: the same eager calls the per-step capture records, back to back
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for i, tr in enumerate(trainers): tr(res[i].planned())
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i, tr in enumerate(trainers): tr(res[i].planned())
multi = lambda j: g.replay()
for _ in range(3): timed(8, multi)
gc.collect(); gc.freeze()
for rnd in range(3):
    a = timed(NB * 64, single) * 1e3
    b = timed(64, multi) / NB * 1e3
    print(f"{cfg_name}: one graph per step {a:.4f} ms/step | one graph per {NB} steps {b:.4f} ms/step  ({(b / a - 1) * 100:+.1f} %)", flush=True)
