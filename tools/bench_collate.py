#!/usr/bin/env python3
"""Dev tool: on-device collate throughput (f1) vs the host PyG-rule collate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, hcatgnet_amd as H
from hcatgnet_amd import synth
sb = synth.make_config("C2", num_graphs=8192)
n = 30
graphs = []
for g in range(sb.num_graphs):
    e0, e1 = g * 64, (g + 1) * 64
    graphs.append(H.Data(x=sb.x[g * n:(g + 1) * n], edge_index=sb.edge_index[:, e0:e1] - g * n, y=sb.y[g:g + 1], idx=g))
t0 = time.perf_counter(); hb = H.collate(graphs[:4096]); t_host = time.perf_counter() - t0
store = H.DeviceGraphStore(graphs, "cuda")
rng = np.random.default_rng(0)
ids = rng.permutation(8192)[:4096]
for _ in range(5): store.collate(ids)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): b = store.collate(ids)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"host collate of 4096 graphs: {t_host*1e3:.1f} ms ({4096/t_host:.3g} graphs/s)")
print(f"device collate of 4096 graphs: {dt*1e3:.3f} ms ({4096/dt:.3g} graphs/s), plan attached (no plan launch)")
m = H.make_network("GCN", H.default_options(), 64).cuda()
def step():
    b = store.collate(ids); m.optimizer.zero_grad(set_to_none=True)
    loss = torch.sqrt(m.loss(m(b), b.y.unsqueeze(1))); loss.backward(); m.optimizer.step()
for _ in range(10): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
print(f"eager train step incl. device collate + FusedAdam: {dt*1e3:.3f} ms ({4096/dt:.3g} graphs/s)")
