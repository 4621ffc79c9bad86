#!/usr/bin/env python3
"""Dev tool: where does the HOST time of one epoch at the reference's operating point go (535 graphs of 57-117 atoms,
batch_size 40, shuffled, DeviceLoader + train.train_network)?  cProfile over 20 epochs."""
import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hcatgnet_amd as H
from hcatgnet_amd import synth
from hcatgnet_amd.train import train_network
from hcatgnet_amd import train as _t
if os.environ.get("HCG_LOOP") == "1":
    _t.EPOCH_WINDOW = False
sb = synth.make_config("REAL", num_graphs=535)
store = H.DeviceGraphStore(sb.as_graph_list(), device="cuda")
loader = H.DeviceLoader(store, batch_size=40, shuffle=True, seed=0)
model = H.make_network("GCN", H.default_options(), 25).cuda()
for _ in range(5): train_network(model, loader, "cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): train_network(model, loader, "cuda")
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host issue {1e3*(t1-t0)/20:.3f} ms/epoch, with drain {1e3*(t2-t0)/20:.3f} ms/epoch ({len(loader)} steps)")
pr = cProfile.Profile(); pr.enable()
for _ in range(20): train_network(model, loader, "cuda")
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
