#!/usr/bin/env python3
"""Dev tool: one training epoch as the reference runs it (scripts_experiments/train_GNN.py:73-80: 535 training graphs,
batch_size 40, shuffle; utils/utils_model.py:55-70) -- hcatgnet_amd.train.train_network over a DeviceLoader vs the CPU
oracle's loop on the same graphs.  Prints ms per epoch and graphs/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcatgnet_amd as H
from hcatgnet_amd import synth
from hcatgnet_amd.train import train_network, eval_network
from oracle import gcn_oracle
G, BS = 535, 40
sb = synth.make_config("REAL", num_graphs=G)
graphs = sb.as_graph_list()
store = H.DeviceGraphStore(graphs, device="cuda")
loader = H.DeviceLoader(store, batch_size=BS, shuffle=True, seed=0)
model = H.make_network("GCN", H.default_options(), 25).cuda()
import gc
for _ in range(3):
    train_network(model, loader, "cuda")
# set-up is over: everything alive now leaves the cyclic collector's sight -- a full collection of a process that holds torch,
# 15 trainers and a captured epoch takes ~85 ms and lands in whichever loop allocates most (the per-batch loop: it measured
# 7.9 ms/epoch instead of 2.3 with the collector walking those objects)
gc.collect(); gc.freeze()
torch.cuda.synchronize(); t0 = time.perf_counter()
EP = 20
for _ in range(EP):
    loss = train_network(model, loader, "cuda")
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / EP
print(f"MI355X  train_network: {dt * 1e3:8.2f} ms/epoch ({len(loader)} batches of {BS}; {G / dt:10.0f} graphs/s), last epoch loss {loss:.4f}"
      f"  [whole epochs as one hipGraph: {getattr(loader, '_hcg_epoch_window', (None, None))[1] is not None}]")
from hcatgnet_amd import train as _train
_train.EPOCH_WINDOW = False
loader_b = H.DeviceLoader(store, batch_size=BS, shuffle=True, seed=0)
model_b = H.make_network("GCN", H.default_options(), 25).cuda()      # (its own model: the window's 14 trainers stay out of the way)
for _ in range(3):
    train_network(model_b, loader_b, "cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(EP):
    train_network(model_b, loader_b, "cuda")
torch.cuda.synchronize(); dtl = (time.perf_counter() - t0) / EP
_train.EPOCH_WINDOW = True
print(f"MI355X  train_network: {dtl * 1e3:8.2f} ms/epoch with the per-batch loop (round 2's form)")
t0 = time.perf_counter()
for _ in range(EP):
    eval_network(model, loader, "cuda")
torch.cuda.synchronize(); dte = (time.perf_counter() - t0) / EP
print(f"MI355X  eval_network : {dte * 1e3:8.2f} ms/epoch (shuffling loader: per-batch loop)")
vloader = H.DeviceLoader(store, batch_size=BS)          # the reference's validation / test loaders do not shuffle
for _ in range(3):
    eval_network(model, vloader, "cuda")
t0 = time.perf_counter()
for _ in range(EP):
    eval_network(model, vloader, "cuda")
torch.cuda.synchronize(); dtv = (time.perf_counter() - t0) / EP
print(f"MI355X  eval_network : {dtv * 1e3:8.2f} ms/epoch (fixed loader: batches collated once, one graph launch per call)")
# CPU: the oracle's loop with the host collate (what the reference does through PyG on the CPU)
torch.set_num_threads(min(16, os.cpu_count() or 1))
params = {k: v.detach().cpu().clone() for k, v in H.make_network("GCN", H.default_options(), 25).state_dict().items()}
p, opt = gcn_oracle.make_train_state(params)
cl = H.DataLoader(graphs, batch_size=BS, shuffle=True)
def cpu_epoch():
    for b in cl:
        gcn_oracle.train_step(p, opt, b.x, b.edge_index, b.batch, b.y, b.num_graphs)
cpu_epoch(); t0 = time.perf_counter()
for _ in range(3): cpu_epoch()
dtc = (time.perf_counter() - t0) / 3
print(f"CPU oracle loop ({torch.get_num_threads()} threads): {dtc * 1e3:8.2f} ms/epoch ({G / dtc:10.0f} graphs/s) -> x{dtc / dt:.1f}")
