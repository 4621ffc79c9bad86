#!/usr/bin/env python3
"""Dev tool: K independent runs of the reference's training regime at once (hcatgnet_amd.train.train_networks /
eval_networks) -- the nested cross-validation of scripts_experiments/train_GNN.py:48-50 trains 90 models one after another, each
on ~535 graphs in batches of 40.  Prints, per K, the time for one epoch of ALL K runs (train + validation + test evaluation,
as the reference's epoch loop does) and the aggregate graphs/s.  usage: python tools/bench_runs.py [K ...]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcatgnet_amd as H
from hcatgnet_amd import synth
from hcatgnet_amd.train import eval_networks, train_networks

G_TRAIN, G_VAL, BS, EP = 535, 60, 40, 20
Ks = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 6, 9, 12]


def make_run(k):
    sb = synth.make_config("REAL", num_graphs=G_TRAIN + 2 * G_VAL, seed=synth.BASE_SEED + k)
    graphs = sb.as_graph_list()
    tr = H.DeviceGraphStore(graphs[:G_TRAIN], device="cuda")
    va = H.DeviceGraphStore(graphs[G_TRAIN:G_TRAIN + G_VAL], device="cuda")
    te = H.DeviceGraphStore(graphs[G_TRAIN + G_VAL:], device="cuda")
    m = H.make_network("GCN", H.default_options(), 25).cuda()
    return m, H.DeviceLoader(tr, batch_size=BS, shuffle=True, seed=k), H.DeviceLoader(va, batch_size=BS), H.DeviceLoader(te, batch_size=BS)


runs = [make_run(k) for k in range(max(Ks))]
for K in Ks:
    ms, trn, val, tst = ([r[i] for r in runs[:K]] for i in range(4))

    def epoch():
        train_networks(ms, trn, "cuda")
        eval_networks(ms, val, "cuda")
        eval_networks(ms, tst, "cuda")
    for _ in range(3):
        epoch()
    gc.collect(); gc.freeze()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(EP):
        epoch()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / EP
    t0 = time.perf_counter()
    for _ in range(EP):
        train_networks(ms, trn, "cuda")
    torch.cuda.synchronize(); dtt = (time.perf_counter() - t0) / EP
    print(f"K = {K:2d} runs at once: {dt * 1e3:7.2f} ms per epoch of all runs (train + 2 evaluations) = {dt * 1e3 / K:6.3f} ms per run; "
          f"training alone {dtt * 1e3:7.2f} ms = {K * G_TRAIN / dtt:10.0f} graphs/s", flush=True)
