#!/usr/bin/env python3
"""Dev tool: where the time of K concurrent runs goes (host issue vs GPU): issue time of K epoch launches, total time, and
the same K epochs issued on ONE stream."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcatgnet_amd as H
from hcatgnet_amd import synth, train
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
runs = []
for k in range(K):
    sb = synth.make_config("REAL", num_graphs=535, seed=synth.BASE_SEED + k)
    st = H.DeviceGraphStore(sb.as_graph_list(), device="cuda")
    m = H.make_network("GCN", H.default_options(), 25).cuda()
    ld = H.DeviceLoader(st, batch_size=40, shuffle=True, seed=k)
    runs.append((m, ld))
ms, lds = [r[0] for r in runs], [r[1] for r in runs]
for _ in range(3):
    train.train_networks(ms, lds, "cuda")
wins = [ld._hcg_epoch_window[1] for ld in lds]
streams = [m._hcg_run_stream for m in ms]
gc.collect(); gc.freeze()
EP = 20
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(EP): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / EP * 1e3
issue = [0.0]
def multi():
    t0 = time.perf_counter()
    fl = []
    for w, s in zip(wins, streams):
        with torch.cuda.stream(s):
            fl.append(w.launch_epoch())
    issue[0] += time.perf_counter() - t0
    torch.cuda.synchronize()
def multi_finish():
    for w, s in zip(wins, streams):
        with torch.cuda.stream(s):
            w.launch_epoch()
    for w, s in zip(wins, streams):
        with torch.cuda.stream(s):
            w.window.value()
def single():
    for w in wins:
        w.launch_epoch()
    torch.cuda.synchronize()
def layout_only():
    for w in wins:
        w._layout(w.draw_order())
    torch.cuda.synchronize()
def replay_only():
    for w, s in zip(wins, streams):
        with torch.cuda.stream(s):
            w.window.graph.replay()
    torch.cuda.synchronize()
def replay_single():
    for w in wins:
        w.window.graph.replay()
    torch.cuda.synchronize()
print(f"K = {K}  HW queues env: {os.environ.get('GPU_MAX_HW_QUEUES')}")
t = timed(multi); print(f"  K streams : {t:7.3f} ms per round of K epochs, host issue part {issue[0] / EP * 1e3:7.3f} ms")
print(f"  K streams, per-run value read: {timed(multi_finish):7.3f} ms;  train_networks: {timed(lambda: train.train_networks(ms, lds, 'cuda')):7.3f} ms")
print(f"  one stream: {timed(single):7.3f} ms")
print(f"  layout + upload only: {timed(layout_only):7.3f} ms")
print(f"  graph replays only, K streams: {timed(replay_only):7.3f} ms;  one stream: {timed(replay_single):7.3f} ms")
