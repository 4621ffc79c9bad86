#!/usr/bin/env python3
"""Dev tool: shader clock / power the box reports while the captured C3 step replays back to back (rocm-smi sampled from
a side thread).  Answers whether the s_memtime tick rates seen by the probes (1.1-2.4 GHz) are DVFS under load."""
import os, sys, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcatgnet_amd as H
from hcatgnet_amd import synth
from hcatgnet_amd.train import FusedTrainStep

sb = synth.make_config("C3")
x, ei, bv, y = sb.x.cuda(), sb.edge_index.cuda(), sb.batch.cuda(), sb.y.cuda()
fresh = lambda: H.Batch(x, ei, bv, sb.num_graphs, y=y, max_nodes=sb.max_nodes, max_edges=sb.max_edges, edges_grouped=True)
m = H.make_network("GCN", H.default_options(), 64).cuda()
step = FusedTrainStep(m)
step.capture(fresh)
samples, stop = [], False

def sample():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "-c", "-P", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.perf_counter(), out.strip().replace("\n", " | ")))
        except Exception as e:
            samples.append((time.perf_counter(), f"rocm-smi failed: {e}"))
        time.sleep(0.4)

print("idle:", subprocess.run(["rocm-smi", "-c", "-P", "--csv"], capture_output=True, text=True).stdout.strip().replace("\n", " | "))
th = threading.Thread(target=sample); th.start()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 6.0:
    for _ in range(500):
        step.replay()
    torch.cuda.synchronize(); n += 500
dt = time.perf_counter() - t0
stop = True; th.join()
print(f"{n} replays in {dt:.2f} s -> {dt / n * 1e3:.4f} ms/step")
for t, s in samples:
    print(f"+{t - t0:5.2f}s {s[:300]}")
