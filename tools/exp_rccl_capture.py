#!/usr/bin/env python3
"""Dev experiment: does an RCCL all-reduce survive hipGraph capture on this stack (world size 1 on one GPU)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones(16643, device="cuda")
dist.all_reduce(x); torch.cuda.synchronize()          # communicator up before any capture
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): dist.all_reduce(x)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        x.mul_(2.0)
        dist.all_reduce(x)
        x.add_(1.0)
    x.fill_(1.0); torch.cuda.synchronize()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    print("captured all_reduce: replayed, x[0] =", float(x[0]), "(expect 63)")
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize(); print(f"replay {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per graph (3 nodes)")
except Exception as exc:
    print("capture FAILED:", type(exc).__name__, exc)
dist.destroy_process_group()
