#!/usr/bin/env python3
"""Dev tool: what the data-parallel step costs on ONE GPU (RCCL world size 1, collective forced): the extra launches of
the exchange path (plain slab reduction, all-reduce kernel, separate Adam) without any cross-GPU latency."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import hcatgnet_amd as H
from hcatgnet_amd import synth
from hcatgnet_amd.ddp import DataParallelGCN
from hcatgnet_amd.train import FusedTrainStep
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
sb = synth.make_config("C2")
x, ei, bv, y = sb.x.cuda(), sb.edge_index.cuda(), sb.batch.cuda(), sb.y.cuda()
fresh = lambda: H.Batch(x, ei, bv, sb.num_graphs, y=y, max_nodes=sb.max_nodes, max_edges=sb.max_edges, edges_grouped=True)
def timeit(fn, k=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
m0 = H.make_network("GCN", H.default_options(), 64).cuda()
s0 = FusedTrainStep(m0)
print(f"single-GPU step (reduction + Adam fused): eager {timeit(lambda: s0(fresh())):.4f} ms")
m1 = H.make_network("GCN", H.default_options(), 64).cuda()
s1 = FusedTrainStep(m1, grad_sync=lambda flat: None)
s1.capture(fresh)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dp = DataParallelGCN(m1, force_collective=True)
s1.grad_sync = dp.reduce_flat
print(f"DP step, world 1 forced collective: eager {timeit(lambda: s1(fresh())):.4f} ms, captured + eager exchange/update {timeit(s1.replay):.4f} ms")
dist.destroy_process_group()
