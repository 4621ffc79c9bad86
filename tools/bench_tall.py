#!/usr/bin/env python3
"""Dev tool: the wide-layer kernels (csrc/tall.hip) against the one-graph-per-workgroup kernels (csrc/mid.hip) on a
BASELINE config, launch by launch (HIP events, 50 repetitions).  usage: python tools/bench_tall.py [C5] [num_graphs]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcatgnet_amd as H
from hcatgnet_amd import synth, _lib
if os.environ.get("HCG_LIB"):          # a library variant (A/B runs on one box)
    _lib.LIB_PATH = os.path.abspath(os.environ["HCG_LIB"])
from hcatgnet_amd.plan import BatchPlan
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
ng = int(sys.argv[2]) if len(sys.argv) > 2 else None
sb = synth.make_config(name, num_graphs=ng)
cfg = synth.CONFIGS[name]
D, F = cfg["hidden"], cfg["feat"]
b = sb.as_batch("cuda")
plan = BatchPlan.build(b.edge_index, b.batch, b.x.shape[0], num_graphs=b.num_graphs, mode="blocked", max_nodes=sb.max_nodes,
                       max_edges=sb.max_edges, validate=False)
lib, p, st = _lib.load(), _lib.ptr, _lib.stream_ptr()
N, B, mxn, mxe, slope = plan.N, plan.B, sb.max_nodes, sb.max_edges, 0.01
g = torch.Generator().manual_seed(1)
rnd = lambda *s: torch.randn(*s, generator=g).cuda()
x, W1, b1, W2, b2 = b.x, rnd(D, F) * 0.1, rnd(D) * 0.1, rnd(D, D) * 0.1, rnd(D) * 0.1
a1, a2 = torch.empty(N, D, device="cuda"), torch.empty(N, D, device="cuda")
emb, demb, dx = torch.empty(B, 2 * D, device="cuda"), rnd(B, 2 * D), torch.empty(N, D, device="cuda")
wst = torch.empty(lib.hcg_tall_workspace_bytes(N, B, D, D), dtype=torch.uint8, device="cuda")
wsm = torch.empty(lib.hcg_mid_workspace_bytes(B, D, D, mxn, mxe), dtype=torch.uint8, device="cuda")
gp, ep, ei, E, stt = p(plan.graph_ptr), p(plan.edge_ptr), p(plan.edge_index), plan.E, p(plan.status)

def timeit(fn, k=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3

chk = _lib.check
cases = {
  "tall fwd L1": lambda: chk(lib.hcg_tall_layer_fwd(p(x), p(W1), p(b1), ei, E, gp, ep, N, B, F, D, mxn, mxe, slope, 1, p(a1), None, None, None, None, stt, p(wst), wst.numel(), st), "f"),
  "mid  fwd L1": lambda: chk(lib.hcg_mid_layer_fwd(p(x), p(W1), p(b1), ei, E, gp, ep, N, B, F, D, mxn, mxe, slope, 1, p(a1), None, None, None, None, stt, st), "f"),
  "tall fwd L2+pool": lambda: chk(lib.hcg_tall_layer_fwd(p(a1), p(W2), p(b2), ei, E, gp, ep, N, B, D, D, mxn, mxe, slope, 1, p(a2), p(emb), None, None, None, stt, p(wst), wst.numel(), st), "f"),
  "mid  fwd L2+pool": lambda: chk(lib.hcg_mid_layer_fwd(p(a1), p(W2), p(b2), ei, E, gp, ep, N, B, D, D, mxn, mxe, slope, 1, p(a2), p(emb), None, None, None, stt, st), "f"),
  "tall bwd L2 (pooled, dx premasked)": lambda: chk(lib.hcg_tall_layer_bwd(None, p(demb), p(emb), p(a2), None, None, None, None, p(a1), p(W2), ei, E, gp, ep, N, B, D, D, mxn, mxe, slope, 3, p(dx), stt, p(wst), wst.numel(), st), "b"),
  "mid  bwd L2 (pooled, dx premasked)": lambda: chk(lib.hcg_mid_layer_bwd(None, p(demb), p(emb), p(a2), p(a1), p(W2), ei, E, gp, ep, N, B, D, D, mxn, mxe, slope, 3, p(dx), stt, p(wsm), wsm.numel(), st), "b"),
  "tall bwd L1 (no dx)": lambda: chk(lib.hcg_tall_layer_bwd(p(dx), None, None, None, None, None, None, None, p(x), p(W1), ei, E, gp, ep, N, B, F, D, mxn, mxe, slope, 0, None, stt, p(wst), wst.numel(), st), "b"),
  "mid  bwd L1 (no dx)": lambda: chk(lib.hcg_mid_layer_bwd(p(dx), None, None, None, p(x), p(W1), ei, E, gp, ep, N, B, F, D, mxn, mxe, slope, 0, None, stt, p(wsm), wsm.numel(), st), "b"),
}
for k, fn in cases.items():
    print(f"{k:40s} {timeit(fn):8.1f} us")
assert plan.check_status() == 0
