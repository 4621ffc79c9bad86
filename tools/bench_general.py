#!/usr/bin/env python3
"""Dev tool: time real-graph-sized batches through the mid-size kernels (use_fused=True) and the any-shape path."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hcatgnet_amd as H
from hcatgnet_amd import synth, _lib
lib = _lib.load()
def run(name, sb, F, fused, fwd_only=False):
    m = H.make_network("GCN", H.default_options(use_fused=fused), F).cuda()
    b = sb.as_batch("cuda")
    def step():
        m.zero_grad(set_to_none=True); b._hcg_plan = None
        if fwd_only:
            with torch.no_grad(): m(b)
            return
        out = m(b); torch.sqrt(m.loss(out, b.y.unsqueeze(1))).backward()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"{name:40s} fused={fused!s:5s} {'fwd' if fwd_only else 'fwd+bwd'} {dt*1e3:8.3f} ms/step  {sb.num_graphs/dt:12.0f} graphs/s", flush=True)
real = synth.make_batch(num_graphs=4096, nodes=87, extra_bonds=4, max_degree=4, feat=25, nodes_jitter=30)
big = synth.make_batch(num_graphs=4096, nodes=150, extra_bonds=6, max_degree=4, feat=32, nodes_jitter=34)
b40 = synth.make_batch(num_graphs=40, nodes=87, extra_bonds=4, max_degree=4, feat=25, nodes_jitter=30)
for fused in (True, False):
    run("real-sized (57-117 nodes, F=25, B=4096)", real, 25, fused)
    run("real-sized (57-117 nodes, F=25, B=4096)", real, 25, fused, True)
    run("biaryl-sized (116-184 nodes, F=32, B=4096)", big, 32, fused)
    run("reference batch (B=40, 57-117 nodes)", b40, 25, fused)
# per-launch timing of the mid kernels
names = ["hcg_mid_layer_fwd", "hcg_mid_layer_bwd"]
ev = {n: [] for n in names}
for n in names:
    orig = getattr(lib, n)
    def wrap(*a, _o=orig, _n=n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); rc = _o(*a); e.record(); ev[_n].append((s, e)); return rc
    setattr(lib, n, wrap)
m = H.make_network("GCN", H.default_options(), 25).cuda(); b = real.as_batch("cuda")
for _ in range(12):
    m.zero_grad(set_to_none=True); out = m(b); torch.sqrt(m.loss(out, b.y.unsqueeze(1))).backward()
torch.cuda.synchronize()
for n in names:
    ms = [s.elapsed_time(e) for s, e in ev[n]][4:]
    print(n, "even/odd launch mean us:", round(1e3 * sum(ms[0::2]) / len(ms[0::2]), 1), round(1e3 * sum(ms[1::2]) / len(ms[1::2]), 1))
