#!/usr/bin/env python3
"""Dev tool: the any-shape kernel family (plan + GEMM + CSR segmented sum + pool, through autograd) on a BASELINE
config, for a per-kernel profile:  rocprofv3 --kernel-trace --stats -- python3 tools/anyshape_c5.py [C5|REAL] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcatgnet_amd as H
from hcatgnet_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sb = synth.make_config(name)
cfg = synth.CONFIGS[name]
opt = H.default_options()
opt.embedding_dim = cfg["hidden"]
opt.use_fused = os.environ.get("USE_FUSED", "0") == "1"
m = H.make_network("GCN", opt, cfg["feat"]).cuda()
b = sb.as_batch("cuda")
def step():
    m.optimizer.zero_grad(set_to_none=True)
    out = m(b)
    loss = torch.sqrt(m.loss(out.squeeze(-1), b.y))
    loss.backward()
    m.optimizer.step()
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
print(f"{name} any-shape autograd step: {(time.perf_counter() - t0) / steps * 1e3:.4f} ms")
