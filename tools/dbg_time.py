"""Dev probe: time the three fused launches of a C3 step with a given library build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hcatgnet_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import hcatgnet_amd as H
from hcatgnet_amd import synth
from hcatgnet_amd.train import FusedTrainStep
lib = _lib.load()
sb = synth.make_config("C2", num_graphs=4096)
m = H.make_network("GCN", H.default_options(), 64).cuda()
batch = sb.as_batch("cuda")
step = FusedTrainStep(m)
for _ in range(5): step(batch)
names = ["hcg_fused_forward", "hcg_head_fwd_bwd", "hcg_fused_layer_bwd", "hcg_step_tail"]
ev = {n: [] for n in names}
for n in names:
    orig = getattr(lib, n)
    def wrap(*a, _o=orig, _n=n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); rc = _o(*a); e.record(); ev[_n].append((s, e)); return rc
    setattr(lib, n, wrap)
for _ in range(50): step(batch)
torch.cuda.synchronize()
print(_lib.LIB_PATH)
for n in names:
    ms = [s.elapsed_time(e) for s, e in ev[n]]
    if n == "hcg_fused_layer_bwd":
        print(f"  {n} L2: {1e3 * sum(ms[0::2]) / len(ms[0::2]):.1f} us   L1: {1e3 * sum(ms[1::2]) / len(ms[1::2]):.1f} us")
    else:
        print(f"  {n}: {1e3 * sum(ms) / len(ms):.1f} us")
