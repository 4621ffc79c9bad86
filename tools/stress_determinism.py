"""Dev tool: many launches of the full fused step (no optimizer update) on C3, the reference-sized batch, the ragged batch and C5; counts
launches whose loss / gradients / activations differ bitwise from the majority."""
import sys, os, collections, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hcatgnet_amd as H
from hcatgnet_amd import synth, _lib
if os.environ.get("HCG_LIB"):          # a library variant (tools/build_variants.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["HCG_LIB"])
print("library", _lib.LIB_PATH, flush=True)
from hcatgnet_amd.train import FusedTrainStep
def digest(ts): 
    h = hashlib.md5()
    for t in ts: h.update(t.detach().cpu().numpy().tobytes())
    return h.hexdigest()
for cfg, reps in (("C3", int(sys.argv[1]) if len(sys.argv) > 1 else 300), ("REAL", int(sys.argv[2]) if len(sys.argv) > 2 else 80),
                  ("RAGGED", int(sys.argv[3]) if len(sys.argv) > 3 else 80), ("C5", int(sys.argv[4]) if len(sys.argv) > 4 else 80)):
    sb = synth.make_config("C3", nodes_jitter=6, group_by_size=True) if cfg == "RAGGED" else synth.make_config(cfg)
    c0 = synth.CONFIGS["C3" if cfg == "RAGGED" else cfg]
    m = H.make_network("GCN", H.default_options(embedding_dim=c0["hidden"]), c0["feat"]).cuda()
    batch = sb.as_batch("cuda")
    step = FusedTrainStep(m, optimizer_step=False)
    c = collections.Counter()
    for rep in range(reps):
        loss = step(batch)
        bufs = [v for k, v in step._bufs.items() if k != "cap"][0]
        c[digest([loss, step._flat, bufs["acts"][0], bufs["acts"][1], bufs["emb"], bufs["demb"], bufs["dacts"][0], bufs["out"]])] += 1
    print(cfg, "launches", reps, "distinct results", len(c), "-> BAD launches:", reps - c.most_common(1)[0][1], flush=True)
