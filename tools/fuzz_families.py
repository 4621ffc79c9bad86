#!/usr/bin/env python3
"""Dev tool: random shapes through the wide-layer kernels (csrc/tall.hip) against the any-shape GPU path -- forward, pooled
embedding and every gradient of one training step, through the autograd path AND through `train.FusedTrainStep`.  Shapes are drawn around the kernels' own boundaries (64 / 65 / 128 / 129 /
224 nodes per graph, node counts that are no multiple of the 32- and 64-row tiles, 1..33 graphs, every K padding).  Batches
with a near-tie in the max pooling or an activation within rounding of the LeakyReLU kink are skipped (there the arg-max / the
slope may legitimately differ between two summation orders: seed 0's case 14 was one such activation, 8e-9 in fp64).
usage: python tools/fuzz_families.py [cases=60] [seed=0] [tall|small]
`small`: 64-wide layers over graphs of 1..130 nodes instead -- whichever of fused.hip / wave.hip / mid.hip the host picks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hcatgnet_amd as H
from hcatgnet_amd import functional as HF, synth
from tests.helpers import rel_inf
from tests.test_gpu_parity import _model_from_params, _rand_params, _step_grads, TOL_DW
from tests.test_gpu_mid import _near_ties
from hcatgnet_amd.train import FusedTrainStep

def run(cases=60, seed=0, small=False, log=print, oracle_step=None):
    """-> (cases run, cases skipped, tags of the failed ones).  `oracle_step(params, sb)` -> (out, emb, {name: grad}) of the
    CPU oracle (given by tests/test_gpu_fuzz.py only: tools never import oracle/): every case is then ALSO compared with it."""
    rng = np.random.default_rng(seed)
    failed = []
    tall_min = HF.TALL_MIN_NODES_D64
    try:
        return _run(cases, rng, small, log, failed, oracle_step)
    finally:
        HF.TALL_MIN_NODES_D64 = tall_min


def _run(cases, rng, small, log, failed, oracle_step=None):
    if not small:
        HF.TALL_MIN_NODES_D64 = 0
    edges = [1, 2, 3, 15, 16, 17, 31, 32, 33, 34, 63, 64, 65, 100, 130] if small else [65, 66, 96, 127, 128, 129, 130, 160, 199, 200, 223, 224]
    skipped = ran = 0
    for case in range(cases):
        D = 64 if small else int(rng.choice([64, 128]))
        top = int(rng.choice(edges)) if rng.random() < 0.6 else (int(rng.integers(1, 131)) if small else int(rng.integers(65, 225)))
        jitter = int(rng.integers(0, min((top - 1) // 2, 60) + 1)) if rng.random() < 0.6 else 0
        nodes = top - jitter                                   # sizes in [top - 2 jitter, top]
        if D == 128:
            feat = int(rng.choice([4, 8, 12, 16, 20, 28, 32, 36, 60, 64, 68, 100, 124, 128]))
        else:
            feat = int(rng.choice([1, 3, 8, 16, 17, 25, 31, 32, 33, 48, 63, 64]))
        B = int(rng.choice([1, 2, 3, 5, 7, 12, 33] + ([64, 203] if small else [])))      # (small batches: the screens below would throw most large ones out)
        deg, extra, seed = int(rng.choice([3, 4, 6])), int(rng.integers(0, 9)), int(rng.integers(0, 10_000))
        tag = f"case {case}: D={D} F={feat} nodes={nodes}+-{jitter} B={B} deg={deg} extra={extra} seed={seed}"
        sb = synth.make_batch(num_graphs=B, nodes=nodes, extra_bonds=extra, max_degree=deg, feat=feat, nodes_jitter=jitter, seed=seed)
        if sb.max_nodes <= 64 and not small:
            skipped += 1
            continue
        params = _rand_params(feat, D, seed=seed + 1)
        m = _model_from_params(H, params)
        batch = sb.as_batch("cuda")
        plan = H.BatchPlan.build(batch.edge_index, batch.batch, batch.x.shape[0], num_graphs=sb.num_graphs, mode="blocked",
                                 max_nodes=sb.max_nodes, max_edges=sb.max_edges)
        batch._hcg_plan = plan
        if not small and not (HF.tall_supported(plan, feat, D) and HF.tall_supported(plan, D, D)):
            log(tag, "-- not supported by the family, skipped")
            skipped += 1
            continue
        m.use_fused = False
        out_g, emb_g, g_g = _step_grads(m, batch, batch.y)
        with torch.no_grad():
            h1 = m.conv1(batch.x, plan, apply_act=True, fused=False)
            h2 = m.conv_layers[0](h1, plan, apply_act=True, fused=False)
        kink = lambda h: bool(((h > -3e-9) & (h < 3e-7)).any())     # an activation AT the LeakyReLU kink: its slope is rounding's call
        if _near_ties(h2.cpu(), sb.batch, sb.num_graphs) != 0 or kink(h1) or kink(h2):
            skipped += 1
            continue
        m.use_fused = True
        out_t, emb_t, g_t = _step_grads(m, batch, batch.y)
        st = plan.check_status()
        errs = {"emb": rel_inf(emb_t, emb_g), "out": rel_inf(out_t, out_g, floor=1.0)}
        errs.update({k: rel_inf(g_t[k], g_g[k]) for k in g_t})
        ok = st == 0 and errs["emb"] <= 2e-6 and errs["out"] <= 2e-6 and all(errs[k] <= TOL_DW for k in g_t)
        ok = ok and all(torch.isfinite(v).all() for v in g_t.values())
        if oracle_step is not None:                            # the CPU oracle (fp64 gradients), not only the any-shape GPU path
            o_out, o_emb, o_g = oracle_step(params, sb)
            errs["oracle:emb"], errs["oracle:out"] = rel_inf(emb_t, o_emb), rel_inf(out_t, o_out, floor=1.0)
            errs.update({"oracle:" + k: rel_inf(g_t[k], o_g[k]) for k in g_t})
            ok = ok and errs["oracle:emb"] <= 1e-5 and errs["oracle:out"] <= 1e-5 and all(errs["oracle:" + k] <= TOL_DW for k in g_t)
        # the same batch through the no-autograd training step (train.FusedTrainStep: its own dispatch, workspaces, jobs)
        step = FusedTrainStep(m, optimizer_step=False)
        if step.unsupported_reason(m, batch) is None:
            m.zero_grad()
            loss = step(batch)
            st |= plan.check_status()
            errs["step:out"] = rel_inf(step.last_out, out_g, floor=1.0)
            errs.update({"step:" + k: rel_inf(v.grad, g_g[k]) for k, v in m.named_parameters()})
            ok = ok and st == 0 and bool(torch.isfinite(loss)) and errs["step:out"] <= 2e-6 and all(
                errs["step:" + k] <= TOL_DW for k in g_g)
        else:
            log(tag, "-- FusedTrainStep:", step.unsupported_reason(m, batch))
        ran += 1
        if not ok:
            failed.append(tag)
            log("FAIL", tag, "status", st, {k: f"{v:.2e}" for k, v in errs.items()})
        else:
            log("ok  ", tag, f"max err {max(errs.values()):.1e}")
    return ran, skipped, failed


if __name__ == "__main__":
    ran, skipped, failed = run(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0,
                               len(sys.argv) > 3 and sys.argv[3] == "small")
    print(f"{ran} run, {skipped} skipped, {len(failed)} failed")
    sys.exit(1 if failed else 0)
