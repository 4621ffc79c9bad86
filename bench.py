#!/usr/bin/env python3
"""bench.py -- graphs/sec of the GCN fwd+bwd hot path on N MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic batch that is already resident in HBM:
    batch plan (CSR / gcn_norm from the int64 edge_index, rebuilt EVERY step)
    -> forward (2 x GCNConv+LeakyReLU, [max, mean] pool, readout MLP)
    -> sqrt(MSE) loss (reference utils/utils_model.py:64)
    -> backward (all weight gradients)
    -> [N > 1] RCCL all-reduce of the flat gradient buffer.
    -> Adam update (the reference's optimiser, model/networks.py:38).
`value` times ALL of it (hipGraph replay of the captured step when capture succeeds); `fwd_bwd_only` reports the
same step without the update, and the CPU baseline runs the same full step.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3|C5|...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

T_START = time.perf_counter()
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3", help="BASELINE.json config: C3 (= configs[2], the metric's), C2, C5, C1; "
                    "REAL / REAL40 = the reference's own graph sizes (57-117 atoms, F = 25) at B = 4096 / 40")
    ap.add_argument("--num-graphs", type=int, default=None, help="override graphs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=60)
    ap.add_argument("--roofline-entry", default=None, help="C-ABI entry point timed for the roofline object")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step into a hipGraph (eager launches)")
    ap.add_argument("--forward-only", action="store_true",
                    help="BASELINE configs[1] (C2): plan build + forward only, no loss / backward (not the headline metric)")
    return ap.parse_args()


class EntryTimer:
    """HIP-event timing of ONE C-ABI entry point on the stream it launches on (torch's current
    stream: the library only enqueues on the stream it is handed)."""

    # entry points that launch the SAME kernel family as the named one (training forms of the pooled layer)
    SAME_KERNEL = {"hcg_fused_layer_bwd": ("hcg_fused_layer_bwd_poolbits",),
                   "hcg_fused_stack2_fwd": ("hcg_fused_stack2_fwd_train",),
                   "hcg_fused_layer_fwd": ("hcg_fused_layer_fwd_train",)}

    def __init__(self, lib, name):
        self.lib, self.name = lib, name
        self.names = (name,) + self.SAME_KERNEL.get(name, ())
        self.orig = {n: getattr(lib, n) for n in self.names}
        self.events, self.enabled = [], False

    def install(self):
        def make(orig):
            def wrapper(*a):
                if not self.enabled:
                    return orig(*a)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                rc = orig(*a)
                e.record()
                self.events.append((s, e))
                return rc
            return wrapper
        for n in self.names:
            setattr(self.lib, n, make(self.orig[n]))

    def uninstall(self):
        for n in self.names:
            setattr(self.lib, n, self.orig[n])

    def mean_ms(self, per_step_calls):
        torch.cuda.synchronize()
        ms = [s.elapsed_time(e) for s, e in self.events]
        return (sum(ms) / len(ms) if ms else float("nan")), len(ms)


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def usable_cpus() -> int:
    """CPUs this process may actually use: min(affinity mask, cgroup CPU quota).  On the GPU box the
    affinity mask can name every core of the host while the container's quota is ~16."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 64))


def cpu_baseline(cfg_name, num_graphs, steps):
    """The reference CPU scatter path (oracle = torch-native restatement, validated bit-exact against
    the reference's embeddings) timed on this box's host cores.  Checker / baseline only."""
    from hcatgnet_amd import synth
    from oracle import gcn_oracle
    import hcatgnet_amd as H
    cores = usable_cpus()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads")
    sb = synth.make_config(cfg_name, num_graphs=num_graphs)
    cfg = synth.CONFIGS[cfg_name]
    model = H.make_network("GCN", H.default_options(embedding_dim=cfg["hidden"]), cfg["feat"])
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    p, opt = gcn_oracle.make_train_state(params)
    ts = []
    for i in range(3 + steps):
        t0 = time.perf_counter()
        gcn_oracle.train_step(p, opt, sb.x, sb.edge_index, sb.batch, sb.y, sb.num_graphs)
        ts.append(time.perf_counter() - t0)
        if i % 5 == 0:
            log(f"cpu_baseline step {i}: {ts[-1] * 1e3:.1f} ms")
    ts = sorted(ts[3:])
    med = ts[len(ts) // 2]
    model_name = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model_name = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": sb.num_graphs / med, "unit": "graphs/s", "cores": cores, "kind": "port",
            "sample": f"{steps} timed training steps (fwd + sqrt(MSE) + bwd + Adam; median, 3 warm-up) of the full {cfg_name} batch "
                      f"({sb.num_graphs} graphs) with the torch CPU restatement of the reference's PyG scatter path, "
                      f"{cores} threads; PyG itself is not installable here",
            "ms_per_step": med * 1e3, "cpu_model": model_name}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 control flow on a ONE-GPU box (tools/rehearse_multi_rank.sh): every rank on device 0,
    # gloo instead of RCCL (RCCL refuses two ranks on one device).  Never set by the driver; the numbers mean nothing.
    rehearsal = os.environ.get("HCG_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import hcatgnet_amd as H
    from hcatgnet_amd import _lib, algbytes, synth
    from hcatgnet_amd.ddp import DataParallelGCN
    lib = _lib.load()
    log(f"rank {rank}/{world}: library loaded")

    cfg_name = args.config
    cfg = synth.CONFIGS[cfg_name]
    sb = synth.make_config(cfg_name, rank=rank, num_graphs=args.num_graphs)
    x, ei, bvec, y = sb.x.to(dev), sb.edge_index.to(dev), sb.batch.to(dev), sb.y.to(dev)
    N, E, B, F, D = x.shape[0], ei.shape[1], sb.num_graphs, cfg["feat"], cfg["hidden"]
    opt = H.default_options(embedding_dim=D)
    model = H.make_network("GCN", opt, F).to(dev)
    dp = None            # created after the hipGraph capture: no RCCL activity while a stream is capturing
    y2 = y.unsqueeze(1)

    from hcatgnet_amd.train import FusedTrainStep

    def make_batch():      # a fresh Batch per step: its plan (graph_ptr / edge_ptr from the int64 inputs) is rebuilt every step
        return H.Batch(x, ei, bvec, B, y=y, max_nodes=sb.max_nodes, max_edges=sb.max_edges, edges_grouped=True)

    fused_ok = (not args.forward_only) and FusedTrainStep.unsupported_reason(model, make_batch()) is None
    # the training step of the reference's loop (utils/utils_model.py:60-68) WITH the Adam update; `fwdbwd` stops
    # after the backward (gradients only), for the secondary "fwd+bwd only" figure
    trainer = FusedTrainStep(model, optimizer_step=True) if fused_ok else None
    fwdbwd = FusedTrainStep(model, optimizer_step=False) if fused_ok else None

    def autograd_step(with_opt=True):
        model.optimizer.zero_grad(set_to_none=True)
        out = model(make_batch())                        # plan build + forward
        loss = torch.sqrt(model.loss(out, y2))
        loss.backward()
        if dp is not None:
            dp.reduce_gradients()
        if with_opt:
            model.optimizer.step()
        return loss

    def forward_step():
        with torch.no_grad():
            return model(make_batch())

    def eager_step(with_opt=True):
        if args.forward_only:
            return forward_step()
        if not fused_ok:
            return autograd_step(with_opt)
        return (trainer if with_opt else fwdbwd)(make_batch())

    replay = {}

    def capture_all():
        """Everything the step enqueues (plan build, forward, head with loss, backward, slab reduction, Adam) goes
        into hipGraphs; the RCCL all-reduce stays an eager call between the backward graph and the update graph."""
        if args.forward_only or not fused_ok:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    forward_step() if args.forward_only else autograd_step(False)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if not args.forward_only:
                model.optimizer.zero_grad(set_to_none=True)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                forward_step() if args.forward_only else autograd_step(False)
            grads = [] if args.forward_only else [p.grad for p in model.parameters()]

            def run(with_opt=True):
                g.replay()
                if dp is not None:
                    dp.reduce_gradients(grads=grads)
                if with_opt and not args.forward_only:
                    model.optimizer.step()
            replay["full"] = run
            replay["fwdbwd"] = lambda: run(False)
            return
        trainer.capture(make_batch)
        fwdbwd.capture(make_batch)
        replay["full"] = trainer.replay
        replay["fwdbwd"] = fwdbwd.replay

    def timed(k, fn):
        if world > 1:
            dist.barrier(**({} if rehearsal else {"device_ids": [local_rank]}))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(**({} if rehearsal else {"device_ids": [local_rank]}))
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    log(f"inputs resident: N={N} E={E} B={B} F={F} D={D}; fused trainer: {fused_ok}")

    def count_launching_calls():
        """Library entry points that enqueue kernels during ONE eager step (launches per step)."""
        skip = ("_bytes", "_job", "_supported", "_per_tile", "hcg_version", "hcg_error_string")
        names = [n for n in _lib.SIGNATURES if not n.endswith(skip) and n not in skip]
        counts, origs = {}, {}
        for n in names:
            origs[n] = getattr(lib, n)
            def wrap(*a, _n=n):
                counts[_n] = counts.get(_n, 0) + 1
                return origs[_n](*a)
            setattr(lib, n, wrap)
        try:
            eager_step()
            torch.cuda.synchronize()
        finally:
            for n in names:
                setattr(lib, n, origs[n])
        return counts
    for _ in range(args.warmup):
        eager_step()
    torch.cuda.synchronize()
    log("warm-up done")
    launch_counts = count_launching_calls()

    launch_mode, graph_err = "eager", None
    if not args.no_graph:
        try:
            if world > 1 and fused_ok:          # the exchange sits between backward and update: two graphs per step
                trainer.grad_sync = fwdbwd.grad_sync = (lambda flat: None)
            capture_all()                       # before RCCL comes up: no collective activity while a stream is capturing
            launch_mode = "hipgraph"
            log("step captured into hipGraphs")
        except Exception as exc:  # report, never hide: the eager number stands
            replay.clear()
            graph_err = f"{type(exc).__name__}: {exc}"
            log(f"graph capture failed, keeping eager launches: {graph_err}")

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        dp = DataParallelGCN(model)           # broadcasts rank-0 weights (in place: the graphs see them)
        if fused_ok:
            trainer.grad_sync = dp.reduce_flat
            fwdbwd.grad_sync = dp.reduce_flat
        for _ in range(3):
            eager_step()
        log(f"RCCL process group up: world {world}")

    # kernel-level roofline: HIP events around the dominant entry point, inside a timed eager loop
    mid = sb.max_nodes > 32
    entry = args.roofline_entry or (("hcg_mid_layer_fwd" if mid else "hcg_fused_stack2_fwd") if args.forward_only
                                    else ("hcg_mid_layer_bwd" if mid else "hcg_fused_layer_bwd"))
    timer = EntryTimer(lib, entry)
    timer.install()
    timer.enabled = True
    dt = timed(args.steps, eager_step)
    timer.enabled = False
    k_ms, k_calls = timer.mean_ms(args.steps)
    timer.uninstall()
    log(f"timed (with kernel events): {dt / args.steps * 1e3:.3f} ms/step")
    dt_eager = timed(args.steps, eager_step)
    log(f"timed eager (full step): {dt_eager / args.steps * 1e3:.3f} ms/step")
    # distribution of single steps (median, p10 / p90): HIP events around every step of one more pass
    evs = []
    for _ in range(args.steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eager_step()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    per_step = sorted(a.elapsed_time(b) for a, b in evs)
    pct = {f"p{q}": per_step[min(len(per_step) - 1, int(len(per_step) * q / 100))] for q in (10, 50, 90)}
    dt_best, dt_fb, dt_graph = dt_eager, None, None
    if "full" in replay:
        for _ in range(max(3, args.warmup // 2)):
            replay["full"]()
        dt_graph = timed(args.steps, replay["full"])
        log(f"timed hipGraph replay (full step): {dt_graph / args.steps * 1e3:.3f} ms/step")
        if dt_graph <= dt_eager:
            dt_best = dt_graph
        else:   # the no-autograd step issues 6 launches from a host loop that runs ahead of the GPU: replay need not win
            launch_mode = "eager"
        if not args.forward_only:
            for _ in range(3):
                replay["fwdbwd"]()
            dt_fb = timed(args.steps, replay["fwdbwd"])
            log(f"timed hipGraph replay (fwd+bwd only, no update): {dt_fb / args.steps * 1e3:.3f} ms/step")
    elif not args.forward_only:
        dt_fb = timed(args.steps, lambda: eager_step(False))

    bd = algbytes.breakdown(N, E, B, F, D, opt.n_convolutions)
    step_bytes = sum(v for k, v in bd.items() if not args.forward_only or k.endswith("_fwd") or k == "csr_build")
    # algorithmic bytes of ONE launch of the timed entry point, averaged over its launches in a step
    # (layer 2 backward also carries the pool backward it fuses; layer 2 forward the pool forward)
    entry_bytes = {"hcg_gcn_layer_bwd": (bd["conv1_bwd"] + bd["conv2_bwd"]) / 2.0,
                   "hcg_gcn_layer_fwd": (bd["conv1_fwd"] + bd["conv2_fwd"]) / 2.0,
                   "hcg_fused_layer_bwd": (bd["conv1_bwd"] + bd["conv2_bwd"] + bd["pool_bwd"]) / 2.0,
                   "hcg_mid_layer_bwd": (bd["conv1_bwd"] + bd["conv2_bwd"] + bd["pool_bwd"]) / 2.0,
                   "hcg_mid_layer_fwd": (bd["conv1_fwd"] + bd["conv2_fwd"] + bd["pool_fwd"]) / 2.0,
                   "hcg_fused_layer_fwd": (bd["conv1_fwd"] + bd["conv2_fwd"] + bd["pool_fwd"]) / 2.0,
                   "hcg_fused_stack2_fwd": bd["conv1_fwd"] + bd["conv2_fwd"] + bd["pool_fwd"]}.get(entry, float("nan"))
    achieved = entry_bytes / (k_ms * 1e-3) / 1e9 if k_ms == k_ms and k_ms > 0 else None
    # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, gfx950
    # correction of MI355X_MICROARCH.md; tools/pmc_traffic.py) -- only valid for the config they were taken on
    traffic = None
    tpath = os.path.join(REPO, "profiles", "traffic_latest.json")
    if cfg_name in ("C2", "C3", "C4") and args.num_graphs is None and os.path.isfile(tpath):
        try:
            tj = json.load(open(tpath))
            prefix = {"hcg_fused_layer_bwd": "k_fused_layer_bwd"}.get(entry)   # (forward traffic: re-profile after STACK2)
            vals = [v["hbm_bytes"] for k, v in tj.items() if prefix and k.startswith(prefix)]
            traffic = sum(vals) / len(vals) if vals else None
        except (OSError, ValueError, KeyError):
            traffic = None

    if rank == 0:
        ms_step = dt_best / args.steps * 1e3
        rec = {
            "metric": "molecular graphs/sec fwd+bwd at 1/2/4/8 MI355X; achieved HBM GB/s" if not args.forward_only
                      else "molecular graphs/sec FORWARD ONLY (configs[1]; not the headline metric)",
            "value": world * B * args.steps / dt_best, "unit": "graphs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg_name}: {B} synthetic graphs/GPU x {N / B:.0f} atoms x {E / B:.0f} directed edges x "
                                   f"{F}-d features, {opt.n_convolutions}xGCNConv({D}) + [max,mean] pool + readout; "
                                   f"full training step: per-step plan/gcn_norm build, forward, sqrt(MSE) loss, backward, "
                                   f"{'RCCL all-reduce, ' if world > 1 else ''}Adam update; launch={launch_mode}",
                       "graphs_per_gpu": B, "nodes": N, "edges": E, "feat": F, "hidden": D,
                       "parallelism": f"dp{world} (batch-of-graphs, RCCL all-reduce of {sum(p.numel() for p in model.parameters())} fp32 grads)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "kernel": entry, "kernel_ms": k_ms, "kernel_launches_timed": k_calls,
                         "algorithmic_bytes_per_launch": entry_bytes},
            "step_roofline": {"algorithmic_bytes_per_step": step_bytes, "achieved": step_bytes / (ms_step * 1e-3) / 1e9,
                              "unit": "GB/s", "frac": step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "breakdown": bd},
            "fwd_bwd_only": None if dt_fb is None else {"value": world * B * args.steps / dt_fb, "unit": "graphs/s",
                                                        "ms_per_step": dt_fb / args.steps * 1e3,
                                                        "note": "same step without the Adam update (gradients only)"},
            "optimizer": type(model.optimizer).__name__ + "(lr=0.01, eps=1e-9), inside the timed step",
            "step_path": "FusedTrainStep (no autograd)" if fused_ok else "autograd",
            "library_launching_calls_per_step": {"total": sum(launch_counts.values()), "by_entry_point": launch_counts},
            "ms_per_step_with_kernel_events": dt / args.steps * 1e3,
            "eager_step_ms_percentiles_hip_events": pct,
            "launch": launch_mode, "eager_ms_per_step": dt_eager / args.steps * 1e3,
            "hipgraph_ms_per_step": (dt_graph / args.steps * 1e3) if "full" in replay else None, "graph_capture_error": graph_err,
        }
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(cfg_name, args.num_graphs, args.cpu_steps)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
