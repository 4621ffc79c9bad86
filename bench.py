#!/usr/bin/env python3
"""bench.py -- graphs/sec of the GCN fwd+bwd hot path on N MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic batch that is already resident in HBM:
    batch plan (graph_ptr / edge_ptr / gcn_norm from the int64 edge_index, rebuilt EVERY step)
    -> forward (2 x GCNConv+LeakyReLU, [max, mean] pool, readout MLP)
    -> sqrt(MSE) loss (reference utils/utils_model.py:64)
    -> backward (all weight gradients)
    -> [N > 1] RCCL all-reduce of the flat gradient buffer
    -> Adam update (the reference's optimiser, model/networks.py:38).

What `value` is.  Consecutive steps run on DISTINCT batches: `--distinct-batches` (16) synthetic batches, each with its
own step buffers (activations, gradients, slabs), are visited round-robin, so that more than 2 GB is touched between two
uses of the same bytes -- nothing of a step is served from the 256 MiB Infinity Cache because an earlier step left it
there.  Before the timed region the same rotation runs for `--sustain` (5) seconds (`sustained`: its own graphs/s), so
the K timed steps run at the clocks the chip holds under load, not in a cold burst.  `value` = the K steps timed right
behind it.  `burst` is the old figure: one batch replayed in place (cache resident), timed from idle.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3|C5|...]
With --gpus N > 1 and no RANK in the environment this script starts its own N ranks (one per GPU, a child
`python -m torch.distributed.run`), relays rank 0's JSON line and exits with the child's code; launched under an
external torch.distributed.run it is one of the ranks.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

T_START = time.perf_counter()
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3", help="BASELINE.json config: C3 (= configs[2], the metric's), C2, C5, C1; "
                    "REAL / REAL40 = the reference's own graph sizes (57-117 atoms, F = 25) at B = 4096 / 40; "
                    "RAGGED = C3 with n_g ~ U{24..36} (SURVEY 8d's variable-size variant)")
    ap.add_argument("--num-graphs", type=int, default=None, help="override graphs per GPU")
    ap.add_argument("--distinct-batches", type=int, default=16, help="distinct resident batches (+ step buffers) visited round-robin")
    ap.add_argument("--sustain", type=float, default=5.0, help="seconds of the same rotation run right before the timed steps")
    ap.add_argument("--combine", default="sse", choices=("sse", "mean"),
                    help="N > 1: 'sse' = gradient of sqrt(MSE) over the concatenated batch of all ranks (the reference's "
                         "semantics at batch N*B); 'mean' = mean of per-rank RMSE gradients (DDP convention)")
    ap.add_argument("--exchange", default="auto", choices=("auto", "rccl", "oneshot"),
                    help="N > 1: 'rccl' = all-reduce of the flat gradient between two launches; 'oneshot' = every rank writes its "
                         "gradients into the peers' inboxes over the direct xGMI links inside the slab reduction + Adam launch; "
                         "'auto' = one-shot if it sets up and passes its self test on this machine, else RCCL")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-gate", action="store_true", help="development only: skip the oracle check in front of the timings "
                    "(the JSON line then says parity_gate: null)")
    ap.add_argument("--no-ragged", action="store_true", help="skip the secondary ragged-batch measurement")
    ap.add_argument("--cpu-steps", type=int, default=40)
    ap.add_argument("--roofline-entry", default=None, help="C-ABI entry point timed for the roofline object")
    ap.add_argument("--plan-overlap", default="fused", choices=("none", "fused", "fork"),
                    help="captured step: 'none' = the step's own plan build in front of its forward; 'fused' = every step derives "
                         "the NEXT batch's plan inside its last launch (slab reduction + Adam; -2.5 % at C3); 'fork' = on a forked "
                         "branch of the graph (measured SLOWER on ROCm 7.2)")
    ap.add_argument("--no-window", action="store_true", help="one hipGraph per step even where a whole rotation could be one graph launch")
    ap.add_argument("--no-graph", action="store_true", help="do not capture the step into a hipGraph (eager launches)")
    ap.add_argument("--forward-only", action="store_true",
                    help="BASELINE configs[1] (C2): plan build + forward only, no loss / backward (not the headline metric)")
    return ap.parse_args()


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 from a bare shell: start the ranks as a CHILD process before anything here touches the GPU
# ---------------------------------------------------------------------------------------------------------------------
def self_launch(args) -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {args.gpus} without RANK in the environment: starting {args.gpus} ranks: {' '.join(cmd)}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:                      # relay; the record is the LAST line that parses as the bench's JSON
        out = out.rstrip("\n")
        try:
            rec = json.loads(out)
            if isinstance(rec, dict) and "metric" in rec:
                line = out
                continue
        except ValueError:
            pass
        print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        log("the ranks exited 0 without a JSON line")
        rc = 1
    return rc


def isolated_probe(world, rehearsal, combine, timeout_s=420.0):
    """Both data-parallel graph forms tried once in a CHILD process per rank (`python -m hcatgnet_amd.xgmi`: its own process
    group on the next port; the one-shot exchange with set-up, self test and 64 free-running real steps; then the RCCL
    collective recorded into the step's hipGraph, 32 replays) before THIS process touches the GPU.  A failure no `try` can
    catch -- a GPU memory fault on a peer mapping aborts the process, a wedged launch or collective never returns -- then ends
    the child, not the bench.  -> {"oneshot": bool, "captured": bool}: what the child had recorded when it ended (a phase it
    never finished counts as failed); the plain RCCL form needs neither.  Stdlib only (nothing here may initialise HIP)."""
    import tempfile
    env = dict(os.environ)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)          # the child group's rank 0 hosts its own store ...
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str((int(os.environ.get("MASTER_PORT", "29500")) - 1024 + 1) % (65535 - 1024) + 1024)   # ... on the next port
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = REPO + os.pathsep + env.get("PYTHONPATH", "")
    fd, out_path = tempfile.mkstemp(prefix="hcg_probe_", suffix=".json")
    os.close(fd)
    env["HCG_PROBE_OUT"] = out_path
    cmd = [sys.executable, "-m", "hcatgnet_amd.xgmi", "--combine", combine]
    if rehearsal:
        cmd += ["--one-device", "--soak-steps", "0"]      # (two ranks on one device starve each other in a free-running soak)
    t0 = time.perf_counter()
    verdicts = {"oneshot": False, "captured": False}
    try:
        proc = subprocess.Popen(cmd, env=env, stdout=sys.stderr, stderr=sys.stderr, start_new_session=True)
    except OSError as exc:
        log(f"data-parallel probe could not start ({exc}): plain RCCL form")
        return verdicts
    try:
        rc = proc.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, 9)                         # exactly the process group started above
        except OSError:
            pass
        proc.wait()
        rc = -9
    try:
        got = json.load(open(out_path))
        verdicts.update({k: bool(got.get(k, False)) for k in verdicts})
    except (OSError, ValueError):
        pass
    try:
        os.unlink(out_path)
    except OSError:
        pass
    log(f"data-parallel probe (child process, world {world}): exit {rc} after {time.perf_counter() - t0:.1f} s -> {verdicts}")
    return verdicts


class EntryTimer:
    """HIP-event timing of ONE C-ABI entry point on the stream it launches on (torch's current
    stream: the library only enqueues on the stream it is handed)."""

    # entry points that launch the SAME kernel family as the named one (training forms of the pooled layer)
    SAME_KERNEL = {"hcg_fused_layer_bwd": ("hcg_fused_layer_bwd_poolbits",),
                   "hcg_fused_stack2_fwd": ("hcg_fused_stack2_fwd_train",),
                   "hcg_fused_layer_fwd": ("hcg_fused_layer_fwd_train",)}

    def __init__(self, lib, name):
        import torch
        self.torch = torch
        self.lib, self.name = lib, name
        self.names = (name,) + self.SAME_KERNEL.get(name, ())
        self.orig = {n: getattr(lib, n) for n in self.names}
        self.events, self.enabled = [], False

    def install(self):
        torch = self.torch

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

    def mean_ms(self):
        self.torch.cuda.synchronize()
        ms = [s.elapsed_time(e) for s, e in self.events]
        return (sum(ms) / len(ms) if ms else float("nan")), len(ms)


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
    the reference's embeddings) timed on this box's host cores.  Checker / baseline only.  Three figures
    (BASELINE.md 3): all cores incl. Adam (`value`), all cores without the optimiser, one thread."""
    import torch
    from hcatgnet_amd import synth
    from oracle import gcn_oracle
    import hcatgnet_amd as H
    cores = usable_cpus()
    sb = synth.make_config(cfg_name, num_graphs=num_graphs)
    cfg = synth.CONFIGS[cfg_name]
    model = H.make_network("GCN", H.default_options(embedding_dim=cfg["hidden"]), cfg["feat"])
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def measure(threads, n_steps, with_opt, budget_s):
        torch.set_num_threads(threads)
        p, opt = gcn_oracle.make_train_state(params)
        ts, t_begin = [], time.perf_counter()
        for i in range(2 + n_steps):
            t0 = time.perf_counter()
            if with_opt:
                gcn_oracle.train_step(p, opt, sb.x, sb.edge_index, sb.batch, sb.y, sb.num_graphs)
            else:
                opt.zero_grad()
                out, _ = gcn_oracle.gcn_forward(p, sb.x, sb.edge_index, sb.batch, sb.num_graphs)
                gcn_oracle.rmse_loss(out, sb.y).backward()
            ts.append(time.perf_counter() - t0)
            if i >= 4 and time.perf_counter() - t_begin > budget_s:      # bounded sample
                break
        ts = sorted(ts[2:])
        return ts[len(ts) // 2], ts[0], len(ts)
    log(f"cpu_baseline: {cores} threads, full step")
    med, best, n_full = measure(cores, steps, True, 14.0)
    log(f"cpu_baseline: {cores} threads, no optimiser")
    med_noopt, _, n_noopt = measure(cores, max(6, steps // 3), False, 7.0)
    log("cpu_baseline: 1 thread")
    med_1t, _, n_1t = measure(1, 6, True, 10.0)
    torch.set_num_threads(cores)
    model_name = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model_name = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": sb.num_graphs / med, "unit": "graphs/s", "cores": cores, "kind": "port",
            "sample": f"{n_full} timed training steps (fwd + sqrt(MSE) + bwd + Adam; median, 2 warm-up) of the full {cfg_name} batch "
                      f"({sb.num_graphs} graphs) with the torch CPU restatement of the reference's PyG scatter path, "
                      f"{cores} threads; PyG itself is not installable here",
            "ms_per_step": med * 1e3, "best_ms_per_step": best * 1e3, "cpu_model": model_name,
            "without_optimizer": {"value": sb.num_graphs / med_noopt, "ms_per_step": med_noopt * 1e3, "cores": cores,
                                  "steps": n_noopt},
            "one_thread": {"value": sb.num_graphs / med_1t, "ms_per_step": med_1t * 1e3, "cores": 1, "steps": n_1t}}


def parity_gate(model, r0, trainer, forward_only, dev):
    """"parity gates -- run before any timing is accepted".  Step 0 on batch 0 of the rotation through the
    product path (the same FusedTrainStep / C-ABI launches the timed loop issues, gradients only) against the fp64 CPU oracle:
    loss, outputs, pooled embedding and EVERY gradient tensor.  Batch 0 is first made decidable (oracle/screen.py: the node
    features of the few graphs that own an activation within 2e-6 of the LeakyReLU kink or a max-pool near-tie are re-drawn
    -- one such element legitimately moves a weight gradient by ~1e-3); the timed rotation uses the screened batch.
    The oracle is the CHECKER here, never the thing measured.  -> the `parity_gate` object of the JSON line."""
    import torch
    from oracle import gcn_oracle, screen
    t0 = time.perf_counter()
    sb = r0.sb
    params = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x_cpu, redrawn = screen.make_decidable(params, sb.x, sb.edge_index, sb.batch, sb.num_graphs, seed=sb.num_graphs)
    sb.x = x_cpu
    r0.x.copy_(x_cpu.to(dev))

    def rel(a, ref, floor=1e-30):
        a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
        return float((a - ref).abs().max()) / max(float(ref.abs().max()), floor)
    tol, tol_w = 1e-5, 1e-4
    errs, limits = {}, {}
    if forward_only:
        with torch.no_grad():
            out, emb = model(r0.fresh(), True)
        o_out, o_emb = gcn_oracle.gcn_forward({k: v.double() for k, v in params.items()}, sb.x.double(), sb.edge_index, sb.batch,
                                              sb.num_graphs)
        errs.update(out=rel(out, o_out, 1.0), emb=rel(emb, o_emb))
        limits.update(out=tol, emb=tol)
    else:
        if trainer is not None:                    # the no-autograd step the timed loop issues
            loss = trainer(r0.fresh())
            out, emb = trainer.last_out, trainer._bufs["cap"]["emb"][:sb.num_graphs]
        else:                                      # models outside the fused step: the autograd path (same kernels per op)
            model.optimizer.zero_grad(set_to_none=True)
            out, emb = model(r0.fresh(), True)
            loss = torch.sqrt(model.loss(out, r0.y2))
            loss.backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        o_loss, o_out, o_emb, o_grads = gcn_oracle.train_step_grads(params, sb.x, sb.edge_index, sb.batch, sb.y, sb.num_graphs,
                                                                    dtype=torch.float64)
        errs.update(loss=abs(float(loss) - float(o_loss)) / abs(float(o_loss)), out=rel(out, o_out, 1.0), emb=rel(emb, o_emb))
        limits.update(loss=tol, out=tol, emb=tol)
        for k, g in grads.items():
            errs["d" + k] = rel(g, o_grads[k])
            limits["d" + k] = tol_w if k.endswith("lin.weight") else tol     # conv weights sum > 1e5 terms
    failed = [k for k in errs if not (errs[k] <= limits[k])]
    return {"passed": not failed, "failed": failed, "errors": errs, "worst": max(errs.values()),
            "tolerance": {"default": tol, "conv_weight_gradients": tol_w, "metric": "||d||_inf / ||ref||_inf per tensor (outputs: max(||ref||_inf, 1))"},
            "oracle": "oracle/gcn_oracle.py in fp64 on the host (torch CPU restatement of the reference's PyG path, pinned bit-exact "
                      "on the reference's embeddings.csv)",
            "what": ("forward of batch 0" if forward_only else
                     "step 0 of batch 0: loss, outputs, pooled embedding, every gradient tensor") +
                    f" ({sb.num_graphs} graphs; {redrawn} graphs' features re-drawn by oracle/screen.py so that LeakyReLU / arg-max "
                    f"branches are decidable at fp32)",
            "seconds": time.perf_counter() - t0}


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(env_world) if env_world is not None else 1
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}` from a bare shell: it starts its own ranks)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE line, the JSON record: whatever libraries print there (RCCL announces its version on stdout at
    # communicator set-up, gloo its peers) goes to stderr from here on
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # N > 1, --exchange auto: the one-shot exchange has to survive a sacrificial child process group first (before anything
    # here touches the GPU); the ranks combine their verdicts once the real process group is up
    probe_ok, probe_captured = None, False
    if (world > 1 and args.exchange == "auto" and not args.forward_only
            and (os.environ.get("HCG_BENCH_REHEARSAL") != "1" or os.environ.get("HCG_PROBE_IN_REHEARSAL") == "1")):
        probe_ok = isolated_probe(world, os.environ.get("HCG_BENCH_REHEARSAL") == "1", args.combine)

    import torch
    import torch.distributed as dist
    # rehearsal of the N > 1 control flow on a ONE-GPU box (tools/rehearse_multi_rank.sh): every rank on device 0,
    # gloo instead of RCCL (RCCL refuses two ranks on one device).  Never set by the driver; the numbers mean nothing.
    rehearsal = os.environ.get("HCG_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU fallback")
    if not rehearsal and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import hcatgnet_amd as H
    from hcatgnet_amd import _lib, algbytes, synth
    from hcatgnet_amd.ddp import DataParallelGCN
    from hcatgnet_amd.train import FusedTrainStep
    if os.environ.get("HCG_LIB"):      # a library variant for A/B measurements (tools/build_variants.sh)
        _lib.LIB_PATH = os.path.abspath(os.environ["HCG_LIB"])
    lib = _lib.load()
    # development A/B switches (tools/ab_env.sh): measurement tooling only, the product has no environment switches
    if os.environ.get("HCG_NO_POOLBITS") == "1":
        FusedTrainStep.POOLBITS = False
    if os.environ.get("HCG_NO_PREMASK") == "1":
        FusedTrainStep.PREMASK = False
    if os.environ.get("HCG_NO_OVERLAP") == "1":
        FusedTrainStep.OVERLAP_GROUPS = False
    log(f"rank {rank}/{world}: library loaded")

    cfg_name = args.config
    ragged_cfg = dict(nodes_jitter=6, group_by_size=True)   # n_g ~ U{24..36}; the loader groups the graphs of a batch by size class
    if cfg_name == "RAGGED":
        cfg_name, extra = "C3", ragged_cfg
    else:
        extra = {}
    cfg = synth.CONFIGS[cfg_name]
    NB = max(1, args.distinct_batches)
    F, D = cfg["feat"], cfg["hidden"]
    opt = H.default_options(embedding_dim=D)
    model = H.make_network("GCN", opt, F).to(dev)

    class Resident:
        """One synthetic batch in HBM (its own seed stream: base + 1000 * (rank + world * i))."""

        def __init__(self, i, **kw):
            sb = synth.make_config(cfg_name, rank=rank + world * i, num_graphs=args.num_graphs, **kw)
            self.sb = sb
            self.x, self.ei, self.bvec, self.y = sb.x.to(dev), sb.edge_index.to(dev), sb.batch.to(dev), sb.y.to(dev)
            self.y2 = self.y.unsqueeze(1)
            self.B, self.N, self.E = sb.num_graphs, self.x.shape[0], self.ei.shape[1]

        def fresh(self):     # a fresh Batch per step: its plan (graph_ptr / edge_ptr from the int64 inputs) is rebuilt every step
            sb = self.sb
            return H.Batch(self.x, self.ei, self.bvec, self.B, y=self.y, max_nodes=sb.max_nodes, max_edges=sb.max_edges,
                           edges_grouped=True, n_small=sb.n_small)

        def make_plan(self):
            """A persistent plan for this batch: `planned()` batches carry it, `plan.rebuild` re-derives it in place."""
            sb = self.sb
            self.plan = H.BatchPlan.build(self.ei, self.bvec, self.N, num_graphs=self.B, mode="blocked", validate=False,
                                          max_nodes=sb.max_nodes, max_edges=sb.max_edges)
            return self.plan

        def planned(self):
            b = self.fresh()
            b._hcg_plan = self.plan
            return b

        def bytes_touched(self):
            return self.x.nbytes + self.ei.nbytes + self.bvec.nbytes + self.y.nbytes

    res = [Resident(i, **extra) for i in range(NB)]
    r0 = res[0]
    N, E, B = r0.N, r0.E, r0.B
    fused_ok = (not args.forward_only) and FusedTrainStep.unsupported_reason(model, r0.fresh()) is None
    # the training step of the reference's loop (utils/utils_model.py:60-68) WITH the Adam update, one trainer (= one set of
    # step buffers) per resident batch; `fwdbwd` stops after the backward (gradients only: secondary figure, batch 0)
    trainers = [FusedTrainStep(model, optimizer_step=True) for _ in res] if fused_ok else []
    fwdbwd = FusedTrainStep(model, optimizer_step=False) if fused_ok else None
    dp = None            # created after the hipGraph capture: no RCCL activity while a stream is capturing

    # ---- parity gate: no timing is accepted unless step 0 of batch 0 matches the oracle
    gate = None
    if not args.no_parity_gate:
        gate = parity_gate(model, r0, fwdbwd, args.forward_only, dev)
        log(f"parity gate: {'PASSED' if gate['passed'] else 'FAILED ' + str(gate['failed'])} (worst {gate['worst']:.2e}, {gate['seconds']:.1f} s)")

    def refuse_timing(why):
        """A failed parity gate: ONE JSON line without a `value` (no timing is accepted), exit code 1."""
        if rank == 0:
            rec = {"metric": "molecular graphs/sec fwd+bwd at 1/2/4/8 MI355X; achieved HBM GB/s", "value": None, "unit": "graphs/s",
                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                   "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                   "config": {"workload": f"{args.config}: not timed -- {why}"}, "parity_gate": gate}
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(rec) + "\n").encode())
        raise SystemExit(1)
    if gate is not None and world == 1 and not gate["passed"]:
        refuse_timing("the parity gate failed")

    def autograd_step(i=0, with_opt=True):
        model.optimizer.zero_grad(set_to_none=True)
        out = model(res[i].fresh())                      # plan build + forward
        loss = torch.sqrt(model.loss(out, res[i].y2))
        loss.backward()
        if dp is not None:
            dp.reduce_gradients()
        if with_opt:
            model.optimizer.step()
        return loss

    def forward_step(i=0):
        with torch.no_grad():
            return model(res[i].fresh())

    def eager_step(i=0, with_opt=True):
        if args.forward_only:
            return forward_step(i)
        if not fused_ok:
            return autograd_step(i, with_opt)
        return (trainers[i] if with_opt else fwdbwd)(res[i].fresh())

    replay = {}

    def capture_all():
        """Everything the step enqueues (plan build, forward, head with loss, backward, slab reduction, Adam) goes
        into hipGraphs, one per resident batch; with N > 1 the RCCL all-reduce stays an eager call between the backward
        graph and the (single-launch) update."""
        if args.forward_only or not fused_ok:
            runs = []
            for i in range(NB):
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        forward_step(i) if args.forward_only else autograd_step(i, False)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                if not args.forward_only:
                    model.optimizer.zero_grad(set_to_none=True)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    forward_step(i) if args.forward_only else autograd_step(i, False)
                grads = [] if args.forward_only else [p.grad for p in model.parameters()]

                def run(with_opt=True, g=g, grads=grads):
                    g.replay()
                    if args.forward_only:
                        return
                    for p, gr in zip(model.parameters(), grads):
                        p.grad = gr
                    if dp is not None:
                        dp.reduce_gradients(grads=grads)
                    if with_opt:
                        model.optimizer.step()
                runs.append(run)
            replay["full"] = runs
            replay["fwdbwd"] = lambda: runs[0](False)
            return
        # every step derives ONE plan (graph_ptr / edge_ptr / validation from the int64 inputs): its own, in front of its forward;
        # with --plan-overlap the NEXT batch's, on a forked branch of the graph beside this step's kernels
        for r in res:
            r.make_plan()
        for i, tr in enumerate(trainers):
            if args.plan_overlap == "none":
                tr.capture(res[i].fresh)
            elif args.plan_overlap == "fork":
                tr.capture(res[i].planned, prefetch=res[(i + 1) % NB].plan.rebuild)
            else:           # "fused": the step's last launch (slab reduction + Adam) also derives the next batch's plan
                tr.capture(res[i].planned, next_plan=res[(i + 1) % NB].plan)
        fwdbwd.capture(r0.fresh)
        replay["full"] = [tr.replay for tr in trainers]
        replay["fwdbwd"] = fwdbwd.replay

    def barrier():
        if world > 1:
            dist.barrier(**({} if rehearsal else {"device_ids": [local_rank]}))

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def timed(k, fn, start=0, runner=None):
        """K calls fn(i), i = start, start + 1, ... (mod NB) -- or runner(start, K), which issues exactly those K steps, several per
        graph launch where it can -- bracketed by barrier + synchronize; MAX over ranks.
        The interpreter's cyclic garbage collector is off inside the bracket: a full collection of a process that has
        imported torch takes ~80 ms -- four hundred steps' worth -- and lands wherever the allocation counters say (seen:
        once in the 200-step burst loop, 0.106 -> 0.50 ms/step).  Nothing of the step is skipped by that."""
        barrier()
        torch.cuda.synchronize()
        gc_was = gc.isenabled()
        gc.disable()
        try:
            t0 = time.perf_counter()
            if runner is not None:
                runner(start, k)
            else:
                for j in range(k):
                    fn((start + j) % NB)
            torch.cuda.synchronize()
            barrier()
            dt = time.perf_counter() - t0
        finally:
            if gc_was:
                gc.enable()
        return max_over_ranks(dt)

    def sustain(seconds, fn, runner=None):
        """The rotation for at least `seconds`: chunks of steps, one synchronize per chunk.  -> (steps, seconds, next index)"""
        barrier()
        torch.cuda.synchronize()
        t0, n, chunk = time.perf_counter(), 0, 256
        while True:
            if runner is not None:
                runner(n % NB, chunk)
            else:
                for j in range(chunk):
                    fn((n + j) % NB)
            n += chunk
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if world > 1:    # every rank must leave the loop after the same chunk
                t = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            if el >= seconds:
                return n, el, n % NB
            if n % (chunk * 16) == 0:
                log(f"sustained rotation: {n} steps, {el:.1f} s")

    log(f"inputs resident: {NB} batches of N={N} E={E} B={B} F={F} D={D}; fused trainer: {fused_ok}")

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
    for j in range(max(args.warmup, NB)):
        eager_step(j % NB)
    torch.cuda.synchronize()
    log("warm-up done")
    launch_counts = count_launching_calls()

    # ---- N > 1: process group, weights replicated, the gradient exchange chosen -- BEFORE the capture, so that a one-shot
    #      exchange (inside the step's last launch) is captured with the step
    rccl_world, exchange_mode, xchg = None, "none", None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        rccl_world = dist.get_world_size()
        if rccl_world != world:
            raise SystemExit(f"bench.py: the process group reports world size {rccl_world}, expected {world}")
        if gate is not None:                                  # every rank gated its own batch 0: one failure refuses the run
            gv = torch.tensor([1 if gate["passed"] else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(gv, op=dist.ReduceOp.MIN)
            gate["passed_on_every_rank"] = bool(int(gv.item()))
            if not gate["passed_on_every_rank"]:
                dist.barrier()
                dist.destroy_process_group()
                refuse_timing("the parity gate failed on at least one rank")
        dp = DataParallelGCN(model, combine=args.combine)    # broadcasts rank-0 weights (in place)
        exchange_mode = "rccl"
        if not fused_ok and probe_ok is not None:             # (the autograd path uses neither graph form)
            probe_ok = bool(probe_ok["oneshot"])
        if fused_ok:
            for tr in trainers + [fwdbwd]:
                dp.attach(tr)
            probe_captured = False
            if probe_ok is not None:       # every rank's child verdicts, combined: one failure anywhere = that form nowhere
                pv = torch.tensor([1 if probe_ok["oneshot"] else 0, 1 if probe_ok["captured"] else 0], device=dev, dtype=torch.int32)
                dist.all_reduce(pv, op=dist.ReduceOp.MIN)
                probe_ok, probe_captured = bool(int(pv[0].item())), bool(int(pv[1].item()))
            if probe_ok is False:
                log("one-shot exchange: its probe in a child process group did not pass on every rank -- RCCL all-reduce stays")
            elif rehearsal and args.exchange == "auto":
                log("rehearsal (ranks share one device): a polling launch of one rank leaves no room for the other rank's conv "
                    "kernels on the same GPU -- the one-shot exchange needs one GPU per rank; RCCL-form exchange (gloo) here")
            elif args.exchange != "rccl":
                # the one-shot xGMI exchange, if it sets up and passes its self test against the process group's own
                # all-reduce on THIS machine (every rank gets the same verdict); else the RCCL collective stays
                from hcatgnet_amd.xgmi import OneShotExchange
                xchg = OneShotExchange(sum(p.numel() for p in model.parameters()))
                ok = xchg.ok and xchg.self_test()
                if ok:
                    # ... and a free-running soak on a throw-away model: 64 real steps back to back, no host synchronisation
                    # in between (what the timed loop does), every rank's verdict combined
                    soak_model = H.make_network("GCN", opt, F).to(dev)
                    soak_dp = DataParallelGCN(soak_model, combine=args.combine)
                    soak = xchg.attach(soak_dp.make_train_step())
                    last = None
                    for j in range(64):
                        last = soak(res[j % NB].fresh())
                    torch.cuda.synchronize()
                    good = int(xchg.err[0].item()) == 0 and bool(torch.isfinite(last).item())
                    verdict = torch.tensor([1 if good else 0], device=dev, dtype=torch.int32)
                    dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
                    ok = bool(int(verdict.item()))
                    xchg.err.zero_()
                    xchg.reset()                    # the soak's step stamps must not meet the real optimiser's
                    del soak, soak_dp, soak_model
                if ok:
                    for tr in trainers:
                        xchg.attach(tr)
                    exchange_mode = "oneshot"
                elif args.exchange == "oneshot":
                    raise SystemExit("bench.py: --exchange oneshot, but the one-shot exchange failed its set-up / self test")
                log(f"one-shot exchange: {'in use' if exchange_mode == 'oneshot' else 'not usable here, RCCL all-reduce stays'}")
            if exchange_mode == "rccl" and probe_captured and not rehearsal and args.exchange == "auto":
                # the RCCL form with the collective and the update recorded INTO the step's graph (passed in the child process
                # group on every rank): the data-parallel step is one graph, and whole rotations one graph launch
                for tr in trainers + [fwdbwd]:
                    tr.capture_exchange = True
                exchange_mode = "rccl-captured"
                log("RCCL all-reduce recorded into the step's hipGraph (its probe in a child process group passed on every rank)")
        for j in range(max(3, NB)):
            eager_step(j % NB)
        torch.cuda.synchronize()
        log(f"{'gloo (rehearsal)' if rehearsal else 'RCCL'} process group up: world {rccl_world}, exchange {exchange_mode}")

    launch_mode, graph_err = "eager", None
    if not args.no_graph:
        try:
            capture_all()     # (RCCL form: the graph ends before the collective; one-shot form: the whole step is one graph)
            launch_mode = "hipgraph"
            log("step captured into hipGraphs")
        except Exception as exc:  # report, never hide: the eager number stands
            replay.clear()
            graph_err = f"{type(exc).__name__}: {exc}"
            log(f"graph capture failed, keeping eager launches: {graph_err}")

    # ---- a window: the NB steps of one rotation as ONE hipGraph (train.StepWindow) -- saves the bubble between two graph
    #      launches.  Only where a whole step is one graph (one GPU, or the one-shot exchange inside the update launch) and every
    #      step derives the next batch's plan itself (--plan-overlap fused).
    window, window_err, tail_window, tail_n = None, None, None, 0
    if ("full" in replay and fused_ok and not args.forward_only and args.plan_overlap == "fused" and not args.no_window
            and exchange_mode in ("none", "oneshot", "rccl-captured") and NB > 1):
        try:
            from hcatgnet_amd.train import StepWindow
            window = StepWindow(trainers, [r.planned for r in res])
            # the timed K steps = ONE shorter window for K mod NB steps (the LAST batches of a rotation, issued first: a short graph
            # reaches the GPU sooner after the bracket's synchronize, and the launch of the long one behind it is hidden) + whole rotations
            tail_n = args.steps % NB
            if tail_n >= 2:
                tail_window = StepWindow(trainers[NB - tail_n:], [r.planned for r in res[NB - tail_n:]])
            log(f"window captured: {NB} steps per graph launch" + (f" (+ one of {tail_n} for the remainder of {args.steps})" if tail_window else ""))
        except Exception as exc:          # report, never hide: the per-step graphs stand
            window, tail_window, window_err = None, None, f"{type(exc).__name__}: {exc}"
            log(f"window capture failed, one graph per step stays: {window_err}")

    def window_runner(start, k):
        """Exactly k consecutive steps of the rotation from index `start`: whole rotations as one window launch, the rest as
        single-step graphs (the same captured launches either way)."""
        j = 0
        while j < k:
            i = (start + j) % NB
            if i == 0 and k - j >= NB:
                window.replay()
                j += NB
            elif i == NB - tail_n and tail_window is not None and k - j >= tail_n:
                tail_window.replay()
                j += tail_n
            else:
                replay["full"][i]()
                j += 1

    # set-up is over: collect once and move everything alive now (torch, the model, 17 trainers, the captured graphs) out of
    # the collector's sight, so that a later collection -- the sustained run keeps the collector on -- walks little
    t_gc = time.perf_counter()
    gc.collect()
    gc.freeze()
    log(f"gc.collect + gc.freeze after set-up: {(time.perf_counter() - t_gc) * 1e3:.0f} ms")

    # kernel-level roofline: HIP events around the dominant entry point, inside a timed eager loop
    mid = r0.sb.max_nodes > 32
    # 128-wide layers over large graphs run through csrc/tall.hip (one entry point = up to three launches)
    from hcatgnet_amd import functional as _HF
    tall = mid and bool(lib.hcg_tall_supported(D, D, r0.sb.max_nodes, r0.sb.max_edges)) and (D != 64 or N >= _HF.TALL_MIN_NODES_D64)
    fam = "hcg_tall" if tall else "hcg_mid"
    entry = args.roofline_entry or ((f"{fam}_layer_fwd" if mid else "hcg_fused_stack2_fwd") if args.forward_only
                                    else (f"{fam}_layer_bwd" if mid else "hcg_fused_layer_bwd"))
    timer = EntryTimer(lib, entry)
    timer.install()
    timer.enabled = True
    dt = timed(args.steps, eager_step)
    timer.enabled = False
    k_ms, k_calls = timer.mean_ms()
    timer.uninstall()
    log(f"timed (with kernel events): {dt / args.steps * 1e3:.3f} ms/step")
    dt_eager = timed(args.steps, eager_step)
    log(f"timed eager (full step, rotating batches): {dt_eager / args.steps * 1e3:.3f} ms/step")
    # distribution of single steps (median, p10 / p90): HIP events around every step of one more pass
    evs = []
    for j in range(args.steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eager_step(j % NB)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    per_step = sorted(a.elapsed_time(b) for a, b in evs)
    pct = {f"p{q}": per_step[min(len(per_step) - 1, int(len(per_step) * q / 100))] for q in (10, 50, 90)}

    # ---- burst figure (round 1's headline): ONE batch replayed in place from an idle chip
    dt_fb, dt_graph_burst = None, None
    if "full" in replay:
        for _ in range(max(3, args.warmup // 2)):
            replay["full"][0]()
        dt_graph_burst = timed(args.steps, lambda i: replay["full"][0]())
        log(f"burst, hipGraph replay of ONE batch in place: {dt_graph_burst / args.steps * 1e3:.3f} ms/step")
        if not args.forward_only:
            for _ in range(3):
                replay["fwdbwd"]()
            dt_fb = timed(args.steps, lambda i: replay["fwdbwd"]())
            log(f"burst, hipGraph replay (fwd+bwd only, no update): {dt_fb / args.steps * 1e3:.3f} ms/step")
    elif not args.forward_only:
        dt_fb = timed(args.steps, lambda i: eager_step(0, False))
    dt_eager_burst = timed(args.steps, lambda i: eager_step(0))

    # ---- the headline: sustained rotation over distinct batches, then EXACTLY K steps right behind it
    use_graph = "full" in replay
    if use_graph:      # the no-autograd step issues 6 launches from a host loop that can run ahead of the GPU: take the faster
        tg = timed(max(args.steps, 4 * NB), lambda i: replay["full"][i]())
        te = timed(max(args.steps, 4 * NB), eager_step)
        use_graph = tg <= te
        log(f"rotation probe: hipGraph {tg / max(args.steps, 4 * NB) * 1e3:.4f} vs eager {te / max(args.steps, 4 * NB) * 1e3:.4f} ms/step")
    launch_mode = "hipgraph" if use_graph else "eager"
    step_fn = (lambda i: replay["full"][i]()) if use_graph else eager_step
    runner, steps_per_graph, per_step_graph_ms = None, (1 if use_graph else None), None
    if window is not None:        # (also when the eager launches edged out the per-step graphs above)
        kp = max(args.steps, 4 * NB) // NB * NB
        timed(kp, None, runner=window_runner)
        tw, tg1, te1 = (timed(kp, None, runner=window_runner), timed(kp, lambda i: replay["full"][i]()),
                        timed(kp, eager_step))
        log(f"rotation probe: {NB} steps per graph launch {tw / kp * 1e3:.4f} vs one graph per step {tg1 / kp * 1e3:.4f} "
            f"vs eager {te1 / kp * 1e3:.4f} ms/step")
        per_step_graph_ms = tg1 / kp * 1e3
        if tw <= min(tg1, te1):
            use_graph, launch_mode = True, "hipgraph"
            step_fn = lambda i: replay["full"][i]()
            runner, steps_per_graph = window_runner, NB
    sus_steps, sus_s, nxt = sustain(args.sustain, step_fn, runner) if args.sustain > 0 else (0, 0.0, 0)
    if runner is not None and tail_window is not None and nxt != NB - tail_n:
        # (untimed) walk the rotation on to where the remainder window starts: the K timed steps are then that window + whole rotations
        runner(nxt, (NB - tail_n - nxt) % NB)
        nxt = NB - tail_n
    dt_best = timed(args.steps, step_fn, start=nxt, runner=runner)
    log(f"sustained {sus_s:.2f} s / {sus_steps} steps = {sus_s / max(sus_steps, 1) * 1e3:.4f} ms/step; "
        f"timed {args.steps} steps right behind: {dt_best / args.steps * 1e3:.4f} ms/step ({launch_mode}"
        f"{', ' + str(steps_per_graph) + ' steps per graph launch' if steps_per_graph and steps_per_graph > 1 else ''})")
    for tr in trainers:
        tr.check_health()
    if xchg is not None and exchange_mode == "oneshot":
        xchg.check()

    bd = algbytes.breakdown(N, E, B, F, D, opt.n_convolutions)
    step_bytes = sum(v for k, v in bd.items() if not args.forward_only or k.endswith("_fwd") or k == "csr_build")
    # algorithmic bytes of ONE launch of the timed entry point, averaged over its launches in a step
    # (layer 2 backward also carries the pool backward it fuses; layer 2 forward the pool forward)
    entry_bytes = {"hcg_gcn_layer_bwd": (bd["conv1_bwd"] + bd["conv2_bwd"]) / 2.0,
                   "hcg_gcn_layer_fwd": (bd["conv1_fwd"] + bd["conv2_fwd"]) / 2.0,
                   "hcg_fused_layer_bwd": (bd["conv1_bwd"] + bd["conv2_bwd"] + bd["pool_bwd"]) / 2.0,
                   "hcg_mid_layer_bwd": (bd["conv1_bwd"] + bd["conv2_bwd"] + bd["pool_bwd"]) / 2.0,
                   "hcg_mid_layer_fwd": (bd["conv1_fwd"] + bd["conv2_fwd"] + bd["pool_fwd"]) / 2.0,
                   "hcg_tall_layer_bwd": (bd["conv1_bwd"] + bd["conv2_bwd"] + bd["pool_bwd"]) / 2.0,
                   "hcg_tall_layer_fwd": (bd["conv1_fwd"] + bd["conv2_fwd"] + bd["pool_fwd"]) / 2.0,
                   "hcg_fused_layer_fwd": (bd["conv1_fwd"] + bd["conv2_fwd"] + bd["pool_fwd"]) / 2.0,
                   "hcg_fused_stack2_fwd": bd["conv1_fwd"] + bd["conv2_fwd"] + bd["pool_fwd"]}.get(entry, float("nan"))
    achieved = entry_bytes / (k_ms * 1e-3) / 1e9 if k_ms == k_ms and k_ms > 0 else None
    # HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, gfx950
    # correction of MI355X_MICROARCH.md; tools/pmc_traffic.py) -- only valid for the config they were taken on
    traffic, traffic_src = None, None
    tpath = os.path.join(REPO, "profiles", "traffic_latest.json")
    if cfg_name in ("C2", "C3", "C4", "C5", "REAL") and not extra and args.num_graphs is None and os.path.isfile(tpath):
        try:
            tj = json.load(open(tpath))
            if "C3" in tj or "C5" in tj or "REAL" in tj:      # per-config sections; (round-1 files: flat, C3 only)
                tj = tj.get(cfg_name if cfg_name in ("C5", "REAL") else "C3", {})
            elif cfg_name in ("C5", "REAL"):
                tj = {}
            prefix = {"hcg_fused_layer_bwd": "k_fused_layer_bwd", "hcg_mid_layer_bwd": "k_mid_layer_bwd",
                      "hcg_fused_stack2_fwd": "k_fused_layer_fwd", "hcg_mid_layer_fwd": "k_mid_layer_fwd"}.get(entry)
            vals = [v["hbm_bytes"] for k, v in tj.items() if prefix and k.startswith(prefix)]
            traffic = sum(vals) / len(vals) if vals else None
            if traffic is not None and entry.startswith("hcg_mid_") and D > 64:
                traffic *= D // 64            # one call of the entry point = one kernel launch per 64-column half
            if entry.startswith("hcg_tall_layer_"):
                # one call = several launches (bwd: k_seg_bwd + k_tall_dw + k_tall_mm; fwd: k_split_weight + k_seg_fwd):
                # bytes of every launch of those kernels in the profiled run / calls of the entry point in it (2 per step)
                bwd = entry.endswith("bwd")
                fams = ("k_seg_bwd", "k_gseg_bwd", "k_tall_dw", "k_tall_mm") if bwd else ("k_seg_fwd", "k_split_weight", "k_mid_layer_fwd")
                tot = sum(v["hbm_bytes"] * v["launches"] for k, v in tj.items() if k.startswith(fams))
                calls = sum(v["launches"] for k, v in tj.items()
                            if k.startswith(("k_seg_bwd", "k_gseg_bwd") if bwd else ("k_seg_fwd", "k_mid_layer_fwd")))
                traffic = tot / calls if calls else None
            traffic_src = "profiles/traffic_latest.json: builder-run rocprofv3 PMC passes of this command, not measured in this run"
        except (OSError, ValueError, KeyError, AttributeError):
            traffic = None

    # ---- secondary: the ragged variant of the same workload (n_g ~ U{24..36}, "report both")
    ragged = None
    if world == 1 and fused_ok and not extra and cfg_name in ("C2", "C3", "C4") and not args.no_ragged and not args.forward_only:
        try:
            rr = [Resident(100 + i, **ragged_cfg) for i in range(NB)]
            rtr = [FusedTrainStep(model, optimizer_step=True) for _ in rr]
            for r in rr:
                r.make_plan()
            for i, (tr, r) in enumerate(zip(rtr, rr)):
                for _ in range(2):
                    tr(r.fresh())
                # the headline's pipeline: the step's last launch also derives the NEXT batch's plan (--plan-overlap fused)
                if args.plan_overlap == "fused":
                    tr.capture(r.planned, next_plan=rr[(i + 1) % NB].plan)
                else:
                    tr.capture(r.fresh)
            rfn = lambda i: rtr[i].replay()
            rrun = None
            if runner is not None and args.plan_overlap == "fused":         # the headline's launch form
                from hcatgnet_amd.train import StepWindow
                rwin = StepWindow(rtr, [r.planned for r in rr])

                def rrun(start, k):
                    j = 0
                    while j < k:
                        i = (start + j) % NB
                        if i == 0 and k - j >= NB:
                            rwin.replay()
                            j += NB
                        else:
                            rtr[i].replay()
                            j += 1
            timed(4 * NB, rfn, runner=rrun)
            rdt = timed(max(args.steps, 200), rfn, runner=rrun)
            k = max(args.steps, 200)
            gpt = lib.hcg_fused_graphs_per_tile(F, D, rr[0].sb.max_nodes)
            ragged = {"value": rr[0].B * k / rdt, "unit": "graphs/s", "ms_per_step": rdt / k * 1e3, "steps": k,
                      "graphs": rr[0].B, "nodes": rr[0].N, "edges": rr[0].E, "max_nodes": rr[0].sb.max_nodes,
                      "kernel_family": "small-graph tiles" if gpt > 0 else ("size-grouped batch: tiles for graphs <= 32 nodes + one graph per wave"
                                                                           if rr[0].sb.n_small else "one graph per wave / workgroup"),
                      "note": f"n_g ~ U{{24..36}}, {NB} distinct batches round-robin, hipGraph replay"
                              f"{' (' + str(NB) + ' steps per graph launch)' if rrun is not None else ''}"}
            log(f"ragged variant: {rdt / k * 1e3:.4f} ms/step")
            del rr, rtr
        except Exception as exc:
            ragged = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        ms_step = dt_best / args.steps * 1e3
        touched = NB * (r0.bytes_touched() + (sum(t.nbytes for t in trainers[0]._bufs["cap"]["acts"] + trainers[0]._bufs["cap"]["dacts"])
                                              if fused_ok else 0))
        fwd_only_note = "plan/gcn_norm build + forward (conv stack, pool, readout) only" if args.forward_only else \
            (f"full training step: per-step plan/gcn_norm build, forward, sqrt(MSE) loss, backward, "
             f"{('one-shot xGMI exchange inside the update launch (' if exchange_mode == 'oneshot' else ('RCCL all-reduce recorded in the step graph (' if exchange_mode == 'rccl-captured' else 'RCCL all-reduce (')) + args.combine + '), ' if world > 1 else ''}Adam update")
        rec = {
            "metric": "molecular graphs/sec fwd+bwd at 1/2/4/8 MI355X; achieved HBM GB/s" if not args.forward_only
                      else "molecular graphs/sec FORWARD ONLY (configs[1]; not the headline metric)",
            "value": world * B * args.steps / dt_best, "unit": "graphs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'C2' if args.forward_only and cfg_name == 'C3' else args.config}: {B} synthetic graphs/GPU x "
                                   f"{N / B:.0f} atoms x {E / B:.0f} directed edges x "
                                   f"{F}-d features, {opt.n_convolutions}xGCNConv({D}) + [max,mean] pool + readout; "
                                   f"{fwd_only_note}; {NB} distinct batches round-robin; launch={launch_mode}"
                                   f"{' (' + str(steps_per_graph) + ' consecutive steps per graph launch)' if steps_per_graph and steps_per_graph > 1 else ''}"
                                   f"{'' if (launch_mode != 'hipgraph' or args.plan_overlap == 'none' or not fused_ok) else ', plan build of the NEXT batch ' + ('inside the last launch of the step' if args.plan_overlap == 'fused' else 'on a forked graph branch')}",
                       "graphs_per_gpu": B, "nodes": N, "edges": E, "feat": F, "hidden": D,
                       "parallelism": f"dp{world} (batch-of-graphs, {'one-shot xGMI exchange' if exchange_mode == 'oneshot' else 'RCCL all-reduce'} "
                                      f"of {sum(p.numel() for p in model.parameters())} fp32 grads)"},
            "parity_gate": gate,
            "rccl_world": rccl_world, "exchange": exchange_mode,
            # the child-process trials: of the one-shot exchange, and of the RCCL collective recorded into the step's graph
            "exchange_probe": (None if probe_ok is None else ("passed" if probe_ok else "failed")),
            "captured_collective_probe": (None if probe_ok is None or world == 1 else ("passed" if probe_captured else "failed")),
            "distinct_batches": NB, "bytes_touched_between_reuse": touched,
            "sustained_s": sus_s, "sustained": {"steps": sus_steps, "seconds": sus_s, "ms_per_step": sus_s / max(sus_steps, 1) * 1e3,
                                                "value": world * B * sus_steps / sus_s if sus_s > 0 else None,
                                                "note": "the same rotation, run right before the timed steps (one synchronize per 256 steps)"},
            "burst": {"note": "ONE batch replayed in place from an idle chip (cache resident): round 1's headline figure",
                      "hipgraph_ms_per_step": dt_graph_burst / args.steps * 1e3 if dt_graph_burst else None,
                      "eager_ms_per_step": dt_eager_burst / args.steps * 1e3,
                      "value": world * B * args.steps / min(x for x in (dt_graph_burst, dt_eager_burst) if x)},
            "ragged": ragged,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": entry, "kernel_ms": k_ms, "kernel_launches_timed": k_calls,
                         "algorithmic_bytes_per_launch": entry_bytes},
            "step_roofline": {"algorithmic_bytes_per_step": step_bytes, "achieved": step_bytes / (ms_step * 1e-3) / 1e9,
                              "unit": "GB/s", "frac": step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "breakdown": bd},
            "fwd_bwd_only": None if dt_fb is None else {"value": world * B * args.steps / dt_fb, "unit": "graphs/s",
                                                        "ms_per_step": dt_fb / args.steps * 1e3,
                                                        "note": "burst form, same step without the Adam update (gradients only)"},
            "optimizer": type(model.optimizer).__name__ + "(lr=0.01, eps=1e-9), inside the timed step",
            "step_path": "forward only" if args.forward_only else ("FusedTrainStep (no autograd)" if fused_ok else "autograd"),
            "library_launching_calls_per_step": {"total": sum(launch_counts.values()), "by_entry_point": launch_counts},
            "ms_per_step_with_kernel_events": dt / args.steps * 1e3,
            "eager_step_ms_percentiles_hip_events": pct,
            "launch": launch_mode, "steps_per_graph_launch": steps_per_graph, "one_graph_per_step_ms": per_step_graph_ms,
            "window_error": window_err, "eager_ms_per_step": dt_eager / args.steps * 1e3,
            "graph_capture_error": graph_err,
        }
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(cfg_name, args.num_graphs, args.cpu_steps)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(rec) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
