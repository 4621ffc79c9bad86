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


PROBE_BUDGET_S = 90.0     # the child-process trial of the two graph forms; + in-process self test and soak (bounded polls,
                          # < 30 s): <= 120 s of a multi-GPU run go to deciding the exchange form


def probe_port(world, rank):
    """A port for the child process groups: rank 0 binds port 0 (a free one), the others read it from the launcher's store
    (torchrun hosts a TCPStore at MASTER_ADDR:MASTER_PORT).  Without such a store: MASTER_PORT + 1."""
    fallback = (int(os.environ.get("MASTER_PORT", "29500")) - 1024 + 1) % (65535 - 1024) + 1024
    try:
        import socket
        from datetime import timedelta
        from torch.distributed import TCPStore
        store = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"]), world, False,
                         timedelta(seconds=20))
        if rank == 0:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            store.set("hcg_probe_port", str(port))
            return port
        return int(store.get("hcg_probe_port").decode())
    except Exception as exc:      # noqa: BLE001  (no agent store: every rank derives the same fallback)
        log(f"probe port: no launcher store ({type(exc).__name__}); using MASTER_PORT + 1")
        return fallback


def isolated_probe(world, rank, rehearsal, combine, timeout_s=PROBE_BUDGET_S):
    """Both data-parallel graph forms tried once in a CHILD process per rank (`python -m hcatgnet_amd.xgmi`: its own process
    group; the one-shot exchange with set-up, self test and 64 free-running real steps; then the RCCL collective recorded
    into the step's hipGraph, 32 replays) before THIS process touches the GPU.  A failure no `try` can catch -- a GPU memory
    fault on a peer mapping aborts the process, a wedged launch or collective never returns -- then ends the child, not the
    bench.  -> {"oneshot": bool, "captured": bool, "exit": code, "seconds": s, "note": last line of the child's log}: a
    phase the child never finished counts as failed; the plain RCCL form needs neither.  No GPU call here."""
    import tempfile
    env = dict(os.environ)
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)          # the child group's rank 0 hosts its own store ...
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(probe_port(world, rank))      # ... on a port rank 0 found free
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = REPO + os.pathsep + env.get("PYTHONPATH", "")
    fd, out_path = tempfile.mkstemp(prefix="hcg_probe_", suffix=".json")
    os.close(fd)
    fd, log_path = tempfile.mkstemp(prefix="hcg_probe_", suffix=".log")
    os.close(fd)
    env["HCG_PROBE_OUT"] = out_path
    cmd = [sys.executable, "-m", "hcatgnet_amd.xgmi", "--combine", combine]
    if rehearsal:
        cmd += ["--one-device", "--soak-steps", "0"]      # (two ranks on one device starve each other in a free-running soak)
    t0 = time.perf_counter()
    verdicts = {"oneshot": False, "captured": False, "exit": None, "seconds": 0.0, "note": None}
    log(f"data-parallel probe: child process group on port {env['MASTER_PORT']}, budget {timeout_s:.0f} s")
    try:
        logf = open(log_path, "w")
        proc = subprocess.Popen(cmd, env=env, stdout=logf, stderr=subprocess.STDOUT, start_new_session=True)
    except OSError as exc:
        verdicts["note"] = f"could not start: {exc}"
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
    logf.close()
    try:
        got = json.load(open(out_path))
        verdicts.update({k: bool(got.get(k, False)) for k in ("oneshot", "captured")})
    except (OSError, ValueError):
        pass
    try:
        lines = [ln.strip() for ln in open(log_path, errors="replace").read().splitlines() if ln.strip()]
        for ln in lines[-12:]:
            print("[probe child] " + ln, file=sys.stderr, flush=True)
        verdicts["note"] = ("killed at the time limit; " if rc == -9 else "") + (lines[-1][:200] if lines else "no output")
    except OSError:
        pass
    for path in (out_path, log_path):
        try:
            os.unlink(path)
        except OSError:
            pass
    verdicts["exit"], verdicts["seconds"] = rc, time.perf_counter() - t0
    log(f"data-parallel probe (child process, world {world}): exit {rc} after {verdicts['seconds']:.1f} s -> "
        f"oneshot {verdicts['oneshot']}, captured {verdicts['captured']}")
    return verdicts


class EntryTimer:
    """HIP-event timing of ONE C-ABI entry point on the stream it launches on (torch's current
    stream: the library only enqueues on the stream it is handed)."""

    # entry points that launch the SAME kernel family as the named one (none since the forward / backward forms of the
    # small-graph tiles are one entry point each)
    SAME_KERNEL = {}

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
    rehearsal = os.environ.get("HCG_BENCH_REHEARSAL") == "1"
    # N > 1, --exchange auto: the two graph forms of the exchange have to survive a sacrificial child process group first
    # (before anything here touches the GPU); the ranks combine their verdicts once the real process group is up
    probe = None
    if (world > 1 and args.exchange == "auto" and not args.forward_only
            and (not rehearsal or os.environ.get("HCG_PROBE_IN_REHEARSAL") == "1")):
        import torch  # noqa: F401  (no GPU call: pages the libraries in, so that the child's import does not eat its time limit)
        probe = isolated_probe(world, rank, rehearsal, args.combine, timeout_s=PROBE_BUDGET_S)

    import torch
    import torch.distributed as dist
    # rehearsal of the N > 1 control flow on a ONE-GPU box (tools/rehearse_multi_rank.sh): every rank on device 0,
    # gloo instead of RCCL (RCCL refuses two ranks on one device).  Never set by the driver; the numbers mean nothing.
    if rehearsal:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); there is no CPU fallback")
    if not rehearsal and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import hcatgnet_amd as H
    from hcatgnet_amd import _lib, algbytes, launch, synth
    from hcatgnet_amd.ddp import DataParallelGCN
    from hcatgnet_amd.train import FusedTrainStep
    if os.environ.get("HCG_LIB"):      # a library variant for A/B measurements (tools/build_variants.sh)
        _lib.LIB_PATH = os.path.abspath(os.environ["HCG_LIB"])
    lib = _lib.load()
    # development A/B switches (tools/ab_env.sh): measurement tooling only, the product has no environment switches
    for env, attr in (("HCG_NO_POOLBITS", "POOLBITS"), ("HCG_NO_PREMASK", "PREMASK"), ("HCG_NO_OVERLAP", "OVERLAP_GROUPS"),
                      ("HCG_NO_HEAD_IN_FORWARD", "HEAD_IN_FORWARD"), ("HCG_NO_TALL_PREMASK", "TALL_PREMASK"), ("HCG_NO_XAGG", "XAGG"), ("HCG_NO_XAGG_MID", "XAGG_MID")):
        if os.environ.get(env) == "1":
            setattr(FusedTrainStep, attr, False)
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
    if os.environ.get("HCG_TALL_MIN_NODES"):          # development A/B: where the 64-wide backward moves to the wide-layer route
        from hcatgnet_amd import functional as _HF
        _HF.TALL_MIN_NODES_D64 = int(os.environ["HCG_TALL_MIN_NODES"])
    if os.environ.get("HCG_FAMILY_MID") == "1":       # development A/B: keep every layer on the one-graph-per-workgroup kernels
        for cv in [model.conv1] + list(model.conv_layers):
            cv.family = "mid"

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
    dp = None

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

    def barrier():
        if world > 1:
            dist.barrier(**({} if rehearsal else {"device_ids": [local_rank]}))

    def max_over_ranks(dt):
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def timed(k, runner, start=0):
        """runner(start, k) issues EXACTLY k steps; bracketed by barrier + synchronize on the way in and synchronize on the
        way out (the closing barrier sits OUTSIDE the bracket: inside it cost ~2 % of a 20-step bracket at N = 8); MAX over
        ranks.  The interpreter's cyclic garbage collector is off inside the bracket: a full collection of a process that
        has imported torch takes ~80 ms -- four hundred steps' worth -- and lands wherever the allocation counters say (seen:
        once in a 200-step burst loop, 0.106 -> 0.50 ms/step).  Nothing of the step is skipped by that."""
        barrier()
        torch.cuda.synchronize()
        gc_was = gc.isenabled()
        gc.disable()
        try:
            t0 = time.perf_counter()
            runner(start, k)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        finally:
            if gc_was:
                gc.enable()
        barrier()
        return max_over_ranks(dt)

    def sustain(seconds, runner):
        """The rotation for at least `seconds`: chunks of steps, one synchronize per chunk.  -> (steps, seconds, next index)"""
        barrier()
        torch.cuda.synchronize()
        t0, n, chunk = time.perf_counter(), 0, 256
        while True:
            runner(n % NB, chunk)
            n += chunk
            torch.cuda.synchronize()
            el = max_over_ranks(time.perf_counter() - t0)      # every rank must leave the loop after the same chunk
            if el >= seconds:
                return n, el, n % NB
            if n % (chunk * 16) == 0:
                log(f"sustained rotation: {n} steps, {el:.1f} s")

    log(f"inputs resident: {NB} batches of N={N} E={E} B={B} F={F} D={D}; fused trainer: {fused_ok}")

    # ---- the autograd / forward-only paths (models or flags outside the fused step): eager or one graph per step
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

    class PlainRotation:
        """The same interface as launch.Rotation for steps that are not a FusedTrainStep (forward only / autograd path)."""
        def __init__(self):
            self.graph_runs, self.errors, self.window, self.graphs, self.NB = None, {}, None, False, NB

        def eager(self, i):
            return forward_step(i) if args.forward_only else autograd_step(i)

        def capture(self):
            runs = []
            try:
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

                    def run(g=g, grads=grads):
                        g.replay()
                        if args.forward_only:
                            return
                        for p, gr in zip(model.parameters(), grads):
                            p.grad = gr
                        if dp is not None:
                            dp.reduce_gradients(grads=grads)
                        model.optimizer.step()
                    runs.append(run)
                err = None
            except Exception as exc:      # noqa: BLE001
                err = f"{type(exc).__name__}: {exc}"
            self.graphs = launch.all_ranks_agree(err is None, dev)
            if self.graphs:
                self.graph_runs = runs
            else:
                self.errors["graph"] = err or "capture failed on another rank"
            return self.graphs

        def build_windows(self, k):
            return False

        def forms(self):
            return ["eager"] + (["graph"] if self.graphs else [])

        def run(self, form, start, k):
            for j in range(k):
                i = (start + j) % NB
                self.eager(i) if form == "eager" else self.graph_runs[i]()

        pick = launch.Rotation.pick

        def losses_finite(self):
            return True

    if fused_ok:
        for r in res:
            r.make_plan()
        rot = launch.Rotation(trainers, [r.fresh for r in res], [r.planned for r in res], [r.plan for r in res],
                              plan_overlap=args.plan_overlap)
    else:
        rot = PlainRotation()

    def count_launching_calls():
        """Library entry points that enqueue kernels during ONE eager step (launches per step)."""
        skip = ("_bytes", "_job", "_supported", "_per_tile", "_blocks", "hcg_version", "hcg_error_string", "hcg_reduce_job_append")
        names = [n for n in _lib.SIGNATURES if not n.endswith(skip) and n not in skip]
        counts, origs = {}, {}
        for n in names:
            origs[n] = getattr(lib, n)

            def wrap(*a, _n=n):
                counts[_n] = counts.get(_n, 0) + 1
                return origs[_n](*a)
            setattr(lib, n, wrap)
        try:
            rot.eager(0)
            torch.cuda.synchronize()
        finally:
            for n in names:
                setattr(lib, n, origs[n])
        return counts
    for j in range(max(args.warmup, NB)):
        rot.eager(j % NB)
    torch.cuda.synchronize()
    log("warm-up done")
    launch_counts = count_launching_calls()

    # ---- N > 1: process group, weights replicated
    rccl_world, xchg = None, None
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
            gate["passed_on_every_rank"] = launch.all_ranks_agree(gate["passed"], dev)
            if not gate["passed_on_every_rank"]:
                dist.barrier()
                dist.destroy_process_group()
                refuse_timing("the parity gate failed on at least one rank")
        dp = DataParallelGCN(model, combine=args.combine)    # broadcasts rank-0 weights (in place)
        log(f"{'gloo (rehearsal)' if rehearsal else 'RCCL'} process group up: world {rccl_world}")

    gc_frozen = [False]

    def freeze_gc():
        # set-up is over: collect once and move everything alive now (torch, the model, the trainers, the captured graphs) out
        # of the collector's sight, so that a later collection -- the sustained run keeps the collector on -- walks little
        if not gc_frozen[0]:
            t_gc = time.perf_counter()
            gc.collect()
            gc.freeze()
            gc_frozen[0] = True
            log(f"gc.collect + gc.freeze after set-up: {(time.perf_counter() - t_gc) * 1e3:.0f} ms")

    def measure_form(sustain_s):
        """Capture (graphs, windows) for the trainers as they are configured now, pick the fastest launch form, run the
        rotation for `sustain_s` and time EXACTLY K steps right behind it.  -> dict (ms_per_step, ...)."""
        if not args.no_graph:
            if rot.capture() and not args.no_window:
                rot.build_windows(args.steps)
        freeze_gc()
        form, per_form = rot.pick(timed, args.steps)
        runner = lambda s, n: rot.run(form, s, n)
        sus_steps, sus_s, nxt = sustain(sustain_s, runner) if sustain_s > 0 else (0, 0.0, 0)
        tail_n = getattr(rot, "tail_n", 0)
        if form == "window" and tail_n and nxt != NB - tail_n:
            # (untimed) walk the rotation on to where the remainder window starts: the K timed steps are then that window + whole rotations
            runner(nxt, (NB - tail_n - nxt) % NB)
            nxt = NB - tail_n
        dt = timed(args.steps, runner, start=nxt)
        return {"launch": form, "ms_per_step": dt / args.steps * 1e3, "launch_forms_ms": per_form, "seconds": dt,
                "sustained": {"steps": sus_steps, "seconds": sus_s, "ms_per_step": sus_s / max(sus_steps, 1) * 1e3},
                "steps_per_graph_launch": NB if form == "window" else (1 if form == "graph" else None),
                "errors": dict(rot.errors)}

    # ---- kernel-level roofline: HIP events around the dominant entry point, inside a timed eager loop (before any graph)
    mid = r0.sb.max_nodes > 32
    # 128-wide layers over large graphs run through csrc/tall.hip (one entry point = up to three launches)
    from hcatgnet_amd import functional as _HF
    tall = mid and bool(lib.hcg_tall_supported(D, D, r0.sb.max_nodes, r0.sb.max_edges)) and (D != 64 or N >= _HF.TALL_MIN_NODES_D64)
    fam = "hcg_tall" if tall else "hcg_mid"
    entry = args.roofline_entry or ((f"{fam}_layer_fwd" if mid else "hcg_fused_forward") if args.forward_only
                                    else (f"{fam}_layer_bwd" if mid else "hcg_fused_layer_bwd"))
    if world > 1 and fused_ok:
        launch.set_exchange_form(trainers + [fwdbwd], dp, "rccl")
        for j in range(max(3, NB)):
            rot.eager(j % NB)
        torch.cuda.synchronize()
    eager_runner = lambda s, n: rot.run("eager", s, n)
    timer = EntryTimer(lib, entry)
    timer.install()
    timer.enabled = True
    dt = timed(args.steps, eager_runner)
    timer.enabled = False
    k_ms, k_calls = timer.mean_ms()
    timer.uninstall()
    log(f"timed (with kernel events): {dt / args.steps * 1e3:.3f} ms/step")
    dt_eager = timed(args.steps, eager_runner)
    log(f"timed eager (full step, rotating batches): {dt_eager / args.steps * 1e3:.3f} ms/step")
    # distribution of single steps (median, p10 / p90): HIP events around every step of one more pass
    evs = []
    for j in range(args.steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rot.eager(j % NB)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    per_step = sorted(a.elapsed_time(b) for a, b in evs)
    pct = {f"p{q}": per_step[min(len(per_step) - 1, int(len(per_step) * q / 100))] for q in (10, 50, 90)}

    # ---- the headline.  One GPU: one form.  N > 1: EVERY exchange form that is usable here is measured, the plain RCCL form
    #      FIRST (so that whatever happens to a later form, a healthy number is banked), and `value` is the best healthy one
    exchange_forms, fallback_reason, probe_note = {}, None, None
    if world == 1 or not fused_ok:
        best = measure_form(args.sustain)
        exchange_mode = "none" if world == 1 else "rccl"
    else:
        verdicts = {"oneshot": False, "captured": False}
        if probe is not None:       # every rank's child verdicts, combined: one failure anywhere = that form nowhere
            verdicts = {k: launch.all_ranks_agree(bool(probe.get(k)), dev) for k in verdicts}
            probe_note = probe.get("note")
        candidates = ["rccl"]
        if args.exchange == "auto" and not rehearsal:
            candidates += [f for f, k in (("rccl-captured", "captured"), ("oneshot", "oneshot")) if verdicts[k]]
        elif args.exchange == "oneshot":
            candidates += ["oneshot"]
        short = min(args.sustain, 1.5)
        for form in candidates:
            t_form = time.perf_counter()
            try:
                if form == "oneshot":
                    from hcatgnet_amd.xgmi import OneShotExchange
                    xchg = OneShotExchange(sum(p.numel() for p in model.parameters()))
                    ok = launch.all_ranks_agree(bool(xchg.ok and xchg.self_test()), dev)
                    if ok:      # a free-running soak on a throw-away model: 64 real steps back to back, no host synchronisation
                        soak_model = H.make_network("GCN", opt, F).to(dev)
                        soak_dp = DataParallelGCN(soak_model, combine=args.combine)
                        soak = xchg.attach(soak_dp.make_train_step())
                        last = None
                        for j in range(64):
                            last = soak(res[j % NB].fresh())
                        torch.cuda.synchronize()
                        ok = launch.all_ranks_agree(int(xchg.err[0].item()) == 0 and bool(torch.isfinite(last).item()), dev)
                        xchg.err.zero_()
                        xchg.reset()                    # the soak's step stamps must not meet the real optimiser's
                        del soak, soak_dp, soak_model
                    if not ok:
                        exchange_forms[form] = {"healthy": False, "reason": "set-up / self test / soak failed on this machine"}
                        continue
                launch.set_exchange_form(trainers, dp, form, xchg)
                for j in range(max(3, NB)):
                    rot.eager(j % NB)
                torch.cuda.synchronize()
                m = measure_form(short)
                healthy = rot.losses_finite() and (form != "oneshot" or int(xchg.err[0].item()) == 0)
                m["healthy"] = launch.all_ranks_agree(healthy, dev)
                if not m["healthy"]:
                    m["reason"] = "non-finite loss or an exchange time-out after the timed loop"
                    if xchg is not None:
                        xchg.err.zero_()
            except Exception as exc:      # noqa: BLE001  (a form that raises is a form that is not used; the banked one stands)
                m = {"healthy": False, "reason": f"{type(exc).__name__}: {exc}"}
                launch.all_ranks_agree(False, dev)
            m["seconds_spent"] = time.perf_counter() - t_form
            exchange_forms[form] = m
            log(f"exchange form {form}: {m.get('ms_per_step')} ms/step, healthy {m['healthy']}" + (f" ({m.get('reason')})" if not m["healthy"] else ""))
        good = {f: m for f, m in exchange_forms.items() if m.get("healthy") and m.get("ms_per_step")}
        if not good:
            # nothing healthy, not even the plain collective: report what happened, no value
            gate = dict(gate or {}, exchange_forms=exchange_forms)
            dist.barrier()
            dist.destroy_process_group()
            refuse_timing("no data-parallel exchange form ran healthy")
        exchange_mode = min(good, key=lambda f: good[f]["ms_per_step"])
        unhealthy = [f for f, m in exchange_forms.items() if not m.get("healthy")]
        if unhealthy:
            fallback_reason = "; ".join(f"{f}: {exchange_forms[f].get('reason')}" for f in unhealthy)
        # the official figure: the chosen form again, full sustained run + EXACTLY K steps
        launch.set_exchange_form(trainers, dp, exchange_mode, xchg)
        for j in range(max(3, NB)):
            rot.eager(j % NB)
        torch.cuda.synchronize()
        best = measure_form(args.sustain)
        healthy = rot.losses_finite() and (exchange_mode != "oneshot" or int(xchg.err[0].item()) == 0)
        if not launch.all_ranks_agree(healthy, dev):
            # the chosen form went bad in the long run: the banked short measurement of the plain collective stands
            fallback_reason = (fallback_reason + "; " if fallback_reason else "") + f"{exchange_mode}: unhealthy in the sustained run"
            exchange_mode, best = "rccl", dict(good["rccl"]) if "rccl" in good else best
    launch_mode, ms_step = best["launch"], best["ms_per_step"]
    log(f"headline: {ms_step:.4f} ms/step ({launch_mode}; sustained {best['sustained']['ms_per_step']:.4f} over {best['sustained']['seconds']:.2f} s)")

    # ---- secondary: burst figures (round 1's headline: ONE batch replayed in place from an idle chip), gradients-only step
    dt_fb = dt_graph_burst = None
    one = lambda fn: (lambda s, n: [fn() for _ in range(n)])
    if fused_ok and world == 1:
        try:
            if rot.graphs:
                for _ in range(3):
                    trainers[0].replay()
                dt_graph_burst = timed(args.steps, one(trainers[0].replay))
            fwdbwd.capture(r0.fresh)
            for _ in range(3):
                fwdbwd.replay()
            dt_fb = timed(args.steps, one(fwdbwd.replay))
        except Exception as exc:      # noqa: BLE001
            log(f"burst figures skipped: {type(exc).__name__}: {exc}")
    dt_eager_burst = timed(args.steps, lambda s, n: [rot.eager(0) for _ in range(n)])

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
                   # (training step: the launch also carries the readout head, forward and backward)
                   "hcg_fused_forward": bd["conv1_fwd"] + bd["conv2_fwd"] + bd["pool_fwd"] +
                                        (0 if args.forward_only else bd["readout_fwd"] + bd["readout_bwd"])}.get(entry, float("nan"))
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
                      "hcg_fused_forward": "k_fused_layer_fwd", "hcg_mid_layer_fwd": "k_mid_layer_fwd"}.get(entry)
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
            rrot = launch.Rotation(rtr, [r.fresh for r in rr], [r.planned for r in rr], [r.plan for r in rr],
                                   plan_overlap=args.plan_overlap)
            for j in range(2 * NB):
                rrot.eager(j % NB)
            if rrot.capture() and best["launch"] == "window":
                rrot.build_windows(0)
            rform = "window" if rrot.window is not None else ("graph" if rrot.graphs else "eager")
            rrun = lambda s, n: rrot.run(rform, s, n)
            timed(4 * NB, rrun)
            k = max(args.steps, 200)
            rdt = timed(k, rrun)
            gpt = lib.hcg_fused_graphs_per_tile(F, D, rr[0].sb.max_nodes)
            ragged = {"value": rr[0].B * k / rdt, "unit": "graphs/s", "ms_per_step": rdt / k * 1e3, "steps": k,
                      "graphs": rr[0].B, "nodes": rr[0].N, "edges": rr[0].E, "max_nodes": rr[0].sb.max_nodes,
                      "kernel_family": "small-graph tiles" if gpt > 0 else ("size-grouped batch: tiles for graphs <= 32 nodes + one graph per wave"
                                                                           if rr[0].sb.n_small else "one graph per wave / workgroup"),
                      "note": f"n_g ~ U{{24..36}}, {NB} distinct batches round-robin, launch form {rform}"}
            log(f"ragged variant: {rdt / k * 1e3:.4f} ms/step")
            del rr, rtr, rrot
        except Exception as exc:
            ragged = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        touched = NB * (r0.bytes_touched() + (sum(t.nbytes for t in trainers[0]._bufs["cap"]["acts"] + trainers[0]._bufs["cap"]["dacts"])
                                              if fused_ok else 0))
        xdesc = {"none": "", "rccl": "RCCL all-reduce (", "rccl-captured": "RCCL all-reduce recorded in the step graph (",
                 "oneshot": "one-shot xGMI exchange inside the last launch ("}[exchange_mode]
        fwd_only_note = "plan/gcn_norm build + forward (conv stack, pool, readout) only" if args.forward_only else \
            (f"full training step: per-step plan/gcn_norm build, forward, sqrt(MSE) loss, backward, "
             f"{xdesc + args.combine + '), ' if world > 1 else ''}Adam update")
        spg = best.get("steps_per_graph_launch")
        rec = {
            "metric": "molecular graphs/sec fwd+bwd at 1/2/4/8 MI355X; achieved HBM GB/s" if not args.forward_only
                      else "molecular graphs/sec FORWARD ONLY (configs[1]; not the headline metric)",
            "value": world * B * args.steps / best["seconds"], "unit": "graphs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'C2' if args.forward_only and cfg_name == 'C3' else args.config}: {B} synthetic graphs/GPU x "
                                   f"{N / B:.0f} atoms x {E / B:.0f} directed edges x "
                                   f"{F}-d features, {opt.n_convolutions}xGCNConv({D}) + [max,mean] pool + readout; "
                                   f"{fwd_only_note}; {NB} distinct batches round-robin; launch={launch_mode}"
                                   f"{' (' + str(spg) + ' consecutive steps per graph launch)' if spg and spg > 1 else ''}"
                                   f"{'' if (launch_mode == 'eager' or args.plan_overlap == 'none' or not fused_ok) else ', plan build of the NEXT batch ' + ('inside the last launch of the step' if args.plan_overlap == 'fused' else 'on a forked graph branch')}",
                       "graphs_per_gpu": B, "nodes": N, "edges": E, "feat": F, "hidden": D,
                       "parallelism": f"dp{world} (batch-of-graphs, {'one-shot xGMI exchange' if exchange_mode == 'oneshot' else 'RCCL all-reduce'} "
                                      f"of {sum(p.numel() for p in model.parameters())} fp32 grads)"},
            "parity_gate": gate,
            "rccl_world": rccl_world, "exchange": exchange_mode,
            # N > 1: every exchange form measured on this machine (the plain collective first), and why a form was not used
            "exchange_forms": exchange_forms or None, "exchange_fallback_reason": fallback_reason,
            # the child-process trials of the two graph forms (budget: PROBE_BUDGET_S seconds)
            "exchange_probe": None if probe is None else {k: probe.get(k) for k in ("oneshot", "captured", "exit", "seconds", "note")},
            "distinct_batches": NB, "bytes_touched_between_reuse": touched,
            "sustained_s": best["sustained"]["seconds"],
            "sustained": dict(best["sustained"], value=(world * B * best["sustained"]["steps"] / best["sustained"]["seconds"]
                                                        if best["sustained"]["seconds"] > 0 else None),
                              note="the same rotation, run right before the timed steps (one synchronize per 256 steps)"),
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
            "launch": launch_mode, "steps_per_graph_launch": spg, "launch_forms_ms": best.get("launch_forms_ms"),
            "launch_errors": best.get("errors") or None, "eager_ms_per_step": dt_eager / args.steps * 1e3,
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
