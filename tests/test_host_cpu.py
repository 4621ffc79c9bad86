"""CPU-side tests (no GPU): the C-ABI library loads and exports every declared symbol, host logic
(collation, synthetic generator, module surface / state-dict compatibility) and loud failure on CPU
tensors.  No compute entry point is called here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import hcatgnet_amd as H
from hcatgnet_amd import _lib, synth
from tests.helpers import golden_files, load_golden

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()


def test_library_exports_every_header_symbol():
    hdr = open(os.path.join(REPO, "include", "hcatgnet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(hcg_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 12
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/hcatgnet_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().hcg_version() == 1
    assert _lib.load().hcg_error_string(-2) == b"workspace too small"
    assert len(declared) <= 55, "the ABI grew: fold fusion combinations into an argument struct instead of a new symbol"


def test_host_structs_match_the_library():
    """The ctypes mirrors of the ABI's HOST structs (hcg_reduce_job, hcg_tail_args, hcg_fused_fwd_args) have the library's
    sizes, and a wrong argument block is refused without touching the GPU."""
    lib = _lib.load()
    assert ctypes.sizeof(_lib.ReduceJob) == lib.hcg_struct_bytes(_lib.HCG_STRUCT_REDUCE_JOB)
    assert ctypes.sizeof(_lib.TailArgs) == lib.hcg_struct_bytes(_lib.HCG_STRUCT_TAIL_ARGS)
    assert ctypes.sizeof(_lib.FusedFwdArgs) == lib.hcg_struct_bytes(_lib.HCG_STRUCT_FUSED_FWD_ARGS)
    assert ctypes.sizeof(_lib.CollateArgs) == lib.hcg_struct_bytes(_lib.HCG_STRUCT_COLLATE_ARGS)
    assert ctypes.sizeof(_lib.CollateSlot) == lib.hcg_struct_bytes(_lib.HCG_STRUCT_COLLATE_SLOT)
    assert lib.hcg_step_tail(None, None) == -1 and lib.hcg_fused_forward(None, None) == -1
    a = _lib.TailArgs()
    assert lib.hcg_step_tail(ctypes.addressof(a), None) == 0            # no jobs, nothing else: nothing to do
    a.njobs = 1                                                          # jobs announced but not given
    assert lib.hcg_step_tail(ctypes.addressof(a), None) == -1
    f = _lib.FusedFwdArgs()
    f.D, f.F, f.graphs_per_tile = 96, 64, 1
    assert lib.hcg_fused_forward(ctypes.addressof(f), None) == -3       # unsupported width


def test_cpu_tensors_fail_loudly_no_fallback():
    sb = synth.make_config("C1")
    m = H.make_network("GCN", H.default_options(), 64)
    with pytest.raises(_lib.HcgError, match="no CPU fallback|MI355X"):
        m(sb.as_batch())
    with pytest.raises(ValueError):
        H.make_network("GAT", H.default_options(), 64)          # reference call_methods.py:11-12


def test_module_surface_matches_reference_contract():
    opt = H.default_options()
    m = H.make_network("GCN", opt, 25)
    assert m.name == "GCN" and type(m.loss).__name__ == "MSELoss"
    assert isinstance(m.optimizer, torch.optim.Optimizer) and type(m.optimizer).__name__ == "FusedAdam"
    assert m.optimizer.defaults["eps"] == 1e-9 and m.optimizer.defaults["lr"] == 0.01 and m.optimizer.defaults["betas"] == (0.9, 0.999)
    assert isinstance(m.scheduler, torch.optim.lr_scheduler.ReduceLROnPlateau)
    keys = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert keys == {"conv1.bias": (64,), "conv1.lin.weight": (64, 25), "conv_layers.0.bias": (64,),
                    "conv_layers.0.lin.weight": (64, 64), "readout.0.0.weight": (64, 128), "readout.0.0.bias": (64,),
                    "readout.1.weight": (1, 64), "readout.1.bias": (1,)}
    assert float(m.conv1.bias.abs().max()) == 0.0                          # PyG zeros(bias)
    a = (6.0 / (25 + 64)) ** 0.5
    assert float(m.conv1.lin.weight.abs().max()) <= a                       # glorot bound
    # reference state-dicts load unchanged
    g = load_golden(golden_files()[0])
    m.load_state_dict(g["params"])
    # seeding at construction (model/networks.py:19)
    m1 = H.make_network("GCN", opt, 25); m2 = H.make_network("GCN", opt, 25)
    assert torch.equal(m1.conv1.lin.weight, m2.conv1.lin.weight)


def test_collate_reproduces_pyg_batch_rule():
    gs = []
    g = torch.Generator().manual_seed(0)
    for n, e in [(3, 4), (1, 0), (5, 8)]:
        ei = torch.randint(0, n, (2, e), generator=g)
        gs.append(H.Data(x=torch.randn(n, 7, generator=g), edge_index=ei, y=torch.randn(1, generator=g), idx=n))
    b = H.collate(gs)
    assert b.num_graphs == 3 and b.x.shape == (9, 7) and b.edge_index.shape == (2, 12)
    assert b.batch.tolist() == [0, 0, 0, 1, 2, 2, 2, 2, 2]
    assert torch.equal(b.edge_index[:, :4], gs[0].edge_index) and torch.equal(b.edge_index[:, 4:], gs[2].edge_index + 4)
    assert b.ptr.tolist() == [0, 3, 4, 9] and b.edge_ptr.tolist() == [0, 4, 4, 12]
    assert b.y.shape == (3,) and b.idx.tolist() == [3, 1, 5] and b.max_nodes == 5 and b.edges_grouped
    dl = H.DataLoader(gs, batch_size=2, shuffle=False)
    assert [bb.num_graphs for bb in dl] == [2, 1] and len(dl) == 2


@pytest.mark.parametrize("name,ng", [("C1", 1), ("C2", 50), ("C5", 3)])
def test_synthetic_graphs_follow_survey_spec(name, ng):
    cfg = synth.CONFIGS[name]
    sb = synth.make_config(name, num_graphs=ng)
    N, E = sb.x.shape[0], sb.edge_index.shape[1]
    assert N == ng * cfg["nodes"] and E == ng * 2 * (cfg["nodes"] - 1 + cfg["extra_bonds"])
    src, dst = sb.edge_index.numpy()
    assert np.array_equal(src[0::2], dst[1::2]) and np.array_equal(dst[0::2], src[1::2])   # [i,j],[j,i] interleaved
    assert (src != dst).all()
    assert (sb.batch.numpy()[src] == sb.batch.numpy()[dst]).all()
    deg = np.bincount(dst, minlength=N)
    assert deg.max() <= cfg["max_degree"] and deg.min() >= 1
    pairs = set(zip(src.tolist(), dst.tolist())); assert len(pairs) == E                    # no duplicate bonds
    assert not np.all(np.diff(dst) >= 0)                                                     # NOT target-sorted
    sb2 = synth.make_config(name, num_graphs=ng)
    assert torch.equal(sb.x, sb2.x) and torch.equal(sb.edge_index, sb2.edge_index)           # seeded
    assert not torch.equal(sb.x, synth.make_config(name, num_graphs=ng, rank=1).x)


def test_ctypes_signatures_match_header_arity():
    hdr = open(os.path.join(REPO, "include", "hcatgnet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for m in re.finditer(r"\b(hcg_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("void", "") else len(args.split(","))
        assert n == len(_lib.SIGNATURES[name][1]), name


def test_algorithmic_bytes_match_survey_numbers():
    from hcatgnet_amd import algbytes
    assert algbytes.structure_bytes(122880, 262144) == 2_031_620
    assert algbytes.conv_fwd(122880, 262144, 64, 64) == 64_962_820
    assert algbytes.forward_bytes(122880, 262144, 4096, 64, 64) == 165_643_280
    assert algbytes.step_bytes(122880, 262144, 4096, 64, 64) == 438_242_860


def test_fused_train_step_support_check_is_host_only():
    """`FusedTrainStep.unsupported_reason` decides on the host (no GPU needed) whether a model / batch can take the
    no-autograd step: small-graph tiles up to 32 nodes, one-graph-per-workgroup kernels up to 224 nodes, D = 64 or 128."""
    import hcatgnet_amd as H
    from hcatgnet_amd.train import FusedTrainStep
    m = H.make_network("GCN", H.default_options(), 25)
    x = torch.zeros(4, 25); ei = torch.zeros(2, 0, dtype=torch.int64); bv = torch.zeros(4, dtype=torch.int64)
    mk = lambda **kw: H.Batch(x, ei, bv, 1, y=torch.zeros(1), **kw)
    assert FusedTrainStep.unsupported_reason(m, mk(max_nodes=30, max_edges=64, edges_grouped=True)) is None
    assert FusedTrainStep.unsupported_reason(m, mk(max_nodes=184, max_edges=390, edges_grouped=True)) is None
    assert "shape" in FusedTrainStep.unsupported_reason(m, mk(max_nodes=300, max_edges=700, edges_grouped=True))
    assert "metadata" in FusedTrainStep.unsupported_reason(m, mk())
    assert "targets" in FusedTrainStep.unsupported_reason(m, H.Batch(x, ei, bv, 1, max_nodes=30, max_edges=64, edges_grouped=True))
    wide = H.make_network("GCN", H.default_options(embedding_dim=128), 25)       # 128-d: one-launch head since round 2
    assert FusedTrainStep.unsupported_reason(wide, mk(max_nodes=200, max_edges=424, edges_grouped=True)) is None
    odd = H.make_network("GCN", H.default_options(embedding_dim=96), 25)         # no conv kernel family covers D = 96
    assert "shape" in FusedTrainStep.unsupported_reason(odd, mk(max_nodes=30, max_edges=64, edges_grouped=True))
    deep = H.make_network("GCN", H.default_options(readout_layers=3), 25)
    assert "readout" in FusedTrainStep.unsupported_reason(deep)


def test_kernel_family_selection_is_host_only():
    """Which kernel family takes a layer is decided on the host from (F, D, largest graph): 128-wide layers over graphs up
    to 224 nodes -> csrc/tall.hip when F is a multiple of 4; 64-wide layers over graphs of 65 .. 224 nodes -> its backward,
    but only for batches that fill the chip (`functional.TALL_MIN_NODES_D64`); everything up to 64 nodes stays with the
    one-graph-per-wave kernels."""
    from types import SimpleNamespace
    from hcatgnet_amd import _lib, functional as HF
    lib = _lib.load()
    assert lib.hcg_tall_supported(128, 128, 200, 424) == 1 and lib.hcg_tall_supported(28, 128, 159, 330) == 1
    assert lib.hcg_tall_supported(126, 128, 200, 424) == 0          # rows of x not 16-byte aligned: the forward needs float4 rows
    assert lib.hcg_tall_supported(25, 64, 117, 250) == 1 and lib.hcg_tall_supported(64, 64, 184, 390) == 1
    assert lib.hcg_tall_supported(25, 64, 60, 130) == 0             # <= 64 nodes: one graph per wave
    assert lib.hcg_tall_supported(25, 64, 225, 500) == 0 and lib.hcg_tall_supported(25, 64, 117, 1025) == 0
    assert lib.hcg_tall_supported(25, 96, 117, 250) == 0
    assert lib.hcg_tall_workspace_bytes(1000, 10, 25, 64) > 1000 * 64 * 4 and lib.hcg_tall_workspace_bytes(1000, 10, 25, 96) == 0
    plan = lambda N: SimpleNamespace(mode="blocked", ew_csr=None, max_nodes=117, max_edges=250, B=40, N=N)
    assert not HF.tall_supported(plan(3473), 25, 64)                # the reference's batch of 40 graphs: launch-bound, stays on mid.hip
    assert HF.tall_supported(plan(356553), 25, 64)
    assert HF.tall_supported(SimpleNamespace(mode="blocked", ew_csr=None, max_nodes=200, max_edges=424, B=24, N=4800), 128, 128)
    assert lib.hcg_head_supported(64, 1) == 1 and lib.hcg_head_supported(128, 8) == 1 and lib.hcg_head_supported(96, 1) == 0


def test_base_network_dispatch_mirrors_the_reference():
    """`BaseNetwork._make_loss / _make_optimizer / _make_scheduler` (reference model/networks.py:28-56): every branch the
    reference's option parser can select builds the same kind of object, unknown names raise the reference's errors."""
    import pytest
    import hcatgnet_amd as H
    from hcatgnet_amd.optim import FusedAdam
    kinds = {"StepLR": torch.optim.lr_scheduler.StepLR, "ExponentialLR": torch.optim.lr_scheduler.ExponentialLR,
             "ReduceLROnPlateau": torch.optim.lr_scheduler.ReduceLROnPlateau}
    for name, cls in kinds.items():
        m = H.make_network("GCN", H.default_options(scheduler=name), 25)
        assert isinstance(m.scheduler, cls) and isinstance(m.optimizer, FusedAdam)
        assert m.optimizer.param_groups[0]["eps"] == 1e-9 and m.optimizer.param_groups[0]["lr"] == 0.01      # networks.py:38
    m = H.make_network("GCN", H.default_options(scheduler="MultiStepLR", step_size=[3, 6]), 25)
    assert isinstance(m.scheduler, torch.optim.lr_scheduler.MultiStepLR)
    assert isinstance(H.make_network("GCN", H.default_options(optimizer="SGD"), 25).optimizer, torch.optim.SGD)
    assert isinstance(H.make_network("GCN", H.default_options(optimizer="rmsprop"), 25).optimizer, torch.optim.RMSprop)
    assert isinstance(H.make_network("GCN", H.default_options(problem_type="classification", n_classes=3), 25).loss,
                      torch.nn.CrossEntropyLoss)
    with pytest.raises(NotImplementedError):
        H.make_network("GCN", H.default_options(optimizer="LBFGS"), 25)
    with pytest.raises(NotImplementedError):
        H.make_network("GCN", H.default_options(scheduler="Cosine"), 25)
    with pytest.raises(ValueError):
        H.make_network("GCN", H.default_options(problem_type="ranking"), 25)
    with pytest.raises(ValueError):
        H.make_network("GAT", H.default_options(), 25)                        # call_methods.py:7-12
    assert m.name == "GCN"


def test_exchange_probe_failure_in_the_child_is_a_verdict_not_an_exception(monkeypatch):
    """bench.py tries the one-shot xGMI exchange in a sacrificial child process per rank first.  Here (no GPU) the child
    dies at its first device call: the caller gets both verdicts False and carries on with the plain RCCL form."""
    import importlib.util
    import socket
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    for k, v in {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_PORT": str(port)}.items():
        monkeypatch.setenv(k, v)
    if torch.cuda.is_available():
        pytest.skip("the point is a child that cannot reach a GPU")
    got = bench.isolated_probe(1, 0, False, "sse", timeout_s=240.0)
    assert got["oneshot"] is False and got["captured"] is False
    assert got["exit"] not in (None, 0) and got["seconds"] > 0 and got["note"]          # which phase failed and why is recorded
    assert bench.PROBE_BUDGET_S <= 120.0


def test_trainer_parameter_cache_follows_the_model():
    """`FusedTrainStep._trainable()` caches `model.parameters()` (the module walk is a quarter of a small step's host time):
    same objects in the same order, and the cache notices a replaced Parameter, a replaced child module and a frozen one."""
    from hcatgnet_amd.train import FusedTrainStep
    m = H.make_network("GCN", H.default_options(), 25)
    t = FusedTrainStep(m, optimizer_step=False)
    same = lambda: all(a is b for a, b in zip(t._trainable(), m.parameters())) and len(t._trainable()) == len(list(m.parameters()))
    assert same() and t._trainable() is t._trainable()
    m.conv1.bias = torch.nn.Parameter(torch.zeros_like(m.conv1.bias))
    assert same()
    m.conv1 = H.make_network("GCN", H.default_options(), 25).conv1
    assert same()
    assert t.reason() is None
    m.conv1.bias.requires_grad_(False)
    assert "frozen" in t.reason()
