"""ctypes binding of libhcatgnet_hip.so (the C ABI declared in include/hcatgnet_hip.h).

There is NO CPU fallback: if the library is missing, or a tensor is not on an MI355X, every
entry point raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported
from this package.)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libhcatgnet_hip.so")

HCG_PLAN_GENERAL, HCG_PLAN_BLOCKED, HCG_PLAN_PTRS_ONLY, HCG_PLAN_KEEP_STATUS = 0, 1, 2, 4
HCG_ACT_NONE, HCG_ACT_LEAKY = 0, 1
HCG_LOSS_MSE, HCG_LOSS_RMSE, HCG_LOSS_SSE = 0, 1, 2      # loss modes of hcg_step_tail / hcg_loss_finalize / hcg_loss_fwd_bwd
HCG_HEAD_FORWARD_ONLY = 1
HCG_WS_PLAN, HCG_WS_LINEAR, HCG_WS_GCN_LAYER_BWD, HCG_WS_READOUT2 = 0, 1, 2, 3        # hcg_general_workspace_bytes kinds
HCG_FUSED_POOLBITS, HCG_FUSED_HEAD_WS = 0, 1                                          # hcg_fused_aux_bytes kinds
HCG_STRUCT_REDUCE_JOB, HCG_STRUCT_TAIL_ARGS, HCG_STRUCT_FUSED_FWD_ARGS, HCG_STRUCT_COLLATE_ARGS, HCG_STRUCT_COLLATE_SLOT = 0, 1, 2, 3, 4   # hcg_struct_bytes
HCG_REDUCE_MAX_JOBS, HCG_REDUCE_MAX_SEGS = 8, 4
HCG_XCHG_MEAN, HCG_XCHG_SSE, HCG_XCHG_ERR_TIMEOUT, HCG_XCHG_MAX_WORLD = 0, 1, 1, 8
STATUS_BITS = {1: "edge_index entry outside [0, N)", 2: "batch vector not sorted (non-decreasing)",
               4: "batch id outside [0, num_graphs)", 8: "edges not grouped by graph / edge crosses graphs",
               16: "a graph exceeds the fused-kernel tile",
               32: "explicit self-loop edge together with edge_weight (its weight as loop weight is not implemented)"}
STATUS_EDGE_UNGROUPED = 8

P, I64, SZ, F32, INT = c_void_p, c_int64, c_size_t, c_float, c_int
I32 = ctypes.c_int32


# ---- host structs of the C ABI (include/hcatgnet_hip.h); tests/test_host_cpu.py checks their sizes against the library
class ReduceSeg(ctypes.Structure):
    _fields_ = [("begin", I32), ("count", I32), ("row_in", I32), ("row_out", I32), ("dst", P)]


class ReduceJob(ctypes.Structure):
    _fields_ = [("slabs", P), ("sse_part", P), ("nslabs", I32), ("slab_floats", I32), ("nseg", I32), ("reserved", I32),
                ("seg", ReduceSeg * HCG_REDUCE_MAX_SEGS)]


class TailArgs(ctypes.Structure):
    """hcg_tail_args: the step's last launch (slab reductions, loss + deferred scale, exchange, Adam, next plan)."""
    _fields_ = [("jobs_host", P), ("njobs", I32), ("loss_mode", I32), ("loss_count", F32), ("beta1", F32), ("beta2", F32),
                ("eps", F32), ("loss", P), ("sse_tail", P), ("grad_flat", P), ("param", P), ("exp_avg", P), ("exp_avg_sq", P),
                ("n", I64), ("lr_dev", P), ("step_dev", P), ("next_edge_index", P), ("next_batch", P), ("next_N", I64),
                ("next_E", I64), ("next_B", I64), ("next_graph_ptr", P), ("next_edge_ptr", P), ("next_status", P), ("inbox", P),
                ("peers_host", P), ("rank", I32), ("world", I32), ("xchg_mode", I32), ("reserved", I32), ("xchg_err", P)]


class FusedFwdArgs(ctypes.Structure):
    """hcg_fused_fwd_args: every forward form of the small-graph tiles."""
    _fields_ = [("x", P), ("W1", P), ("b1", P), ("W2", P), ("b2", P), ("edge_index", P), ("E", I64), ("graph_ptr", P),
                ("edge_ptr", P), ("N", I64), ("B", I64), ("F", I64), ("D", I64), ("graphs_per_tile", I32), ("apply_act", I32),
                ("slope", F32), ("head_flags", I32), ("out1", P), ("out2", P), ("emb", P), ("poolbits", P), ("status", P),
                ("y", P), ("head_W0", P), ("head_b0", P), ("head_W1", P), ("head_b1", P), ("C", I64), ("z", P), ("out", P),
                ("demb", P), ("head_workspace", P), ("head_workspace_bytes", SZ), ("step_counter", P)]


class CollateSlot(ctypes.Structure):
    """hcg_collate_slot: one batch of a collate launch."""
    _fields_ = [("ids", P), ("graph_ptr", P), ("edge_ptr", P), ("x_out", P), ("edge_index_out", P), ("batch_out", P),
                ("y_out", P), ("idx_out", P), ("B", I64), ("N_out", I64), ("E_out", I64)]


class CollateArgs(ctypes.Structure):
    """hcg_collate_args: the dataset + one slot (host values) or a device array of slots."""
    _fields_ = [("x_all", P), ("src_all", P), ("dst_all", P), ("node_ptr_all", P), ("edge_ptr_all", P), ("y_all", P),
                ("idx_all", P), ("F", I64), ("nslots", I32), ("reserved", I32), ("slot", CollateSlot), ("slots_dev", P),
                ("max_B", I64)]


# name -> (restype, argtypes); must list every symbol of include/hcatgnet_hip.h
SIGNATURES = {
    "hcg_version": (INT, []),
    "hcg_struct_bytes": (SZ, [INT]),
    "hcg_general_workspace_bytes": (SZ, [INT, I64, I64, I64, INT]),
    "hcg_fused_aux_bytes": (SZ, [INT, I64, INT]),
    "hcg_error_string": (c_char_p, [INT]),
    "hcg_plan_build": (INT, [P, P, P, I64, I64, I64, F32, INT, P, P, P, P, P, P, P, P, P, P, P, P, P, P, SZ, P]),
    "hcg_linear_fwd": (INT, [P, P, P, P, I64, I64, I64, INT, F32, P]),
    "hcg_linear_bwd": (INT, [P, P, P, P, P, P, P, P, I64, I64, I64, INT, F32, P, SZ, P]),
    "hcg_gcn_layer_fwd": (INT, [P, P, P, P, P, P, P, F32, F32, INT, P, P, I64, I64, I64, I64, P]),
    "hcg_gcn_layer_bwd": (INT, [P, P, P, P, P, P, P, P, F32, F32, INT, P, P, P, P, I64, I64, I64, I64, P, SZ, P]),
    "hcg_gcn_edge_weight_grad": (INT, [P, P, P, P, P, P, F32, INT, P, I64, I64, I64, P]),
    "hcg_pool_fwd": (INT, [P, P, P, I64, I64, I64, P]),
    "hcg_pool_bwd": (INT, [P, P, P, P, P, I64, I64, I64, P]),
    "hcg_fused_graphs_per_tile": (INT, [I64, I64, I64]),
    "hcg_fused_workspace_bytes": (SZ, [I64, I64, I64, INT]),
    "hcg_fused_forward": (INT, [P, P]),
    "hcg_fused_head_reduce_job": (INT, [P, SZ, I64, INT, I64, P, P, P, P, P]),
    "hcg_fused_layer_bwd": (INT, [P, P, P, P, P, P, P, P, I64, P, P, I64, I64, I64, I64, INT, F32, INT, P, P, P, SZ, P]),
    "hcg_mid_supported": (INT, [I64, I64, I64, I64]),
    "hcg_mid_workspace_bytes": (SZ, [I64, I64, I64, I64, I64]),
    "hcg_mid_layer_fwd": (INT, [P, P, P, P, I64, P, P, I64, I64, I64, I64, I64, I64, F32, INT, P, P, P, P, P, P, P]),
    "hcg_mid_layer_bwd": (INT, [P, P, P, P, P, P, P, I64, P, P, I64, I64, I64, I64, I64, I64, F32, INT, P, P, P, SZ, P]),
    "hcg_mid_reduce_job": (INT, [P, SZ, I64, I64, I64, I64, I64, INT, P, P, P]),
    "hcg_tall_supported": (INT, [I64, I64, I64, I64]),
    "hcg_tall_workspace_bytes": (SZ, [I64, I64, I64, I64]),
    "hcg_tall_layer_fwd": (INT, [P, P, P, P, I64, P, P, I64, I64, I64, I64, I64, I64, F32, INT, P, P, P, P, P, P, P, SZ, P]),
    "hcg_tall_layer_bwd": (INT, [P, P, P, P, P, P, P, P, P, P, P, I64, P, P, I64, I64, I64, I64, I64, I64, F32, INT, P, P, P, SZ, P]),
    "hcg_tall_reduce_jobs": (INT, [P, SZ, I64, I64, I64, I64, INT, P, P, P]),
    "hcg_fused_reduce_job": (INT, [P, SZ, I64, I64, I64, I64, INT, P, P, P]),
    "hcg_readout2_bwd_partial": (INT, [P, P, P, P, P, I64, I64, I64, F32, P, P, SZ, P]),
    "hcg_readout2_reduce_job": (INT, [P, SZ, I64, I64, P, P, P, P, P]),
    "hcg_reduce_job_append": (INT, [P, P]),
    "hcg_step_tail": (INT, [P, P]),
    "hcg_loss_finalize": (INT, [P, F32, INT, P, P, P]),
    "hcg_collate": (INT, [P, P]),
    "hcg_adam_step": (INT, [P, P, P, P, I64, F32, F32, F32, F32, I64, P]),
    "hcg_mse_fwd": (INT, [P, P, I64, P, P]),
    "hcg_mse_bwd": (INT, [P, P, P, I64, P, P, P]),
    "hcg_loss_fwd_bwd": (INT, [P, P, I64, INT, P, P, P, P]),
    "hcg_readout2_fwd": (INT, [P, P, P, P, P, I64, I64, I64, F32, P, P, P]),
    "hcg_head_supported": (INT, [I64, I64]),
    "hcg_head_workspace_bytes": (SZ, [I64, I64]),
    "hcg_head_fwd_bwd": (INT, [P, P, P, P, P, P, I64, I64, I64, F32, INT, P, P, P, P, SZ, P, P]),
    "hcg_head_reduce_job": (INT, [P, SZ, I64, I64, I64, P, P, P, P, P]),
    "hcg_sse_finalize": (INT, [P, I64, P, P]),
    "hcg_adam_step_dev_sse": (INT, [P, P, P, P, I64, P, F32, F32, F32, P, P, P]),
    "hcg_adam_step_dev": (INT, [P, P, P, P, I64, P, F32, F32, F32, P, P]),
    "hcg_xchg_inbox_bytes": (SZ, [I64, INT]),
    "hcg_xchg_resident_blocks": (INT, []),
    "hcg_xchg_alloc": (INT, [SZ, P]),
    "hcg_xchg_free": (INT, [P]),
    "hcg_xchg_zero": (INT, [P, SZ]),
    "hcg_xchg_ipc_export": (INT, [P, P]),
    "hcg_xchg_ipc_open": (INT, [P, P]),
    "hcg_xchg_ipc_close": (INT, [P]),
}

_lib = None


class HcgError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HcgError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C hcatgnet_amd/csrc`). hcatgnet_amd has no CPU/PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI/header drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.hcg_version() != 1:
        raise HcgError(f"ABI version mismatch: library reports {lib.hcg_version()}, binding expects 1")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().hcg_error_string(rc)
        raise HcgError(f"{what} failed: {msg.decode() if msg else rc} (code {rc})")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = None
_gpu_ok = None


def stream_ptr():
    """hipStream_t of torch's CURRENT stream on the current device (so that work enqueued here is
    ordered with the surrounding torch ops, also under torch.cuda.graph capture on a side stream).
    `torch._C._cuda_getCurrentRawStream` is the cheap accessor (~1 us vs ~11 us for the Stream object)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*tensors):
    """Every tensor on the GPU -- and on the CURRENT device: the library enqueues on the current device's stream, a
    pointer into another GPU's memory would fault there."""
    global _gpu_ok
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise HcgError("hcatgnet_amd runs on MI355X (ROCm) tensors only; got a CPU tensor. "
                           "There is no CPU fallback: move the model and the batch to the GPU.")
        if cur is None:
            import torch
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise HcgError(f"tensor on cuda:{t.device.index} but the current device is cuda:{cur}: "
                           "call torch.cuda.set_device (one process per GPU) before using hcatgnet_amd")
    if _gpu_ok is None:
        import torch
        _gpu_ok = bool(torch.cuda.is_available())
    if not _gpu_ok:
        raise HcgError("no ROCm GPU visible")


def describe_status(word: int) -> str:
    return "; ".join(msg for bit, msg in STATUS_BITS.items() if word & bit) or "ok"


# ---- struct-argument entry points ------------------------------------------------------------------------------------
def fused_forward(**kw):
    """hcg_fused_forward: keyword = field of hcg_fused_fwd_args (tensors or None for pointers, numbers otherwise)."""
    a = FusedFwdArgs()
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if hasattr(v, "data_ptr") else v)
    check(load().hcg_fused_forward(ctypes.addressof(a), stream_ptr()), "hcg_fused_forward")


def step_tail(jobs_addr: int, njobs: int, *, loss=None, loss_mode: int = HCG_LOSS_RMSE, loss_count: float = 0.0, sse_tail=None,
              adam=None, next_plan=None, xchg=None):
    """hcg_step_tail.  `jobs_addr`: host address of `njobs` hcg_reduce_job; `loss` [2] / `sse_tail` [2] device tensors;
    `adam`: dict(grad_flat, param, exp_avg, exp_avg_sq (tensors), n, lr_dev, step_dev (tensors), beta1, beta2, eps);
    `next_plan`: a pointers-only blocked BatchPlan; `xchg`: dict(inbox (address), peers_host (address of the host pointer
    array), rank, world, mode, err (tensor))."""
    a = TailArgs()
    a.jobs_host, a.njobs, a.loss_mode, a.loss_count = jobs_addr, njobs, loss_mode, float(loss_count)
    a.loss, a.sse_tail = ptr(loss), ptr(sse_tail)
    if adam is not None:
        a.grad_flat, a.param, a.exp_avg, a.exp_avg_sq = (adam["grad_flat"].data_ptr(), adam["param"].data_ptr(),
                                                         adam["exp_avg"].data_ptr(), adam["exp_avg_sq"].data_ptr())
        a.n, a.lr_dev, a.step_dev = adam["n"], adam["lr_dev"].data_ptr(), adam["step_dev"].data_ptr()
        a.beta1, a.beta2, a.eps = adam["beta1"], adam["beta2"], adam["eps"]
    if next_plan is not None:
        np_ = next_plan
        a.next_edge_index, a.next_batch = np_.edge_index.data_ptr(), np_.batch.data_ptr()
        a.next_N, a.next_E, a.next_B = np_.N, np_.E, np_.B
        a.next_graph_ptr, a.next_edge_ptr, a.next_status = np_.graph_ptr.data_ptr(), np_.edge_ptr.data_ptr(), np_.status.data_ptr()
    if xchg is not None:
        a.inbox, a.peers_host, a.rank, a.world, a.xchg_mode = xchg["inbox"], xchg["peers_host"], xchg["rank"], xchg["world"], xchg["mode"]
        a.xchg_err = xchg["err"].data_ptr()
    check(load().hcg_step_tail(ctypes.addressof(a), stream_ptr()), "hcg_step_tail")


def job_bytes() -> int:
    """sizeof(hcg_reduce_job): the stride of a host array of jobs (tests/test_host_cpu.py checks the mirror against the library)."""
    return ctypes.sizeof(ReduceJob)


def reduce_jobs(jobs_addr: int, njobs: int):
    """The slab reductions alone (no loss scale, no update)."""
    step_tail(jobs_addr, njobs)
