"""Shared test helpers: golden-fixture loading and error metrics (tests only)."""
import glob
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "golden_*.npz")))


def load_golden(path):
    """-> dict(params, x, edge_index[int64 2,E], batch[int64 N], node_ptr, ref_emb, ref_pred, y)."""
    z = np.load(path)
    params = {k[len("param/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param/")}
    node_ptr = z["node_ptr"]; edge_ptr = z["edge_ptr"]
    ei_local = z["edge_index_local"].astype(np.int64)
    ei = ei_local.copy()
    batch = np.zeros(node_ptr[-1], np.int64)
    for g in range(len(node_ptr) - 1):
        ei[:, edge_ptr[g]:edge_ptr[g + 1]] += node_ptr[g]
        batch[node_ptr[g]:node_ptr[g + 1]] = g
    return dict(params=params, x=torch.from_numpy(z["x"]), edge_index=torch.from_numpy(ei),
                batch=torch.from_numpy(batch), node_ptr=node_ptr, edge_ptr=edge_ptr,
                ref_emb=torch.from_numpy(z["ref_emb"]), ref_pred=torch.from_numpy(z["ref_pred"]),
                y=torch.from_numpy(z["y"]), num_graphs=len(node_ptr) - 1)


def rel_inf(a: torch.Tensor, ref: torch.Tensor, floor: float = 1e-30) -> float:
    """||a - ref||_inf / max(||ref||_inf, floor)  (SURVEY 8d parity metric).

    `floor` is only raised for the model OUTPUT (predictions, O(1..10) kJ/mol): a prediction is a
    64-term dot product of O(1) terms that may cancel to ~1e-2, and with a single graph in the batch
    ||ref||_inf is that one cancelled value; floor=1.0 turns the bound into abs <= 1e-5 there
    (the golden test applies abs <= 5e-5 to the reference's own ddG_pred for the same reason)."""
    a = a.detach().double().cpu(); ref = ref.detach().double().cpu()
    den = ref.abs().max().item() if ref.numel() else 0.0
    num = (a - ref).abs().max().item() if ref.numel() else 0.0
    return num / max(den, floor)


def elementwise_ok(a: torch.Tensor, ref: torch.Tensor, rtol=1e-5, floor=1e-3) -> bool:
    """abs(d) <= rtol * max(abs(ref), floor)  (SURVEY 8d spot-check form)."""
    a = a.detach().double().cpu(); ref = ref.detach().double().cpu()
    return bool(((a - ref).abs() <= rtol * ref.abs().clamp_min(floor)).all())
