#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz from the reference's artefacts.

Runs ONLY in the build container (needs /root/reference).  Nothing here travels to the GPU
box except the small derived `.npz` fixtures it writes (data: inputs + expected outputs).

What is read, and how (nothing from any reference file is executed or unpickled):
  * `results/<set>/results_GNN/Fold_o_test_set/Fold_i_val_set/model_params.pth`
        -> `torch.load(weights_only=True)`                     (trained weights, 8 tensors)
  * `.../embeddings.csv` (written by reference utils/utils_model.py:95-106,199-205)
        -> CSV text: columns 0..127 = graph_emb [max, mean], ddG_pred = model output,
           ddG_exp = y, index = N of reaction_N.pt                       (expected outputs)
  * `data/datasets/<ds>/processed/reaction_N.pt` (written by reference data/rhcaa.py:78-92)
        `torch.load(weights_only=True)` REFUSES these files (their `data.pkl` names
        torch_geometric classes), and unpickling them is not allowed.  They are zip
        containers whose members `data/0` and `data/1` are the raw little-endian tensor
        storages of `x` (float32 [N, F]) and `edge_index` (int64 [2, E]); this script reads
        exactly those two members (plus `data/2`'s *size* as a consistency check and
        `data/3` = y, 4 bytes) with `zipfile` + `numpy.frombuffer`.  `data.pkl` is never
        opened.  F is fixed per dataset by the weights (`conv1.lin.weight.shape[1]`); the
        interpretation is cross-checked three ways: byte sizes (x: 4NF, edge_index: 16E,
        edge_attr: 28E), index range (0 <= id < N), and y == the CSV's ddG_exp.

Usage:
    python oracle/make_golden.py                # write tests/golden/golden_*.npz (few graphs)
    python oracle/make_golden.py --verify-all   # check the oracle on EVERY graph of EVERY
                                                # embeddings.csv; writes tests/golden/ORACLE_PIN.txt
"""
from __future__ import annotations

import argparse
import csv
import os
import sys
import zipfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import gcn_oracle  # noqa: E402

REF = "/root/reference"
# results set -> (dataset dir holding the graphs, results set holding the weights)
SETS = {
    "learning": ("rhcaa_learning", "learning"),
    "all_data": ("all_data", "all_data"),
    "half_data": ("half_data", "half_data"),
    "biaryl": ("biAryl", "biaryl"),
    "final_test": ("rhcaa_final_test", "learning"),          # predict_test.py:55-70
    "half_final_test": ("half_final_test", "half_data"),
}
# (results set, outer, inner) used for the committed fixtures
PICKS = [("learning", 8, 10), ("all_data", 1, 2), ("biaryl", 3, 5), ("half_data", 10, 1), ("final_test", 8, 10)]
GRAPHS_PER_PICK = 8


def read_graph_raw(path: str, n_feat: int):
    """x, edge_index, y from the raw storage members of a torch zip container (no unpickling)."""
    with zipfile.ZipFile(path) as z:
        members = {i.filename.split("/", 1)[1]: i for i in z.infolist()}
        bx = z.read(members["data/0"])
        be = z.read(members["data/1"])
        by = z.read(members["data/3"])
        attr_bytes = members["data/2"].file_size
    x = np.frombuffer(bx, dtype="<f4")
    if x.size % n_feat:
        raise ValueError(f"{path}: x storage {x.size} floats not divisible by F={n_feat}")
    x = x.reshape(-1, n_feat)
    ei = np.frombuffer(be, dtype="<i8")
    if ei.size % 2:
        raise ValueError(f"{path}: edge_index storage odd")
    ei = ei.reshape(2, -1)
    n, e = x.shape[0], ei.shape[1]
    if attr_bytes != e * 7 * 4:
        raise ValueError(f"{path}: edge_attr bytes {attr_bytes} != 28*E ({e})")
    if e and (ei.min() < 0 or ei.max() >= n):
        raise ValueError(f"{path}: edge index out of range")
    y = np.frombuffer(by, dtype="<f4")
    return x.copy(), ei.copy(), float(y[0])


def read_embeddings_csv(path: str):
    with open(path, newline="") as f:
        rd = csv.reader(f)
        hdr = next(rd)
        col = {name: i for i, name in enumerate(hdr)}
        emb_cols = [col[str(k)] for k in range(128) if str(k) in col]
        rows = []
        for r in rd:
            rows.append((int(float(r[col["index"]])),
                         np.array([np.float32(r[c]) for c in emb_cols], dtype=np.float32),
                         np.float32(r[col["ddG_pred"]]), np.float32(r[col["ddG_exp"]]), r[col["set"]]))
    return rows


def fold_dir(rset: str, outer: int, inner: int) -> str:
    return os.path.join(REF, "results", rset, "results_GNN", f"Fold_{outer}_test_set", f"Fold_{inner}_val_set")


def row_dataset(rset: str, row_set: str) -> str:
    """Dataset dir a CSV row's `index` points into.  In the *_final_test results the
    'training'/'val' rows are the training set's graphs (loaders re-loaded from the weight
    set, predict_test.py:58-61) and only the 'test' rows come from the final-test dataset."""
    ds, wset = SETS[rset]
    if wset != rset and row_set != "test":
        return SETS[wset][0]
    return ds


def graph_path(rset: str, row) -> str:
    return os.path.join(REF, "data", "datasets", row_dataset(rset, row[4]), "processed", f"reaction_{row[0]}.pt")


def load_fold(rset: str, outer: int, inner: int):
    ds, wset = SETS[rset]
    params = torch.load(os.path.join(fold_dir(wset, outer, inner), "model_params.pth"),
                        weights_only=True, map_location="cpu")
    rows = read_embeddings_csv(os.path.join(fold_dir(rset, outer, inner), "embeddings.csv"))
    return ds, params, rows


def eval_rows(rset, params, rows):
    """Run the oracle on all graphs of `rows` as ONE batch; return per-graph relative errors."""
    n_feat = params["conv1.lin.weight"].shape[1]
    xs, eis, bs, off = [], [], [], 0
    for gi, row in enumerate(rows):
        idx, yexp = row[0], row[3]
        x, ei, y = read_graph_raw(graph_path(rset, row), n_feat)
        if abs(y - float(yexp)) > 1e-6 * max(1.0, abs(y)):
            raise ValueError(f"{rset}/reaction_{idx}: y {y} != ddG_exp {yexp}")
        xs.append(x); eis.append(ei + off); bs.append(np.full(x.shape[0], gi, np.int64)); off += x.shape[0]
    x = torch.from_numpy(np.concatenate(xs)); ei = torch.from_numpy(np.concatenate(eis, 1)); b = torch.from_numpy(np.concatenate(bs))
    with torch.no_grad():
        out, emb = gcn_oracle.gcn_forward(params, x, ei, b, len(rows))
    ref_emb = torch.from_numpy(np.stack([r[1] for r in rows])); ref_out = torch.tensor([float(r[2]) for r in rows])
    emb_rel = ((emb - ref_emb).abs().amax(1) / ref_emb.abs().amax(1).clamp_min(1e-30))
    out_abs = (out[:, 0] - ref_out).abs()
    return emb_rel, out_abs


def write_fixtures():
    outdir = os.path.join(REPO, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    for rset, outer, inner in PICKS:
        ds, params, rows = load_fold(rset, outer, inner)
        n_feat = params["conv1.lin.weight"].shape[1]
        # deterministic spread over the CSV (first, last and evenly spaced rows)
        sel = sorted(set(np.linspace(0, len(rows) - 1, GRAPHS_PER_PICK).astype(int).tolist()))
        if SETS[rset][1] != rset:   # final-test sets: take the rows of the unseen test graphs
            test_rows = [i for i, r in enumerate(rows) if r[4] == "test"]
            sel = [test_rows[i] for i in sorted(set(np.linspace(0, len(test_rows) - 1, GRAPHS_PER_PICK).astype(int).tolist()))]
        arrs = {f"param/{k}": v.numpy() for k, v in params.items()}
        ptr, eptr, xs, eis, emb, pred, yexp, idxs = [0], [0], [], [], [], [], [], []
        for r in sel:
            idx, e, p, yv, _ = rows[r]
            x, ei, _ = read_graph_raw(graph_path(rset, rows[r]), n_feat)
            xs.append(x); eis.append(ei.astype(np.int32)); ptr.append(ptr[-1] + x.shape[0]); eptr.append(eptr[-1] + ei.shape[1])
            emb.append(e); pred.append(p); yexp.append(yv); idxs.append(idx)
        arrs.update(x=np.concatenate(xs), edge_index_local=np.concatenate(eis, 1), node_ptr=np.array(ptr, np.int64),
                    edge_ptr=np.array(eptr, np.int64), ref_emb=np.stack(emb), ref_pred=np.array(pred, np.float32),
                    y=np.array(yexp, np.float32), reaction_index=np.array(idxs, np.int64))
        name = f"golden_{rset}_o{outer}_i{inner}.npz"
        np.savez_compressed(os.path.join(outdir, name), **arrs)
        print("wrote", name, "graphs", len(sel), "nodes", ptr[-1], "edges", eptr[-1],
              "bytes", os.path.getsize(os.path.join(outdir, name)))


def verify_all():
    lines, worst_e, worst_o, n_graphs, n_files = [], 0.0, 0.0, 0, 0
    for rset in SETS:
        base = os.path.join(REF, "results", rset, "results_GNN")
        for outer in range(1, 11):
            for inner in range(1, 11):
                d = fold_dir(rset, outer, inner)
                if not os.path.isfile(os.path.join(d, "embeddings.csv")):
                    continue
                ds, params, rows = load_fold(rset, outer, inner)
                emb_rel, out_abs = eval_rows(rset, params, rows)
                n_graphs += len(rows); n_files += 1
                worst_e = max(worst_e, float(emb_rel.max())); worst_o = max(worst_o, float(out_abs.max()))
                lines.append(f"{rset:16s} o{outer:<2d} i{inner:<2d} graphs {len(rows):4d}  emb_rel_max {float(emb_rel.max()):.3e}  pred_abs_max {float(out_abs.max()):.3e}")
        print(rset, "done", n_files, "files so far; worst emb rel", worst_e, flush=True)
    hdr = [
        "Oracle pin: oracle/gcn_oracle.py vs EVERY embeddings.csv of the reference (oracle/make_golden.py --verify-all)",
        f"files {n_files}  graphs {n_graphs}  worst per-graph max|d emb|/max|emb| {worst_e:.3e}  worst |d pred| {worst_o:.3e}",
        "(CSV values are float32 shortest-repr; tolerance gate in tests: emb rel <= 1e-5, pred abs <= 5e-5)", ""]
    with open(os.path.join(REPO, "tests", "golden", "ORACLE_PIN.txt"), "w") as f:
        f.write("\n".join(hdr + lines) + "\n")
    print("\n".join(hdr))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--verify-all", action="store_true")
    a = ap.parse_args()
    if a.verify_all:
        verify_all()
    else:
        write_fixtures()
