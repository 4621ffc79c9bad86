"""On-disk formats either side of the path: the reference's processed graphs in, its
`embeddings.csv` out.

* IN  -- `data/datasets/<set>/processed/reaction_<N>.pt`, written by reference data/rhcaa.py:78-92 with
  `torch.save(Data(x, edge_index, edge_attr, y, ...))` and read back by data/datasets.py:74-78 with `torch.load`.
  Those files pickle `torch_geometric` classes, so `torch.load(weights_only=True)` refuses them and a plain
  `torch.load` would execute the pickle.  This reader does neither: a torch checkpoint is a zip container whose
  members `<root>/data/<k>` are the raw little-endian tensor storages in the order the tensors were pickled; for
  the reference's graphs that order is x (float32 [N, F]), edge_index (int64 [2, E]), edge_attr (float32 [E, 7]),
  y (float32 [1]).  Only those members are read (`zipfile` + `numpy.frombuffer`); `data.pkl` is never opened.
  F is not stored in a raw storage: pass the model's `n_node_features`.  Every interpretation is cross-checked
  (byte sizes: 4NF / 16E / 28E / 4, index range) and a mismatch raises.
* OUT -- `embeddings.csv` as reference utils/utils_model.py:95-106,172-203 writes it: pandas `to_csv` of a frame
  with columns 0..2D-1 (graph_emb = [max, mean]), `ddG_exp`, `ddG_pred`, `index`, `set`.
"""
from __future__ import annotations

import os
import re
import zipfile
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch

from .batch import Data

EDGE_ATTR_WIDTH = 7    # bond one-hots, reference data/rhcaa.py:134-166


def read_reaction_graph(path: str, n_node_features: int, idx: Optional[int] = None) -> Data:
    """One `reaction_<N>.pt` -> Data(x, edge_index, edge_attr, y, idx) without unpickling anything."""
    F = int(n_node_features)
    if F <= 0:
        raise ValueError("n_node_features must be positive")
    with zipfile.ZipFile(path) as z:
        members: Dict[str, zipfile.ZipInfo] = {}
        for info in z.infolist():
            parts = info.filename.split("/", 1)
            if len(parts) == 2:
                members[parts[1]] = info
        need = ["data/0", "data/1", "data/2", "data/3"]
        missing = [m for m in need if m not in members]
        if missing:
            raise ValueError(f"{path}: not a processed reaction graph (missing storage members {missing})")
        bx, be = z.read(members["data/0"]), z.read(members["data/1"])
        ba, by = z.read(members["data/2"]), z.read(members["data/3"])
    if len(bx) % (4 * F):
        raise ValueError(f"{path}: x storage of {len(bx)} bytes is not a whole number of {F}-float rows")
    if len(be) % 16:
        raise ValueError(f"{path}: edge_index storage of {len(be)} bytes is not int64 [2, E]")
    x = np.frombuffer(bx, dtype="<f4").reshape(-1, F)
    ei = np.frombuffer(be, dtype="<i8").reshape(2, -1)
    n, e = x.shape[0], ei.shape[1]
    if len(ba) != 4 * EDGE_ATTR_WIDTH * e:
        raise ValueError(f"{path}: edge_attr storage {len(ba)} bytes != {4 * EDGE_ATTR_WIDTH} * E ({e}): wrong n_node_features?")
    if len(by) != 4:
        raise ValueError(f"{path}: y storage is {len(by)} bytes, expected one float32")
    if e and (int(ei.min()) < 0 or int(ei.max()) >= n):
        raise ValueError(f"{path}: edge_index refers to a node outside [0, {n})")
    if idx is None:
        m = re.search(r"reaction_(\d+)\.pt$", os.path.basename(path))
        idx = int(m.group(1)) if m else -1
    return Data(x=torch.from_numpy(x.copy()), edge_index=torch.from_numpy(ei.copy()),
                edge_attr=torch.from_numpy(np.frombuffer(ba, dtype="<f4").reshape(e, EDGE_ATTR_WIDTH).copy()),
                y=torch.from_numpy(np.frombuffer(by, dtype="<f4").copy()), idx=int(idx))


def load_processed_dir(processed_dir: str, n_node_features: int, indices: Optional[Iterable[int]] = None) -> List[Data]:
    """All (or the given) `reaction_<N>.pt` of a `processed/` directory, ordered by N -- the dataset the reference's
    `reaction_graph.get(idx)` serves one file at a time (data/datasets.py:74-78)."""
    if indices is None:
        found = []
        for name in os.listdir(processed_dir):
            m = re.fullmatch(r"reaction_(\d+)\.pt", name)
            if m:
                found.append(int(m.group(1)))
        indices = sorted(found)
    return [read_reaction_graph(os.path.join(processed_dir, f"reaction_{i}.pt"), n_node_features, i) for i in indices]


def embeddings_frame(model, loaders: Dict[str, object], device=None):
    """The frame the reference saves as `embeddings.csv`: `predict_network(..., True)` over each named loader
    ('training' / 'val' / 'test' in network_report, utils/utils_model.py:172-199), a `set` column, concatenated."""
    import pandas as pd
    from .train import predict_network
    frames = []
    for name, loader in loaders.items():
        _, _, _, emb = predict_network(model, loader, True, device=device)
        emb["set"] = name
        frames.append(emb)
    return pd.concat(frames, axis=0)


def write_embeddings_csv(model, loaders: Dict[str, object], path: str, device=None):
    frame = embeddings_frame(model, loaders, device=device)
    frame.to_csv(path)          # reference: emb_all.to_csv("{}/embeddings.csv".format(log_dir))
    return frame


def read_embeddings_csv(path: str):
    """-> dict(index [G] int64, emb [G, 2D] float32, pred [G], exp [G], set [G] str) from a reference-format file
    (extra columns such as the reference's tSNE1 / tSNE2 are ignored)."""
    import pandas as pd
    df = pd.read_csv(path, index_col=0)
    cols = [c for c in df.columns if str(c).isdigit()]
    cols.sort(key=lambda c: int(c))
    return dict(index=df["index"].to_numpy(np.int64), emb=df[cols].to_numpy(np.float32),
                pred=df["ddG_pred"].to_numpy(np.float32), exp=df["ddG_exp"].to_numpy(np.float32),
                set=df["set"].to_numpy() if "set" in df.columns else None)
