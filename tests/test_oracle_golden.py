"""The oracle (oracle/gcn_oracle.py) against the reference's own golden vectors.

Fixtures: tests/golden/golden_*.npz, derived by oracle/make_golden.py from the reference's
committed `model_params.pth` + processed graphs + `embeddings.csv` (reference
utils/utils_model.py:82-111).  tests/golden/ORACLE_PIN.txt records the same check over ALL
540 CSVs / 348 372 graphs (run once in the build container).
"""
import os

import pytest
import torch

from oracle import gcn_oracle
from tests.helpers import golden_files, load_golden, rel_inf

FILES = golden_files()


def test_fixtures_present():
    assert len(FILES) >= 4


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p) for p in FILES])
def test_oracle_reproduces_reference_embeddings(path):
    g = load_golden(path)
    with torch.no_grad():
        out, emb = gcn_oracle.gcn_forward(g["params"], g["x"], g["edge_index"], g["batch"], g["num_graphs"])
    # tolerance: 1e-5 relative (north_star); observed: bit-exact embeddings, <= 8e-6 abs on predictions
    assert rel_inf(emb, g["ref_emb"]) <= 1e-5
    assert (out[:, 0] - g["ref_pred"]).abs().max().item() <= 5e-5


@pytest.mark.parametrize("path", FILES[:2], ids=[os.path.basename(p) for p in FILES[:2]])
def test_self_loop_weight_two_does_not_reproduce(path):
    """`improved=True` is a no-op in the reference's results (fill must be 1.0)."""
    g = load_golden(path)
    ew = torch.ones(g["edge_index"].shape[1])
    with torch.no_grad():
        _, emb2 = gcn_oracle.gcn_forward(g["params"], g["x"], g["edge_index"], g["batch"], g["num_graphs"],
                                         edge_weight=ew, improved=True)
        _, emb1 = gcn_oracle.gcn_forward(g["params"], g["x"], g["edge_index"], g["batch"], g["num_graphs"],
                                         edge_weight=ew, improved=False)
    assert rel_inf(emb1, g["ref_emb"]) <= 1e-5          # explicit unit weights, fill 1 == no weights
    assert rel_inf(emb2, g["ref_emb"]) > 1e-2           # fill 2 is visibly wrong


def test_oracle_pin_record():
    """The full-corpus pin must have been recorded with a passing bound."""
    p = os.path.join(os.path.dirname(FILES[0]), "ORACLE_PIN.txt")
    head = open(p).read().splitlines()[1]
    worst = float(head.split("max|emb|")[1].split()[0])
    assert "files 540" in head and worst <= 1e-5


def test_fp64_matches_fp32_oracle():
    g = load_golden(FILES[0])
    p64 = {k: v.double() for k, v in g["params"].items()}
    with torch.no_grad():
        o32, e32 = gcn_oracle.gcn_forward(g["params"], g["x"], g["edge_index"], g["batch"], g["num_graphs"])
        o64, e64 = gcn_oracle.gcn_forward(p64, g["x"].double(), g["edge_index"], g["batch"], g["num_graphs"])
    assert rel_inf(e32, e64) <= 1e-5 and rel_inf(o32, o64) <= 1e-5


def test_max_pool_backward_splits_ties_evenly():
    """SURVEY hard part: torch amax backward = even split among tied rows (the oracle's rule)."""
    x = torch.tensor([[1.0, 2.0], [1.0, 0.0], [0.5, 2.0]], requires_grad=True)
    b = torch.zeros(3, dtype=torch.long)
    gcn_oracle.global_max_pool(x, b, 1).sum().backward()
    assert torch.allclose(x.grad, torch.tensor([[0.5, 0.5], [0.5, 0.0], [0.0, 0.5]]))
