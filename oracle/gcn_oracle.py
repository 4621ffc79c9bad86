"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's GCN hot path.

This file is the *oracle* for `hcatgnet_amd`: a torch-only (CPU, eager) restatement of
what `GCN.forward` of the reference executes.  The reference delegates the arithmetic
to the third-party package ``torch_geometric>=2.4.0`` (reference `requirements.txt:14`,
un-pinned, not vendored, not installable here).  The op sequence below is the published
GCNConv algorithm (Kipf & Welling; PyG `GCNConv` + `gcn_norm` + `global_{max,mean}_pool`
semantics) expressed with the stock torch ops PyG's CPU path reduces to
(`index_select` -> scale -> `scatter_add_`, `scatter_reduce_('amax')`, `nn.functional.linear`).

Parity status: **forward is PINNED** against the reference's own committed artefacts
(`results/*/results_GNN/Fold_*/Fold_*/embeddings.csv` produced by the reference's
`utils/utils_model.py:82-111`, from `model_params.pth` + `data/datasets/*/processed/reaction_N.pt`):
see `oracle/make_golden.py`, `tests/golden/` and `tests/test_oracle_golden.py`.
**Backward is NOT pinned by any reference artefact**; its oracle is torch autograd through
this restatement (fp32 and fp64).

Reference call sites followed (file:line into /root/reference):
  * wiring  ............ model/gcn.py:54-76   (conv1 -> LeakyReLU -> convs -> cat[gmp, gap] -> readout)
  * layer shapes ....... model/gcn.py:18-45
  * explain variant .... model/gcn.py:124-140 (tensor-style arguments, optional edge_weight)
  * loss / step ........ utils/utils_model.py:55-70, model/networks.py:28-44
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

LEAKY_SLOPE = 0.01  # nn.LeakyReLU() default, reference model/gcn.py:21,63


def gcn_norm(edge_index: torch.Tensor, num_nodes: int, edge_weight: Optional[torch.Tensor] = None,
             improved: bool = False, dtype=torch.float32) -> Tuple[torch.Tensor, torch.Tensor]:
    """Symmetric GCN normalisation with self loops (call sites model/gcn.py:58,62,127,131).

    Appends one self loop (i, i) per node, ``deg[i] = sum of weights of edges with dst == i``,
    ``norm_e = deg[src]^-1/2 * w_e * deg[dst]^-1/2`` (inf -> 0).

    Self-loop weight: the reference passes ``edge_weight=None`` (model/gcn.py:56) and its
    committed embeddings only reproduce with self-loop weight **1.0** even though
    ``improved=True`` is recorded.  So: fill = 1.0 when ``edge_weight is None``;
    fill = 2.0 only when explicit weights are given together with ``improved``.
    """
    src, dst = edge_index[0], edge_index[1]
    E = src.numel()
    if edge_weight is None:
        fill = 1.0
        w = torch.ones(E, dtype=dtype, device=src.device)
    else:
        fill = 2.0 if improved else 1.0
        w = edge_weight.to(dtype)
    # PyG `add_remaining_self_loops`: explicit (i, i) edges are REMOVED and every node gets exactly one
    # self loop; a node that had an explicit one keeps that edge's weight as its loop weight (published
    # torch_geometric.utils.loop semantics; the reference's molecular graphs contain no self loops, so this
    # branch is not pinned by any artefact).
    keep = src != dst
    loop = torch.arange(num_nodes, dtype=src.dtype, device=src.device)
    loop_w = torch.full((num_nodes,), fill, dtype=dtype, device=src.device)
    if edge_weight is not None and bool((~keep).any()):
        loop_w = loop_w.clone()
        loop_w[src[~keep]] = w[~keep]
    src_f = torch.cat([src[keep], loop])
    dst_f = torch.cat([dst[keep], loop])
    w_f = torch.cat([w[keep], loop_w])
    deg = torch.zeros(num_nodes, dtype=dtype, device=src.device).scatter_add_(0, dst_f, w_f)
    dinv = deg.pow(-0.5)
    dinv = torch.where(torch.isinf(dinv), torch.zeros_like(dinv), dinv)
    norm = dinv[src_f] * w_f * dinv[dst_f]
    return torch.stack([src_f, dst_f]), norm


def gcn_conv(x: torch.Tensor, edge_index: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor,
             edge_weight: Optional[torch.Tensor] = None, improved: bool = False,
             edge_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One GCNConv: ``scatter_add(norm * (x W^T)[src] -> dst) + bias``.

    `edge_mask` ([E], already passed through its sigmoid): explain mode.  PyG's Explainer multiplies every MESSAGE
    by the mask inside `MessagePassing.propagate` -- after gcn_norm, the self loops gcn_norm appended get mask 1
    (published torch_geometric behaviour; the reference uses it through `Explainer(..., edge_mask_type='object')`,
    scripts_experiments/explain_gnn.py:39-50).  No reference artefact pins this branch: parity unpinned."""
    n = x.shape[0]
    ei, norm = gcn_norm(edge_index, n, edge_weight, improved, x.dtype)
    if edge_mask is not None:
        keep = edge_index[0] != edge_index[1]             # same edges gcn_norm kept, then one loop per node
        norm = norm * torch.cat([edge_mask[keep].to(norm.dtype), torch.ones(n, dtype=norm.dtype, device=norm.device)])
    h = F.linear(x, weight)                               # a4: Linear(bias=False)
    hj = h.index_select(0, ei[0])                         # a5: gather of source rows
    m = norm.unsqueeze(1) * hj                            # a6: message
    y = torch.zeros_like(h).scatter_add_(0, ei[1].unsqueeze(1).expand_as(m), m)  # a7
    return y + bias                                       # a8 (bias)


def global_max_pool(x: torch.Tensor, batch: torch.Tensor, num_graphs: int) -> torch.Tensor:
    """Per-graph feature-wise max (torch `scatter_reduce_('amax', include_self=False)`;
    empty graph slot -> 0).  Backward: torch `amax` semantics = even split among ties."""
    idx = batch.unsqueeze(1).expand_as(x)
    return x.new_zeros(num_graphs, x.shape[1]).scatter_reduce(0, idx, x, reduce="amax", include_self=False)


def global_mean_pool(x: torch.Tensor, batch: torch.Tensor, num_graphs: int) -> torch.Tensor:
    idx = batch.unsqueeze(1).expand_as(x)
    s = x.new_zeros(num_graphs, x.shape[1]).scatter_add(0, idx, x)
    cnt = torch.zeros(num_graphs, dtype=x.dtype, device=x.device).scatter_add_(
        0, batch, torch.ones_like(batch, dtype=x.dtype)).clamp_(min=1)
    return s / cnt.unsqueeze(1)


def conv_param_names(n_convolutions: int) -> List[Tuple[str, str]]:
    """State-dict key pairs (weight, bias) of the conv stack."""
    names = [("conv1.lin.weight", "conv1.bias")]
    for i in range(n_convolutions - 1):
        names.append((f"conv_layers.{i}.lin.weight", f"conv_layers.{i}.bias"))
    return names


def readout_param_names(readout_layers: int) -> List[Tuple[str, str]]:
    names = []
    for i in range(readout_layers - 1):
        names.append((f"readout.{i}.0.weight", f"readout.{i}.0.bias"))
    names.append((f"readout.{readout_layers - 1}.weight", f"readout.{readout_layers - 1}.bias"))
    return names


def infer_depths(params: Dict[str, torch.Tensor]) -> Tuple[int, int]:
    n_conv = 1 + sum(1 for k in params if k.startswith("conv_layers.") and k.endswith(".lin.weight"))
    n_read = sum(1 for k in params if k.startswith("readout.") and k.endswith("weight"))
    return n_conv, n_read


def gcn_forward(params: Dict[str, torch.Tensor], x: torch.Tensor, edge_index: torch.Tensor,
                batch: Optional[torch.Tensor] = None, num_graphs: Optional[int] = None,
                edge_weight: Optional[torch.Tensor] = None, improved: bool = False,
                return_intermediates: bool = False, edge_mask: Optional[torch.Tensor] = None):
    """`GCN.forward` (model/gcn.py:54-76).  Returns ``(out[B,n_classes], graph_emb[B,2D])``
    (+ list of post-activation node embeddings per conv when asked).  `edge_mask`: explain mode, applied in EVERY
    conv layer (see gcn_conv)."""
    if batch is None:
        batch = torch.zeros(x.shape[0], dtype=torch.long, device=x.device)
    if num_graphs is None:
        num_graphs = int(batch.max().item()) + 1 if batch.numel() else 0
    n_conv, n_read = infer_depths(params)
    acts = []
    h = x
    for li, (wk, bk) in enumerate(conv_param_names(n_conv)):
        # reference passes edge_weight to conv1 only (model/gcn.py:58 vs :62, and :127 vs :131)
        ew = edge_weight if li == 0 else None
        h = gcn_conv(h, edge_index, params[wk], params[bk], ew, improved, edge_mask)
        h = F.leaky_relu(h, LEAKY_SLOPE)
        acts.append(h)
    emb = torch.cat([global_max_pool(h, batch, num_graphs), global_mean_pool(h, batch, num_graphs)], dim=1)
    z = emb
    rn = readout_param_names(n_read)
    for i, (wk, bk) in enumerate(rn):
        z = F.linear(z, params[wk], params[bk])
        if i < len(rn) - 1:
            z = F.leaky_relu(z, LEAKY_SLOPE)
    if return_intermediates:
        return z, emb, acts
    return z, emb


def rmse_loss(out: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """``sqrt(MSELoss(out, y.unsqueeze(1)))`` -- utils/utils_model.py:64, model/networks.py:32."""
    return torch.sqrt(F.mse_loss(out, y.unsqueeze(1)))


def train_step_grads(params: Dict[str, torch.Tensor], x, edge_index, batch, y, num_graphs=None,
                     dtype=torch.float32, x_requires_grad: bool = False):
    """zero_grad -> forward -> sqrt(MSE) -> backward (utils/utils_model.py:62-65), no optimiser.
    Returns (loss, out, emb, {name: grad}[, dx])."""
    p = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in params.items()}
    xx = x.detach().to(dtype).clone().requires_grad_(x_requires_grad)
    out, emb = gcn_forward(p, xx, edge_index, batch, num_graphs)
    loss = rmse_loss(out, y.to(dtype))
    loss.backward()
    grads = {k: v.grad for k, v in p.items()}
    if x_requires_grad:
        return loss.detach(), out.detach(), emb.detach(), grads, xx.grad
    return loss.detach(), out.detach(), emb.detach(), grads


def make_train_state(params: Dict[str, torch.Tensor], lr: float = 0.01, eps: float = 1e-9, dtype=torch.float32):
    """Leaf parameters + the reference's optimiser: `torch.optim.Adam(self.parameters(), lr=lr, eps=1e-9)`
    (model/networks.py:38).  Returns (p, opt)."""
    p = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in params.items()}
    opt = torch.optim.Adam(list(p.values()), lr=lr, eps=eps)
    return p, opt


def train_step(p: Dict[str, torch.Tensor], opt, x, edge_index, batch, y, num_graphs=None):
    """One full step of the reference's loop (utils/utils_model.py:62-66): zero_grad -> forward -> sqrt(MSE) ->
    backward -> Adam step, in place on `p`.  Returns the loss (detached)."""
    opt.zero_grad()
    out, _ = gcn_forward(p, x, edge_index, batch, num_graphs)
    loss = rmse_loss(out, y.to(out.dtype))
    loss.backward()
    opt.step()
    return loss.detach()
