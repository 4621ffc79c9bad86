"""BaseNetwork: same constructor contract and attributes as the reference's `model/networks.py`
(:8-61): stores the `opt` dimensions, seeds globally, and attaches `loss` / `optimizer` /
`scheduler` to the module, because the training loop reads them from the model
(utils/utils_model.py:62-66, scripts_experiments/train_GNN.py:88)."""
from __future__ import annotations

import argparse
import random

import numpy as np
import torch
import torch.nn as nn


class MSELoss(nn.Module):
    """`nn.MSELoss()` (mean reduction) with the reference's call contract `loss(out, y.unsqueeze(1))`
    (utils/utils_model.py:64); on MI355X tensors it is one HIP launch each way (csrc/loss.hip) instead
    of torch's elementwise + reduction chain.  Same-shape float32 inputs only."""

    def forward(self, inp, target):
        from . import functional as HF
        return HF.mse_loss(inp, target)


class BaseNetwork(nn.Module):
    def __init__(self, opt: argparse.Namespace, n_node_features: int):
        super().__init__()
        self._name = "BaseNetwork"
        self._opt = opt
        self.n_node_features = n_node_features
        self._n_classes = opt.n_classes
        self.n_convolutions = opt.n_convolutions
        self.embedding_dim = opt.embedding_dim
        self.readout_layers = opt.readout_layers
        self._seed_everything(opt.global_seed)

    def forward(self):
        raise NotImplementedError

    @property
    def name(self):
        return self._name

    def _make_loss(self, problem_type, mae=None):
        if problem_type == "classification":
            self.loss = nn.CrossEntropyLoss()
        elif problem_type == "regression" and mae is None:
            self.loss = MSELoss()
        else:
            raise ValueError(f"Problem type {problem_type} not supported")

    def _make_optimizer(self, optimizer, lr):
        if optimizer == "Adam":
            # torch.optim.Adam(lr, eps=1e-9) in the reference (model/networks.py:38); same rule, one HIP launch
            from .optim import FusedAdam
            self.optimizer = FusedAdam(self.parameters(), lr=lr, eps=1e-9)
        elif optimizer == "SGD":
            self.optimizer = torch.optim.SGD(self.parameters(), lr=lr)
        elif optimizer == "rmsprop":
            self.optimizer = torch.optim.RMSprop(self.parameters(), lr=lr)
        else:
            raise NotImplementedError(f"Optimizer type {optimizer} not implemented")

    def _make_scheduler(self, scheduler, step_size, gamma, min_lr):
        sched = torch.optim.lr_scheduler
        if scheduler == "StepLR":
            self.scheduler = sched.StepLR(self.optimizer, step_size=step_size, gamma=gamma)
        elif scheduler == "MultiStepLR":
            self.scheduler = sched.MultiStepLR(self.optimizer, milestones=step_size, gamma=gamma)
        elif scheduler == "ExponentialLR":
            self.scheduler = sched.ExponentialLR(self.optimizer, gamma=gamma)
        elif scheduler == "ReduceLROnPlateau":
            self.scheduler = sched.ReduceLROnPlateau(self.optimizer, mode="min", factor=gamma, patience=step_size,
                                                     min_lr=min_lr)
        else:
            raise NotImplementedError(f"Scheduler type {scheduler} not implemented")

    def _seed_everything(self, seed):
        # what torch_geometric.seed.seed_everything does (reference model/networks.py:58-61)
        random.seed(seed)
        np.random.seed(seed % (2 ** 32))
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)
