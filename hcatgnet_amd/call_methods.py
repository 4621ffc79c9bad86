"""Factory with the reference's name and contract (reference call_methods.py:7-12)."""
import argparse


def make_network(network_name: str, opt: argparse.Namespace, n_node_features: int):
    if network_name == "GCN":
        from .gcn import GCN
        return GCN(opt=opt, n_node_features=n_node_features)
    else:
        raise ValueError(f"Network {network_name} not implemented")


def default_options(**overrides) -> argparse.Namespace:
    """The hot-path-relevant defaults of the reference's `BaseOptions`
    (options/base_options.py:178-274, :369-374)."""
    d = dict(n_classes=1, n_convolutions=2, readout_layers=2, embedding_dim=64, improved=True,
             problem_type="regression", optimizer="Adam", lr=0.01, scheduler="ReduceLROnPlateau",
             step_size=7, gamma=0.7, min_lr=1e-8, batch_size=40, global_seed=20232023)
    d.update(overrides)
    return argparse.Namespace(**d)
