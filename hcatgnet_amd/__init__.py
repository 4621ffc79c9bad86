"""hcatgnet_amd -- MI355X-native drop-in for the GCN message-passing path of EdAguilarB/hcatgnet.

Public surface (mirrors the reference's names for this path):
    make_network("GCN", opt, n_node_features)   reference call_methods.py:7-12
    GCN, GCN_explain, BaseNetwork               reference model/gcn.py, model/networks.py
    Data, Batch, collate, DataLoader            stand-ins for the PyG containers the loops touch
    BatchPlan                                   per-batch CSR / gcn_norm plan (GPU)
    DeviceGraphStore, DeviceLoader              dataset resident in HBM, batches collated on the GPU
    DataParallelGCN                             one-process-per-GPU gradient all-reduce (RCCL)
    train.FusedTrainStep, train.train_network / eval_network / predict_network
                                                the reference's loops (utils/utils_model.py:55-111); the step as 6 launches
    io.load_processed_dir / write_embeddings_csv the reference's on-disk formats (reaction_N.pt in, embeddings.csv out)
    explain.set_masks / clear_masks             explain-mode edge masks (PyG Explainer hooks)
Compute lives in csrc/libhcatgnet_hip.so (hand-written HIP for gfx950) behind include/hcatgnet_hip.h.
"""
from .batch import Batch, Data, DataLoader, collate  # noqa: F401
from .call_methods import default_options, make_network  # noqa: F401
from .gcn import GCN, GCN_explain, GCNConv  # noqa: F401
from .networks import BaseNetwork  # noqa: F401
from .plan import BatchPlan  # noqa: F401
from .store import DeviceGraphStore, DeviceLoader  # noqa: F401

__all__ = ["make_network", "default_options", "GCN", "GCN_explain", "GCNConv", "BaseNetwork", "Data", "Batch",
           "collate", "DataLoader", "BatchPlan", "DeviceGraphStore", "DeviceLoader"]
