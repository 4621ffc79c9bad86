#!/usr/bin/env python3
"""One inner run of the reference's nested cross-validation (scripts_experiments/train_GNN.py:66-134) on MI355X, with
hcatgnet_amd standing in for `make_network` / `train_network` / `eval_network` / `predict_network` and the PyG loaders.

    python examples/train_like_reference.py                          # synthetic reaction-sized graphs (57-117 atoms, F = 25)
    python examples/train_like_reference.py --processed DIR --n-node-features 25
                                                                     # a `processed/` directory of the reference's own
                                                                     # reaction_N.pt files (read without unpickling)
The loop body is the reference's: per epoch train / validate / test, every 5th epoch `model.scheduler.step(val_loss)`
(ReduceLROnPlateau) + early-stopping bookkeeping + a copy of the best state-dict, then predictions and `embeddings.csv`.
"""
import argparse
import os
import sys
import time
from copy import deepcopy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import hcatgnet_amd as H  # noqa: E402
from hcatgnet_amd import io as hio, synth  # noqa: E402
from hcatgnet_amd.train import eval_network, predict_network, train_network  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--processed", default=None, help="directory with the reference's reaction_N.pt files")
    ap.add_argument("--n-node-features", type=int, default=25)
    ap.add_argument("--graphs", type=int, default=668, help="synthetic dataset size (the reference's learning set has 668)")
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--early-stopping", type=int, default=6)
    ap.add_argument("--out", default=None, help="where to write embeddings.csv")
    args = ap.parse_args(argv)
    device = torch.device("cuda")
    opt = H.default_options()

    if args.processed:
        graphs = hio.load_processed_dir(args.processed, args.n_node_features)
    else:
        graphs = synth.make_config("REAL", num_graphs=args.graphs).as_graph_list()
    F = graphs[0].x.shape[1]
    order = np.random.default_rng(opt.global_seed).permutation(len(graphs))
    n_test, n_val = len(graphs) // 10, len(graphs) // 10           # one outer / inner fold of the 10 x 9 scheme
    parts = {"test": order[:n_test], "val": order[n_test:n_test + n_val], "training": order[n_test + n_val:]}
    stores = {k: H.DeviceGraphStore([graphs[i] for i in v], device) for k, v in parts.items()}   # datasets resident in HBM
    train_loader = H.DeviceLoader(stores["training"], batch_size=opt.batch_size, shuffle=True, seed=opt.global_seed)
    val_loader = H.DeviceLoader(stores["val"], batch_size=opt.batch_size)
    test_loader = H.DeviceLoader(stores["test"], batch_size=opt.batch_size)

    model = H.make_network("GCN", opt, F).to(device)
    val_best, best_epoch, stall, best_params = float("inf"), 0, 0, deepcopy(model.state_dict())
    # an epoch here is ~2 ms of host work; a full cyclic-GC pass over a process that imported torch is ~85 ms: keep what
    # exists now (torch, the datasets, the model) out of the collector's sight
    import gc
    gc.collect()
    gc.freeze()
    t0 = time.time()
    for epoch in range(args.epochs):
        if stall > args.early_stopping:
            print("Early stopping limit reached")
            break
        train_loss = train_network(model, train_loader, device)
        val_loss = eval_network(model, val_loader, device)
        test_loss = eval_network(model, test_loader, device)
        print(f"Epoch {epoch:03d} | Train loss: {train_loss:.3f} | Validation loss: {val_loss:.3f} | Test loss: {test_loss:.3f}")
        if epoch % 5 == 0:
            model.scheduler.step(val_loss)
            if val_loss < val_best:
                val_best, best_epoch, stall = val_loss, epoch, 0
                best_params = deepcopy(model.state_dict())
            else:
                stall += 1
    torch.cuda.synchronize()
    print(f"training time: {time.time() - t0:.2f} s, best validation loss {val_best:.4f} at epoch {best_epoch}")

    model.load_state_dict(best_params)
    y_pred, y_true, idx = predict_network(model, test_loader)
    print(f"test RMSE {float(np.sqrt(np.mean((y_pred - y_true) ** 2))):.4f} over {len(idx)} graphs")
    if args.out:
        frame = hio.write_embeddings_csv(model, {"training": H.DeviceLoader(stores["training"], batch_size=opt.batch_size),
                                                 "val": val_loader, "test": test_loader}, args.out)
        print(f"wrote {args.out}: {frame.shape[0]} rows x {frame.shape[1]} columns")
    return val_best


if __name__ == "__main__":
    main()
