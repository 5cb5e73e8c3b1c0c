#!/usr/bin/env python3
"""Host-side cost of one TrainStep call at a KITTIMOTS-size graph (cfg-D): the time the Python side needs to ENQUEUE a step (no
synchronisation inside the loop) against the step's device time, and the cost of building the native model description
(MOTMPNet.c_model).  usage: python tools/diag/host_cost.py [config]"""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from mpntrackseg_amd import synth, train as mtrain  # noqa: E402
from mpntrackseg_amd.mpn import MOTMPNet, _prepared  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "D"
    c = dict(synth.CONFIGS[cfg])
    dev = torch.device("cuda:0")
    params = synth.model_params(c["d"], c["L"], "sum")
    W = synth.make_weights(params, seed=7)
    g = synth.make_knn_graph(seed=1, **c["knn"]) if c.get("knn") else synth.make_graph(c["N"], c["E"], seed=1)
    model = MOTMPNet(params)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
    model = model.to(dev)
    x, ei, ea = (torch.from_numpy(g[k]).to(dev) for k in ("x", "edge_index", "edge_attr"))

    class H:
        pass
    h = H()
    _prepared(ei, x.shape[0], h)
    step = mtrain.TrainStep(model, world_size=1)
    for _ in range(30):
        step(x, ei, ea, holder=h)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        step(x, ei, ea, holder=h)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(n):
        model.c_model([], n_edges=ea.shape[0])
    t_cm = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(n):
        model.c_model([], grads=step.bucket.views, n_edges=ea.shape[0])
    t_cmg = time.perf_counter() - t0
    print("cfg-%s: enqueue %.1f us per step, step (device) %.1f us, c_model %.1f us, c_model with grads %.1f us" %
          (cfg, t_enq / n * 1e6, t_all / n * 1e6, t_cm / n * 1e6, t_cmg / n * 1e6))


if __name__ == "__main__":
    main()
