#!/usr/bin/env python3
"""Host enqueue time against GPU time of one training step (are the small configurations host-bound?): `steps` steps are enqueued
without a synchronisation; host_ms = time until the loop returns, total_ms = until the device is idle.
usage: python tools/diag/host_vs_gpu.py [--config D] [--steps 200]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpntrackseg_amd import synth, train as mtrain
from mpntrackseg_amd.mpn import MOTMPNet, _prepared


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="D")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--mode", default="train")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    c = synth.CONFIGS[a.config]
    params = synth.model_params(c["d"], c["L"], "sum")
    g = synth.make_knn_graph(seed=1, **c["knn"]) if c.get("knn") else synth.make_graph(c["N"], c["E"], seed=1)
    W = synth.make_weights(params, seed=7, gain=0.6)
    model = MOTMPNet(params)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
    model = model.to(dev)
    x, ei, ea = (torch.from_numpy(g[k]).to(dev) for k in ("x", "edge_index", "edge_attr"))

    class H:
        pass
    holder = H()
    _prepared(ei, x.shape[0], holder)
    if a.mode == "train":
        model.train()
        stepper = mtrain.TrainStep(model, world_size=1)
        step = lambda: stepper(x, ei, ea, holder=holder)
    else:
        model.eval()
        model.keep_packed_weights = True

        def step():
            with torch.no_grad():
                return model.hot_path(x, ei, ea, holder=holder)
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(json.dumps({"config": a.config, "mode": a.mode, "host_ms_per_step": (t1 - t0) * 1e3 / a.steps, "total_ms_per_step": (t2 - t0) * 1e3 / a.steps}))


if __name__ == "__main__":
    main()
