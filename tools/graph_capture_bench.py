#!/usr/bin/env python3
"""Inference forward replayed from a captured HIP graph (torch.cuda.CUDAGraph around the C-ABI call) against the same
forward launched kernel by kernel.  usage: python tools/graph_capture_bench.py [--config C] [--precision fp32]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpntrackseg_amd import synth
from mpntrackseg_amd.mpn import MOTMPNet


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C")
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    c = synth.CONFIGS[a.config]
    params = synth.model_params(c["d"], c["L"], "sum")
    g = synth.make_knn_graph(seed=1, **c["knn"]) if c.get("knn") else synth.make_graph(c["N"], c["E"], seed=1)
    W = synth.make_weights(params, seed=7, gain=0.6)
    model = MOTMPNet(params)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
    model = model.to(dev).eval()
    model.gemm_precision = a.precision
    x, ei, ea = (torch.from_numpy(g[k]).to(dev) for k in ("x", "edge_index", "edge_attr"))

    class H:
        pass
    holder = H()

    def fwd():
        with torch.no_grad():
            return model.hot_path(x, ei, ea, holder=holder)
    for _ in range(5):
        ref = fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fwd()
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) * 1e3 / a.iters
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fwd()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fwd()
    graph.replay()
    torch.cuda.synchronize()
    ok = bool(torch.equal(out, ref))
    t0 = time.perf_counter()
    for _ in range(a.iters):
        graph.replay()
    torch.cuda.synchronize()
    graph_ms = (time.perf_counter() - t0) * 1e3 / a.iters
    print(json.dumps({"config": a.config, "precision": a.precision, "eager_ms": eager_ms, "graph_replay_ms": graph_ms,
                      "identical_output": ok}))


if __name__ == "__main__":
    main()
