#!/usr/bin/env python3
"""Steady-state cost of mpnhip_graph_prep (one stable sort family per NEW graph): what a training loop that sees a
different graph every step pays on top of bench.py's step time (which caches the prep of its one graph)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpntrackseg_amd import capi, synth

def main():
    dev = torch.device("cuda:0")
    out = {}
    for name in ("B", "C"):
        c = synth.CONFIGS[name]
        g = synth.make_knn_graph(**c["knn"]) if c.get("knn") else synth.make_graph(c["N"], c["E"], seed=1)
        ei = torch.from_numpy(g["edge_index"]).to(dev)
        n = g["x"].shape[0]
        for _ in range(3):
            capi.PreparedGraph(ei, n)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            capi.PreparedGraph(ei, n)
        torch.cuda.synchronize()
        out["cfg-" + name] = {"nodes": n, "edges": int(ei.shape[1]), "graph_prep_us": (time.perf_counter() - t0) / 50 * 1e6}
    print(json.dumps(out))

if __name__ == "__main__":
    main()
