#!/usr/bin/env python3
"""Achieved HBM bandwidth of the mask branch's attention aggregation (mpnhip_attention_aggregate) on a
MOTS20-02-like graph (SURVEY.md section 8d cfg-C: 500 nodes, reciprocal top-k kNN, 64x14x14 features per node).
Algorithmic bytes: every sorted edge of the two directions reads one 50,176-byte neighbour row, every (node,
direction) writes one row."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpntrackseg_amd import capi, synth

def main():
    dev = torch.device("cuda:0")
    lib = capi.load()
    g = synth.make_knn_graph(frames=20, dets=25, top_k=150, seed=3, node_in_dim=8)
    N, E = g["x"].shape[0], g["edge_index"].shape[1]
    F = 64 * 14 * 14
    pg = capi.PreparedGraph(torch.from_numpy(g["edge_index"]).to(dev), N)
    x = torch.from_numpy(synth.normal(1, (N, F))).to(dev)
    lg = torch.from_numpy(synth.normal(2, (E,))).to(dev)
    oi, oo = torch.empty_like(x), torch.empty_like(x)
    w = torch.empty(E, device=dev)
    def run():
        capi.check(lib.mpnhip_attention_aggregate(capi.ptr(pg.buf), N, E, capi.ptr(x), F, capi.ptr(lg), capi.ptr(oi),
                                                  capi.ptr(oo), capi.ptr(w), capi.stream_ptr()), "attention")
    for _ in range(3): run()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20): run()
    t1.record(); torch.cuda.synchronize()
    us = t0.elapsed_time(t1) * 1000 / 20
    byts = E * F * 4 + 2 * N * F * 4 + E * 12
    print("N %d E %d  %.1f us  %.2f GB gathered  %.0f GB/s (%.0f %% of 8 TB/s); table %.1f MB is L2/Infinity-Cache resident"
          % (N, E, us, byts / 1e9, byts / us / 1e3, byts / us / 1e3 / 80, N * F * 4 / 1e6))

if __name__ == "__main__":
    main()
