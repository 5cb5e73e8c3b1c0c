#!/usr/bin/env python3
"""Measure the f-3 / f-4 rows on one MI355X: device-side graph construction and sliding-window inference over a
synthetic sequence (reference dims, d = 32, L = 12), next to the CPU oracle on a bounded sample of the same work.

    python tools/track_bench.py [--frames 150] [--dets 25] [--fpg 15] [--top-k 50]
Prints one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mpntrackseg_amd import graph as G, synth, tracker
from mpntrackseg_amd.mpn import MOTMPNet
from oracle import mpn_oracle as O, tracker_oracle as T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--dets", type=int, default=25)
    ap.add_argument("--fpg", type=int, default=15)
    ap.add_argument("--top-k", type=int, default=50)
    ap.add_argument("--cpu-windows", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    det = synth.make_detections(frames=a.frames, dets_lo=a.dets - 3, dets_hi=a.dets + 3, seed=21, emb_dim=256, node_in_dim=2048)
    names = list(G.EDGE_FEAT_NAMES) + ["emb_dist"]
    emb = torch.from_numpy(det["reid"]).to(dev)
    maxd = a.fpg - 1  # mpn_tracker.py:80: max_frame_dist = step_size * (frames_per_graph - 1)
    G.construct_graph(det, emb, 25.0, maxd, names)  # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g = G.construct_graph(det, emb, 25.0, maxd, names)
    torch.cuda.synchronize(); t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    gc = T.construct_graph(det, torch.from_numpy(det["reid"]), 25.0, maxd, names)
    t_build_cpu = time.perf_counter() - t0
    assert np.array_equal(g["edge_index"].cpu().numpy(), gc["edge_index"].numpy())

    params = synth.model_params(32, 12, "sum", num_class_steps=1, node_in_dim=2048, edge_in_dim=6)
    W = synth.make_weights(params, seed=7, gain=0.5)
    model = MOTMPNet(params)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    model = model.to(dev).eval()
    x = torch.from_numpy(det["x"]).to(dev)
    args = (model, x, g["edge_index"], g["edge_attr"], g["reid_emb_dists"], det["frame"])
    kw = dict(frames_per_graph=a.fpg, top_k_nns=a.top_k, reciprocal_k_nns=True)
    nwin = len(tracker.frame_windows(det["frame"], a.fpg))
    res = {}
    for wpl in (1, 8):
        tracker.evaluate_graph_in_batches(*args, windows_per_launch=wpl, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = tracker.evaluate_graph_in_batches(*args, windows_per_launch=wpl, **kw)
        torch.cuda.synchronize(); res[wpl] = time.perf_counter() - t0
    # (the runs above are in the default 'auto' precision: per launch, by its edge count -- MOTMPNet.operand_precision)
    # the same, eight windows per launch, pinned to each fp32 mode
    outs = {}
    for prec in ("fp32", "fp32_split"):
        model.gemm_precision = prec
        tracker.evaluate_graph_in_batches(*args, windows_per_launch=8, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs[prec] = tracker.evaluate_graph_in_batches(*args, windows_per_launch=8, **kw)
        torch.cuda.synchronize(); res[prec] = time.perf_counter() - t0
    model.gemm_precision = "auto"
    res["split8"] = res["fp32_split"]
    split_dev = float((outs["fp32_split"] - outs["fp32"]).abs().max())
    # CPU oracle: the first few windows only (bounded sample), same weights
    Wt = O.to_tensors(W)
    def fwd(xs, ei, ea):
        _, logits, _, _ = O.forward(params, Wt, xs, ei, ea, return_state=True)
        return logits[-1]
    wins = tracker.frame_windows(det["frame"], a.fpg)[:a.cpu_windows]
    n_end = wins[-1][1]
    keep = (gc["edge_index"][0] < n_end) & (gc["edge_index"][1] < n_end)
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = T.evaluate_graph_in_batches(fwd, torch.from_numpy(det["x"])[:n_end], gc["edge_index"][:, keep], gc["edge_attr"][keep],
                                          gc["reid_emb_dists"].view(-1)[keep], det["frame"][:n_end], a.fpg, a.top_k, True)
    t_cpu = time.perf_counter() - t0
    n_cpu_win = len(tracker.frame_windows(det["frame"][:n_end], a.fpg))
    print(json.dumps({
        "sequence": {"frames": a.frames, "nodes": int(x.shape[0]), "directed_edges": int(g["edge_index"].shape[1]),
                     "windows": nwin, "frames_per_graph": a.fpg, "top_k_nns": a.top_k},
        "construct_graph_ms": {"mi355x": 1e3 * t_build, "cpu_oracle": 1e3 * t_build_cpu},
        "sliding_window_s": {"windows_per_launch_1": res[1], "windows_per_launch_8": res[8]},
        "windows_per_s": {"windows_per_launch_1": nwin / res[1], "windows_per_launch_8": nwin / res[8],
                          "windows_per_launch_8_fp32_split": nwin / res["split8"], "windows_per_launch_8_fp32_mfma": nwin / res["fp32"], "fp32_split_max_abs_dev_of_probabilities": split_dev,
                          "cpu_oracle": n_cpu_win / t_cpu, "cpu_threads": torch.get_num_threads(), "cpu_sample_windows": n_cpu_win},
        "max_pred": float(out.max())}))


if __name__ == "__main__":
    main()
