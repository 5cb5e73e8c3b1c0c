#!/usr/bin/env python3
"""Where does the HIP path's gradient error come from?  HIP (fp32 / fp32_split) and the fp32 oracle against the float64 oracle
on one configuration, with the fused kernels switched off one at a time (environment toggles of libmpnhip.so).
    python tools/diag/grad_accuracy.py --config A --agg sum --L 1,2,3,6 --gain 1.0"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from mpntrackseg_amd import synth
from mpntrackseg_amd.mpn import MOTMPNet
from oracle import mpn_oracle as O


def oracle(params, W, g, r, dtype):
    Wt = {k: torch.from_numpy(v).to(dtype).requires_grad_(True) for k, v in W.items()}
    x = torch.from_numpy(g["x"]).to(dtype).requires_grad_(True)
    ea = torch.from_numpy(g["edge_attr"]).to(dtype).requires_grad_(True)
    _, lg, _, _ = O.forward(params, Wt, x, torch.from_numpy(g["edge_index"]), ea, return_state=True)
    lg = torch.stack([l.view(-1) for l in lg])
    keys = list(Wt)
    gr = torch.autograd.grad((lg * torch.from_numpy(r).to(dtype)).sum(), [x, ea] + [Wt[k] for k in keys], allow_unused=True)
    gr = [v if v is not None else torch.zeros_like(t) for v, t in zip(gr, [x, ea] + [Wt[k] for k in keys])]
    out = {"logits": lg.detach().double().numpy(), "grad_x": gr[0].double().numpy(), "grad_ea": gr[1].double().numpy()}
    out.update({k: v.double().numpy() for k, v in zip(keys, gr[2:])})
    return out


def hip(params, W, g, r, precision):
    dev = torch.device("cuda:0")
    model = MOTMPNet(params)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
    model = model.to(dev).train()
    model.gemm_precision = precision
    x = torch.from_numpy(g["x"]).to(dev).requires_grad_(True)
    ea = torch.from_numpy(g["edge_attr"]).to(dev).requires_grad_(True)
    lg = model.hot_path(x, torch.from_numpy(g["edge_index"]).to(dev), ea)
    (lg * torch.from_numpy(r).to(dev)).sum().backward()
    torch.cuda.synchronize()
    out = {"logits": lg.detach().double().cpu().numpy(), "grad_x": x.grad.double().cpu().numpy(), "grad_ea": ea.grad.double().cpu().numpy()}
    out.update({k: p.grad.double().cpu().numpy() for k, p in model.named_parameters()})
    return out


def rl2(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


KEYS = ["logits", "grad_x", "grad_ea", "encoder.node_model.fc_layers.0.weight", "encoder.edge_model.fc_layers.0.weight",
        "MPNet.edge_model.edge_model.fc_layers.0.weight", "MPNet.edge_model.edge_model.fc_layers.2.weight",
        "MPNet.node_model.flow_in_model.fc_layers.0.weight", "MPNet.node_model.flow_out_model.fc_layers.2.weight",
        "MPNet.node_model.node_model.0.weight", "classifier.edge_model.fc_layers.0.weight"]
SHORT = ["logits", "dx", "dea", "encN0", "encE0", "edge0", "edge2", "fin0", "fout2", "node", "cls0"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="A")
    ap.add_argument("--agg", default="sum")
    ap.add_argument("--L", default="1,2,3,6")
    ap.add_argument("--gain", type=float, default=1.0)
    ap.add_argument("--toggles", default="", help="comma list of env names to set to 1, one run each")
    ap.add_argument("--knn", default="")
    args = ap.parse_args()
    c = synth.CONFIGS[args.config]
    if c.get("knn"):
        g = synth.make_knn_graph(seed=1, **c["knn"])
    else:
        g = synth.make_graph(c["N"], c["E"], seed=1)
    E = g["edge_index"].shape[1]
    print("graph N %d E %d d %d agg %s gain %g" % (g["x"].shape[0], E, c["d"], args.agg, args.gain))
    print("%-34s" % "rel-L2 error vs float64 oracle" + " ".join("%8s" % s for s in SHORT))
    for L in [int(v) for v in args.L.split(",")]:
        params = synth.model_params(c["d"], L, args.agg)
        W = synth.make_weights(params, seed=7, gain=args.gain)
        r = synth.normal(11, (L, E))
        ref64 = oracle(params, W, g, r, torch.float64)
        ref32 = oracle(params, W, g, r, torch.float32)
        print("L=%d max|logit| %.3g" % (L, np.abs(ref64["logits"]).max()))
        print("%-34s" % "  oracle fp32" + " ".join("%8.1e" % rl2(ref32[k], ref64[k]) for k in KEYS))
        runs = [("hip fp32", "fp32", None), ("hip fp32_split", "fp32_split", None)]
        for t in [t for t in args.toggles.split(",") if t]:
            # "NAME" (fp32, NAME=1), "split:NAME" (fp32_split, NAME=1), "split:NAME=0"
            prec = "fp32"
            if t.startswith("split:"):
                prec, t = "fp32_split", t[6:]
            runs.append(("hip %s %s" % (prec, t.replace("MPNHIP_", "")), prec, t))
        for name, prec, tog in runs:
            if tog:
                k, _, v = tog.partition("=")
                os.environ[k] = v or "1"
            try:
                h = hip(params, W, g, r, prec)
                print("%-34s" % ("  " + name) + " ".join("%8.1e" % rl2(h[k], ref64[k]) for k in KEYS))
            finally:
                if tog:
                    del os.environ[tog.partition("=")[0]]


if __name__ == "__main__":
    main()
