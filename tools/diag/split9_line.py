#!/usr/bin/env python3
"""Six- against nine-product form of MPNHIP_PREC_FP32_SPLIT on the HEADLINE workload: accuracy of the cfg-B logits
against the float64 oracle, and timing of the forward and of the training step.

    python tools/diag/split9_line.py            # the library as built (six piece products: h_a h_b ... without m l, l m, l l)
    make clean && make EXTRA=-DMPNHIP_SPLIT9 && python tools/diag/split9_line.py    # diagnostic build: all nine products (edge_chain.hip mfma6)

cfg-B graph and widths, node_agg_fn=sum, 12 steps, weights scaled to O(1) logits (the g11 fixture's gain): the float64 oracle's forward is
~20 s of host time.  Prints one line; profiles/r06/split6_vs_split9.txt holds both."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from mpntrackseg_amd import synth
from mpntrackseg_amd import train as mtrain
from mpntrackseg_amd.mpn import MOTMPNet
from oracle import mpn_oracle as O

dev = torch.device("cuda:0")
c = synth.CONFIGS["B"]
params = synth.model_params(c["d"], c["L"], "sum")
W = synth.make_weights(params, seed=7, gain=0.7)
g = synth.make_graph(c["N"], c["E"], seed=1)
model = MOTMPNet(params)
model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
model = model.to(dev).eval()
x, ei, ea = (torch.from_numpy(g[k]).to(dev) for k in ("x", "edge_index", "edge_attr"))
out = {}
for prec in ("fp32_split", "fp32"):
    model.gemm_precision = prec
    with torch.no_grad():
        lg = model.hot_path(x, ei, ea).double().cpu().numpy()
    out[prec] = lg
with torch.no_grad():
    Wt = O.to_tensors(W, dtype=torch.float64)
    _, ref, _, _ = O.forward(params, Wt, torch.from_numpy(g["x"]).double(), torch.from_numpy(g["edge_index"]), torch.from_numpy(g["edge_attr"]).double(),
                             return_state=True)
ref = np.stack([l.view(-1).numpy() for l in ref])


def err(a):
    d = a - ref
    return float(np.linalg.norm(d) / np.linalg.norm(ref)), float(np.abs(d).max() / max(1.0, np.abs(ref).max())), float(d.mean() / np.sqrt((d * d).mean()))


model.gemm_precision = "fp32_split"
with torch.no_grad(), model.frozen_weights():
    for _ in range(5):
        model.hot_path(x, ei, ea)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        model.hot_path(x, ei, ea)
    torch.cuda.synchronize()
fwd_ms = (time.perf_counter() - t0) * 1e3 / 30
model.train()
stepper = mtrain.TrainStep(model, world_size=1)


class H:
    pass


h = H()
for _ in range(5):
    stepper(x, ei, ea, holder=h)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    stepper(x, ei, ea, holder=h)
torch.cuda.synchronize()
train_ms = (time.perf_counter() - t0) * 1e3 / 20
r2, mx, bias = err(out["fp32_split"])
r2f, mxf, biasf = err(out["fp32"])
print("cfg-B sum L=12 gain 0.7 (max |logit| %.1f): fp32_split logits vs float64 oracle rel_l2 %.3e max %.3e mean/rms %+.3f | fp32 MFMAs rel_l2 %.3e max %.3e mean/rms %+.3f | "
      "fp32_split forward %.3f ms, training step %.3f ms" % (np.abs(ref).max(), r2, mx, bias, r2f, mxf, biasf, fwd_ms, train_ms))
