#!/usr/bin/env python3
"""Forward logits of the fp32 and fp32_split modes against the float64 oracle at cfg-B (12 steps): relative L2 per step and the
MEAN of the error in units of its rms (the bf16 MFMA's accumulate bias, tools/micro/mfma_bias.hip, shows up as a mean)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mpntrackseg_amd import synth  # noqa: E402
from mpntrackseg_amd.mpn import MOTMPNet  # noqa: E402
from oracle import mpn_oracle as O  # noqa: E402

dev = torch.device("cuda:0")
c = synth.CONFIGS["B"]
L = 12
params = synth.model_params(c["d"], L, "sum")
W = synth.make_weights(params, seed=7, gain=0.7)
g = synth.make_graph(c["N"], c["E"], seed=1)
model = MOTMPNet(params)
model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
model = model.to(dev).eval()
x, ei, ea = (torch.from_numpy(g[k]).to(dev) for k in ("x", "edge_index", "edge_attr"))
Wt = {k: torch.from_numpy(v).double() for k, v in W.items()}
with torch.no_grad():
    _, lg, _, _ = O.forward(params, Wt, torch.from_numpy(g["x"]).double(), torch.from_numpy(g["edge_index"]),
                            torch.from_numpy(g["edge_attr"]).double(), return_state=True)
ref = torch.stack([l.view(-1) for l in lg]).numpy()
for prec in ("fp32", "fp32_split"):
    model.gemm_precision = prec
    with torch.no_grad():
        out = model.hot_path(x, ei, ea)
    got = out.double().cpu().numpy()
    for s in (0, 5, 11):
        d = got[s] - ref[s]
        print("%-10s step %2d  rel L2 %.2e   mean(err)/rms(err) %+.3f" % (prec, s + 1, np.linalg.norm(d) / np.linalg.norm(ref[s]), d.mean() / np.sqrt((d * d).mean())))
