#!/usr/bin/env python3
"""grad_x error (rel-L2 against the float64 oracle) of the HIP variants over several graph / weight seeds: tells a code-path bug
(persists for one variant) from knife-edge ReLU decisions (comes and goes between variants and seeds)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from mpntrackseg_amd import synth
from tools.diag.grad_accuracy import hip, oracle, rl2

c = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "A"]
agg = sys.argv[2] if len(sys.argv) > 2 else "sum"
gain = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
Ls = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "3,6").split(",")]
variants = [("fp32", "fp32", None), ("split", "fp32_split", None), ("noNODE", "fp32", "MPNHIP_NO_NODE_FUSION"),
            ("noENC", "fp32", "MPNHIP_NO_ENCODER_FUSION"), ("noCHAIN", "fp32", "MPNHIP_NO_CHAIN")]
print("L seed  oracle32 " + " ".join("%9s" % v[0] for v in variants))
for L in Ls:
    for seed in range(1, 7):
        g = synth.make_graph(c["N"], c["E"], seed=seed)
        params = synth.model_params(c["d"], L, agg)
        W = synth.make_weights(params, seed=6 + seed, gain=gain)
        r = synth.normal(10 + seed, (L, c["E"]))
        r64 = oracle(params, W, g, r, torch.float64)
        r32 = oracle(params, W, g, r, torch.float32)
        row = []
        for name, prec, tog in variants:
            if tog:
                os.environ[tog] = "1"
            h = hip(params, W, g, r, prec)
            if tog:
                del os.environ[tog]
            row.append(max(rl2(h["grad_x"], r64["grad_x"]), rl2(h["grad_ea"], r64["grad_ea"])))
        print("%d %4d  %8.1e " % (L, seed, max(rl2(r32["grad_x"], r64["grad_x"]), rl2(r32["grad_ea"], r64["grad_ea"]))) + " ".join("%9.1e" % v for v in row))
