#!/usr/bin/env python3
"""Where does the MPNHIP_PREC_FP32_SPLIT backward lose accuracy?  ONE fp32 training forward (saved activations, ReLU masks);
then mpnhip_backward twice on that same forward -- precision fp32, precision fp32_split -- and every per-step block of
pre-activation gradients both runs keep (mpnhip_debug_backward_saved) compared: relative L2 of (split - fp32) per buffer and
step, in the order the backward computes them (last step first).  fp32 vs fp32 (run twice) is printed as the noise floor.

usage: python tools/diag/split_bwd_locate.py [--config B] [--L 12] [--agg sum]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mpntrackseg_amd import capi, synth  # noqa: E402
from mpntrackseg_amd.autograd import native_backward, native_forward_saved  # noqa: E402
from mpntrackseg_amd.mpn import MOTMPNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="B")
    ap.add_argument("--L", type=int, default=12)
    ap.add_argument("--agg", default="sum")
    ap.add_argument("--gain", type=float, default=0.7)
    ap.add_argument("--variant", default="all", choices=["all", "gemm_only", "chain_only"])
    ap.add_argument("--brief", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    c = synth.CONFIGS[args.config]
    params = synth.model_params(c["d"], args.L, args.agg)
    W = synth.make_weights(params, seed=7, gain=args.gain)
    g = synth.make_graph(c["N"], c["E"], seed=1)
    model = MOTMPNet(params)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
    model = model.to(dev).train()
    x = torch.from_numpy(g["x"]).to(dev)
    ea = torch.from_numpy(g["edge_attr"]).to(dev)
    ei = torch.from_numpy(g["edge_index"]).to(dev)
    N, E, L = x.shape[0], ea.shape[0], args.L
    pg = capi.PreparedGraph(ei, N, validate=True)
    logits = torch.empty((L, E), dtype=torch.float32, device=dev)
    model.gemm_precision = "fp32"
    ws = native_forward_saved(model, pg, x, ea, logits)
    r = torch.from_numpy(synth.normal(11, (L, E))).to(dev)
    lib = capi.load()

    def run(prec):
        model.gemm_precision = prec
        prm = model.hot_path_parameters()
        grads = {id(p): torch.zeros_like(p) for p in prm}
        native_backward(model, pg, x, ea, r, ws, grads)
        torch.cuda.synchronize()
        m = model.c_model([])
        bws = capi.workspace(lib.mpnhip_backward_workspace_bytes(m, N, E), dev, "bwd")
        out = {}
        for s in range(L, 0, -1):
            for what, layers in (("dz_flow", (1, 0)), ("dz_cls", (0,)), ("dz_edge", (1, 0)), ("dp", (0,)), ("dz_node", (0,))):
                for ly in layers:
                    out[(s, what, ly)] = capi.backward_saved(model, pg, bws, what, s, ly).double().cpu().numpy()
        names = {id(p): k for k, p in model.named_parameters()}
        return out, {names[i]: t.double().cpu().numpy() for i, t in grads.items()}

    a, ga = run("fp32")
    a2, _ = run("fp32")
    if args.variant == "gemm_only":      # split GEMMs, fp32 chain kernel
        os.environ["MPNHIP_CHAIN_SPLIT"] = "0"
        b, gb = run("fp32_split")
    elif args.variant == "chain_only":   # split chain kernel, fp32 GEMMs
        os.environ["MPNHIP_CHAIN_SPLIT"] = "1"
        b, gb = run("fp32")
    else:
        b, gb = run("fp32_split")
    os.environ.pop("MPNHIP_CHAIN_SPLIT", None)

    def rel(u, v):
        return float(np.linalg.norm(u - v) / max(np.linalg.norm(v), 1e-300))

    print("%-5s %-10s %-3s %12s %12s   |block|" % ("step", "block", "ly", "split-fp32", "fp32-fp32"))
    for k in a:
        if args.brief and not (k[0] in (1, L // 2, L) and k[1] in ("dz_flow", "dz_edge", "dp")):
            continue
        dlt = (b[k] - a[k]).ravel()
        av = a[k].ravel()
        nd, na = np.linalg.norm(dlt), np.linalg.norm(av)
        corr = float(dlt @ av / (nd * na)) if nd > 0 and na > 0 else 0.0
        # is the difference a SCALING of the block (corr -> +-1), a shift (mean), or noise?
        print("%-5d %-10s %-3d %12.2e %12.2e   %.3e   corr(diff, value) %+.3f   mean(diff)/rms(diff) %+.3f   sign-agreement %.3f" % (
            k[0], k[1], k[2], rel(b[k], a[k]), rel(a2[k], a[k]), na, corr, float(dlt.mean() / (nd / np.sqrt(dlt.size))) if nd > 0 else 0.0,
            float(np.mean(np.sign(dlt[av != 0]) == np.sign(av[av != 0]))) if nd > 0 else 0.0))
    print("parameter gradients, split vs fp32:")
    for k in ga:
        print("  %-52s %.2e" % (k, rel(gb[k], ga[k])))


if __name__ == "__main__":
    main()
