#!/usr/bin/env python3
"""Micro-benchmark (and float64 check) of the weight-gradient product dW += dZ^T H at the cfg-B shapes of one group of steps.
Usage: python tools/wgrad_bench.py [--iters 20] [--nbatch 5] [--check]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpntrackseg_amd import capi, synth

CFG_B = [  # (rows, n_out, k_in, what)
    (50000, 320, 128, "edge L1 [e0|e] part"), (50000, 320, 64, "edge L1 e part"), (50000, 64, 320, "edge L2"), (25000, 224, 64, "flow L1 e-part (one dir)"),
    (25000, 128, 224, "flow L2 (one dir)"), (50000, 32, 64, "classifier L1"), (5000, 1088, 128, "node projections"),
    (5000, 128, 256, "node update"), (5000, 512, 2048, "node encoder L1 (nbatch 1)"), (5000, 128, 512, "node encoder L2 (nbatch 1)"),
]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--nbatch", type=int, default=5)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--only", type=str, default="", help="comma-separated indices into the shape list")
    ap.add_argument("--split", action="store_true", help="MPNHIP_PREC_FP32_SPLIT: three-piece bf16 operands (wgrad_panel.hip)")
    ap.add_argument("--bf16", action="store_true", help="MPNHIP_PREC_BF16: operands rounded to bf16, one product per k block")
    a = ap.parse_args()
    lib = capi.load()
    dev = torch.device("cuda:0")
    tot = 0.0
    prec = capi.PRECISIONS["bf16"] if a.bf16 else (capi.PRECISIONS["fp32_split"] if a.split else capi.PRECISIONS["fp32"])
    shapes = [CFG_B[int(i)] for i in a.only.split(",")] if a.only else CFG_B
    for rows, n_out, k_in, what in shapes:
        nb = 1 if "nbatch 1" in what else a.nbatch
        dz = torch.from_numpy(synth.normal(1, (nb, rows, n_out))).to(dev)
        h = torch.from_numpy(synth.normal(2, (nb, rows, k_in))).to(dev)
        gw = torch.zeros((n_out, k_in), device=dev)
        gb = torch.zeros(n_out, device=dev)
        ws = torch.empty(lib.mpnhip_weight_grad_workspace_bytes(n_out, k_in, rows, nb), dtype=torch.uint8, device=dev)
        err = ""
        if a.check:
            capi.check(lib.mpnhip_weight_grad_prec(capi.ptr(dz), capi.ptr(h), rows, n_out, k_in, nb, prec, capi.ptr(gw), capi.ptr(gb),
                                                   capi.ptr(ws), ws.numel(), capi.stream_ptr()), "weight_grad")
            ref = torch.einsum("bmo,bmc->oc", dz.double(), h.double())
            refb = dz.double().sum((0, 1))
            dw = gw.double() - ref
            err = "  rel_l2 %.1e  mean/rms %+.3f  bias %.1e" % (float(dw.norm() / ref.norm()), float(dw.mean() / dw.pow(2).mean().sqrt()),
                                                              float((gb.double() - refb).norm() / refb.norm()))
        us = ctypes.c_float(0)
        capi.check(lib.mpnhip_time_weight_grad_prec(capi.ptr(dz), capi.ptr(h), rows, n_out, k_in, nb, prec, capi.ptr(gw), capi.ptr(gb),
                                                    capi.ptr(ws), ws.numel(), a.iters, ctypes.byref(us), capi.stream_ptr()), "time_weight_grad")
        fl = 2.0 * nb * rows * n_out * k_in
        by = 4.0 * nb * rows * (n_out + k_in)
        print("%6d x %4d x %4d x%d %-28s %8.1f us  %6.1f TFLOP/s  %5.2f TB/s operand bytes%s" % (rows, n_out, k_in, nb, what, us.value,
              fl / us.value / 1e6, by / us.value / 1e6, err))
        tot += us.value
    print("sum %.1f us" % tot)

if __name__ == "__main__":
    main()
