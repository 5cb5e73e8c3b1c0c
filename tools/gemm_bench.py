#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM at the shapes the MPN hot path uses (cfg-B by default).
Usage: python tools/gemm_bench.py [--iters 30] [--shapes M,N,K ...]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpntrackseg_amd import capi, synth

CFG_B = [  # (M, N, K, what)
    (50000, 320, 128, "edge L1 (e part)"), (50000, 64, 320, "edge L2"), (50000, 224, 64, "flow L1 (both dirs)"),
    (50000, 128, 224, "flow L2"), (50000, 32, 64, "classifier L1"), (5000, 1088, 256, "node projections"),
    (5000, 128, 256, "node update"), (5000, 512, 2048, "node encoder L1"), (5000, 128, 512, "node encoder L2"),
    (50000, 72, 72, "edge encoder L2"), (50000, 64, 72, "edge encoder L3"),
]

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--shapes", nargs="*")
    ap.add_argument("--check", action="store_true", help="also print the max error against a float64 product")
    a = ap.parse_args()
    shapes = CFG_B
    if a.shapes:
        shapes = [tuple(int(v) for v in s.split(",")) + ("",) for s in a.shapes]
    lib = capi.load()
    dev = torch.device("cuda:0")
    tot = 0.0
    for M, N, K, what in shapes:
        x = torch.from_numpy(synth.normal(1, (M, K))).to(dev)
        w = torch.from_numpy(synth.normal(2, (N, K), std=(2.0 / K) ** 0.5)).to(dev)
        b = torch.zeros(N, device=dev)
        y = torch.empty((M, N), device=dev)
        us = ctypes.c_float(0)
        capi.check(lib.mpnhip_time_linear(capi.ptr(x), capi.ptr(w), capi.ptr(b), capi.ptr(y), M, N, K, a.iters,
                                          ctypes.byref(us), capi.stream_ptr()), "time_linear")
        fl = 2.0 * M * N * K
        err = ""
        if a.check:
            ref = torch.relu(x[:4096].double() @ w.double().t())
            err = "  max|err| %.3e (fp32 eps * max|y| = %.3e)" % ((y[:4096].double() - ref).abs().max().item(),
                                                                 2.0 ** -24 * ref.abs().max().item())
        print("%6d x %4d x %4d  %-24s %8.1f us  %6.1f TFLOP/s%s" % (M, N, K, what, us.value, fl / us.value / 1e6, err))
        tot += us.value
    print("sum %.1f us" % tot)

if __name__ == "__main__":
    main()
