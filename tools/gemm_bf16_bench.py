#!/usr/bin/env python3
"""Micro-benchmark of the bf16-operand node-side products of BASELINE.json configs[4] (cfg-E: 20,000 nodes / 400,000 edges / 256-d)
through mpnhip_linear_bf16 / mpnhip_time_linear_bf16: the tiled kernel (csrc/gemm_bf16.hip) against the older strip kernel
(MPNHIP_NO_GEMM_BF16_TILED=1), with fp32 rows or bf16 rows in memory.  Prints us, TFLOP/s and the operand + result bytes / time.
usage: python tools/gemm_bf16_bench.py [--iters 20]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpntrackseg_amd import capi, synth

# (M, N, K, ksplit, c_in, relu, what)
CFG_E = [
    (20000, 2176, 512, 256, 0, 0, "projections [x0|x] (K = 512, no P0 stream)"),
    (20000, 2176, 256, 256, 1, 0, "projections x + P0 (round 4's form)"),
    (20000, 256, 512, 512, 0, 1, "node update"),
    (20000, 256, 2176, 2176, 0, 0, "dX = dP Wx"),
    (20000, 512, 256, 256, 0, 0, "dAGG = dZn Wu"),
    (20000, 512, 2048, 2048, 0, 1, "node encoder L1"),
    (400000, 128, 640, 640, 0, 0, "dE0 = S W1e"),
    (400000, 160, 144, 144, 0, 1, "edge encoder L2"),
]


def bits(t):
    out = torch.empty(t.shape, dtype=torch.int16, device=t.device)
    capi.check(capi.load().mpnhip_to_bf16(capi.ptr(t), capi.ptr(out), t.numel(), capi.stream_ptr()), "to_bf16")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a_ = ap.parse_args()
    lib = capi.load()
    dev = torch.device("cuda:0")
    for M, N, K, ks, cin, relu, what in CFG_E:
        x = torch.from_numpy(synth.normal(1, (M, K))).to(dev)
        w = torch.from_numpy(synth.normal(2, (N, K), std=(2.0 / K) ** 0.5)).to(dev)
        b = torch.zeros(N, device=dev)
        y = torch.empty((M, N), device=dev)
        c = torch.from_numpy(synth.normal(3, (M, N))).to(dev) if cin else None
        xa, xb = x[:, :ks].contiguous(), (x[:, ks:].contiguous() if ks < K else None)
        row = "%6d x %4d x %4d  %-44s" % (M, N, K, what)
        for form in ("old", "f32", "w16", "x16w16"):
            if form == "old":
                os.environ["MPNHIP_NO_GEMM_BF16_TILED"] = "1"
            else:
                os.environ.pop("MPNHIP_NO_GEMM_BF16_TILED", None)
            x16 = form == "x16w16"
            w16 = form in ("w16", "x16w16")
            xa_, xb_ = (bits(xa), bits(xb) if xb is not None else None) if x16 else (xa, xb)
            w_ = bits(w) if w16 else w
            a = capi.LinearBf16Args()
            a.x, a.ldx = capi.ptr(xa_).value, xa_.shape[1]
            if xb_ is not None:
                a.x2, a.ldx2 = capi.ptr(xb_).value, xb_.shape[1]
            a.w, a.ldw, a.b = capi.ptr(w_).value, K, capi.ptr(b).value
            if c is not None:
                a.c_in, a.ldc_in = capi.ptr(c).value, N
            a.y, a.ldy, a.m, a.n, a.k, a.ksplit = capi.ptr(y).value, N, M, N, K, ks
            a.x_bf16, a.w_bf16, a.relu = int(x16), int(w16), relu
            us = C.c_float(0)
            capi.check(lib.mpnhip_time_linear_bf16(C.byref(a), a_.iters, C.byref(us), capi.stream_ptr()), "time_linear_bf16")
            byts = M * K * (2 if x16 else 4) + N * K * (2 if w16 else 4) + M * N * 4 * (2 if cin else 1)
            row += "  %s %7.1f us %5.0f TF %4.2f TB/s" % (form, us.value, 2.0 * M * N * K / us.value / 1e6, byts / us.value / 1e6)
        print(row, flush=True)


if __name__ == "__main__":
    main()
