#!/usr/bin/env python3
"""Accuracy of the six-product split GEMM (MPNHIP_GEMM_PREC=2) against the fp32 MFMA GEMM (=0), both against float64:
single products with heavy cancellation, and a chain of 48 norm-preserving layers (does the error grow like n or sqrt(n)?)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from mpntrackseg_amd import capi, synth

lib = capi.load()
dev = torch.device("cuda:0")


def linear(x, w, prec):
    os.environ["MPNHIP_GEMM_PREC"] = str(prec)
    y = torch.empty((x.shape[0], w.shape[0]), device=dev)
    capi.check(lib.mpnhip_linear(capi.ptr(x), x.shape[1], capi.ptr(w), None, capi.ptr(y), w.shape[0], x.shape[0], w.shape[0], x.shape[1],
                                 0, capi.stream_ptr()), "linear")
    torch.cuda.synchronize()
    del os.environ["MPNHIP_GEMM_PREC"]
    return y


for M, N, K in [(8192, 320, 128), (8192, 256, 320), (8192, 256, 1088), (4096, 512, 2048)]:
    x = torch.from_numpy(synth.normal(1, (M, K))).to(dev)
    w = torch.from_numpy(synth.normal(2, (N, K), std=(1.0 / K) ** 0.5)).to(dev)
    ref = x.double() @ w.double().t()
    for prec in (0, 2):
        y = linear(x, w, prec).double()
        d = y - ref
        print("%5d x %4d x %4d prec %d: rel_l2 %.2e  max %.2e  bias toward zero %.2e" % (
            M, N, K, prec, float(d.norm() / ref.norm()), float(d.abs().max() / ref.abs().max()),
            float((d * ref.sign()).mean() / ref.abs().mean())))

# chain of norm-preserving layers
n = 256
q, _ = np.linalg.qr(synth.normal(5, (n, n)).astype(np.float64))
w = torch.from_numpy(q.astype(np.float32)).to(dev)
x0 = torch.from_numpy(synth.normal(6, (8192, n))).to(dev)
for prec in (0, 2):
    y, ref = x0.clone(), x0.double()
    out = []
    for i in range(1, 49):
        y = linear(y, w, prec)
        ref = ref @ w.double().t()
        if i in (1, 2, 4, 8, 16, 32, 48):
            out.append("%d: %.2e" % (i, float((y.double() - ref).norm() / ref.norm())))
    print("chain prec %d rel_l2 after n layers: %s" % (prec, "  ".join(out)))
