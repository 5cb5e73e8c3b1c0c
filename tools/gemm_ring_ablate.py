#!/usr/bin/env python3
"""Timing ablations of the LDS-DMA ring GEMM (csrc/gemm_bf16.hip) at the projections' shape (MPNHIP_GEMM_RING_DEBUG bits; results of
the ablated runs are wrong by construction): which of DMA / operand fetch + MFMA / stores the launch time follows."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpntrackseg_amd import capi, synth
lib = capi.load()
dev = torch.device("cuda:0")
def bits(t):
    out = torch.empty(t.shape, dtype=torch.int16, device=t.device)
    capi.check(lib.mpnhip_to_bf16(capi.ptr(t), capi.ptr(out), t.numel(), capi.stream_ptr()), "to_bf16")
    return out
for (M, N, K) in ((20000, 2176, 512), (20000, 256, 2176), (20000, 512, 2048)):
    x = bits(torch.from_numpy(synth.normal(1, (M, K))).to(dev)); w = bits(torch.from_numpy(synth.normal(2, (N, K), std=0.05)).to(dev))
    y = torch.empty((M, N), device=dev)
    a = capi.LinearBf16Args()
    a.x, a.ldx, a.w, a.ldw, a.y, a.ldy, a.m, a.n, a.k, a.ksplit, a.x_bf16, a.w_bf16 = capi.ptr(x).value, K, capi.ptr(w).value, K, capi.ptr(y).value, N, M, N, K, K, 1, 1
    for tile in ("128", "256"):
        row = "%d x %d x %d tile %s:" % (M, N, K, tile)
        for dbg in (0, 1, 2, 4):
            os.environ["MPNHIP_GEMM_RING_TILE"] = tile
            os.environ["MPNHIP_GEMM_RING_DEBUG"] = str(dbg)
            us = C.c_float(0)
            capi.check(lib.mpnhip_time_linear_bf16(C.byref(a), 10, C.byref(us), capi.stream_ptr()), "time")
            row += "  dbg%d %.1f" % (dbg, us.value)
        print(row, flush=True)
