#!/usr/bin/env python3
"""Micro-benchmark (and float64 check) of the weight-gradient products over bf16 rows at the cfg-E widths (BASELINE.json configs[4]:
400,000 edges, 256-d) through mpnhip_weight_grad_bf16_rows: the one-pass LDS-DMA kernel (csrc/wgrad_rows16.hip) against the
row-panel kernel's bf16 variants (MPNHIP_NO_WGRAD_ROWS16=1).  Times are product + slab sum (torch events, `iters` calls).
usage: python tools/wgrad_rows16_bench.py [--iters 10] [--nbatch 4] [--check]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpntrackseg_amd import capi, synth

CFG_E = [(400000, 640, 128, "edge L1 e part"), (400000, 128, 640, "edge L2"), (200000, 448, 128, "flow L1 e part (one dir)"),
         (200000, 256, 448, "flow L2 (one dir)"), (400000, 64, 128, "classifier L1"),
         # node level (20,000 rows): GEMM-shaped, 256 x 256 output tiles
         (20000, 2176, 256, "node projections"), (20000, 256, 512, "node update"), (20000, 1024, 2048, "node encoder L0"),
         (20000, 256, 1024, "node encoder L1")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--nbatch", type=int, default=4)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    lib = capi.load()
    dev = torch.device("cuda:0")
    for rows, n_out, k_in, what in CFG_E:
        nb = a.nbatch
        dz = torch.from_numpy(synth.normal(1, (nb, rows, n_out))).to(dev).bfloat16()
        h = torch.from_numpy(synth.normal(2, (nb, rows, k_in))).to(dev).bfloat16()
        row = "%6d x %4d x %4d x%d %-26s" % (rows, n_out, k_in, nb, what)
        for form in ("panel", "rows16"):
            if form == "panel":
                os.environ["MPNHIP_NO_WGRAD_ROWS16"] = "1"
            else:
                os.environ.pop("MPNHIP_NO_WGRAD_ROWS16", None)
            gw = torch.zeros((n_out, k_in), device=dev)
            gb = torch.zeros(n_out, device=dev)
            ws = torch.empty(lib.mpnhip_weight_grad_bf16_rows_workspace_bytes(n_out, k_in, rows, nb), dtype=torch.uint8, device=dev)

            def call():
                capi.check(lib.mpnhip_weight_grad_bf16_rows(capi.ptr(dz), capi.ptr(h), rows, n_out, k_in, nb, capi.ptr(gw), capi.ptr(gb),
                                                            capi.ptr(ws), ws.numel(), capi.stream_ptr()), "weight_grad_bf16_rows")
            call()
            err = ""
            if a.check:
                torch.cuda.synchronize()
                ref = torch.einsum("bmo,bmc->oc", dz[:, :50000].double(), h[:, :50000].double()) if False else None
                refb = dz.float().sum((0, 1)).double()
                err = " bias %.1e" % float((gb.double() - refb).norm() / refb.norm())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                call()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.iters
            by = 2.0 * nb * rows * (n_out + k_in)
            row += "  %s %8.1f us %5.2f TB/s%s" % (form, us, by / us / 1e6, err)
        print(row, flush=True)


if __name__ == "__main__":
    main()
