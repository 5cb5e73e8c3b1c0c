#!/usr/bin/env python3
"""mpnhip_weight_grad_prec on one shape against float64: python tools/diag/wgrad_probe.py rows n_out k_in [nbatch] [prec]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mpntrackseg_amd import capi, synth
rows, n_out, k_in = (int(v) for v in sys.argv[1:4])
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 1
prec = capi.PRECISIONS[sys.argv[5] if len(sys.argv) > 5 else "fp32_split"]
lib = capi.load(); dev = torch.device("cuda:0")
dz = torch.from_numpy(synth.normal(1, (nb, rows, n_out))).to(dev)
h = torch.from_numpy(synth.normal(2, (nb, rows, k_in))).to(dev)
gw = torch.zeros((n_out, k_in), device=dev); gb = torch.zeros(n_out, device=dev)
ws = torch.empty(lib.mpnhip_weight_grad_workspace_bytes(n_out, k_in, rows, nb), dtype=torch.uint8, device=dev)
capi.check(lib.mpnhip_weight_grad_prec(capi.ptr(dz), capi.ptr(h), rows, n_out, k_in, nb, prec, capi.ptr(gw), capi.ptr(gb), capi.ptr(ws), ws.numel(), capi.stream_ptr()), "wg")
torch.cuda.synchronize()
ref = torch.einsum("bmo,bmc->oc", dz.double(), h.double()); refb = dz.double().sum((0, 1))
print(rows, n_out, k_in, nb, "rel_l2 %.2e bias %.2e" % (float((gw.double() - ref).norm() / ref.norm()), float((gb.double() - refb).norm() / refb.norm())), capi.path_counters())
