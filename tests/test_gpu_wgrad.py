"""Operator-level parity of the weight-gradient products (include/mpnhip.h mpnhip_weight_grad_prec; models/mlp.py:27-28 under
autograd: dW = dZ^T H, db = column sums of dZ) against float64, over every block variant of the row-panel kernel
(csrc/wgrad_panel.hip), row counts that give empty / one / odd / even numbers of full 16-row stages and partial last stages,
batched products, and all three operand forms (fp32 MFMAs, three bf16 pieces, one bf16 piece)."""
import numpy as np
import pytest
import torch

from mpntrackseg_amd import capi, synth

pytestmark = pytest.mark.gpu

SHAPES = [(320, 64), (64, 320), (224, 64), (32, 64), (128, 224), (128, 256), (1088, 128), (80, 32), (16, 80), (56, 16), (8, 16),
          (18, 18), (1, 32), (1, 8), (640, 128), (448, 128), (72, 6), (144, 6), (160, 7)]
ROWS = [3, 16, 17, 44, 300, 1000]


@pytest.mark.parametrize("precision,tol", [("fp32", 3e-6), ("fp32_split", 3e-6), ("bf16", 1.5e-2)])
@pytest.mark.parametrize("n_out,k_in", SHAPES)
def test_weight_grad_matches_float64(n_out, k_in, precision, tol):
    lib = capi.load()
    dev = torch.device("cuda:0")
    for rows in ROWS:
        for nb in (1, 3):
            dz = torch.from_numpy(synth.normal(3 + rows, (nb, rows, n_out))).to(dev)
            h = torch.from_numpy(synth.normal(5 + rows, (nb, rows, k_in))).to(dev)
            gw = torch.full((n_out, k_in), 0.25, device=dev)      # "+=" into existing values
            gb = torch.full((n_out,), -0.5, device=dev)
            ws = torch.empty(lib.mpnhip_weight_grad_workspace_bytes(n_out, k_in, rows, nb), dtype=torch.uint8, device=dev)
            capi.check(lib.mpnhip_weight_grad_prec(capi.ptr(dz), capi.ptr(h), rows, n_out, k_in, nb, capi.PRECISIONS[precision],
                                                   capi.ptr(gw), capi.ptr(gb), capi.ptr(ws), ws.numel(), capi.stream_ptr()), "weight_grad")
            torch.cuda.synchronize()
            ref = torch.einsum("bmo,bmc->oc", dz.double(), h.double())
            refb = dz.double().sum((0, 1))
            scale = float(ref.abs().max()) + 1e-30
            err = float((gw.double() - 0.25 - ref).abs().max()) / scale
            errb = float((gb.double() + 0.5 - refb).abs().max()) / (float(refb.abs().max()) + 1e-30)
            assert err < tol and errb < 3e-6, (rows, nb, err, errb)


def test_weight_grad_is_bitwise_reproducible():
    lib = capi.load()
    dev = torch.device("cuda:0")
    rows, n_out, k_in, nb = 5000, 320, 64, 2
    dz = torch.from_numpy(synth.normal(1, (nb, rows, n_out))).to(dev)
    h = torch.from_numpy(synth.normal(2, (nb, rows, k_in))).to(dev)
    outs = []
    for _ in range(3):
        gw = torch.zeros((n_out, k_in), device=dev)
        gb = torch.zeros(n_out, device=dev)
        ws = torch.empty(lib.mpnhip_weight_grad_workspace_bytes(n_out, k_in, rows, nb), dtype=torch.uint8, device=dev)
        capi.check(lib.mpnhip_weight_grad_prec(capi.ptr(dz), capi.ptr(h), rows, n_out, k_in, nb, capi.PRECISIONS["fp32_split"],
                                               capi.ptr(gw), capi.ptr(gb), capi.ptr(ws), ws.numel(), capi.stream_ptr()), "weight_grad")
        torch.cuda.synchronize()
        outs.append((gw.cpu().numpy().copy(), gb.cpu().numpy().copy()))
    assert all(np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]) for o in outs[1:])


SHAPES16 = [(640, 128), (128, 640), (448, 128), (256, 448), (64, 128), (320, 64), (64, 320), (224, 64), (128, 224), (32, 64), (80, 16),
            (16, 80), (56, 16), (32, 56), (8, 16), (160, 32), (32, 160), (112, 32), (64, 112), (16, 32), (24, 8),
            # GEMM-shaped node-level products: 256 x 256 output tiles of the LDS-DMA kernel (partial last tiles in both directions)
            (2176, 256), (680, 328), (264, 648), (1024, 2048)]


@pytest.mark.parametrize("n_out,k_in", SHAPES16)
def test_weight_grad_over_bf16_rows_matches_float64(n_out, k_in):
    """mpnhip_weight_grad_bf16_rows (the row-panel kernel's bf16-source variants: stages loaded as 8-byte pieces and stored to LDS as
    they are, one product per k block) over every shape the bf16-operand training path of the 256-d / 128-d / 64-d / 32-d models
    produces: the operands ARE bf16 values, so every product is exact and only the fp32 accumulation order differs from float64."""
    lib = capi.load()
    dev = torch.device("cuda:0")
    for rows in ROWS + [4097]:
        for nb in (1, 3):
            dz = torch.from_numpy(synth.normal(3 + rows, (nb, rows, n_out))).to(dev).bfloat16()
            h = torch.from_numpy(synth.normal(5 + rows, (nb, rows, k_in))).to(dev).bfloat16()
            gw = torch.full((n_out, k_in), 0.25, device=dev)
            gb = torch.full((n_out,), -0.5, device=dev)
            ws = torch.empty(lib.mpnhip_weight_grad_bf16_rows_workspace_bytes(n_out, k_in, rows, nb), dtype=torch.uint8, device=dev)
            capi.check(lib.mpnhip_weight_grad_bf16_rows(capi.ptr(dz), capi.ptr(h), rows, n_out, k_in, nb, capi.ptr(gw), capi.ptr(gb), capi.ptr(ws),
                                                        ws.numel(), capi.stream_ptr()), "weight_grad_bf16_rows")
            torch.cuda.synchronize()
            ref = torch.einsum("bmo,bmc->oc", dz.double(), h.double())
            refb = dz.double().sum((0, 1))
            err = float((gw.double() - 0.25 - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
            errb = float((gb.double() + 0.5 - refb).abs().max()) / (float(refb.abs().max()) + 1e-30)
            assert err < 3e-6 and errb < 3e-6, (rows, nb, err, errb)
