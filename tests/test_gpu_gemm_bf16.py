"""mpnhip_linear_bf16 (csrc/gemm_bf16.hip: the 128 x 128 tiled bf16-operand GEMM of MPNHIP_PREC_BF16's node-side products,
models/mpn.py:69,87,93,97-99 / mlp.py:27) against a float64 product of the SAME bf16-rounded operands: the only differences left are
the fp32 accumulation order and the fp32 epilogue, so the bound is fp32 noise (2e-5 of the result's scale), not the 2e-2 of the
bf16 mode against an fp32 reference.  Every operand form (fp32 rows converted while staged / bf16 rows in memory), both K segments,
the plain and the full epilogue (c_in, mask, accumulate), the bf16 mirror of the result, ragged M / N / K."""
import ctypes as C

import numpy as np
import pytest
import torch

from mpntrackseg_amd import capi, synth

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda:0")


def to_bf16_bits(t):
    """device fp32 tensor -> int16 tensor of bfloat16 bit patterns through the library's own conversion kernel"""
    t = t.contiguous()
    out = torch.empty(t.shape, dtype=torch.int16, device=t.device)
    capi.check(capi.load().mpnhip_to_bf16(capi.ptr(t), capi.ptr(out), t.numel(), capi.stream_ptr()), "to_bf16")
    return out


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


def run_case(m, n, k, ksplit, x16, w16, relu, c_in, mask, accumulate, y16, seed=3):
    d = dev()
    x = torch.from_numpy(synth.normal(seed, (m, k), stream=1)).to(d)
    w = torch.from_numpy(synth.normal(seed, (n, k), stream=2, std=(2.0 / k) ** 0.5)).to(d)
    b = torch.from_numpy(synth.normal(seed, (n,), stream=3, std=0.1)).to(d)
    xa, xb = x[:, :ksplit].contiguous(), (x[:, ksplit:].contiguous() if ksplit < k else None)
    a = capi.LinearBf16Args()
    keep = []
    if x16:
        xa_, xb_ = to_bf16_bits(xa), (to_bf16_bits(xb) if xb is not None else None)
    else:
        xa_, xb_ = xa, xb
    w_ = to_bf16_bits(w) if w16 else w
    keep += [xa_, xb_, w_]
    a.x, a.ldx = capi.ptr(xa_).value, xa_.shape[1]
    if xb_ is not None:
        a.x2, a.ldx2 = capi.ptr(xb_).value, xb_.shape[1]
    a.w, a.ldw = capi.ptr(w_).value, k
    a.b = capi.ptr(b).value
    y0 = torch.from_numpy(synth.normal(seed, (m, n), stream=4)).to(d)
    y = y0.clone() if accumulate else torch.full((m, n), float("nan"), device=d)
    a.y, a.ldy = capi.ptr(y).value, n
    cin = mk = None
    if c_in:
        cin = torch.from_numpy(synth.normal(seed, (m, n), stream=5)).to(d)
        a.c_in, a.ldc_in = capi.ptr(cin).value, n
    if mask:
        mk = (torch.from_numpy(synth.normal(seed, (m, n), stream=6)).to(d) > 0).float()
        a.mask, a.ldmask = capi.ptr(mk).value, n
    ymir = None
    if y16:
        ymir = torch.zeros((m, n), dtype=torch.int16, device=d)
        a.y16, a.ldy16 = capi.ptr(ymir).value, n
    a.m, a.n, a.k, a.ksplit = m, n, k, ksplit
    a.x_bf16, a.w_bf16, a.relu, a.accumulate = int(x16), int(w16), int(relu), int(accumulate)
    capi.path_counters(reset=True)
    capi.check(capi.load().mpnhip_linear_bf16(C.byref(a), capi.stream_ptr()), "linear_bf16")
    torch.cuda.synchronize()
    counts = capi.path_counters(reset=True)
    ref = bf16_round(x.cpu()) @ bf16_round(w.cpu()).t() + b.cpu().double()
    if c_in:
        ref = ref + cin.cpu().double()
    if relu:
        ref = ref.relu()
    if accumulate:
        ref = ref + y0.cpu().double()
    if mask:
        ref = torch.where(mk.cpu() > 0, ref, torch.zeros_like(ref))
    got = y.cpu().double()
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max()) / scale
    assert np.isfinite(got.numpy()).all() and err < 2e-5, (err, counts)
    if y16:
        mir = ymir.view(torch.bfloat16).float().cpu()
        assert torch.equal(mir, y.cpu().to(torch.bfloat16).float())
    return counts


SHAPES = [(4200, 256, 512, 256), (4100, 2176, 256, 256), (4097, 132, 72, 72), (5000, 64, 640, 640), (4608, 128, 64, 64),
          (4300, 36, 200, 200), (9000, 512, 256, 128)]


@pytest.mark.parametrize("x16,w16", [(0, 0), (1, 1), (0, 1), (1, 0)])
@pytest.mark.parametrize("m,n,k,ksplit", SHAPES)
def test_linear_bf16_plain(m, n, k, ksplit, x16, w16):
    if (x16 or w16) and (k % 8 or ksplit % 8):
        pytest.skip("bf16 rows need K and ksplit multiples of 8")
    counts = run_case(m, n, k, ksplit, x16, w16, relu=1, c_in=0, mask=0, accumulate=0, y16=(n % 4 == 0 and x16))
    # fp32 rows: the tiled kernel from 4,096 rows where the shape fills its 128-column tiles and K steps (gemm.hip launch_gemm);
    # bf16 rows: always
    tiled = bool(x16 or w16) or (k >= 192 and (n % 128 == 0 or n >= 384))
    assert counts["gemm_bf16_tiled"] == int(tiled) and counts["gemm_bf16_ring"] == 0 and counts["gemm_bf16"] == 1, counts


@pytest.mark.parametrize("relu", [0, 1])
@pytest.mark.parametrize("m,n,k,ksplit", [(4200, 256, 512, 256), (4100, 2176, 256, 256), (5000, 64, 640, 640), (9000, 512, 256, 128),
                                          (20000, 2176, 512, 256), (300, 132, 64, 64), (1, 36, 128, 64), (129, 3072, 64, 64)])
def test_linear_bf16_ring_kernel(m, n, k, ksplit, relu):
    """both operands bf16 rows, whole 64-deep K steps, plain epilogue: the persistent LDS-DMA ring kernel (ragged M / N, two K
    segments, more tiles than blocks and fewer)"""
    counts = run_case(m, n, k, ksplit, 1, 1, relu=relu, c_in=0, mask=0, accumulate=0, y16=0)
    assert counts["gemm_bf16_ring"] == 1, counts


@pytest.mark.parametrize("x16,w16", [(0, 0), (1, 1)])
@pytest.mark.parametrize("c_in,mask,accumulate,relu", [(1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 1), (1, 1, 1, 1)])
def test_linear_bf16_full_epilogue(c_in, mask, accumulate, relu, x16, w16):
    counts = run_case(4500, 264, 320, 192, x16, w16, relu, c_in, mask, accumulate, y16=1)
    assert counts["gemm_bf16_tiled"] == 1, counts


def test_linear_bf16_small_row_counts_with_bf16_rows():
    """bf16 rows in memory are served by the tiled kernel at ANY row count (the older kernel cannot read them)"""
    for m in (1, 33, 130, 1000):
        counts = run_case(m, 128, 64, 64, 1, 1, relu=1, c_in=0, mask=0, accumulate=0, y16=1)
        assert counts["gemm_bf16_tiled"] == 1, counts


def test_linear_bf16_narrow_outputs_with_bf16_rows():
    """fewer than 32 output columns: the older strip kernel's shape -- except with bf16 rows, which only the tiled kernel reads
    (the 32-d model's hoisted e0 share: [E, 80] bf16 x [16, 80], accumulated)"""
    counts = run_case(1000, 16, 80, 80, 1, 0, relu=0, c_in=0, mask=0, accumulate=1, y16=0)
    assert counts["gemm_bf16_tiled"] == 1, counts


def test_linear_bf16_rejects_misaligned_bf16_rows():
    d = dev()
    a = capi.LinearBf16Args()
    x = torch.zeros((64, 36), dtype=torch.int16, device=d)
    w = torch.zeros((32, 36), dtype=torch.int16, device=d)
    y = torch.zeros((64, 32), device=d)
    a.x, a.ldx, a.w, a.ldw, a.y, a.ldy = capi.ptr(x).value, 36, capi.ptr(w).value, 36, capi.ptr(y).value, 32
    a.m, a.n, a.k, a.ksplit, a.x_bf16, a.w_bf16 = 64, 32, 36, 36, 1, 1
    assert capi.load().mpnhip_linear_bf16(C.byref(a), capi.stream_ptr()) != 0
    assert b"bf16" in capi.load().mpnhip_last_error()
