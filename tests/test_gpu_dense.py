"""Parity of forward AND backward on the DENSE graphs of BASELINE.json configs[2] / configs[3] (reciprocal-kNN graphs with
E / N >= 48: the long-segment kernels k_segment_reduce_block / k_segment_reduce_block3, the reference-width fusions
k_node_step32 / k_node_step32_bwd / k_edge_encoder*), and the headline cfg-B workload with O(1) logits -- in both fp32
precisions (MPNHIP_PREC_FP32 and MPNHIP_PREC_FP32_SPLIT), against

  * fixtures produced by the REFERENCE's own forward + autograd (tests/golden/g11_*, g12_*; tools/make_golden.py), and
  * torch autograd of the CPU oracle (itself pinned to those fixtures by tests/test_oracle_golden.py).

Every test asserts, through mpnhip_debug_counters, that the kernel variant it means to check is the one that ran.
Criteria (tests/gradcheck.py): logits absolute 1e-4 for mean / max, per element 1e-4 * max(1, |ref|) for sum with O(1) logits.
Gradients here are UNPINNED comparisons on deep / dense networks, where single ReLU decisions at the noise level move whole
tensors (gradcheck.py, measured in profiles/r02/grad_seed_sweep.txt): they are held to the loose bound (relative L2 < 1e-2)
and REPORTED; the sharp gradient statement for the same configurations (2e-5 on the branch taken + decision agreement) is
tests/test_gpu_pinned.py."""
import numpy as np
import pytest
import torch

from gradcheck import (GTOL, check_grads_against_fixture, grad_close, logits_close_abs, logits_close_per_element, nerr)
from pinned import explain_loose_failures
from mpntrackseg_amd import capi, synth
from mpntrackseg_amd.mpn import MOTMPNet
from oracle import mpn_oracle as O

pytestmark = pytest.mark.gpu
PRECISIONS = ["fp32", "fp32_split"]


def dev():
    return torch.device("cuda:0")


def make_model(params, W, precision="fp32"):
    model = MOTMPNet(params)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
    model = model.to(dev()).train()
    model.gemm_precision = precision
    return model


def native_fwd_bwd(model, g, r):
    """(logits, grad_x, grad_edge_attr, {param: grad}, path counters of the forward + backward)"""
    xd = torch.from_numpy(g["x"]).to(dev()).requires_grad_(True)
    ead = torch.from_numpy(g["edge_attr"]).to(dev()).requires_grad_(True)
    model.zero_grad()
    capi.path_counters(reset=True)
    logits = model.hot_path(xd, torch.from_numpy(g["edge_index"]).to(dev()), ead)
    (logits * torch.from_numpy(r).to(dev())).sum().backward()
    torch.cuda.synchronize()
    counts = capi.path_counters(reset=True)
    pg = {k: p.grad.cpu().numpy() for k, p in model.named_parameters()}
    return logits.detach().cpu().numpy(), xd.grad.cpu().numpy(), ead.grad.cpu().numpy(), pg, counts


def oracle_fwd_bwd(params, W, g, r):
    Wt = O.to_tensors(W, requires_grad=True)
    xt = torch.from_numpy(g["x"]).requires_grad_(True)
    eat = torch.from_numpy(g["edge_attr"]).requires_grad_(True)
    _, logits, _, _ = O.forward(params, Wt, xt, torch.from_numpy(g["edge_index"]), eat, return_state=True)
    lg = torch.stack([l.view(-1) for l in logits])
    keys = list(Wt.keys())
    gr = torch.autograd.grad((lg * torch.from_numpy(r)).sum(), [xt, eat] + [Wt[k] for k in keys])
    return lg.detach().numpy(), gr[0].numpy(), gr[1].numpy(), {k: v.numpy() for k, v in zip(keys, gr[2:])}


def assert_dense_paths(counts, L, agg, split, block3=True, panel=None):
    panel = split if panel is None else panel
    """The kernels the dense reference-width configuration is supposed to take."""
    assert counts["edge_chain_fwd_split" if split else "edge_chain_fwd"] == L, counts
    assert counts["edge_chain_bwd_split" if split else "edge_chain_bwd"] == L, counts
    if block3:
        assert counts["segment_reduce_block3"] == L, counts        # the three scatter-adds of a step in one launch (E >= 96 N)
    else:
        assert counts["segment_reduce_block"] >= 2 * L, counts     # E / N = 64: the two by-node lists take the block kernel
    assert counts["edge_encoder"] == 1 and counts["edge_encoder_bwd"] == 1, counts
    if agg == "max":
        assert counts["aggregate_block" if block3 else "aggregate"] == L, counts   # max keeps the separate (arg-max recording) kernels
    else:
        assert counts["node_step32"] == L and counts["node_step32_bwd"] == L - 1, counts
    if panel:   # MPNHIP_PREC_FP32_SPLIT / FP32_WGSPLIT: every weight-gradient product is a job of the row-panel launches
        assert counts["gemm_tn_panel"] > 0 and counts["gemm_tn_small"] == 0 and counts["gemm_tn_mfma"] == 0 and counts["gemm_tn_generic"] == 0, counts
    else:
        assert counts["gemm_tn_small"] > 0 and counts["gemm_tn_mfma"] > 0, counts


def check_logits(got, ref, agg, what=""):
    if agg == "sum":
        ok, q = logits_close_per_element(got, ref)
        assert ok, "%s sum logits: max per-element error %.3g (x 1e-4 * max(1, |ref|))" % (what, q)
    else:
        ok, d = logits_close_abs(got, ref)
        assert ok, "%s %s logits: max abs error %.3g > 1e-4" % (what, agg, d)


# ------------------------------------------------------------------------------------ configs[2] stand-in, reference fixtures
@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("agg", ["sum", "mean", "max"])
def test_g12_dense_knn_against_reference_autograd(golden, agg, precision):
    z = golden(f"g12_dense_knn_{agg}.npz")
    g = synth.make_knn_graph(frames=20, dets=25, top_k=60, seed=3, node_in_dim=64)
    N, E = g["x"].shape[0], g["edge_index"].shape[1]
    assert E == int(z["E"]) and E >= 48 * N
    params = synth.model_params(32, 12, agg, node_in_dim=64)
    W = synth.make_weights(params, seed=7, gain=float(z["gain"]))
    model = make_model(params, W, precision)
    lg, gx, gea, pg, counts = native_fwd_bwd(model, g, synth.normal(11, (12, E)))
    assert_dense_paths(counts, 12, agg, precision == "fp32_split", block3=False)
    check_logits(lg[:, z["edge_ids"]], z["logits"], agg, "g12")
    for s in range(12):   # whole-tensor checksums of the reference run
        scale = max(1.0, float(z["step_max"][s]))
        assert abs(float(np.abs(lg[s]).astype(np.float64).sum()) - float(z["step_abssum"][s])) / (scale * E) < 1e-5, s
    bad, log = check_grads_against_fixture(z, pg, gx, gea, tol=GTOL, robust="loose")
    print("\n".join(log))
    assert not bad, "\n".join(bad)
    #  whatever left the STRICT bound must come with a counted decision flip against the fp32 arithmetic
    explain_loose_failures(check_grads_against_fixture(z, pg, gx, gea, tol=GTOL, robust=False)[0], model, params, W, g,
                           synth.normal(11, (12, E)), dev())


# ------------------------------------------------------------------------------------ configs[2] stand-in at its bench size
@pytest.mark.parametrize("precision", PRECISIONS)
def test_cfgC_standin_backward_against_oracle(precision):
    """The graph bench.py --config C runs (20 x 25 detections, top-150 kNN: E / N = 155, reference dims incl. the 2048-d
    node input), sum aggregation, weights scaled to O(1) logits; forward + every gradient against oracle autograd."""
    c = synth.CONFIGS["C"]
    g = synth.make_knn_graph(seed=1, **c["knn"])
    N, E = g["x"].shape[0], g["edge_index"].shape[1]
    assert E >= 48 * N
    params = synth.model_params(c["d"], c["L"], "sum")
    W = synth.make_weights(params, seed=7, gain=0.35)
    r = synth.normal(11, (c["L"], E))
    model = make_model(params, W, precision)
    lg, gx, gea, pg, counts = native_fwd_bwd(model, g, r)
    assert_dense_paths(counts, c["L"], "sum", precision == "fp32_split")
    lr, rx, rea, rpg = oracle_fwd_bwd(params, W, g, r)
    assert 0.5 < float(np.abs(lr[-1]).max()) < 50.0        # the logits carry signal and stay O(1)
    check_logits(lg, lr, "sum", "cfg-C")
    log, bad = [], []
    strict = []
    for name, a, b in [("grad_x", gx, rx), ("grad_edge_attr", gea, rea)] + [(k, pg[k], rpg[k]) for k in W]:
        ok, msg = grad_close(a, b, GTOL, "loose", name, log)
        if not ok:
            bad.append(msg)
        if not grad_close(a, b, GTOL, False, name)[0]:
            strict.append(msg)
    print("\n".join(log))
    assert not bad, "\n".join(bad)
    explain_loose_failures(strict, model, params, W, g, r, dev())


# ------------------------------------------------------------------------------------ configs[3] stand-in
@pytest.mark.parametrize("precision", PRECISIONS + ["fp32_wgsplit"])
def test_cfgD_graphs_and_their_batch_backward(precision):
    """The 8 graphs bench.py --config D gives the 8 ranks (20 frames x 7 detections, top-100 kNN, E / N = 103, d = 32, L = 4):
    each graph alone, and all 8 as one torch_geometric-style batch, forward + backward against oracle autograd."""
    c = synth.CONFIGS["D"]
    graphs = [synth.make_knn_graph(seed=1 + rank, node_in_dim=64, **c["knn"]) for rank in range(8)]
    params = synth.model_params(c["d"], c["L"], "sum", num_class_steps=3, node_in_dim=64)
    W = synth.make_weights(params, seed=7, gain=0.5)
    model = make_model(params, W, precision)
    split = precision == "fp32_split"
    worst = {}
    for gi, g in enumerate(graphs + [synth.batch_graphs(graphs)]):
        N, E = g["x"].shape[0], g["edge_index"].shape[1]
        assert E >= 48 * N
        r = synth.normal(20 + gi, (c["L"], E))
        lg, gx, gea, pg, counts = native_fwd_bwd(model, g, r)
        assert_dense_paths(counts, c["L"], "sum", split, panel=precision != "fp32")
        lr, rx, rea, rpg = oracle_fwd_bwd(params, W, g, r)
        check_logits(lg, lr, "sum", "cfg-D graph %d" % gi)
        strict = []
        for name, a, b in [("grad_x", gx, rx), ("grad_edge_attr", gea, rea)] + [(k, pg[k], rpg[k]) for k in W]:
            e = nerr(a, b)
            worst[name] = max(worst.get(name, 0.0), e)
            ok, msg = grad_close(a, b, GTOL, "loose", name)
            assert ok, "graph %d %s" % (gi, msg)
            if not grad_close(a, b, GTOL, False, name)[0]:
                strict.append(msg)
        explain_loose_failures(strict, model, params, W, g, r, dev())
    print("worst normalised gradient errors over the 9 graphs:", {k: "%.2g" % v for k, v in worst.items()})


# ------------------------------------------------------------------------------------ the headline workload, O(1) logits
@pytest.mark.parametrize("precision", PRECISIONS)
def test_g11_cfgB_sum_o1_against_reference_autograd(golden, precision):
    """BASELINE.json configs[1] graph and widths, node_agg_fn = 'sum' (shipped default), 12 steps, TRAINING: logits per element
    against the reference (|d| <= 1e-4 max(1, |ref|) on every sampled edge, O(1) magnitudes) and every gradient against the
    reference's autograd (unpinned: loose bound, statistics printed; sharp version: test_gpu_pinned.py::test_cfgB)."""
    z = golden("g11_cfgB_sum_o1.npz")
    c = synth.CONFIGS["B"]
    g = synth.make_graph(c["N"], c["E"], seed=1)
    assert synth.checksum(g["x"]) == int(z["cs_x"])
    params = synth.model_params(c["d"], c["L"], "sum")
    W = synth.make_weights(params, seed=7, gain=float(z["gain"]))
    model = make_model(params, W, precision)
    lg, gx, gea, pg, counts = native_fwd_bwd(model, g, synth.normal(11, (c["L"], c["E"])))
    split = precision == "fp32_split"
    assert counts["edge_chain_fwd_split" if split else "edge_chain_fwd"] == c["L"], counts
    assert counts["edge_chain_bwd_split" if split else "edge_chain_bwd"] == c["L"], counts
    ok, q = logits_close_per_element(lg[:, z["edge_ids"]], z["logits"])
    assert ok, "per-element logit error %.3g" % q
    for s in range(c["L"]):
        scale = max(1.0, float(z["step_max"][s]))
        assert abs(float(np.abs(lg[s]).max()) - float(z["step_max"][s])) / scale < 1e-4
        assert abs(float(np.abs(lg[s]).astype(np.float64).sum()) - float(z["step_abssum"][s])) / (scale * c["E"]) < 1e-5
    bad, log = check_grads_against_fixture(z, pg, gx, gea, tol=GTOL, robust="loose")
    print("\n".join(log))
    assert not bad, "\n".join(bad)
    explain_loose_failures(check_grads_against_fixture(z, pg, gx, gea, tol=GTOL, robust=False)[0], model, params, W, g,
                           synth.normal(11, (c["L"], c["E"])), dev())


# ------------------------------------------------------------------------------------ long-segment kernels in isolation
@pytest.mark.parametrize("agg", ["sum", "mean", "max"])
def test_block_per_segment_aggregation_matches_sequential_order_oracle(agg):
    """node_agg_fn over long segments (the block-per-segment kernel with its fixed LDS tree) against the oracle's sequential
    scatter; max must be exact (and its ties at 0 resolve to the earliest edge like the sequential scan)."""
    from mpntrackseg_amd.mpn import NodeAggFn
    m, dim, x_size = 20000, 32, 100
    src = np.maximum(synth.normal(4, (m, dim), stream=1), 0)
    row = (synth.uniform01(4, m, stream=2) * x_size).astype(np.int64)
    row[row == 7] = 8   # an empty segment
    capi.path_counters(reset=True)
    out = NodeAggFn(agg)(torch.from_numpy(src).to(dev()), torch.from_numpy(row).to(dev()), x_size).cpu().numpy()
    counts = capi.path_counters(reset=True)
    ref = O.AGG[agg](torch.from_numpy(src), torch.from_numpy(row), x_size).numpy()
    if agg == "max":
        assert np.array_equal(out, ref)
    else:
        assert nerr(out, ref) < 2e-6
    assert (out[7] == 0).all()
    print(counts)
