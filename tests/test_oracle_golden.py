"""The CPU oracle (oracle/mpn_oracle.py) against fixtures produced by the imported reference
(tools/make_golden.py).  Tolerances: the oracle uses the same torch CPU ops in the same order, so
agreement is expected to be exact on the authoring machine; a different host CPU may pick other
GEMM kernels, hence 2e-6 relative-to-max slack."""
import numpy as np
import pytest
import torch

from mpntrackseg_amd import synth
from oracle import mpn_oracle as O


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(1.0, float(np.abs(b).max())))


def weights_of(z):
    return {k[2:]: z[k] for k in z.files if k.startswith("W:")}


@pytest.mark.parametrize("agg", ["sum", "mean", "max"])
def test_g1_tiny_forward_and_grads(golden, agg):
    z = golden(f"g1_tiny_{agg}.npz")
    L = int(z["L"])
    params = synth.model_params(32, L, agg, num_class_steps=3, node_in_dim=int(z["node_in_dim"]))
    # regenerated inputs are bit-identical to the stored ones
    W0 = synth.make_weights(params, seed=7)
    for k, v in weights_of(z).items():
        assert np.array_equal(W0[k], v), k
    g = synth.make_graph(int(z["N"]), int(z["E"]), T=10, seed=1, node_in_dim=int(z["node_in_dim"]))
    assert np.array_equal(g["edge_index"], z["edge_index"]) and np.array_equal(g["edge_attr"], z["edge_attr"])

    W = O.to_tensors(W0, requires_grad=True)
    x4 = torch.from_numpy(z["x4"])
    cls = O.forward(params, W, x4, torch.from_numpy(z["edge_index"]), torch.from_numpy(z["edge_attr"]))
    assert len(cls) == 3 and cls[0].shape == (int(z["E"]), 1)
    for s in range(3):
        assert rel_err(cls[s].detach().numpy().reshape(-1), z["logits"][L - 3 + s]) < 2e-6

    xp = torch.from_numpy(z["x_pooled"]).requires_grad_(True)
    ea = torch.from_numpy(z["edge_attr"]).requires_grad_(True)
    _, logits, xL, eL = O.forward(params, W, xp, torch.from_numpy(z["edge_index"]), ea, return_state=True)
    assert rel_err(xL.detach().numpy(), z["x_final"]) < 2e-6
    assert rel_err(eL.detach().numpy(), z["e_final"]) < 2e-6
    r = torch.from_numpy(z["r"])
    loss = sum((logits[s].view(-1) * r[s]).sum() for s in range(L))
    keys = list(W.keys())
    grads = torch.autograd.grad(loss, [xp, ea] + [W[k] for k in keys])
    assert rel_err(grads[0].numpy(), z["grad_x"]) < 1e-5
    assert rel_err(grads[1].numpy(), z["grad_edge_attr"]) < 1e-5
    for k, gr in zip(keys, grads[2:]):
        assert rel_err(gr.numpy(), z["G:" + k]) < 1e-5, k


@pytest.mark.parametrize("agg", ["sum", "mean", "max"])
def test_g4_structure(golden, agg):
    z = golden("g4_structure.npz")
    params = synth.model_params(32, 3, agg, node_in_dim=64)
    W = O.to_tensors(synth.make_weights(params, seed=8))
    ei = torch.from_numpy(z["edge_index"])
    # the case really has a self loop, an isolated node and interleaved halves
    assert int((ei[0] == ei[1]).sum()) == 1
    assert 0 not in set(ei.flatten().tolist())
    d = (ei[0] < ei[1]).numpy()
    assert not d[: d.size // 2].all()
    with torch.no_grad():
        _, logits, xL, eL = O.forward(params, W, torch.from_numpy(z["x"]), ei, torch.from_numpy(z["edge_attr"]),
                                      return_state=True)
    assert rel_err(torch.stack(logits).numpy().reshape(3, -1), z[f"logits_{agg}"]) < 2e-6
    assert rel_err(xL.numpy(), z[f"x_final_{agg}"]) < 2e-6
    assert rel_err(eL.numpy(), z[f"e_final_{agg}"]) < 2e-6


def test_g4_empty_graph(golden):
    z = golden("g4_structure.npz")
    params = synth.model_params(32, 2, "sum", node_in_dim=64)
    W = O.to_tensors(synth.make_weights(params, seed=8))
    with torch.no_grad():
        cls, logits, xL, eL = O.forward(params, W, torch.from_numpy(z["x"][:5]), torch.zeros((2, 0), dtype=torch.int64),
                                        torch.zeros((0, 6)), return_state=True)
    assert logits[0].shape == (0, 1)
    assert rel_err(xL.numpy(), z["empty_x_final"]) < 2e-6


@pytest.mark.parametrize("agg", ["sum", "mean", "max"])
def test_g5_modules(golden, agg):
    z = golden("g5_modules.npz")
    params = synth.model_params(32, 1, agg, node_in_dim=64)
    W = O.to_tensors(synth.make_weights(params, seed=9))
    ei = torch.from_numpy(z["edge_index"])
    with torch.no_grad():
        xo, eo = O.meta_layer(torch.from_numpy(z["x_in"]), ei, torch.from_numpy(z["e_in"]), W, agg)
        ao = O.AGG[agg](torch.from_numpy(z["msg"]), ei[0], 40)
    assert rel_err(xo.numpy(), z[f"meta_x_{agg}"]) < 2e-6
    assert rel_err(eo.numpy(), z[f"meta_e_{agg}"]) < 2e-6
    assert np.array_equal(ao.numpy(), z[f"agg_{agg}"]) or rel_err(ao.numpy(), z[f"agg_{agg}"]) < 1e-6


def test_g0_zero_steps(golden):
    z = golden("g0_l0.npz")
    params = synth.model_params(32, 0, "sum", num_class_steps=0, node_in_dim=64)
    W = O.to_tensors(synth.make_weights(params, seed=7))
    g = synth.make_graph(30, 100, T=5, seed=2, node_in_dim=64)
    with torch.no_grad():
        cls = O.forward(params, W, torch.from_numpy(g["x"]).view(30, 64, 1, 1), torch.from_numpy(g["edge_index"]),
                        torch.from_numpy(g["edge_attr"]))
    assert len(cls) == 1
    assert rel_err(cls[0].numpy().reshape(-1), z["logits"]) < 2e-6


@pytest.mark.parametrize("agg", ["sum", "mean", "max"])
def test_g2_cfgA(golden, agg):
    z = golden(f"g2_cfgA_{agg}.npz")
    c = synth.CONFIGS["A"]
    g = synth.make_graph(c["N"], c["E"], seed=1)
    params = synth.model_params(c["d"], c["L"], agg)
    W0 = synth.make_weights(params, seed=7)
    assert synth.checksum(g["x"]) == int(z["cs_x"])
    assert synth.checksum(g["edge_index"]) == int(z["cs_edge_index"])
    assert synth.checksum(g["edge_attr"]) == int(z["cs_edge_attr"])
    assert synth.checksum(np.concatenate([v.ravel() for v in W0.values()])) == int(z["cs_weights"])
    with torch.no_grad():
        _, logits, _, _ = O.forward(params, O.to_tensors(W0), torch.from_numpy(g["x"]),
                                    torch.from_numpy(g["edge_index"]), torch.from_numpy(g["edge_attr"]),
                                    return_state=True)
    lg = torch.stack(logits).numpy().reshape(c["L"], -1)
    for s in range(c["L"]):
        assert rel_err(lg[s], z["logits"][s]) < 2e-6, s


def test_g3_cfgB_mean(golden):
    """cfg-B on the oracle takes ~1-2 s per aggregation on 8 cores; one aggregation keeps the CPU
    suite short, the others are covered on the GPU side."""
    agg = "mean"
    z = golden(f"g3_cfgB_{agg}.npz")
    c = synth.CONFIGS["B"]
    g = synth.make_graph(c["N"], c["E"], seed=1)
    params = synth.model_params(c["d"], c["L"], agg)
    W0 = synth.make_weights(params, seed=7)
    assert synth.checksum(g["x"]) == int(z["cs_x"])
    assert synth.checksum(np.concatenate([v.ravel() for v in W0.values()])) == int(z["cs_weights"])
    with torch.no_grad():
        _, logits, _, _ = O.forward(params, O.to_tensors(W0), torch.from_numpy(g["x"]),
                                    torch.from_numpy(g["edge_index"]), torch.from_numpy(g["edge_attr"]),
                                    return_state=True)
    lg = torch.stack(logits).numpy().reshape(c["L"], -1)
    assert rel_err(lg[:, z["edge_ids"]], z["logits"]) < 5e-6
    assert np.allclose(np.abs(lg).max(1), z["step_max"], rtol=1e-5)


def test_g7_graph_utils_oracle_matches_reference(golden):
    """oracle/tracker_oracle.py against the reference's utils/graph.py outputs (tests/golden/g7_graph_utils.npz)."""
    import torch
    from oracle import tracker_oracle as T
    z = golden("g7_graph_utils.npz")
    det = {k[4:]: z[k] for k in z.files if k.startswith("det:")}
    emb = torch.from_numpy(det["reid"])
    for tag, mfd in (("max", "max"), ("d3", 3)):
        ei = T.get_time_valid_conn_ixs(det["frame"], mfd)
        assert np.array_equal(ei.numpy(), z[f"{tag}:edge_ixs"])
        feats = T.compute_edge_feats_dict(ei, det, float(z["fps"]))
        got = torch.stack([feats[k] for k in T.EDGE_FEAT_NAMES]).T.numpy()
        assert np.allclose(got, z[f"{tag}:feats"], rtol=1e-6, atol=1e-7)
        d = T.pairwise_distance(emb, ei).view(-1)
        assert np.allclose(d.numpy(), z[f"{tag}:emb_dist"], rtol=1e-6)
        ei2 = torch.cat((ei, torch.stack((ei[1], ei[0]))), dim=1)
        for k in (3, 8):
            for rec in (0, 1):
                m = T.get_knn_mask(d, ei, len(det["frame"]), k, bool(rec), symmetric_edges=False)
                assert np.array_equal(m.numpy(), z[f"{tag}:knn_k{k}_r{rec}_pairs"])
                m2 = T.get_knn_mask(torch.cat((d, d)), ei2, len(det["frame"]), k, bool(rec), symmetric_edges=True)
                assert np.array_equal(m2.numpy(), z[f"{tag}:knn_k{k}_r{rec}_sym"])


def test_g8_embedding_loader_oracle_matches_reference(golden):
    """oracle/embeddings_oracle.py against the reference's load_precomputed_embeddings (tests/golden/g8_embedding_files.npz:
    per-frame files with detections the query does not contain, one frame none of whose detections survives)."""
    from oracle import embeddings_oracle as EO
    z = golden("g8_embedding_files.npz")
    for tag in ("1d", "3d"):
        out = EO.load_precomputed_embeddings(z["stored_" + tag], z["stored_frame"], z["det_frame"], z["det_id"])
        assert out.shape == z["out_" + tag].shape and np.array_equal(out, z["out_" + tag])
    # the reference's assertion fires when the query is not in stored order
    with pytest.raises(AssertionError):
        EO.load_precomputed_embeddings(z["stored_1d"], z["stored_frame"], z["det_frame"][::-1], z["det_id"][::-1])


def _oracle_fwd_bwd(params, W0, g, r):
    W = O.to_tensors(W0, requires_grad=True)
    xp = torch.from_numpy(g["x"]).requires_grad_(True)
    ea = torch.from_numpy(g["edge_attr"]).requires_grad_(True)
    _, logits, xL, eL = O.forward(params, W, xp, torch.from_numpy(g["edge_index"]), ea, return_state=True)
    lg = torch.stack([l.view(-1) for l in logits])
    loss = (lg * torch.from_numpy(r)).sum()
    keys = list(W.keys())
    grads = torch.autograd.grad(loss, [xp, ea] + [W[k] for k in keys])
    return lg.detach().numpy(), xL.detach().numpy(), {k: v.numpy() for k, v in zip(keys, grads[2:])}, grads[0].numpy(), grads[1].numpy()


@pytest.mark.parametrize("agg", ["sum", "mean", "max"])
def test_g12_dense_knn_forward_and_reference_autograd(golden, agg):
    """BASELINE.json configs[2] stand-in: the oracle's forward and its autograd against the REFERENCE's on the dense kNN graph."""
    from gradcheck import check_grads_against_fixture
    z = golden(f"g12_dense_knn_{agg}.npz")
    g = synth.make_knn_graph(frames=20, dets=25, top_k=60, seed=3, node_in_dim=64)
    E = g["edge_index"].shape[1]
    assert E == int(z["E"]) and synth.checksum(g["edge_index"]) == int(z["cs_edge_index"]) and E >= 48 * 500
    params = synth.model_params(32, 12, agg, node_in_dim=64)
    W0 = synth.make_weights(params, seed=7, gain=float(z["gain"]))
    lg, xL, pg, gx, gea = _oracle_fwd_bwd(params, W0, g, synth.normal(11, (12, E)))
    assert np.abs(lg[:, z["edge_ids"]] - z["logits"]).max() < 2e-6 * max(1.0, float(z["step_max"].max()))
    assert rel_err(xL, z["x_final"]) < 2e-6
    bad, log = check_grads_against_fixture(z, pg, gx, gea, tol=1e-5)
    assert not bad, "\n".join(bad)


def test_g11_cfgB_sum_o1_forward_and_reference_autograd(golden):
    """The headline workload (cfg-B, sum, 12 steps) with O(1) logits: oracle forward + autograd against the reference's."""
    from gradcheck import check_grads_against_fixture
    z = golden("g11_cfgB_sum_o1.npz")
    c = synth.CONFIGS["B"]
    g = synth.make_graph(c["N"], c["E"], seed=1)
    assert synth.checksum(g["x"]) == int(z["cs_x"])
    params = synth.model_params(c["d"], c["L"], "sum")
    W0 = synth.make_weights(params, seed=7, gain=float(z["gain"]))
    assert synth.checksum(np.concatenate([v.ravel() for v in W0.values()])) == int(z["cs_weights"])
    lg, xL, pg, gx, gea = _oracle_fwd_bwd(params, W0, g, synth.normal(11, (c["L"], c["E"])))
    q = np.abs(lg[:, z["edge_ids"]] - z["logits"]) / np.maximum(1.0, np.abs(z["logits"]))
    assert q.max() < 2e-6
    assert rel_err(xL[:64], z["x_final_rows"]) < 2e-6
    bad, log = check_grads_against_fixture(z, pg, gx, gea, tol=1e-5)
    assert not bad, "\n".join(bad)


# ------------------------------------------------------------------------------------ the callers' steps
@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_g9_tracking_loss_oracle_against_reference_compute_loss(golden, tag):
    """oracle/loss_oracle.py against MOTNeuralSolver._compute_loss of the reference (value and autograd gradient)."""
    from oracle import loss_oracle as LO
    z = golden("g9_loss_metrics.npz")
    logits = torch.from_numpy(z[f"{tag}:logits"]).clone().requires_grad_(True)
    labels = torch.from_numpy(z[f"{tag}:labels"])
    k, E = logits.shape
    loss = LO.tracking_loss([logits[s].view(E, 1) for s in range(k)], labels, weight=float(z[f"{tag}:weight"]))
    loss.backward()
    assert abs(float(loss) - float(z[f"{tag}:loss"])) <= 1e-6 * max(1.0, abs(float(z[f"{tag}:loss"])))
    assert np.abs(logits.grad.numpy() - z[f"{tag}:grad"]).max() <= 1e-7 * max(1.0, np.abs(z[f"{tag}:grad"]).max())


@pytest.mark.parametrize("tag", ["m1", "m2", "m3"])
def test_g9_step_metrics_oracle_against_reference(golden, tag):
    from oracle import loss_oracle as LO
    z = golden("g9_loss_metrics.npz")
    ei = torch.from_numpy(z["edge_index"])
    m = LO.compute_perform_metrics([torch.from_numpy(z[f"{tag}:logit"])], ei, torch.from_numpy(z[f"{tag}:labels"]), 120)
    got = np.array([m["accuracy"], m["recall"], m["precision"], m["constr_sr"]])
    assert np.abs(got - z[f"{tag}:metrics"]).max() < 1e-6


@pytest.mark.parametrize("tag", ["w1", "w2"])
def test_g10_sliding_window_oracle_against_reference_tracker(golden, tag):
    """oracle/tracker_oracle.evaluate_graph_in_batches (driving oracle/mpn_oracle.forward) against
    MPNTracker._evaluate_graph_in_batches of the reference driving the reference model."""
    from oracle import tracker_oracle as TO
    z = golden("g10_windows.npz")
    inactive, recip, fpg, top_k = [int(v) for v in z[f"{tag}:cfg"]]
    params = synth.model_params(32, 4, "sum", num_class_steps=2, node_in_dim=64)
    W = O.to_tensors(synth.make_weights(params, seed=7, gain=0.6))

    def fwd(x, ei, ea):
        with torch.no_grad():
            return O.forward(params, W, x, ei, ea)[-1].view(-1)
    got = TO.evaluate_graph_in_batches(fwd, torch.from_numpy(z[f"{tag}:x"]), torch.from_numpy(z[f"{tag}:edge_index"]),
                                       torch.from_numpy(z[f"{tag}:edge_attr"]), torch.from_numpy(z[f"{tag}:reid_emb_dists"]),
                                       z[f"{tag}:frame"], fpg, top_k, reciprocal_k_nns=bool(recip),
                                       set_pruned_edges_to_inactive=bool(inactive))
    ref = z[f"{tag}:final_edge_preds"]
    assert got.shape == ref.shape and float(np.abs(ref).max()) > 0.05
    assert np.abs(got.numpy() - ref).max() < 2e-6


def _g14_case(z, tag):
    det = {k: z[f"{tag}:{k}"] for k in ("frame", "bb_height", "bb_width", "feet_x", "feet_y", "reid")}
    inference, top_k, recip, mfd = (int(v) for v in z[f"{tag}:cfg"])
    return det, bool(inference), (None if top_k < 0 else top_k), bool(recip), ("max" if mfd < 0 else mfd)


@pytest.mark.parametrize("tag", ["train_recip", "train_plain", "infer", "infer_mfd"])
def test_g14_construct_graph_oracle_against_reference_motgraph(golden, tag):
    """The reference's own MOTGraph._get_edge_ixs + construct_graph_object (data/mot_graph.py:195-218, 283-317; tools/make_golden.py
    gen_g14) -- training mode with the kNN pruning inside (reciprocal or not) and inference mode -- pins the oracle's composition:
    identical edge lists, features to fp32 rounding."""
    from oracle import tracker_oracle as T
    z = golden("g14_construct_graph.npz")
    det, inference, top_k, recip, mfd = _g14_case(z, tag)
    names = ["secs_time_dists", "norm_feet_x_dists", "norm_feet_y_dists", "bb_height_dists", "bb_width_dists", "emb_dist"]
    got = T.construct_graph(det, torch.from_numpy(det["reid"]), 25.0, mfd, names, top_k_nns=top_k, reciprocal_k_nns=recip,
                            inference_mode=inference)
    assert np.array_equal(got["edge_index"].numpy(), z[f"{tag}:edge_index"])
    assert np.allclose(got["edge_attr"].numpy(), z[f"{tag}:edge_attr"], rtol=1e-6, atol=1e-7)
    if inference:
        assert np.allclose(got["reid_emb_dists"].numpy(), z[f"{tag}:reid_emb_dists"], rtol=1e-6)


@pytest.mark.parametrize("agg", ["sum", "mean", "max"])
def test_g15_stock_torch_replica_against_reference_batchnorm_training(golden, agg):
    """tests/modular_ref.py (the float64 stock-torch composition tests/test_gpu_modular.py checks the HIP layer-by-layer path against)
    reproduces the REFERENCE's own training-mode forward, autograd and BatchNorm running statistics (g15, tools/make_golden.py
    gen_g15): the mirror's modules ARE the reference's (same Sequential, same keys), composed the same way."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from modular_ref import ref_forward
    from mpntrackseg_amd.mpn import MOTMPNet
    z = golden("g15_batchnorm_train.npz")
    N, E, L, nin = 90, 700, 2, 48
    params = synth.model_params(32, L, agg, node_in_dim=nin)
    for k in ("encoder_feats_dict", "edge_model_feats_dict", "node_model_feats_dict", "classifier_feats_dict"):
        params[k] = dict(params[k], use_batchnorm=True, dropout_p=0)
    g = synth.make_graph(N, E, seed=4, node_in_dim=nin)
    model = MOTMPNet(params)
    model.load_state_dict({k[len(agg) + 7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(agg + ":state:")}, strict=True)
    model = model.double().train()
    x = torch.from_numpy(g["x"]).double().requires_grad_(True)
    ea = torch.from_numpy(g["edge_attr"]).double().requires_grad_(True)
    lg = ref_forward(model, x, torch.from_numpy(g["edge_index"]), ea, agg)
    (lg * torch.from_numpy(synth.normal(12, (L, E))).double()).sum().backward()
    assert np.abs(lg.detach().numpy() - z[agg + ":logits"]).max() < 1e-9
    assert np.abs(x.grad.numpy() - z[agg + ":grad_x"]).max() < 1e-9 * max(1.0, np.abs(z[agg + ":grad_x"]).max())
    named = dict(model.named_parameters())
    for k in z.files:
        if k.startswith(agg + ":grad:"):
            ref = z[k]
            assert np.abs(named[k[len(agg) + 6:]].grad.numpy() - ref).max() < 1e-9 * max(1.0, np.abs(ref).max()), k
    bufs = dict(model.named_buffers())
    for k in z.files:
        if k.startswith(agg + ":after:"):
            assert np.abs(bufs[k[len(agg) + 7:]].double().numpy() - z[k]).max() < 1e-9, k


def test_recorded_decisions_compare_like_a_direct_compare_run():
    """tests/pinned.py::oracle_compare keeps ONE free-running float64 forward per test case ("record" mode: every site's own ReLU
    decisions bit-packed + the pre-activations near zero) and compares the decisions of several HIP runs against it; the
    statistics must be those of a direct "compare" run: same units, same mismatch counts per site, same worst margin while the
    differing units lie inside the recorded band, and >= the band's edge (far beyond any accepted margin) otherwise."""
    from mpntrackseg_amd import synth
    g = synth.make_graph(60, 420, T=6, seed=3, node_in_dim=24)
    params = synth.model_params(32, 3, "mean", node_in_dim=24)
    W = synth.make_weights(params, seed=4)
    Wt = {k: torch.from_numpy(v).double() for k, v in W.items()}
    x, ei, ea = torch.from_numpy(g["x"]).double(), torch.from_numpy(g["edge_index"]), torch.from_numpy(g["edge_attr"]).double()
    rec = O.Decisions(None, "record")
    with torch.no_grad(), O.decisions(rec):
        O.forward(params, Wt, x, ei, ea)
    # "given" = the oracle's own decisions with some units flipped: those with the smallest |z| of a site (knife-edge) and,
    # in the second variant, one unit far from zero
    sites = {}
    for site, rows, packed, shape, scale, near_idx, near_abs in rec.recorded:
        own = torch.from_numpy(np.unpackbits(packed, count=int(np.prod(shape))).astype(bool)).view(shape)
        full = sites.get(site)
        if rows is None:
            sites[site] = own.clone()
        else:
            if full is None:
                full = sites[site] = torch.zeros((rows.shape[0], shape[1]), dtype=torch.bool)
            full[rows] = own
    for far in (False, True):
        given = {k: v.clone() for k, v in sites.items()}
        flipped = 0
        for site, rows, packed, shape, scale, near_idx, near_abs in rec.recorded[:6]:
            if len(near_idx) and rows is None:
                j = int(near_idx[int(np.argmin(near_abs))])
                given[site].view(-1)[j] = ~given[site].view(-1)[j]
                flipped += 1
        if far:
            site, rows, packed, shape, scale, near_idx, near_abs = rec.recorded[0]
            cand = np.setdiff1d(np.arange(int(np.prod(shape))), near_idx)
            given[site].view(-1)[int(cand[0])] = ~given[site].view(-1)[int(cand[0])]
            flipped += 1
        assert flipped >= 2
        direct = O.Decisions(given, "compare")
        with torch.no_grad(), O.decisions(direct):
            O.forward(params, Wt, x, ei, ea)
        got = rec.compare_recorded(given)
        assert got.units == direct.units and got.mismatches == direct.mismatches == flipped and got.per_site == direct.per_site
        if far:
            assert direct.worst_margin >= O.Decisions.NEAR and got.worst_margin >= O.Decisions.NEAR
        else:
            assert abs(got.worst_margin - direct.worst_margin) <= 1e-12 * max(1.0, direct.worst_margin)


@pytest.mark.parametrize("tag", ["g3", "g8"])
def test_g16_graph_batched_loss_oracle_against_the_reference_per_graph_loss(golden, tag):
    """oracle/loss_oracle.py::tracking_loss_graphs against the reference's _compute_loss evaluated graph by graph and averaged
    (accumulate_grad_batches in space; tests/golden/g16_loss_graphs.npz): value, per-graph values and autograd gradient."""
    from oracle import loss_oracle as LO
    z = golden("g16_loss_graphs.npz")
    logits = torch.from_numpy(z[f"{tag}:logits"]).clone().requires_grad_(True)
    labels = torch.from_numpy(z[f"{tag}:labels"])
    ptr = z[f"{tag}:edge_ptr"]
    k, E = logits.shape
    w = float(z[f"{tag}:weight"])
    loss = LO.tracking_loss_graphs([logits[s].view(E, 1) for s in range(k)], labels, ptr, weight=w)
    loss.backward()
    assert abs(float(loss) - float(z[f"{tag}:loss"])) <= 1e-6 * max(1.0, abs(float(z[f"{tag}:loss"])))
    assert float(np.abs(logits.grad.numpy() - z[f"{tag}:grad"]).max()) <= 1e-7 * max(1.0, float(np.abs(z[f"{tag}:grad"]).max()))
    for g in range(len(ptr) - 1):
        a, b = int(ptr[g]), int(ptr[g + 1])
        li = LO.tracking_loss([logits.detach()[s, a:b] for s in range(k)], labels[a:b], weight=w)
        assert abs(float(li) - float(z[f"{tag}:per_graph"][g])) <= 1e-6 * max(1.0, abs(float(z[f"{tag}:per_graph"][g])))
