"""Graph construction and sliding-window inference (8f-3) through the C ABI.

(a) `mpntrackseg_amd.graph` against the reference's own utils/graph.py outputs (tests/golden/g7_graph_utils.npz):
    integer / boolean results bit-exact, fp32 features within 1e-6 relative (logf vs torch.log: 1 ulp; the embedding
    norm is re-associated across a wavefront);
(b) `mpntrackseg_amd.tracker.evaluate_graph_in_batches` against the oracle's restatement of
    mpn_tracker.py:143-198 driven by the CPU oracle forward, same weights: |dp| <= 2e-5 on the averaged probabilities;
(c) properties: windows batched block-diagonally == one by one; windows sharded over 2 "ranks" and summed == 1 rank."""
import numpy as np
import pytest
import torch

from mpntrackseg_amd import graph as G, synth, tracker
from mpntrackseg_amd.mpn import MOTMPNet
from oracle import mpn_oracle as O, tracker_oracle as T

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def test_graph_utils_match_reference(golden):
    z = golden("g7_graph_utils.npz")
    det = {k[4:]: z[k] for k in z.files if k.startswith("det:")}
    emb = torch.from_numpy(det["reid"]).to(dev())
    n = len(det["frame"])
    for tag, mfd in (("max", "max"), ("d3", 3)):
        ei = G.get_time_valid_conn_ixs(torch.from_numpy(det["frame"]).to(dev()), mfd)
        assert ei.dtype == torch.int64 and np.array_equal(ei.cpu().numpy(), z[f"{tag}:edge_ixs"])
        feats = G.compute_edge_feats_dict(ei, det, float(z["fps"]))
        got = torch.stack([feats[k] for k in G.EDGE_FEAT_NAMES]).T.cpu().numpy()
        assert np.allclose(got, z[f"{tag}:feats"], rtol=2e-6, atol=1e-6)
        d = G.pairwise_distance(emb, ei)
        assert np.allclose(d.cpu().numpy(), z[f"{tag}:emb_dist"], rtol=1e-5)
        dref = torch.from_numpy(z[f"{tag}:emb_dist"]).to(dev())  # identical distances -> identical ranks
        ei2 = torch.cat((ei, torch.stack((ei[1], ei[0]))), dim=1)
        for k in (3, 8):
            for rec in (0, 1):
                m = G.get_knn_mask(dref, ei, n, k, reciprocal_k_nns=bool(rec), symmetric_edges=False)
                assert np.array_equal(m.cpu().numpy(), z[f"{tag}:knn_k{k}_r{rec}_pairs"]), (tag, k, rec)
                m2 = G.get_knn_mask(torch.cat((dref, dref)), ei2, n, k, reciprocal_k_nns=bool(rec), symmetric_edges=True)
                assert np.array_equal(m2.cpu().numpy(), z[f"{tag}:knn_k{k}_r{rec}_sym"]), (tag, k, rec)


def test_graph_utils_edge_cases():
    d0 = dev()
    # no nodes / one frame only: no pairs
    assert G.get_time_valid_conn_ixs(torch.zeros(0, dtype=torch.int64, device=d0), 'max').shape == (2, 0)
    assert G.get_time_valid_conn_ixs(torch.ones(5, dtype=torch.int64, device=d0), 'max').shape == (2, 0)
    # unsorted frames are handled like the reference's dense formulation
    f = torch.tensor([3, 1, 2, 1, 7], dtype=torch.int64)
    assert np.array_equal(G.get_time_valid_conn_ixs(f.to(d0), 2).cpu().numpy(), T.get_time_valid_conn_ixs(f, 2).numpy())
    # ties in distance: ranked by column index, like a stable argsort
    ei = torch.tensor([[0, 0, 0, 1, 2, 3], [1, 2, 3, 0, 0, 0]], dtype=torch.int64)
    dist = torch.tensor([1.0, 1.0, 1.0, 1.0, 1.0, 1.0])
    for rec in (False, True):
        want = T.get_knn_mask(dist, ei, 4, 2, rec, True)
        got = G.get_knn_mask(dist.to(d0), ei.to(d0), 4, 2, reciprocal_k_nns=rec, symmetric_edges=True)
        assert np.array_equal(got.cpu().numpy(), want.numpy())
    assert G.get_knn_mask(torch.zeros(0, device=d0), torch.zeros((2, 0), dtype=torch.int64, device=d0), 3, 2).numel() == 0


def test_construct_graph_matches_oracle():
    det = synth.make_detections(frames=10, seed=11)
    emb = torch.from_numpy(det["reid"])
    names = list(G.EDGE_FEAT_NAMES) + ["emb_dist"]
    for inference, mfd in ((True, 'max'), (False, 4)):
        want = T.construct_graph(det, emb, 30.0, mfd, names, top_k_nns=6, reciprocal_k_nns=True, inference_mode=inference)
        got = G.construct_graph(det, emb.to(dev()), 30.0, mfd, names, top_k_nns=6, reciprocal_k_nns=True,
                                inference_mode=inference)
        assert np.array_equal(got["edge_index"].cpu().numpy(), want["edge_index"].numpy())
        assert np.allclose(got["edge_attr"].cpu().numpy(), want["edge_attr"].numpy(), rtol=1e-5, atol=1e-6)
        assert np.allclose(got["reid_emb_dists"].cpu().numpy(), want["reid_emb_dists"].numpy(), rtol=1e-5)


@pytest.mark.parametrize("tag", ["train_recip", "train_plain", "infer", "infer_mfd"])
def test_construct_graph_matches_the_reference_motgraph(golden, tag):
    """g14: the reference's MOTGraph._get_edge_ixs + construct_graph_object run on synthetic detections (tools/make_golden.py
    gen_g14): the device-side construct_graph gives the same edge list (bit-exact) and the same features (<= 2e-6 relative)."""
    z = golden("g14_construct_graph.npz")
    det = {k: z[f"{tag}:{k}"] for k in ("frame", "bb_height", "bb_width", "feet_x", "feet_y", "reid")}
    inference, top_k, recip, mfd = (int(v) for v in z[f"{tag}:cfg"])
    names = list(G.EDGE_FEAT_NAMES) + ["emb_dist"]
    got = G.construct_graph(det, torch.from_numpy(det["reid"]).to(dev()), 25.0, "max" if mfd < 0 else mfd, names,
                            top_k_nns=None if top_k < 0 else top_k, reciprocal_k_nns=bool(recip), inference_mode=bool(inference))
    assert np.array_equal(got["edge_index"].cpu().numpy(), z[f"{tag}:edge_index"])
    assert np.allclose(got["edge_attr"].cpu().numpy(), z[f"{tag}:edge_attr"], rtol=2e-6, atol=2e-6)
    if inference:
        assert np.allclose(got["reid_emb_dists"].cpu().numpy(), z[f"{tag}:reid_emb_dists"], rtol=2e-6)


def _sequence(frames=14, seed=5):
    det = synth.make_detections(frames=frames, seed=seed, node_in_dim=64)
    names = list(G.EDGE_FEAT_NAMES) + ["emb_dist"]
    g = T.construct_graph(det, torch.from_numpy(det["reid"]), 25.0, 'max', names)
    params = synth.model_params(32, 4, "sum", num_class_steps=2, node_in_dim=64, edge_in_dim=6)
    W = synth.make_weights(params, seed=7, gain=0.5)
    return det, g, params, W


def _native_model(params, W):
    model = MOTMPNet(params)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    return model.to(dev()).eval()


@pytest.mark.parametrize("set_inactive", [False, True])
def test_sliding_window_matches_oracle(set_inactive):
    det, g, params, W = _sequence()
    Wt = O.to_tensors(W)

    def fwd(xs, ei, ea):
        _, logits, _, _ = O.forward(params, Wt, xs, ei, ea, return_state=True)
        return logits[-1]

    x = torch.from_numpy(det["x"])
    want = T.evaluate_graph_in_batches(fwd, x, g["edge_index"], g["edge_attr"], g["reid_emb_dists"].view(-1), det["frame"],
                                       frames_per_graph=5, top_k_nns=6, reciprocal_k_nns=True,
                                       set_pruned_edges_to_inactive=set_inactive)
    model = _native_model(params, W)
    args = (model, x.to(dev()), g["edge_index"].to(dev()), g["edge_attr"].to(dev()), g["reid_emb_dists"].to(dev()), det["frame"])
    got = tracker.evaluate_graph_in_batches(*args, frames_per_graph=5, top_k_nns=6, reciprocal_k_nns=True,
                                            set_pruned_edges_to_inactive=set_inactive)
    assert got.shape == want.shape
    assert float((got.cpu() - want).abs().max()) <= 2e-5
    assert float(want.max()) > 0.05  # the comparison is not vacuous
    # (c) batching windows block-diagonally changes nothing
    got4 = tracker.evaluate_graph_in_batches(*args, frames_per_graph=5, top_k_nns=6, reciprocal_k_nns=True,
                                             set_pruned_edges_to_inactive=set_inactive, windows_per_launch=4)
    assert float((got4 - got).abs().max()) <= 1e-6


def test_sliding_window_sharded_over_ranks():
    det, g, params, W = _sequence(frames=12, seed=9)
    model = _native_model(params, W)
    args = (model, torch.from_numpy(det["x"]).to(dev()), g["edge_index"].to(dev()), g["edge_attr"].to(dev()),
            g["reid_emb_dists"].to(dev()), det["frame"])
    one = tracker.evaluate_graph_in_batches(*args, frames_per_graph=4, top_k_nns=5)
    # two "ranks": capture each rank's accumulators through reduce_fn and add them, as all_reduce(sum) would
    acc = []

    def grab(t):
        acc.append(t.clone())

    for r in range(2):
        tracker.evaluate_graph_in_batches(*args, frames_per_graph=4, top_k_nns=5, rank=r, world_size=2, reduce_fn=grab)
    preds, num = acc[0] + acc[2], acc[1] + acc[3]
    both = torch.where(num > 0, preds / num, torch.zeros_like(preds))
    assert float((both - one).abs().max()) <= 1e-6


@pytest.mark.parametrize("n,deg,k,seed", [(50, 12, 3, 1), (200, 40, 10, 2), (400, 150, 50, 3), (64, 5, 8, 4)])
def test_knn_mask_random_graphs_match_oracle(n, deg, k, seed):
    """get_knn_mask on random symmetric graphs of several densities (k below, near and above the typical degree),
    reciprocal and union forms, both edge-list conventions, against the dense-matrix restatement of graph.py:40-87."""
    rng = np.random.default_rng(seed)
    pairs = set()
    while len(pairs) < n * deg // 2:
        i, j = rng.integers(0, n, 2)
        if i != j:
            pairs.add((min(i, j), max(i, j)))
    pr = np.array(sorted(pairs), dtype=np.int64).T
    d = rng.random(pr.shape[1]).astype(np.float32) + 0.01
    ei_pairs, d_pairs = torch.from_numpy(pr), torch.from_numpy(d)
    ei_sym = torch.cat((ei_pairs, torch.stack((ei_pairs[1], ei_pairs[0]))), dim=1)
    d_sym = torch.cat((d_pairs, d_pairs))
    # shuffle the symmetric list: the halves need not be aligned for the native lookup
    p = torch.from_numpy(rng.permutation(ei_sym.shape[1]))
    ei_sym, d_sym = ei_sym[:, p], d_sym[p]
    for rec in (False, True):
        want = T.get_knn_mask(d_pairs, ei_pairs, n, k, rec, symmetric_edges=False)
        got = G.get_knn_mask(d_pairs.to(dev()), ei_pairs.to(dev()), n, k, reciprocal_k_nns=rec, symmetric_edges=False)
        assert np.array_equal(got.cpu().numpy(), want.numpy())
        want = T.get_knn_mask(d_sym, ei_sym, n, k, rec, symmetric_edges=True)
        got = G.get_knn_mask(d_sym.to(dev()), ei_sym.to(dev()), n, k, reciprocal_k_nns=rec, symmetric_edges=True)
        assert np.array_equal(got.cpu().numpy(), want.numpy())


def test_sliding_window_degenerate_sequences():
    det, g, params, W = _sequence(frames=6, seed=13)
    model = _native_model(params, W)
    args = (model, torch.from_numpy(det["x"]).to(dev()), g["edge_index"].to(dev()), g["edge_attr"].to(dev()),
            g["reid_emb_dists"].to(dev()), det["frame"])
    # more frames per window than the sequence has: no window, every edge unpredicted -> 0 (NaN -> 0, mpn_tracker.py:197)
    out = tracker.evaluate_graph_in_batches(*args, frames_per_graph=10, top_k_nns=5)
    assert out.shape[0] == g["edge_index"].shape[1] and float(out.abs().max()) == 0.0
    # one window covering everything == one forward over the kNN-pruned full graph
    out = tracker.evaluate_graph_in_batches(*args, frames_per_graph=6, top_k_nns=5)
    keep = G.get_knn_mask(g["reid_emb_dists"].view(-1).to(dev()), g["edge_index"].to(dev()), len(det["frame"]), 5,
                          reciprocal_k_nns=True, symmetric_edges=True)
    with torch.no_grad():
        logits = model.hot_path(args[1], args[2][:, keep].contiguous(), args[3][keep].contiguous())[-1]
    want = torch.zeros_like(out)
    want[keep] = torch.sigmoid(logits)
    assert float((out - want).abs().max()) <= 1e-6
    # top_k = 0 prunes every edge: nothing is predicted
    assert float(tracker.evaluate_graph_in_batches(*args, frames_per_graph=3, top_k_nns=0).abs().max()) == 0.0


def test_graph_build_more_shapes():
    """Pair enumeration on a larger unsorted frame vector, distances for embedding widths off the vector path, one-edge and
    duplicate-edge kNN inputs."""
    d0 = dev()
    rng = np.random.default_rng(7)
    f = torch.from_numpy(rng.integers(0, 40, 1500))
    for mfd in (0, 1, 7, 'max'):
        want = T.get_time_valid_conn_ixs(f, mfd)
        got = G.get_time_valid_conn_ixs(f.to(d0), mfd)
        assert np.array_equal(got.cpu().numpy(), want.numpy()), mfd
    ei = T.get_time_valid_conn_ixs(f[:200], 3)
    for dim in (1, 3, 30, 257):
        emb = torch.from_numpy(rng.standard_normal((200, dim)).astype(np.float32))
        want = T.pairwise_distance(emb, ei).view(-1)
        got = G.pairwise_distance(emb.to(d0), ei.to(d0))
        assert np.allclose(got.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-6), dim
    # one edge; top_k larger than any degree keeps everything
    one = torch.tensor([[0], [1]], dtype=torch.int64)
    assert bool(G.get_knn_mask(torch.tensor([0.5]).to(d0), one.to(d0), 2, 1, reciprocal_k_nns=True, symmetric_edges=False)[0])
    big = G.get_knn_mask(torch.rand(ei.shape[1]).to(d0), ei.to(d0), 200, 10 ** 6, reciprocal_k_nns=True, symmetric_edges=False)
    assert bool(big.all())


@pytest.mark.parametrize("windows_per_launch", [1, 3])
@pytest.mark.parametrize("tag", ["w1", "w2"])
def test_sliding_window_against_reference_tracker(golden, tag, windows_per_launch):
    """tracker.evaluate_graph_in_batches (HIP hot path, native window selection / kNN / accumulation) against the reference's own
    MPNTracker._evaluate_graph_in_batches run with the reference model (tests/golden/g10_windows.npz)."""
    z = golden("g10_windows.npz")
    inactive, recip, fpg, top_k = [int(v) for v in z[f"{tag}:cfg"]]
    params = synth.model_params(32, 4, "sum", num_class_steps=2, node_in_dim=64)
    W = synth.make_weights(params, seed=7, gain=0.6)
    model = MOTMPNet(params)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
    model = model.to(dev()).eval()
    t = {k: torch.from_numpy(z[f"{tag}:{k}"]).to(dev()) for k in ("x", "edge_index", "edge_attr", "reid_emb_dists")}
    got = tracker.evaluate_graph_in_batches(model, t["x"], t["edge_index"], t["edge_attr"], t["reid_emb_dists"], z[f"{tag}:frame"], fpg, top_k,
                                            reciprocal_k_nns=bool(recip), set_pruned_edges_to_inactive=bool(inactive),
                                            windows_per_launch=windows_per_launch)
    ref = z[f"{tag}:final_edge_preds"]
    assert float(np.abs(got.cpu().numpy() - ref).max()) < 2e-5
