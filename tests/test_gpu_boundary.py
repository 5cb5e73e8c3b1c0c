"""The drop-in boundary with the caller's REAL shapes.

(a) the disk path of BASELINE.json configs[2] ("precomputed detection graph ... loaded from disk"): a graph written with
    ``graphfile.save_graph``, loaded back and run through the HIP hot path against the oracle; ``bench.py --graph-file``;
(b) the container types ``MOTMPNet.forward`` receives in the reference: a ``Graph(torch_geometric.data.Data)`` sample
    (data/mot_graph.py:21-83) moved with its own ``.to(device)`` / ``.cuda()``, and a 3-graph ``Batch`` out of a torch_geometric
    ``DataLoader`` (pl_module/pl_module.py:54) -- tests/pyg_standin.py restates that attribute protocol (torch_geometric is not
    installed here).  Per-graph results of the batched forward must equal the graphs run alone."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from mpntrackseg_amd import graphfile, synth
from mpntrackseg_amd.mpn import MOTMPNet
from oracle import mpn_oracle as O
from pyg_standin import Batch, Graph

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


def dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda:0")


def make_model(params, W, precision="fp32"):
    model = MOTMPNet(params)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
    model = model.to(dev()).eval()
    model.gemm_precision = precision
    return model


def oracle_logits(params, W, x, ei, ea):
    with torch.no_grad():
        _, logits, _, _ = O.forward(params, O.to_tensors(W), torch.from_numpy(x), torch.from_numpy(ei), torch.from_numpy(ea), return_state=True)
    return np.stack([l.view(-1).numpy() for l in logits])


def err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1.0, float(np.abs(b).max()))) if a.size else 0.0


# ------------------------------------------------------------------------------------------------ (a) the disk path
@pytest.mark.parametrize("precision", ["fp32", "fp32_split"])
@pytest.mark.parametrize("agg", ["sum", "mean", "max"])
def test_graph_file_runs_the_hip_hot_path_like_the_oracle(tmp_path, agg, precision):
    """save_graph (4-D ReID maps pooled on the DEVICE at write time) -> load_graph(device) -> MOTMPNet.hot_path == oracle on the
    arrays that went in."""
    g = synth.make_knn_graph(seed=5, frames=10, dets=12, top_k=20, node_in_dim=64)
    N = g["x"].shape[0]
    params = synth.model_params(32, 4, agg, node_in_dim=g["x"].shape[1])
    W = synth.make_weights(params, seed=7)
    x4 = torch.from_numpy(g["x"]).to(dev())[:, :, None, None].expand(N, g["x"].shape[1], 2, 2).contiguous()
    path = str(tmp_path / "MOTS20-02-standin.npz")
    labels = (np.arange(g["edge_index"].shape[1]) % 7 == 0).astype(np.float32)
    graphfile.save_graph(path, x4, g["edge_index"], g["edge_attr"], edge_labels=labels)
    z = graphfile.load_graph(path, device=dev())
    assert z["x"].is_cuda and z["x"].shape == (N, g["x"].shape[1]) and z["edge_index"].dtype == torch.int64
    model = make_model(params, W, precision)
    with torch.no_grad():
        logits = model.hot_path(z["x"], z["edge_index"], z["edge_attr"])
    ref = oracle_logits(params, W, g["x"], g["edge_index"], g["edge_attr"])
    for s in range(ref.shape[0]):
        assert err(logits[s].cpu().numpy(), ref[s]) < TOL, (s, agg, precision)


def test_bench_runs_from_a_graph_file(tmp_path):
    """BASELINE.json configs[2]'s shape of run: `bench.py --graph-file <npz> --config C` (fwd+bwd on a graph loaded from disk)."""
    g = synth.make_knn_graph(seed=2, frames=12, dets=15, top_k=30, node_in_dim=256)
    path = str(tmp_path / "seq.npz")
    graphfile.save_graph(path, g["x"], g["edge_index"], g["edge_attr"])
    r = subprocess.run([sys.executable, "bench.py", "--graph-file", path, "--config", "C", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-split-line", "--no-extras"], cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["edges"] == g["edge_index"].shape[1] and d["config"]["nodes"] == g["x"].shape[0], d["config"]
    assert d["config"]["mode"] == "train" and d["value"] > 0 and "graph file" in d["data"], d


# ------------------------------------------------------------------------------------------------ (b) container types
def _sample(seed, n_frames, dets, top_k, with_maps=False):
    g = synth.make_knn_graph(seed=seed, frames=n_frames, dets=dets, top_k=top_k, node_in_dim=64)
    E = g["edge_index"].shape[1]
    x = torch.from_numpy(g["x"])
    if with_maps:
        x = x[:, :, None, None].expand(x.shape[0], x.shape[1], 2, 2).contiguous()
    return Graph(x=x, edge_index=torch.from_numpy(g["edge_index"]), edge_attr=torch.from_numpy(g["edge_attr"]),
                 edge_labels=torch.from_numpy((np.arange(E) % 5 == 0).astype(np.float32)),
                 reid_emb_dists=torch.from_numpy(synth.normal(seed + 100, (E,)).astype(np.float32)),
                 node_names=torch.arange(x.shape[0]))


@pytest.mark.parametrize("agg", ["sum", "max"])
def test_graph_sample_through_forward(agg):
    """A ``Graph`` sample with the reference's extra attributes, moved by its own ``.to(device)`` (returns None, mot_graph.py:71) --
    forward reads x / edge_index / edge_attr only, leaves the sample's own key list alone and caches its graph prep under a dunder name."""
    s = _sample(11, 8, 9, 12, with_maps=True)
    params = synth.model_params(32, 3, agg, num_class_steps=2, node_in_dim=s.x.shape[1])
    W = synth.make_weights(params, seed=7)
    model = make_model(params, W)
    keys_before = sorted(s.keys)
    ref = oracle_logits(params, W, s.x.mean(dim=(2, 3)).numpy(), s.edge_index.numpy(), s.edge_attr.numpy())
    assert s.to(dev()) is None and s.x.is_cuda and s.edge_index.is_cuda and s.edge_index.dtype == torch.int64
    snap = {k: s[k].clone() for k in ("x", "edge_index", "edge_attr")}
    with torch.no_grad():
        out = model(s)
        out2 = model(s)                                   # the cached prep is found again
    assert sorted(s.keys) == keys_before                  # the cache attribute is invisible to Data.keys
    assert all(torch.equal(s[k], v) for k, v in snap.items())          # inputs are borrowed, never mutated
    cls = out["classified_edges"]
    assert len(cls) == 2 and cls[0].shape == (s.edge_index.shape[1], 1) and out["mask_predictions"] == []
    for i in range(2):
        assert err(cls[i].view(-1).cpu().numpy(), ref[1 + i]) < TOL
        assert torch.equal(cls[i], out2["classified_edges"][i])
    s.cpu()
    assert not s.x.is_cuda
    s.cuda()                                              # moved again: new tensors, the stale cache must not be used
    with torch.no_grad():
        out3 = model(s)
    assert torch.equal(out3["classified_edges"][-1], cls[-1])


@pytest.mark.parametrize("precision", ["fp32", "fp32_split"])
@pytest.mark.parametrize("agg", ["sum", "mean", "max"])
def test_three_graph_batch_equals_the_graphs_alone(agg, precision):
    """``Batch.from_data_list`` of three samples of different sizes (edge_index shifted by the running node count and concatenated
    along dim -1 -- so the two direction halves of the sub-graphs INTERLEAVE --, a ``batch`` vector, per-graph extras) through
    ``MOTMPNet.forward``: every sub-graph's logits equal the forward of that sample alone."""
    samples = [_sample(21, 6, 8, 10), _sample(22, 9, 5, 8), _sample(23, 5, 11, 14)]
    params = synth.model_params(32, 4, agg, num_class_steps=4, node_in_dim=samples[0].x.shape[1])
    W = synth.make_weights(params, seed=7)
    model = make_model(params, W, precision)
    b = Batch.from_data_list(samples)
    assert b.num_graphs == 3 and b.batch.shape[0] == sum(s.x.shape[0] for s in samples)
    b = b.to(dev())
    assert b.edge_index.dtype == torch.int64 and int(b.edge_index.max()) == b.x.shape[0] - 1
    with torch.no_grad():
        out = model(b)["classified_edges"]
    e0 = 0
    for s in samples:
        ref = oracle_logits(params, W, s.x.numpy(), s.edge_index.numpy(), s.edge_attr.numpy())
        s.to(dev())
        with torch.no_grad():
            alone = model(s)["classified_edges"]
        E = s.edge_index.shape[1]
        for i in range(4):
            part = out[i][e0:e0 + E].view(-1).cpu().numpy()
            assert err(part, ref[i]) < TOL, (agg, precision, i)
            assert err(part, alone[i].view(-1).cpu().numpy()) < (2e-5 if agg != "sum" else TOL)
        e0 += E
    assert e0 == b.edge_index.shape[1]
