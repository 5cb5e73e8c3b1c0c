"""Host-side mirror of the reference interface: constructor arguments, attribute names and
state_dict keys -- checked on CPU; forward refuses to run without a HIP device."""
import numpy as np
import pytest
import torch

from mpntrackseg_amd import capi, synth
from mpntrackseg_amd.mpn import MOTMPNet, MetaLayer, EdgeModel, TimeAwareNodeModel, MLPGraphIndependent
from mpntrackseg_amd.mlp import MLP

REFERENCE_DEFAULT_KEYS = [
    # [probe] list of SURVEY.md section 8b for configs/tracking_cfg.yaml:134-168
    "encoder.node_model.fc_layers.0.weight", "encoder.node_model.fc_layers.0.bias",
    "encoder.node_model.fc_layers.2.weight", "encoder.node_model.fc_layers.2.bias",
    "encoder.edge_model.fc_layers.0.weight", "encoder.edge_model.fc_layers.0.bias",
    "encoder.edge_model.fc_layers.2.weight", "encoder.edge_model.fc_layers.2.bias",
    "encoder.edge_model.fc_layers.4.weight", "encoder.edge_model.fc_layers.4.bias",
    "MPNet.edge_model.edge_model.fc_layers.0.weight", "MPNet.edge_model.edge_model.fc_layers.0.bias",
    "MPNet.edge_model.edge_model.fc_layers.2.weight", "MPNet.edge_model.edge_model.fc_layers.2.bias",
    "MPNet.node_model.flow_in_model.fc_layers.0.weight", "MPNet.node_model.flow_in_model.fc_layers.0.bias",
    "MPNet.node_model.flow_in_model.fc_layers.2.weight", "MPNet.node_model.flow_in_model.fc_layers.2.bias",
    "MPNet.node_model.flow_out_model.fc_layers.0.weight", "MPNet.node_model.flow_out_model.fc_layers.0.bias",
    "MPNet.node_model.flow_out_model.fc_layers.2.weight", "MPNet.node_model.flow_out_model.fc_layers.2.bias",
    "MPNet.node_model.node_model.0.weight", "MPNet.node_model.node_model.0.bias",
    "classifier.edge_model.fc_layers.0.weight", "classifier.edge_model.fc_layers.0.bias",
    "classifier.edge_model.fc_layers.2.weight", "classifier.edge_model.fc_layers.2.bias",
]


def default_params():
    p = synth.model_params(32, 4, "sum", num_class_steps=3)
    return p


def test_state_dict_keys_and_shapes_match_reference():
    model = MOTMPNet(default_params())
    sd = model.state_dict()
    assert sorted(sd.keys()) == sorted(REFERENCE_DEFAULT_KEYS)
    shapes = synth.hot_path_param_shapes(default_params())
    assert {k: tuple(v.shape) for k, v in sd.items()} == shapes
    assert sd["MPNet.edge_model.edge_model.fc_layers.0.weight"].shape == (80, 160)
    assert sd["MPNet.node_model.flow_in_model.fc_layers.0.weight"].shape == (56, 80)
    assert sd["encoder.node_model.fc_layers.0.weight"].shape == (128, 2048)
    assert sum(v.numel() for v in sd.values()) == 296293  # hot-path parameter count


def test_operand_precision_policy():
    """MOTMPNet.gemm_precision: 'auto' (default) = fp32 results from three-piece bf16 operands where the chain kernels are MFMA-bound
    (128-d class), fp32 MFMAs at the reference's widths; explicit choices pass through; anything else raises."""
    from mpntrackseg_amd import capi
    ref = MOTMPNet(default_params())
    assert ref.gemm_precision == "auto" and ref.operand_precision() == "fp32_wgsplit"
    assert ref.operand_precision(14400) == "fp32_wgsplit" and ref.operand_precision(77800) == "fp32_split"   # (cfg-D / cfg-C stand-ins)
    wide = MOTMPNet(synth.model_params(128, 2, "sum", node_in_dim=64))
    assert wide.operand_precision() == "fp32_split"
    for prec in ("fp32", "fp32_split", "bf16"):
        wide.gemm_precision = prec
        assert wide.operand_precision() == prec
    wide.gemm_precision = "fp16"
    with pytest.raises(capi.MpnhipError):
        wide.operand_precision()


def test_attributes_like_reference():
    m = MOTMPNet(default_params())
    assert m.num_enc_steps == 4 and m.num_class_steps == 3
    assert m.edge_factor == 2 and m.node_factor == 2
    assert isinstance(m.MPNet, MetaLayer) and isinstance(m.MPNet.edge_model, EdgeModel)
    assert isinstance(m.MPNet.node_model, TimeAwareNodeModel)
    assert isinstance(m.encoder, MLPGraphIndependent) and isinstance(m.classifier.edge_model, MLP)
    assert m.classifier.node_model is None
    assert m.MPNet.node_model.node_agg_fn.name == "sum"
    p = default_params()
    p["reattach_initial_nodes"] = False
    p["reattach_initial_edges"] = False
    m2 = MOTMPNet(p)
    assert m2.MPNet.edge_model.edge_model.fc_layers[0].weight.shape == (80, 2 * 32 + 16)


def test_bad_agg_asserts_like_reference():
    p = default_params()
    p["node_agg_fn"] = "median"
    with pytest.raises(AssertionError):
        MOTMPNet(p)
    with pytest.raises(AssertionError):
        MLP(4, 8)


def test_mlp_layer_rule():
    # ReLU after every layer whose out dim != 1 (mlp.py:17); BN / dropout as requested
    m = MLP(6, [18, 1, 4], dropout_p=0.5, use_batchnorm=True)
    kinds = [type(l).__name__ for l in m.fc_layers]
    assert kinds == ["Linear", "BatchNorm1d", "ReLU", "Dropout", "Linear", "Linear", "BatchNorm1d", "ReLU", "Dropout"]
    assert not m.fast_path


def test_forward_refuses_cpu_tensors():
    model = MOTMPNet(default_params())

    class D:
        pass
    d = D()
    d.x = torch.zeros(4, 2048)
    d.edge_index = torch.zeros(2, 0, dtype=torch.int64)
    d.edge_attr = torch.zeros(0, 6)
    with pytest.raises(capi.MpnhipError):
        model(d)
    with pytest.raises(capi.MpnhipError):
        model.MPNet(torch.zeros(4, 64), d.edge_index, torch.zeros(0, 32))
    with pytest.raises(capi.MpnhipError):
        model.encoder.node_model(torch.zeros(4, 2048))


def test_synth_is_deterministic_and_structured():
    g = synth.make_graph(100, 600, seed=3, node_in_dim=8)
    g2 = synth.make_graph(100, 600, seed=3, node_in_dim=8)
    assert all(np.array_equal(g[k], g2[k]) for k in g)
    ei = g["edge_index"]
    assert ei.shape == (2, 600)
    assert (ei[0, :300] < ei[1, :300]).all() and np.array_equal(ei[0, :300], ei[1, 300:])
    assert np.array_equal(g["edge_attr"][:300], g["edge_attr"][300:])
    assert (g["frame"][ei[0]] != g["frame"][ei[1]]).all()
    assert len({(a, b) for a, b in ei[:, :300].T.tolist()}) == 300
    v = synth.normal(5, (200000,))
    assert abs(float(v.mean())) < 0.01 and abs(float(v.std()) - 1.0) < 0.01


def test_frame_windows_match_reference_loop():
    """tracker.frame_windows: node ranges of the sliding windows of mpn_tracker.py:166-170 (no GPU needed)."""
    from mpntrackseg_amd import tracker
    frames = np.array([1, 1, 1, 3, 3, 4, 6, 6, 6, 6, 9, 10, 10])
    wins = tracker.frame_windows(frames, 3)
    all_frames = np.unique(frames)
    want = []
    for s, e in zip(all_frames, all_frames[2:]):
        m = np.nonzero((s <= frames) & (frames <= e))[0]
        want.append((int(m[0]), int(m[-1]) + 1))
    assert wins == want and len(wins) == len(all_frames) - 2
    # round-robin sharding of the windows covers each exactly once
    assert sorted(wins[0::2] + wins[1::2]) == sorted(wins)
    with pytest.raises(Exception):
        tracker.frame_windows(np.array([2, 1, 3]), 2)


def test_graph_file_round_trip(tmp_path):
    """mpntrackseg_amd.graphfile: save / load of a precomputed detection graph, pooling of 4-D node inputs, validation."""
    from mpntrackseg_amd import graphfile
    g = synth.make_graph(40, 200, seed=3, node_in_dim=16)
    x4 = np.repeat(np.repeat(g["x"][:, :, None, None], 2, axis=2), 3, axis=3) + synth.normal(4, (40, 16, 2, 3)) * 0
    labels = (np.arange(200) % 5 == 0).astype(np.float32)
    path = str(tmp_path / "seq.npz")
    graphfile.save_graph(path, x4, g["edge_index"], g["edge_attr"], edge_labels=labels, frame=np.arange(40) // 4)
    z = graphfile.load_graph(path)
    assert z["x"].shape == (40, 16) and np.allclose(z["x"].numpy(), g["x"], atol=1e-6)      # pooled (mpn.py:351-352)
    assert np.array_equal(z["edge_index"].numpy(), g["edge_index"]) and z["edge_index"].dtype == torch.int64
    assert np.array_equal(z["edge_labels"].numpy(), labels) and "reid_emb_dists" not in z
    bad = g["edge_index"].copy()
    bad[0, 0] = 40
    with pytest.raises(ValueError):
        graphfile.save_graph(path, g["x"], bad, g["edge_attr"])
    np.savez(str(tmp_path / "junk.npz"), a=np.zeros(3))
    with pytest.raises(ValueError):
        graphfile.load_graph(str(tmp_path / "junk.npz"))


def test_frame_embedding_files_have_the_reference_layout(tmp_path):
    """write_frame_embeddings: one <frame>.pt per frame, detection id in column 0 ('1D') / broadcast into channel 0 ('3D'),
    as seq_processor.py stores them (what load_precomputed_embeddings, utils/rgb.py:150-188, expects).  Host-only."""
    import torch
    from mpntrackseg_amd import embeddings as E
    frames = np.array([4, 4, 9])
    ids = np.array([10, 11, 12])
    E.write_frame_embeddings(str(tmp_path), "reid", frames, ids, np.arange(12, dtype=np.float32).reshape(3, 4))
    E.write_frame_embeddings(str(tmp_path), "node", frames, ids, np.ones((3, 2, 2, 2), np.float32))
    a = torch.load(str(tmp_path / "processed_data" / "reid" / "4.pt"))
    assert tuple(a.shape) == (2, 5) and a[:, 0].tolist() == [10.0, 11.0] and a[1, 1:].tolist() == [4.0, 5.0, 6.0, 7.0]
    b = torch.load(str(tmp_path / "processed_data" / "node" / "9.pt"))
    assert tuple(b.shape) == (1, 3, 2, 2) and bool((b[:, 0] == 12).all()) and bool((b[:, 1:] == 1).all())


def test_bench_gpus_flag_refuses_more_rccl_ranks_than_devices():
    """--backend nccl with more ranks than visible devices exits non-zero instead of silently sharing a device."""
    import os, subprocess, sys
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = max(2, torch.cuda.device_count() + 1)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--config", "D", "--steps", "2", "--warmup", "1"],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "one GPU per rank" in (r.stderr + r.stdout), r.stdout[-1000:] + r.stderr[-1000:]
