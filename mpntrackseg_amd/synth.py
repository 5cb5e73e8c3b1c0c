"""Deterministic synthetic tracking graphs, weights and gradient seeds.

Everything here is generated from a counter-based splitmix64 stream using only
integer arithmetic, exact float conversions and single IEEE multiplications, so
the container that produced ``tests/golden/*.npz`` and the GPU box regenerate
bit-identical inputs (no libm transcendental is evaluated on arrays).

Graph layout follows the reference's graph object
(``/root/reference/src/mot_neural_solver/data/mot_graph.py:283-317``):
nodes are sorted by frame (index order == time order, ``mot_graph.py:145``),
``edge_index = [pairs(i<j) || flipped pairs]`` (``:312``) and ``edge_attr`` is the
same block duplicated for both directions, not sign-flipped (``:311``).

Dimension rule for "d-d feats" (SURVEY.md section 8d):
dn=d, de=d/2, he=2.5d, hn=1.75d, hc=d/4, encoder hidden 4d (node) / 18*d/32 (edge).
"""
import math

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n, stream=0):
    """n 64-bit outputs of splitmix64 for counters (seed, stream, 0..n-1)."""
    with np.errstate(over="ignore"):
        base = (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
                + np.uint64(stream) * np.uint64(0xD1342543DE82EF95))
        z = base + (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed, n, stream=0):
    """float64 uniforms in [0,1) with 53 random bits (exact conversion)."""
    return (splitmix64(seed, n, stream) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


_IH_SCALE = math.sqrt(3.0) / 65536.0  # Irwin-Hall(4) of 16-bit uniforms -> unit variance


def normal(seed, shape, stream=0, std=1.0, dtype=np.float32):
    """Approximately N(0, std^2) values: Irwin-Hall sum of the four 16-bit fields of one
    splitmix64 output. Integer sum, one float64 multiply, one cast: reproducible anywhere."""
    n = int(np.prod(shape)) if len(shape) else 1
    z = splitmix64(seed, n, stream)
    s = ((z & np.uint64(0xFFFF)) + ((z >> np.uint64(16)) & np.uint64(0xFFFF))
         + ((z >> np.uint64(32)) & np.uint64(0xFFFF)) + (z >> np.uint64(48))).astype(np.int64)
    v = (s - 2 * 65535).astype(np.float64) * (_IH_SCALE * float(std))
    return v.astype(dtype).reshape(shape)


def dims_for(d, node_in_dim=2048, edge_in_dim=6):
    """The reference ``graph_model_params`` dict (``configs/tracking_cfg.yaml:134-168``)
    scaled by d/32. d=32 reproduces the shipped config's hot-path dims."""
    assert d % 4 == 0
    return {
        "node_agg_fn": "sum",
        "num_enc_steps": 12,
        "num_class_steps": 12,
        "reattach_initial_nodes": True,
        "reattach_initial_edges": True,
        "encoder_feats_dict": {
            "edge_in_dim": edge_in_dim, "edge_dims": [18 * d // 32, 18 * d // 32], "edge_out_dim": d // 2,
            "node_in_dim": node_in_dim, "node_dims": [4 * d], "node_out_dim": d,
            "dropout_p": 0, "use_batchnorm": False,
        },
        "edge_model_feats_dict": {"dims": [5 * d // 2, d // 2], "dropout_p": 0, "use_batchnorm": False},
        "node_model_feats_dict": {"dims": [7 * d // 4, d], "dropout_p": 0, "use_batchnorm": False},
        "classifier_feats_dict": {
            "edge_in_dim": d // 2, "edge_dims": [d // 4], "edge_out_dim": 1,
            "dropout_p": 0, "use_batchnorm": False,
        },
    }


def model_params(d, L, agg="sum", num_class_steps=None, node_in_dim=2048, edge_in_dim=6):
    p = dims_for(d, node_in_dim, edge_in_dim)
    p["node_agg_fn"] = agg
    p["num_enc_steps"] = L
    p["num_class_steps"] = L if num_class_steps is None else num_class_steps
    return p


def hot_path_param_shapes(p):
    """Ordered {state_dict key: shape} of the hot-path parameters, names exactly as the
    reference's ``MOTMPNet.state_dict()``."""
    enc = p["encoder_feats_dict"]
    dn, de = enc["node_out_dim"], enc["edge_out_dim"]
    nf = 2 if p["reattach_initial_nodes"] else 1
    ef = 2 if p["reattach_initial_edges"] else 1
    shapes = {}

    def mlp(prefix, in_dim, dims):
        # mlp.py:12-23 with no BN / dropout: Sequential indices 0,2,4,.. (Linear, ReLU) and a
        # trailing Linear without ReLU when dim == 1
        idx = 0
        for dim in dims:
            shapes[f"{prefix}.fc_layers.{idx}.weight"] = (dim, in_dim)
            shapes[f"{prefix}.fc_layers.{idx}.bias"] = (dim,)
            idx += 2 if dim != 1 else 1
            in_dim = dim

    mlp("encoder.node_model", enc["node_in_dim"], list(enc["node_dims"]) + [dn])
    mlp("encoder.edge_model", enc["edge_in_dim"], list(enc["edge_dims"]) + [de])
    mlp("MPNet.edge_model.edge_model", nf * 2 * dn + ef * de, p["edge_model_feats_dict"]["dims"])
    mlp("MPNet.node_model.flow_in_model", nf * dn + de, p["node_model_feats_dict"]["dims"])
    mlp("MPNet.node_model.flow_out_model", nf * dn + de, p["node_model_feats_dict"]["dims"])
    shapes["MPNet.node_model.node_model.0.weight"] = (dn, 2 * dn)
    shapes["MPNet.node_model.node_model.0.bias"] = (dn,)
    cls = p["classifier_feats_dict"]
    mlp("classifier.edge_model", cls["edge_in_dim"], list(cls["edge_dims"]) + [cls["edge_out_dim"]])
    return shapes


# mask-branch dicts of the reference config (configs/tracking_cfg.yaml:168-218)
MASK_PARAMS = {
    "node_ext_encoder_feats_dict": dict(input_dim=256, dims=[128, 32], kernel_sizes=[1, 1], strides=[1, 1],
                                        paddings=[0, 0], dropout_p=0, use_batchnorm=False),
    "attention_model_feats_dict": dict(fc_dims=[16, 1], dropout_p=0, use_batchnorm=False),
    "node_ext_model_feats_dict": dict(dims=[96, 32], kernel_sizes=[3, 3], strides=[1, 1], paddings=[1, 1],
                                      dropout_p=0, use_batchnorm=False),
    "mask_model_feats_dict": {
        "feature_encoder_feats_dict": dict(input_dim=256, dims=[32], kernel_sizes=[1], strides=[1], paddings=[0],
                                           dropout_p=0, use_batchnorm=False),
        "mask_head_feats_dict": dict(input_dim=64, dims=[64, 64, 64], kernel_sizes=[3, 3, 3], strides=[1, 1, 1],
                                     paddings=[1, 1, 1], dropout_p=0, use_batchnorm=False),
        "mask_predictor_feats_dict": dict(input_dim=64, dims=[64, 64, 64, 1], kernel_sizes=[2, 3, 2, 1],
                                          strides=[2, 1, 2, 1], paddings=[0, 1, 0, 0],
                                          transposed=[True, False, True, False]),
    },
}


def mask_param_shapes():
    """{state_dict key: shape} of the mask branch for MASK_PARAMS with reattach_initial_nodes = True
    (names and shapes as the reference's MOTMPNet.state_dict())."""
    sh = {}

    def conv(prefix, idx, cout, cin, k, transposed=False):
        sh[f"{prefix}.layers.{idx}.weight"] = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        sh[f"{prefix}.layers.{idx}.bias"] = (cout,)

    conv("node_ext_encoder", 0, 128, 256, 1)
    conv("node_ext_encoder", 2, 32, 128, 1)
    conv("mask_predictor.feature_encoder", 0, 32, 256, 1)
    sh["mask_predictor.layer_norm.weight"] = (64, 14, 14)
    sh["mask_predictor.layer_norm.bias"] = (64, 14, 14)
    for i in (0, 2, 4):
        conv("mask_predictor.mask_head", i, 64, 64, 3)
    conv("mask_predictor.mask_predictor", 0, 64, 64, 2, transposed=True)
    conv("mask_predictor.mask_predictor", 2, 64, 64, 3)
    conv("mask_predictor.mask_predictor", 4, 64, 64, 2, transposed=True)
    conv("mask_predictor.mask_predictor", 6, 1, 64, 1)
    conv("MPAttentionNet.node_model", 0, 96, 192, 3)
    conv("MPAttentionNet.node_model", 2, 32, 96, 3)
    return sh


def make_mask_weights(seed=17, bias_std=0.05):
    """He-scale conv weights (std = sqrt(2 / (c_in k k))), LayerNorm weight 1 + N(0, 0.1^2), small biases."""
    out = {}
    for i, (k, shp) in enumerate(mask_param_shapes().items()):
        if "layer_norm.weight" in k:
            out[k] = (1.0 + normal(seed, shp, stream=i, std=0.1, dtype=np.float64)).astype(np.float32)
        elif k.endswith(".weight"):
            transposed = k in ("mask_predictor.mask_predictor.layers.0.weight", "mask_predictor.mask_predictor.layers.4.weight")
            cin = shp[0] if transposed else shp[1]
            out[k] = normal(seed, shp, stream=i, std=math.sqrt(2.0 / (cin * shp[2] * shp[3])))
        else:
            out[k] = normal(seed, shp, stream=i, std=bias_std)
    return out


def make_weights(p, seed=7, bias_std=0.1, gain=1.0):
    """He-normal weights (std = gain*sqrt(2/fan_in)), biases N(0, bias_std^2); one RNG stream per
    tensor in ``hot_path_param_shapes`` order. Returns {key: float32 ndarray}."""
    out = {}
    for i, (k, shp) in enumerate(hot_path_param_shapes(p).items()):
        if k.endswith(".weight"):
            out[k] = normal(seed, shp, stream=i, std=gain * math.sqrt(2.0 / shp[1]))
        else:
            out[k] = normal(seed, shp, stream=i, std=bias_std)
    return out


def make_graph(N, E, T=30, seed=1, node_in_dim=2048, edge_in_dim=6, pooled=True):
    """Synthetic tracking graph: ``frame = sort(randint(0,T))``; E/2 distinct
    unordered cross-frame pairs stored (min,max) then mirrored; edge_attr duplicated.

    Returns dict with x [N,node_in_dim] (or [N,node_in_dim,1,1] when pooled=False), edge_index
    int64 [2,E], edge_attr [E,edge_in_dim], frame int64 [N]."""
    assert E % 2 == 0
    half = E // 2
    frame = np.sort((uniform01(seed, N, stream=0) * T).astype(np.int64))
    pairs = np.empty((0,), dtype=np.int64)
    rnd = 0
    while pairs.size < half:
        rnd += 1
        m = int((half - pairs.size) * 1.3) + 64
        a = (uniform01(seed, m, stream=100 + 2 * rnd) * N).astype(np.int64)
        b = (uniform01(seed, m, stream=101 + 2 * rnd) * N).astype(np.int64)
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        ok = frame[lo] != frame[hi]
        key = lo[ok] * N + hi[ok]
        allk = np.concatenate([pairs, key])
        # keep first occurrences in draw order (deterministic)
        _, first = np.unique(allk, return_index=True)
        pairs = allk[np.sort(first)]
        if rnd > 64:
            raise ValueError("cannot draw enough distinct cross-frame pairs")
    pairs = pairs[:half]
    lo, hi = pairs // N, pairs % N
    edge_index = np.stack([np.concatenate([lo, hi]), np.concatenate([hi, lo])]).astype(np.int64)
    ea = normal(seed, (half, edge_in_dim), stream=2)
    edge_attr = np.concatenate([ea, ea], axis=0)
    x = normal(seed, (N, node_in_dim), stream=3)
    if not pooled:
        x = x.reshape(N, node_in_dim, 1, 1)
    return {"x": x, "edge_index": edge_index, "edge_attr": edge_attr, "frame": frame}


def make_knn_graph(frames=20, dets=25, top_k=150, seed=1, node_in_dim=2048, edge_in_dim=6):
    """MOTS20-02-like dense graph (SURVEY.md section 8d cfg-C): ``frames x dets`` nodes, all cross-frame
    pairs scored by a synthetic distance, kept when inside the top-k of either endpoint
    (the reference keeps the union, ``utils/graph.py:40-87``)."""
    N = frames * dets
    frame = np.repeat(np.arange(frames, dtype=np.int64), dets)
    dist = uniform01(seed, N * N, stream=5).reshape(N, N)
    dist = np.minimum(dist, dist.T)
    valid = frame[:, None] != frame[None, :]
    dist = np.where(valid, dist, np.inf)
    k = min(top_k, N - dets)
    nn = np.argsort(dist, axis=1, kind="stable")[:, :k]
    mask = np.zeros((N, N), dtype=bool)
    mask[np.arange(N)[:, None], nn] = True
    mask = (mask | mask.T) & valid
    lo, hi = np.nonzero(np.triu(mask, 1))
    half = lo.size
    edge_index = np.stack([np.concatenate([lo, hi]), np.concatenate([hi, lo])]).astype(np.int64)
    ea = normal(seed, (half, edge_in_dim), stream=2)
    return {"x": normal(seed, (N, node_in_dim), stream=3), "edge_index": edge_index,
            "edge_attr": np.concatenate([ea, ea], axis=0), "frame": frame}


def make_detections(frames=12, dets_lo=4, dets_hi=9, seed=3, emb_dim=64, node_in_dim=64, frame_stride=1):
    """Synthetic detection table of one sequence, ordered by frame (data/mot_graph.py:145), with the columns the
    reference's graph utilities read (utils/graph.py:104-113) plus ReID embeddings and pooled node inputs."""
    u = uniform01(seed, frames, stream=1)
    counts = (dets_lo + np.floor(u * (dets_hi - dets_lo + 1))).astype(np.int64)
    frame = np.repeat(1 + frame_stride * np.arange(frames, dtype=np.int64), counts)
    n = int(frame.shape[0])
    bb_h = (80.0 + 120.0 * uniform01(seed, n, stream=2)).astype(np.float32)
    bb_w = (30.0 + 60.0 * uniform01(seed, n, stream=3)).astype(np.float32)
    feet_x = (1900.0 * uniform01(seed, n, stream=4)).astype(np.float32)
    feet_y = (200.0 + 800.0 * uniform01(seed, n, stream=5)).astype(np.float32)
    return dict(frame=frame, bb_height=bb_h, bb_width=bb_w, feet_x=feet_x, feet_y=feet_y,
                reid=normal(seed, (n, emb_dim), stream=6), x=normal(seed, (n, node_in_dim), stream=7))


def batch_graphs(graphs):
    """torch_geometric-style collation: node offsets added to edge_index, tensors concatenated.
    The (i<j) / (j<i) halves of the sub-graphs end up interleaved, so direction masks must be
    computed from the indices, never assumed half/half."""
    off, xs, eis, eas, frs = 0, [], [], [], []
    for g in graphs:
        xs.append(g["x"]); eas.append(g["edge_attr"]); eis.append(g["edge_index"] + off); frs.append(g["frame"])
        off += g["x"].shape[0]
    # `batch` / `edge_graph`: torch_geometric's per-node graph id, and the graph id of every edge (int32)
    return {"x": np.concatenate(xs), "edge_index": np.concatenate(eis, axis=1),
            "edge_attr": np.concatenate(eas), "frame": np.concatenate(frs),
            "batch": np.concatenate([np.full(g["x"].shape[0], i, np.int64) for i, g in enumerate(graphs)]),
            "edge_graph": np.concatenate([np.full(g["edge_index"].shape[1], i, np.int32) for i, g in enumerate(graphs)])}


def checksum(a):
    """Order-sensitive 64-bit checksum of an array's bytes (pins regenerated inputs to a fixture)."""
    b = np.ascontiguousarray(a).view(np.uint8).ravel()
    pad = (-b.size) % 8
    if pad:
        b = np.concatenate([b, np.zeros(pad, np.uint8)])
    w = b.view(np.uint64)
    with np.errstate(over="ignore"):
        idx = np.arange(1, w.size + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        return int(np.bitwise_xor.reduce((w + idx) * np.uint64(0xBF58476D1CE4E5B9)))


CONFIGS = {
    # name: N, E, d, L  (BASELINE.json configs / SURVEY.md section 8d)
    "A": dict(N=500, E=4000, d=32, L=6),
    "B": dict(N=5000, E=50000, d=128, L=12),
    "E": dict(N=20000, E=400000, d=256, L=12),
    # MOTS20-02-like stand-in (SURVEY.md section 8d cfg-C): 20 frames x 25 detections, reciprocal top-150 kNN, reference dims
    "C": dict(N=500, E=None, d=32, L=12, knn=dict(frames=20, dets=25, top_k=150)),
    # KITTIMOTS-like sequence graph (SURVEY.md section 8d cfg-D, configs/kitti.yaml:66,71,136): 20 frames x ~7 detections,
    # top-100 kNN, reference dims, L = 4; one graph per GPU, a different seed per rank
    "D": dict(N=140, E=None, d=32, L=4, knn=dict(frames=20, dets=7, top_k=100)),
}
