#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation (authoring container only).

The reference's Python never travels to the GPU box; this script imports
``/root/reference/src/mot_neural_solver/models/mpn.py`` here, drives it on inputs from the
repo's own deterministic generator (``mpntrackseg_amd/synth.py``) and stores the outputs as
small fixtures.  The only thing injected is a stand-in for the un-vendored third-party
``torch_scatter`` 2.0.4 package (``environment.yml:146``), restating its documented semantics with
stock torch ops (scatter_add_ / clamp / scatter_reduce amax, empty segments -> 0).

Fixtures:
  g1_tiny_{sum,mean,max}.npz   full ``MOTMPNet.forward`` (mask branch included) on N=60/E=800, default
                               dims with node_in_dim=64; inputs, weights, logits of every step, final
                               x/e, and autograd gradients of loss = sum_steps sum_edges logit*r.
  g4_structure.npz             isolated nodes, only-past / only-future nodes, 3 batched sub-graphs
                               (interleaved halves), self loops, ties at 0 under max.
  g5_modules.npz               MetaLayer.forward single step and node_agg_fn alone.
  g2_cfgA_{agg}.npz            cfg-A (500/4000/d32/L6): logits [6,4000]; inputs regenerated (checksums).
  g3_cfgB_{agg}.npz            cfg-B (5000/50000/d128/L12): logits at 4096 fixed edges x 12 steps +
                               per-step sum / abs-sum / max checksums.
  g0_l0.npz                    num_enc_steps == 0 special case (mpn.py:387-389).
  g11_cfgB_sum_o1.npz          cfg-B, sum aggregation, 12 steps, weights scaled to O(1) logits: sampled logits, checksums and the
                               reference's autograd gradients (the headline training workload).
  g12_dense_knn_{agg}.npz      dense reciprocal-kNN graph (E / N = 64, d = 32, 12 steps): sampled logits + reference autograd.
  g9_loss_metrics.npz          MOTNeuralSolver._compute_loss (+ autograd) and compute_perform_metrics / compute_constr_satisfaction_rate.
  g10_windows.npz              MPNTracker._evaluate_graph_in_batches on a synthetic sequence with the reference model.
  g7_graph_utils.npz           the reference's utils/graph.py on a synthetic detection table (synth.make_detections):
                               get_time_valid_conn_ixs ('max' and 3 frames), compute_edge_feats_dict, F.pairwise_distance,
                               get_knn_mask (reciprocal on/off; one direction per pair and both directions).
  g15_batchnorm_train.npz      MLPs with BatchNorm1d in TRAINING mode: the reference's forward, autograd (incl. BatchNorm weights) and
                               running statistics in float64 (three aggregations).
  g6_mask_branch.npz           full forward WITH the attention / mask branch (deterministic weights for all 54
                               tensors): mask predictions + reference-autograd gradients through both branches.

Usage:  python tools/make_golden.py [--only g1,g2,...]
"""
import sys
sys.dont_write_bytecode = True  # never leave __pycache__ inside the read-only reference tree
import argparse
import os
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mpntrackseg_amd import synth  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


# ------------------------------------------------------------------ torch_scatter 2.0.4 stand-in
def _bcast(index, src, dim):
    if index.dim() == 1:
        shape = [1] * src.dim()
        shape[dim] = -1
        index = index.view(shape)
    return index.expand_as(src)


def _scatter_add(src, index, dim=-1, out=None, dim_size=None):
    dim = dim % src.dim()
    size = list(src.size())
    size[dim] = dim_size if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
    out = torch.zeros(size, dtype=src.dtype, device=src.device)
    return out.scatter_add_(dim, _bcast(index, src, dim), src)


def _scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    dim = dim % src.dim()
    out = _scatter_add(src, index, dim, None, dim_size)
    ones = torch.ones(index.size(), dtype=src.dtype, device=src.device)
    count = _scatter_add(ones, index, 0, None, out.size(dim)).clamp_(1)
    shape = [1] * out.dim()
    shape[dim] = -1
    return out / count.view(shape)


def _scatter_max(src, index, dim=-1, out=None, dim_size=None):
    dim = dim % src.dim()
    size = list(src.size())
    size[dim] = dim_size if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
    out = torch.zeros(size, dtype=src.dtype, device=src.device)
    if src.numel():
        out = out.scatter_reduce(dim, _bcast(index, src, dim), src, reduce="amax", include_self=False)
    return out, None


def _scatter_min(src, index, dim=-1, out=None, dim_size=None):
    raise NotImplementedError   # (imported by data/mot_graph.py at module level; no function under test calls it)


def _scatter_softmax(src, index, dim=-1, eps=1e-12):
    dim = dim % src.dim()
    n = int(index.max()) + 1 if index.numel() else 0
    idx = _bcast(index, src, dim)
    mx = torch.zeros([n] + list(src.shape[1:]), dtype=src.dtype).scatter_reduce(
        dim, idx, src, reduce="amax", include_self=False)
    ex = (src - mx.gather(dim, idx)).exp()
    sm = _scatter_add(ex, index, dim, None, n)
    return ex / (sm.gather(dim, idx) + eps)


def install_shim():
    ts = types.ModuleType("torch_scatter")
    ts.scatter_add, ts.scatter_mean, ts.scatter_max, ts.scatter_min = (
        _scatter_add, _scatter_mean, _scatter_max, _scatter_min)
    comp = types.ModuleType("torch_scatter.composite")
    comp.scatter_softmax = _scatter_softmax
    ts.composite = comp
    sys.modules["torch_scatter"] = ts
    sys.modules["torch_scatter.composite"] = comp


def import_reference():
    install_shim()
    sys.path.insert(0, "/root/reference/src")
    from mot_neural_solver.models import mpn  # noqa
    return mpn


MASK_PARAMS = synth.MASK_PARAMS  # mask-branch dicts of configs/tracking_cfg.yaml:168-218


def build_reference_model(mpn, params, weights):
    full = dict(params)
    full.update(MASK_PARAMS)
    torch.manual_seed(0)
    model = mpn.MOTMPNet(full)
    sd = {k: torch.from_numpy(v) for k, v in weights.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(not any(m.startswith(p) for p in ("encoder.", "MPNet.", "classifier.")) for m in missing), missing
    return model


def ref_hot_path(model, x, edge_index, edge_attr, want_all=True):
    """Drive the REFERENCE modules (encoder / MPNet / classifier) with the loop of mpn.py:355-381,
    leaving out the x_ext lines.  Verified equal to the full forward in g1 below."""
    e, xn = model.encoder(edge_attr, x)
    e0, x0 = e, xn
    logits = []
    for _ in range(model.num_enc_steps):
        e = torch.cat((e0, e), dim=1)
        xn = torch.cat((x0, xn), dim=1)
        xn, e = model.MPNet(xn, edge_index, e)
        dec, _ = model.classifier(e)
        logits.append(dec)
    if model.num_enc_steps == 0:
        dec, _ = model.classifier(e)
        logits.append(dec)
    return logits, xn, e


class Data:
    pass


def gen_g1(mpn):
    for agg in ("sum", "mean", "max"):
        N, E, L, nin = 60, 800, 4, 64
        params = synth.model_params(32, L, agg, num_class_steps=3, node_in_dim=nin)
        W = synth.make_weights(params, seed=7)
        g = synth.make_graph(N, E, T=10, seed=1, node_in_dim=nin)
        model = build_reference_model(mpn, params, W)
        d = Data()
        # reference input layout: x is [N, C, 8, 4] before the avg-pool (seq_processor.py:445)
        x4 = synth.normal(3, (N, nin, 8, 4), stream=9)
        d.x = torch.from_numpy(x4)
        d.x_ext = torch.from_numpy(synth.normal(3, (N, 256, 14, 14), stream=10, std=0.5))
        d.edge_index = torch.from_numpy(g["edge_index"])
        d.edge_attr = torch.from_numpy(g["edge_attr"])
        with torch.no_grad():
            out = model(d)
        full_logits = [t.numpy() for t in out["classified_edges"]]
        assert len(full_logits) == 3 and full_logits[0].shape == (E, 1)

        # hot-path driver on the pooled input, with autograd
        xp = d.x.mean(dim=(2, 3)).clone().requires_grad_(True)
        ea = d.edge_attr.clone().requires_grad_(True)
        logits, xL, eL = ref_hot_path(model, xp, d.edge_index, ea)
        for a, b in zip(full_logits, logits[-3:]):
            assert np.array_equal(a, b.detach().numpy()), "driver loop != MOTMPNet.forward"
        r = torch.from_numpy(synth.normal(11, (L, E), stream=0))
        loss = sum((logits[s].view(-1) * r[s]).sum() for s in range(L))
        hot = {k: p for k, p in model.named_parameters() if k in W}
        grads = torch.autograd.grad(loss, [xp, ea] + list(hot.values()))
        rec = {
            "agg": agg, "N": N, "E": E, "L": L, "d": 32, "node_in_dim": nin,
            "x4": x4, "x_pooled": xp.detach().numpy(), "edge_index": g["edge_index"], "edge_attr": g["edge_attr"],
            "logits": np.stack([t.detach().numpy().reshape(-1) for t in logits]),
            "x_final": xL.detach().numpy(), "e_final": eL.detach().numpy(),
            "r": r.numpy(), "grad_x": grads[0].numpy(), "grad_edge_attr": grads[1].numpy(),
        }
        for k, v in W.items():
            rec["W:" + k] = v
        for k, gr in zip(hot.keys(), grads[2:]):
            rec["G:" + k] = gr.numpy()
        np.savez_compressed(os.path.join(GOLD, f"g1_tiny_{agg}.npz"), **rec)
        print("g1", agg, "max|logit|", float(np.abs(rec["logits"]).max()))


def gen_g6(mpn):
    """Full MOTMPNet.forward INCLUDING the attention / mask branch with deterministic weights for every parameter:
    mask_predictions of the classified steps (12 nodes + checksums of all), and reference-autograd gradients of
    loss = sum logits*r + sum mask_preds*r2 (hot-path parameters, the first attention conv, x_ext)."""
    N, E, L, nin = 40, 360, 3, 64
    params = synth.model_params(32, L, "sum", num_class_steps=2, node_in_dim=nin)
    W = synth.make_weights(params, seed=7)
    W.update(synth.make_mask_weights(seed=17))
    full = dict(params)
    full.update(MASK_PARAMS)
    model = mpn.MOTMPNet(full)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
    g = synth.make_graph(N, E, T=8, seed=4, node_in_dim=nin)
    d = Data()
    d.x = torch.from_numpy(g["x"]).view(N, nin, 1, 1)
    d.x_ext = torch.from_numpy(synth.normal(5, (N, 256, 14, 14), stream=1, std=0.5)).requires_grad_(True)
    d.edge_index = torch.from_numpy(g["edge_index"])
    d.edge_attr = torch.from_numpy(g["edge_attr"])
    out = model(d)
    masks = out["mask_predictions"]
    assert len(masks) == 2 and masks[0].shape == (N, 1, 56, 56)
    r = torch.from_numpy(synth.normal(11, (2, E), stream=0))
    r2 = torch.from_numpy(synth.normal(12, (2, N, 1, 56, 56), stream=0, std=0.05))
    loss = sum((out["classified_edges"][s].view(-1) * r[s]).sum() for s in range(2)) + \
        sum((masks[s] * r2[s]).sum() for s in range(2))
    names = [k for k in W if k.startswith(("encoder.", "MPNet.", "classifier."))] + ["MPAttentionNet.node_model.layers.0.weight",
                                                                                    "mask_predictor.mask_head.layers.0.bias"]
    pd = dict(model.named_parameters())
    grads = torch.autograd.grad(loss, [pd[k] for k in names] + [d.x_ext])
    rec = {"N": N, "E": E, "L": L, "node_in_dim": nin,
           "logits": np.stack([t.detach().numpy().reshape(-1) for t in out["classified_edges"]]),
           "mask_rows": np.stack([m.detach().numpy()[:12] for m in masks]),
           "mask_sum": np.array([float(m.detach().double().sum()) for m in masks]),
           "mask_abssum": np.array([float(m.detach().double().abs().sum()) for m in masks]),
           "grad_x_ext_rows": grads[-1].numpy()[:4, :8],
           "grad_x_ext_abssum": np.float64(grads[-1].double().abs().sum())}
    for k, gr in zip(names, grads[:-1]):
        a = gr.numpy()
        rec["G:" + k] = a if a.size < 20000 else a.reshape(-1)[:20000]
        rec["Gn:" + k] = np.float64(np.sqrt((a.astype(np.float64) ** 2).sum()))
    np.savez_compressed(os.path.join(GOLD, "g6_mask_branch.npz"), **rec)
    print("g6 ok; |mask| per step", rec["mask_abssum"])


def structure_graph():
    """Hand-built corner cases.  Sub-graph 0 (nodes 0..7): node 0 isolated, node 1 only future
    neighbours, node 7 only past neighbours, one self loop (3,3) -- contributes to the edge update but
    to neither aggregate (mpn.py:85,91).  Sub-graphs 1 and 2 are generated and batched behind it so
    the (i<j) / (j<i) halves interleave."""
    lo = np.array([1, 1, 2, 2, 4, 5, 6, 2], dtype=np.int64)
    hi = np.array([2, 4, 5, 7, 7, 6, 7, 6], dtype=np.int64)
    ei = np.stack([np.concatenate([lo, hi, [3]]), np.concatenate([hi, lo, [3]])]).astype(np.int64)
    nin = 64
    g0 = {"x": synth.normal(21, (8, nin), stream=0), "edge_index": ei,
          "edge_attr": synth.normal(21, (ei.shape[1], 6), stream=1), "frame": np.arange(8)}
    g1 = synth.make_graph(20, 60, T=5, seed=22, node_in_dim=nin)
    g2 = synth.make_graph(12, 30, T=4, seed=23, node_in_dim=nin)
    return synth.batch_graphs([g0, g1, g2])


def gen_g4(mpn):
    g = structure_graph()
    rec = {"x": g["x"], "edge_index": g["edge_index"], "edge_attr": g["edge_attr"]}
    for agg in ("sum", "mean", "max"):
        params = synth.model_params(32, 3, agg, node_in_dim=64)
        W = synth.make_weights(params, seed=8)
        model = build_reference_model(mpn, params, W)
        with torch.no_grad():
            logits, xL, eL = ref_hot_path(model, torch.from_numpy(g["x"]), torch.from_numpy(g["edge_index"]),
                                          torch.from_numpy(g["edge_attr"]))
        rec[f"logits_{agg}"] = np.stack([t.numpy().reshape(-1) for t in logits])
        rec[f"x_final_{agg}"] = xL.numpy()
        rec[f"e_final_{agg}"] = eL.numpy()
    # empty graph (E = 0)
    params = synth.model_params(32, 2, "sum", node_in_dim=64)
    model = build_reference_model(mpn, params, synth.make_weights(params, seed=8))
    with torch.no_grad():
        logits, xL, eL = ref_hot_path(model, torch.from_numpy(g["x"][:5]), torch.zeros((2, 0), dtype=torch.int64),
                                      torch.zeros((0, 6)))
    rec["empty_x_final"] = xL.numpy()
    assert logits[0].shape == (0, 1)
    np.savez_compressed(os.path.join(GOLD, "g4_structure.npz"), **rec)
    print("g4 ok, E =", g["edge_index"].shape[1])


def gen_g5(mpn):
    rec = {}
    g = synth.make_graph(40, 300, T=6, seed=31, node_in_dim=64)
    ei = torch.from_numpy(g["edge_index"])
    for agg in ("sum", "mean", "max"):
        params = synth.model_params(32, 1, agg, node_in_dim=64)
        W = synth.make_weights(params, seed=9)
        model = build_reference_model(mpn, params, W)
        x = torch.from_numpy(synth.normal(32, (40, 64), stream=0))   # [N, 2dn]
        e = torch.from_numpy(synth.normal(32, (300, 32), stream=1))  # [E, 2de]
        with torch.no_grad():
            xo, eo = model.MPNet(x, ei, e)                           # MetaLayer.forward mpn.py:33-54
            m = torch.from_numpy(np.maximum(synth.normal(33, (300, 32), stream=2), 0))  # post-ReLU: ties at 0
            row = ei[0]
            ao = model.MPNet.node_model.node_agg_fn(m, row, 40)      # mpn.py:266-273
        rec.update({f"meta_x_{agg}": xo.numpy(), f"meta_e_{agg}": eo.numpy(), f"agg_{agg}": ao.numpy()})
    rec.update({"x_in": x.numpy(), "e_in": e.numpy(), "edge_index": g["edge_index"], "msg": m.numpy()})
    np.savez_compressed(os.path.join(GOLD, "g5_modules.npz"), **rec)
    print("g5 ok")


def gen_g0(mpn):
    params = synth.model_params(32, 0, "sum", num_class_steps=0, node_in_dim=64)
    W = synth.make_weights(params, seed=7)
    g = synth.make_graph(30, 100, T=5, seed=2, node_in_dim=64)
    model = build_reference_model(mpn, params, W)
    d = Data()
    d.x = torch.from_numpy(g["x"]).view(30, 64, 1, 1)
    d.x_ext = torch.from_numpy(synth.normal(3, (30, 256, 14, 14), stream=10, std=0.5))
    d.edge_index = torch.from_numpy(g["edge_index"])
    d.edge_attr = torch.from_numpy(g["edge_attr"])
    with torch.no_grad():
        out = model(d)
    assert len(out["classified_edges"]) == 1
    np.savez_compressed(os.path.join(GOLD, "g0_l0.npz"), logits=out["classified_edges"][0].numpy().reshape(-1))
    print("g0 ok")


def gen_cfg(mpn, name, tag, sample=None):
    c = synth.CONFIGS[name]
    g = synth.make_graph(c["N"], c["E"], seed=1)
    for agg in ("sum", "mean", "max"):
        params = synth.model_params(c["d"], c["L"], agg)
        W = synth.make_weights(params, seed=7)
        model = build_reference_model(mpn, params, W)
        with torch.no_grad():
            logits, xL, eL = ref_hot_path(model, torch.from_numpy(g["x"]), torch.from_numpy(g["edge_index"]),
                                          torch.from_numpy(g["edge_attr"]))
        lg = np.stack([t.numpy().reshape(-1) for t in logits])          # [L, E]
        rec = {"N": c["N"], "E": c["E"], "d": c["d"], "L": c["L"], "agg": agg,
               "cs_x": np.uint64(synth.checksum(g["x"])), "cs_edge_index": np.uint64(synth.checksum(g["edge_index"])),
               "cs_edge_attr": np.uint64(synth.checksum(g["edge_attr"])),
               "cs_weights": np.uint64(synth.checksum(np.concatenate([v.ravel() for v in W.values()]))),
               "step_sum": lg.astype(np.float64).sum(1), "step_abssum": np.abs(lg).astype(np.float64).sum(1),
               "step_max": np.abs(lg).max(1)}
        if sample is None:
            rec["logits"] = lg
        else:
            ids = (synth.uniform01(99, sample, stream=0) * c["E"]).astype(np.int64)
            rec["edge_ids"] = ids
            rec["logits"] = lg[:, ids]
            rec["x_final_rows"] = xL.numpy()[:64]
        np.savez_compressed(os.path.join(GOLD, f"{tag}_{agg}.npz"), **rec)
        print(tag, agg, "max|logit| per step", rec["step_max"][[0, -1]])


def gen_g13(mpn):
    """BASELINE.json configs[4] at FULL size (20,000 nodes / 400,000 edges / 256-d), two message-passing steps: the reference's
    forward (fp32, CPU) -- 4,096 sampled logits per step + whole-tensor checksums.  'mean' has O(1) logits; 'sum' with gain-0.8
    weights keeps them O(10)."""
    c = synth.CONFIGS["E"]
    g = synth.make_graph(c["N"], c["E"], seed=1)
    L = 2
    for agg, gain in (("mean", 1.0), ("sum", 0.8)):
        params = synth.model_params(c["d"], L, agg)
        W = synth.make_weights(params, seed=7, gain=gain)
        model = build_reference_model(mpn, params, W)
        with torch.no_grad():
            logits, xL, eL = ref_hot_path(model, torch.from_numpy(g["x"]), torch.from_numpy(g["edge_index"]),
                                          torch.from_numpy(g["edge_attr"]))
        lg = np.stack([t.numpy().reshape(-1) for t in logits])
        ids = (synth.uniform01(113, 4096, stream=0) * c["E"]).astype(np.int64)
        rec = {"N": c["N"], "E": c["E"], "d": c["d"], "L": L, "agg": agg, "gain": gain,
               "cs_x": np.uint64(synth.checksum(g["x"])), "cs_edge_index": np.uint64(synth.checksum(g["edge_index"])),
               "cs_edge_attr": np.uint64(synth.checksum(g["edge_attr"])),
               "cs_weights": np.uint64(synth.checksum(np.concatenate([v.ravel() for v in W.values()]))),
               "step_sum": lg.astype(np.float64).sum(1), "step_abssum": np.abs(lg).astype(np.float64).sum(1), "step_max": np.abs(lg).max(1),
               "edge_ids": ids, "logits": lg[:, ids], "x_final_rows": xL.numpy()[:32], "e_final_rows": eL.numpy()[:32]}
        np.savez_compressed(os.path.join(GOLD, f"g13_cfgE_{agg}.npz"), **rec)
        print("g13", agg, "max|logit| per step", rec["step_max"])


def _grad_record(rec, names, grads, gx, gea, ids_e):
    """Gradient fixtures: small tensors whole, large ones as their first 20,000 elements + the Euclidean norm of all."""
    for k, gr in zip(names, grads):
        a = gr.numpy()
        rec["G:" + k] = a if a.size <= 20000 else a.reshape(-1)[:20000].copy()
        rec["Gn:" + k] = np.float64(np.sqrt((a.astype(np.float64) ** 2).sum()))
    a = gx.numpy()
    rec["grad_x"] = a if a.size <= 40000 else a[:16].copy()
    rec["grad_x_norm"] = np.float64(np.sqrt((a.astype(np.float64) ** 2).sum()))
    rec["grad_x_rownorm"] = np.sqrt((a.astype(np.float64) ** 2).sum(1))
    b = gea.numpy()
    rec["grad_edge_attr"] = b[ids_e]
    rec["grad_edge_attr_norm"] = np.float64(np.sqrt((b.astype(np.float64) ** 2).sum()))


def _ref_fwd_bwd(mpn, params, W, g, r):
    model = build_reference_model(mpn, params, W)
    xp = torch.from_numpy(g["x"]).clone().requires_grad_(True)
    ea = torch.from_numpy(g["edge_attr"]).clone().requires_grad_(True)
    logits, xL, eL = ref_hot_path(model, xp, torch.from_numpy(g["edge_index"]), ea)
    L = len(logits)
    loss = sum((logits[s].view(-1) * torch.from_numpy(r[s])).sum() for s in range(L))
    hot = {k: p for k, p in model.named_parameters() if k in W}
    grads = torch.autograd.grad(loss, [xp, ea] + list(hot.values()))
    lg = np.stack([t.detach().numpy().reshape(-1) for t in logits])
    return lg, xL.detach(), eL.detach(), list(hot.keys()), grads


def gen_g15(mpn):
    """MLPs with BatchNorm1d in TRAINING mode (models/mlp.py:14; `use_batchnorm: True` in every feats dict, dropout 0): the
    reference's own forward (batch statistics), its autograd incl. the BatchNorm weights / biases, and the running statistics it
    leaves behind, in float64 -- the fixture of the layer-by-layer path (mpntrackseg_amd/modular.py, tests/test_gpu_modular.py)."""
    rec = {}
    for agg in ("sum", "mean", "max"):
        N, E, L, nin = 90, 700, 2, 48
        params = synth.model_params(32, L, agg, node_in_dim=nin)
        for k in ("encoder_feats_dict", "edge_model_feats_dict", "node_model_feats_dict", "classifier_feats_dict"):
            params[k] = dict(params[k], use_batchnorm=True, dropout_p=0)
        g = synth.make_graph(N, E, seed=4, node_in_dim=nin)
        full = dict(params)
        full.update(MASK_PARAMS)
        torch.manual_seed(5)
        model = mpn.MOTMPNet(full)
        hot = ("encoder.", "MPNet.", "classifier.")
        for name, mod in model.named_modules():      # non-trivial BatchNorm state (float32-representable)
            if isinstance(mod, torch.nn.BatchNorm1d) and name.startswith(hot):
                mod.weight.data.uniform_(0.6, 1.4)
                mod.bias.data.normal_(0, 0.2)
                mod.running_mean.normal_(0, 0.3)
                mod.running_var.uniform_(0.5, 1.5)
        state = {k: v.detach().clone() for k, v in model.state_dict().items() if k.startswith(hot)}
        model = model.double().train()
        xp = torch.from_numpy(g["x"]).double().requires_grad_(True)
        ea = torch.from_numpy(g["edge_attr"]).double().requires_grad_(True)
        logits, _, _ = ref_hot_path(model, xp, torch.from_numpy(g["edge_index"]), ea)
        r = synth.normal(12, (L, E))
        loss = sum((logits[s].view(-1) * torch.from_numpy(r[s]).double()).sum() for s in range(L))
        named = [(k, p) for k, p in model.named_parameters() if k.startswith(hot)]
        grads = torch.autograd.grad(loss, [xp, ea] + [p for _, p in named])
        rec["%s:logits" % agg] = np.stack([t.detach().numpy().reshape(-1) for t in logits])
        rec["%s:grad_x" % agg] = grads[0].numpy()
        rec["%s:grad_edge_attr" % agg] = grads[1].numpy()
        for (k, _), gr in zip(named, grads[2:]):
            rec["%s:grad:%s" % (agg, k)] = gr.numpy()
        for k, v in state.items():
            rec["%s:state:%s" % (agg, k)] = v.numpy()
        for k, v in model.state_dict().items():       # buffers AFTER the training-mode forward
            if k.startswith(hot) and ("running_" in k or "num_batches" in k):
                rec["%s:after:%s" % (agg, k)] = v.detach().numpy()
        print("g15", agg, "max|logit|", float(np.abs(rec["%s:logits" % agg]).max()))
    np.savez_compressed(os.path.join(GOLD, "g15_batchnorm_train.npz"), **rec)


def gen_g11(mpn):
    """The HEADLINE workload with O(1) logits: cfg-B (5k nodes / 50k edges / 128-d / 12 steps), node_agg_fn = 'sum' (the shipped
    default), He weights scaled by 0.7 so that the sum-aggregated magnitudes stay O(1) over 12 steps (max |logit| 9.4 at step
    12) -- per-element logit parity means something there -- plus the REFERENCE's autograd of loss = sum logits * r."""
    c = synth.CONFIGS["B"]
    g = synth.make_graph(c["N"], c["E"], seed=1)
    params = synth.model_params(c["d"], c["L"], "sum")
    W = synth.make_weights(params, seed=7, gain=0.7)
    r = synth.normal(11, (c["L"], c["E"]))
    lg, xL, eL, names, grads = _ref_fwd_bwd(mpn, params, W, g, r)
    ids = (synth.uniform01(99, 4096, stream=0) * c["E"]).astype(np.int64)
    rec = {"gain": np.float64(0.7), "cs_x": np.uint64(synth.checksum(g["x"])),
           "cs_weights": np.uint64(synth.checksum(np.concatenate([v.ravel() for v in W.values()]))),
           "edge_ids": ids, "logits": lg[:, ids], "step_sum": lg.astype(np.float64).sum(1),
           "step_abssum": np.abs(lg).astype(np.float64).sum(1), "step_max": np.abs(lg).max(1),
           "x_final_rows": xL.numpy()[:64], "e_final_rows": eL.numpy()[ids[:256]]}
    _grad_record(rec, names, grads[2:], grads[0], grads[1], ids)
    np.savez_compressed(os.path.join(GOLD, "g11_cfgB_sum_o1.npz"), **rec)
    print("g11 max|logit| per step", rec["step_max"])


def gen_g12(mpn):
    """BASELINE.json configs[2] stand-in (SURVEY.md section 8d cfg-C): dense reciprocal-kNN graph, 20 frames x 25 detections,
    top-60 (E / N = 64: long segments -- the block-per-segment reductions of the HIP backward), reference dims d = 32, 12 steps,
    all three aggregations, weights scaled so that the logits stay O(1); reference forward AND reference autograd."""
    g = synth.make_knn_graph(frames=20, dets=25, top_k=60, seed=3, node_in_dim=64)
    E = g["edge_index"].shape[1]
    ids = (synth.uniform01(98, 8192, stream=0) * E).astype(np.int64)
    for agg, gain in (("sum", 0.45), ("mean", 1.0), ("max", 1.0)):
        params = synth.model_params(32, 12, agg, node_in_dim=64)
        W = synth.make_weights(params, seed=7, gain=gain)
        r = synth.normal(11, (12, E))
        lg, xL, eL, names, grads = _ref_fwd_bwd(mpn, params, W, g, r)
        rec = {"gain": np.float64(gain), "E": E, "cs_edge_index": np.uint64(synth.checksum(g["edge_index"])),
               "edge_ids": ids, "logits": lg[:, ids], "step_sum": lg.astype(np.float64).sum(1),
               "step_abssum": np.abs(lg).astype(np.float64).sum(1), "step_max": np.abs(lg).max(1),
               "x_final": xL.numpy(), "e_final_rows": eL.numpy()[ids[:1024]]}
        _grad_record(rec, names, grads[2:], grads[0], grads[1], ids)
        np.savez_compressed(os.path.join(GOLD, f"g12_dense_knn_{agg}.npz"), **rec)
        print("g12", agg, "E", E, "max|logit|", rec["step_max"][[0, -1]])


def gen_g7():
    """utils/graph.py of the reference (graph construction / kNN pruning helpers)."""
    import pandas as pd
    import torch.nn.functional as F
    from mot_neural_solver.utils import graph as G
    det = synth.make_detections()
    df = pd.DataFrame({k: det[k] for k in ("frame", "bb_height", "bb_width", "feet_x", "feet_y")})
    emb = torch.from_numpy(det["reid"])
    fps = 25.0
    out = dict(fps=np.float32(fps), **{"det:" + k: det[k] for k in ("frame", "bb_height", "bb_width", "feet_x", "feet_y", "reid")})
    for tag, mfd in (("max", "max"), ("d3", 3)):
        ei = G.get_time_valid_conn_ixs(torch.from_numpy(det["frame"]), mfd, use_cuda=False)
        out[f"{tag}:edge_ixs"] = ei.numpy()
        feats = G.compute_edge_feats_dict(ei, df, fps, use_cuda=False)
        out[f"{tag}:feats"] = torch.stack([feats[k] for k in ("secs_time_dists", "norm_feet_x_dists", "norm_feet_y_dists",
                                                              "bb_height_dists", "bb_width_dists")]).T.numpy()
        d = F.pairwise_distance(emb[ei[0]], emb[ei[1]])
        out[f"{tag}:emb_dist"] = d.numpy()
        for k in (3, 8):
            for rec in (0, 1):
                m = G.get_knn_mask(d, ei, len(df), k, use_cuda=False, reciprocal_k_nns=bool(rec), symmetric_edges=False)
                out[f"{tag}:knn_k{k}_r{rec}_pairs"] = m.numpy()
                ei2 = torch.cat((ei, torch.stack((ei[1], ei[0]))), dim=1)
                m2 = G.get_knn_mask(torch.cat((d, d)), ei2, len(df), k, use_cuda=False, reciprocal_k_nns=bool(rec),
                                    symmetric_edges=True)
                out[f"{tag}:knn_k{k}_r{rec}_sym"] = m2.numpy()
    np.savez_compressed(os.path.join(GOLD, "g7_graph_utils.npz"), **out)
    print("g7_graph_utils.npz", {k: v.shape for k, v in out.items() if k.startswith("max:")})


def gen_g8():
    """load_precomputed_embeddings (utils/rgb.py:150-188): per-frame ``.pt`` files whose column / channel 0 carries the
    detection id.  utils/rgb.py imports skimage / torchvision / pycocotools / matplotlib at module level for its OTHER
    functions (image cropping, mask decoding, plotting); none is installed here and none is touched by the function under
    test, so empty placeholder modules are registered for the import only."""
    import tempfile
    import types
    import pandas as pd
    for name, attrs in (("skimage", ()), ("skimage.io", ("imread",)), ("torchvision", ()),
                        ("torchvision.transforms", ("Compose", "Resize", "ToTensor", "Normalize")),
                        ("pycocotools", ()), ("pycocotools.mask", ()), ("matplotlib", ()), ("matplotlib.pyplot", ())):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, None)
            sys.modules[name] = m
    from mot_neural_solver.utils import rgb as R
    rng = np.random.RandomState(8)
    frames = [3, 4, 6, 7, 9, 10]
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        next_id, stored1, stored3, fr_of, keep_ids, keep_frames = 0, [], [], [], [], []
        os.makedirs(os.path.join(tmp, "processed_data", "emb1d"))
        os.makedirs(os.path.join(tmp, "processed_data", "emb3d"))
        for f in frames:
            n = int(rng.randint(2, 7))
            ids = np.arange(next_id, next_id + n)
            next_id += n
            e1 = np.concatenate([ids[:, None].astype(np.float32), rng.randn(n, 16).astype(np.float32)], axis=1)
            e3 = rng.randn(n, 5, 3, 2).astype(np.float32)
            e3[:, 0] = ids[:, None, None]
            torch.save(torch.from_numpy(e1), os.path.join(tmp, "processed_data", "emb1d", f"{f}.pt"))
            torch.save(torch.from_numpy(e3), os.path.join(tmp, "processed_data", "emb3d", f"{f}.pt"))
            stored1.append(e1); stored3.append(e3); fr_of += [f] * n
            kept = ids[rng.rand(n) < 0.7]           # the detections that survived the reference's filtering steps
            keep_ids += kept.tolist(); keep_frames += [f] * len(kept)
        # a frame none of whose detections survives is not opened at all (frames_to_retrieve = det_df.frame.unique())
        det_df = pd.DataFrame({"frame": keep_frames, "detection_id": keep_ids})
        info = {"seq_path": tmp}
        o1 = R.load_precomputed_embeddings(det_df, info, "emb1d", use_cuda=False, embedding_dim='1D')
        o3 = R.load_precomputed_embeddings(det_df, info, "emb3d", use_cuda=False, embedding_dim='3D')
    out.update(stored_1d=np.concatenate(stored1), stored_3d=np.concatenate(stored3), stored_frame=np.asarray(fr_of, np.int64),
               det_frame=np.asarray(keep_frames, np.int64), det_id=np.asarray(keep_ids, np.int64),
               out_1d=o1.numpy(), out_3d=o3.numpy())
    np.savez_compressed(os.path.join(GOLD, "g8_embedding_files.npz"), **out)
    print("g8_embedding_files.npz", {k: v.shape for k, v in out.items()})


def _placeholder_modules(specs):
    """Empty stand-ins for third-party / unrelated packages a reference module imports at MODULE level for its OTHER functions
    (nothing of them is touched by the function under test); attributes are set to None unless a value is given."""
    for name, attrs in specs:
        if name in sys.modules:
            m = sys.modules[name]
        else:
            m = types.ModuleType(name)
            sys.modules[name] = m
        for a in attrs:
            if isinstance(a, tuple):
                setattr(m, a[0], a[1])
            elif not hasattr(m, a):
                setattr(m, a, None)
        if "." in name:   # make `import a.b` find b as an attribute of a
            parent, child = name.rsplit(".", 1)
            if parent in sys.modules:
                setattr(sys.modules[parent], child, m)


class _GeoData:
    """Stand-in for torch_geometric.data.Data (third party, not installed): an attribute container with the two derived
    properties the reference code reads -- num_nodes (settable, else rows of x) and num_edges (columns of edge_index)."""
    def __init__(self, **kwargs):
        self._num_nodes = None
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def num_nodes(self):
        return self._num_nodes if self._num_nodes is not None else self.x.shape[0]

    @num_nodes.setter
    def num_nodes(self, v):
        self._num_nodes = v

    @property
    def num_edges(self):
        return self.edge_index.shape[1]


def _import_tracking_stack():
    """utils/evaluation.py, pl_module/pl_module.py and tracker/mpn_tracker.py of the reference.  Their module-level imports pull in
    pytorch_lightning, torch_geometric, motmetrics, the MOTS / KITTI evaluation kits, pulp, pycocotools, torchvision, skimage,
    matplotlib, tracktor -- none installed here and none used by the functions under test (_compute_loss,
    compute_perform_metrics, compute_constr_satisfaction_rate, _predict_edges_and_masks, _evaluate_graph_in_batches)."""
    install_shim()
    sys.modules["torch_scatter"].scatter_min = _scatter_min
    if "/root/reference/src" not in sys.path:
        sys.path.insert(0, "/root/reference/src")

    class _Base:
        def __init__(self, *a, **k):
            pass
    _placeholder_modules([
        ("pytorch_lightning", [("LightningModule", _Base), ("Callback", _Base)]),
        ("torch_geometric", []), ("torch_geometric.data", [("Data", _GeoData), ("DataLoader", None)]),
        # (evaluation.py builds its MOT-metric report formatters at import time: mm.metrics.create().formatters, mm.io.*_names)
        ("motmetrics", [("metrics", types.SimpleNamespace(create=lambda: types.SimpleNamespace(formatters={"mota": None}))),
                        ("io", types.SimpleNamespace(motchallenge_metric_names={"mota": "MOTA"}))]),
        ("MOTChallengeEvalKit", []), ("MOTChallengeEvalKit.MOTS", []),
        ("MOTChallengeEvalKit.MOTS.evalMOTS", ["MOTS_evaluator"]),
        ("TrackEval", []), ("TrackEval.scripts", []), ("TrackEval.scripts.run_kitti_mots", ["eval_kitti_mots"]),
        ("pulp", []), ("pycocotools", []), ("pycocotools.mask", []),
        ("skimage", []), ("skimage.io", ["imread"]),
        ("matplotlib", []), ("matplotlib.pyplot", []),
        ("torchvision", []), ("torchvision.ops", ["roi_align"]), ("torchvision.transforms", ["Compose", "Resize", "ToTensor", "Normalize"]),
        ("torchvision.models", []), ("torchvision.models.detection", []), ("torchvision.models.detection.roi_heads", ["paste_masks_in_image"]),
        ("torchvision.models.utils", ["load_state_dict_from_url"]),
        ("tracktor_masked", []), ("tracktor_masked.maskrcnn_fpn", ["MaskRCNN_FPN"]),
        # reference modules that are unrelated to the functions under test and need yet more packages
        ("mot_neural_solver.data.augmentation", ["MOTGraphAugmentor"]),
        ("mot_neural_solver.data.mot_graph_dataset", ["MOTGraphDataset"]),
        ("mot_neural_solver.models.resnet", ["resnet50_fc256", "load_pretrained_weights"]),
    ])
    import mot_neural_solver.data  # noqa: F401  (package first, so that the placeholders above hang off it)
    from mot_neural_solver.utils import evaluation as EV
    from mot_neural_solver.tracker import mpn_tracker as TR
    from mot_neural_solver.pl_module import pl_module as PL
    return EV, TR, PL


def gen_g9():
    """MOTNeuralSolver._compute_loss (pl_module/pl_module.py:88-120; tracking term: no matched masks) with its autograd gradient
    w.r.t. every classified step's logits, and compute_perform_metrics / compute_constr_satisfaction_rate
    (utils/evaluation.py:340-437), on seeded inputs incl. the no-positive-label and single-edge cases."""
    EV, TR, PL = _import_tracking_stack()
    rec = {}
    cases = [("a", 1000, 3, 0.2), ("b", 6000, 12, 0.02), ("c", 777, 4, 0.0), ("d", 1, 1, 1.0)]
    for tag, E, k, frac in cases:
        logits = synth.normal(3, (k, E), std=3.0)
        labels = (synth.uniform01(4, E) < frac).astype(np.float32)
        lg = torch.from_numpy(logits).clone().requires_grad_(True)
        outputs = {"classified_edges": [lg[s].view(E, 1) for s in range(k)],
                   "mask_predictions": [torch.zeros((2, 1, 4, 4)) for _ in range(k)]}
        batch = types.SimpleNamespace(edge_labels=torch.from_numpy(labels), mask_labels=torch.zeros((2, 1, 4, 4)),
                                      mask_gt_ixs=torch.zeros(0, dtype=torch.long))
        solver = types.SimpleNamespace(hparams={"train_params": {"loss_weights": {"tracking": 0.75, "segmentation": 1.0}}})
        loss = PL.MOTNeuralSolver._compute_loss(solver, outputs, batch)
        loss.backward()
        rec.update({f"{tag}:logits": logits, f"{tag}:labels": labels, f"{tag}:loss": np.float64(float(loss)),
                    f"{tag}:grad": lg.grad.numpy(), f"{tag}:weight": np.float64(0.75)})
    # metrics: two batched tracking graphs (both edge directions present), one self loop, thresholded logits
    g = synth.batch_graphs([synth.make_graph(60, 400, T=6, seed=sd, node_in_dim=4) for sd in (1, 2)])
    ei = g["edge_index"].copy()
    ei[:, 3] = [7, 7]
    for tag, seed, frac in (("m1", 8, 0.25), ("m2", 18, 0.6), ("m3", 28, 0.0)):
        E = ei.shape[1]
        logit = synth.normal(seed, (E, 1))
        labels = (synth.uniform01(seed + 1, E) < frac).astype(np.float32)
        go = types.SimpleNamespace(edge_index=torch.from_numpy(ei), edge_labels=torch.from_numpy(labels), num_nodes=120)
        m = EV.compute_perform_metrics({"classified_edges": [torch.from_numpy(logit)]}, go)
        sr, flow_in, flow_out = EV.compute_constr_satisfaction_rate(go, (torch.from_numpy(logit).view(-1) > 0).float(), return_flow_vals=True)
        rec.update({f"{tag}:logit": logit, f"{tag}:labels": labels,
                    f"{tag}:metrics": np.array([m["accuracy"], m["recall"], m["precision"], m["constr_sr"]], np.float64),
                    f"{tag}:flow_in": flow_in.numpy(), f"{tag}:flow_out": flow_out.numpy()})
    rec["edge_index"] = ei
    np.savez_compressed(os.path.join(GOLD, "g9_loss_metrics.npz"), **rec)
    print("g9 ok:", {k: float(v) for k, v in rec.items() if k.endswith(":loss")}, rec["m1:metrics"])


def gen_g16():
    """accumulate_grad_batches (configs/tracking_cfg.yaml:3-4) executed in space on ONE device: K graphs as one block-diagonal batch.
    The reference calls MOTNeuralSolver._compute_loss (pl_module/pl_module.py:88-107) once per graph -- its own pos_weight, its own
    mean -- and Lightning averages the K backward passes: expected loss = mean over the graphs of the reference's loss on the graph's
    slice of the logits, gradient = its autograd.  Cases: 3 graphs of different sizes, one of them without a positive label; 8 equal
    graphs (the shipped accumulate_grad_batches)."""
    EV, TR, PL = _import_tracking_stack()
    rec = {}
    for tag, sizes, k, fracs in (("g3", [700, 1900, 333], 3, [0.15, 0.0, 0.4]), ("g8", [500] * 8, 4, [0.05 * (i + 1) for i in range(8)])):
        E = sum(sizes)
        logits = synth.normal(7, (k, E), std=2.5)
        labels = np.concatenate([(synth.uniform01(11 + i, n) < f).astype(np.float32) for i, (n, f) in enumerate(zip(sizes, fracs))])
        ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        lg = torch.from_numpy(logits).clone().requires_grad_(True)
        solver = types.SimpleNamespace(hparams={"train_params": {"loss_weights": {"tracking": 0.75, "segmentation": 1.0}}})
        total = 0
        per_graph = []
        for i in range(len(sizes)):
            a, b = int(ptr[i]), int(ptr[i + 1])
            outputs = {"classified_edges": [lg[s, a:b].view(b - a, 1) for s in range(k)],
                       "mask_predictions": [torch.zeros((2, 1, 4, 4)) for _ in range(k)]}
            batch = types.SimpleNamespace(edge_labels=torch.from_numpy(labels[a:b]), mask_labels=torch.zeros((2, 1, 4, 4)),
                                          mask_gt_ixs=torch.zeros(0, dtype=torch.long))
            li = PL.MOTNeuralSolver._compute_loss(solver, outputs, batch)
            per_graph.append(float(li))
            total = total + li
        loss = total / len(sizes)
        loss.backward()
        rec.update({f"{tag}:logits": logits, f"{tag}:labels": labels, f"{tag}:edge_ptr": ptr, f"{tag}:loss": np.float64(float(loss)),
                    f"{tag}:per_graph": np.array(per_graph, np.float64), f"{tag}:grad": lg.grad.numpy(), f"{tag}:weight": np.float64(0.75)})
    np.savez_compressed(os.path.join(GOLD, "g16_loss_graphs.npz"), **rec)
    print("g16 ok:", {k: float(v) for k, v in rec.items() if k.endswith(":loss")})


def gen_g10():
    """MPNTracker._evaluate_graph_in_batches + _predict_edges_and_masks (tracker/mpn_tracker.py:96-210) on a synthetic sequence,
    driven with the REFERENCE model (mask branch included) on CPU.  Two redirections, because the code hard-codes the device:
    the module's `torch.device('cuda')` resolves to the CPU and get_knn_mask is called with use_cuda=False."""
    EV, TR, PL = _import_tracking_stack()
    from mot_neural_solver.utils import graph as G
    import torch.nn.functional as F
    mpn = import_reference()

    class _TorchProxy:
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def device(*a, **k):
            return torch.device("cpu")
    TR.torch = _TorchProxy()
    real_knn = G.get_knn_mask
    TR.get_knn_mask = lambda **kw: real_knn(**dict(kw, use_cuda=False))
    captured = {}
    real_undirected = TR.to_undirected_graph

    def capture_then_undirected(mot_graph, attrs_to_update=("edge_preds", "edge_labels")):
        captured["final_edge_preds"] = mot_graph.graph_obj.edge_preds.clone()
        return real_undirected(mot_graph, attrs_to_update=attrs_to_update)
    TR.to_undirected_graph = capture_then_undirected

    rec = {}
    for tag, inactive, recip, fpg, top_k in (("w1", False, True, 5, 6), ("w2", True, False, 4, 4)):
        det = synth.make_detections(frames=9, dets_lo=3, dets_hi=6, seed=5, emb_dim=32, node_in_dim=64, frame_stride=2)
        n = det["frame"].shape[0]
        import pandas as pd
        df = pd.DataFrame({k: det[k] for k in ("frame", "bb_height", "bb_width", "feet_x", "feet_y")})
        ei = G.get_time_valid_conn_ixs(torch.from_numpy(det["frame"]), "max", use_cuda=False)
        feats = G.compute_edge_feats_dict(ei, df, 25.0, use_cuda=False)
        ef = torch.stack([feats[k] for k in ("secs_time_dists", "norm_feet_x_dists", "norm_feet_y_dists", "bb_height_dists",
                                             "bb_width_dists")]).T
        emb = torch.from_numpy(det["reid"])
        dist = F.pairwise_distance(emb[ei[0]], emb[ei[1]]).view(-1, 1)
        ef = torch.cat((ef, dist), dim=1)
        edge_index = torch.cat((ei, torch.stack((ei[1], ei[0]))), dim=1)
        edge_attr = torch.cat((ef, ef), dim=0)
        emb_dists = torch.cat((dist, dist))
        params = synth.model_params(32, 4, "sum", num_class_steps=2, node_in_dim=64)
        W = synth.make_weights(params, seed=7, gain=0.6)
        W.update(synth.make_mask_weights(seed=17))
        full = dict(params)
        full.update(MASK_PARAMS)
        model = mpn.MOTMPNet(full)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
        x = torch.from_numpy(det["x"]).view(n, 64, 1, 1)
        x_ext = torch.from_numpy(synth.normal(9, (n, 256, 14, 14), stream=1, std=0.5))
        from mot_neural_solver.data.mot_graph import Graph
        graph_obj = Graph(x=x, x_ext=x_ext, edge_attr=edge_attr, reid_emb_dists=emb_dists, edge_index=edge_index)
        full_graph = types.SimpleNamespace(frames=sorted(set(det["frame"].tolist())), graph_df=df, graph_obj=graph_obj,
                                           frames_per_graph=fpg)
        tracker = TR.MPNTracker(dataset=None, graph_model=model, use_gt=False,
                                eval_params={"set_pruned_edges_to_inactive": inactive},
                                dataset_params={"top_k_nns": top_k, "reciprocal_k_nns": recip, "gt_mask_spatial_size": [56, 56]})
        tracker.full_graph = full_graph
        tracker._evaluate_graph_in_batches()
        rec.update({f"{tag}:frame": det["frame"], f"{tag}:x": det["x"], f"{tag}:edge_index": edge_index.numpy(),
                    f"{tag}:edge_attr": edge_attr.numpy(), f"{tag}:reid_emb_dists": emb_dists.numpy(),
                    f"{tag}:final_edge_preds": captured["final_edge_preds"].numpy(),
                    f"{tag}:cfg": np.array([int(inactive), int(recip), fpg, top_k], np.int64)})
        print("g10", tag, "nodes", n, "edges", edge_index.shape[1], "mean pred", float(captured["final_edge_preds"].mean()),
              "windows", len(full_graph.frames) - fpg + 1)
    np.savez_compressed(os.path.join(GOLD, "g10_windows.npz"), **rec)


def gen_g14():
    """MOTGraph._get_edge_ixs + construct_graph_object (data/mot_graph.py:195-218, 283-317) of the reference itself, on synthetic
    detections with the appearance data supplied directly (``_load_appearance_data`` replaced: it reads image crops / .pt files,
    the rows before and after this path): training mode (kNN pruning inside, reciprocal and not) and inference mode (all
    time-valid pairs + reid_emb_dists), 'max' and bounded frame distances.  The MOTGraph is built without its __init__ (which
    slices a sequence data frame): the attributes construct_graph_object reads are set by hand."""
    _import_tracking_stack()
    import pandas as pd
    from mot_neural_solver.data import mot_graph as MG
    from mot_neural_solver.utils import graph as G
    # (the reference moves the index computation to 'cuda' in inference mode: keep it on the CPU here)
    real_tv, real_knn, real_feats = G.get_time_valid_conn_ixs, G.get_knn_mask, G.compute_edge_feats_dict
    MG.get_time_valid_conn_ixs = lambda **kw: real_tv(**dict(kw, use_cuda=False))
    MG.get_knn_mask = lambda **kw: real_knn(**dict(kw, use_cuda=False))
    MG.compute_edge_feats_dict = lambda **kw: real_feats(**dict(kw, use_cuda=False))
    names = ["secs_time_dists", "norm_feet_x_dists", "norm_feet_y_dists", "bb_height_dists", "bb_width_dists", "emb_dist"]
    rec = {}
    cases = [("train_recip", False, 5, True, "max"), ("train_plain", False, 4, False, 6), ("infer", True, 5, True, "max"),
             ("infer_mfd", True, None, True, 4)]
    for tag, inference, top_k, recip, mfd in cases:
        det = synth.make_detections(frames=10, dets_lo=3, dets_hi=7, seed=21, emb_dim=32, node_in_dim=64, frame_stride=2)
        n = det["frame"].shape[0]
        df = pd.DataFrame({k: det[k] for k in ("frame", "bb_height", "bb_width", "feet_x", "feet_y")})
        df["frame_path"] = "synthetic/img1/000001.jpg"
        mg = object.__new__(MG.MOTGraph)
        mg.graph_df = df
        mg.max_frame_dist = mfd
        mg.inference_mode = inference
        mg.seq_info_dict = {"fps": 25.0}
        mg.dataset_params = {"top_k_nns": top_k, "reciprocal_k_nns": recip, "edge_feats_to_use": names}
        emb = torch.from_numpy(det["reid"])
        mg._load_appearance_data = lambda emb=emb, det=det, n=n: (emb, torch.from_numpy(det["x"]).view(n, 64, 1, 1), None)
        mg.construct_graph_object()
        go = mg.graph_obj
        rec.update({f"{tag}:frame": det["frame"], f"{tag}:bb_height": det["bb_height"], f"{tag}:bb_width": det["bb_width"],
                    f"{tag}:feet_x": det["feet_x"], f"{tag}:feet_y": det["feet_y"], f"{tag}:reid": det["reid"],
                    f"{tag}:edge_index": go.edge_index.numpy(), f"{tag}:edge_attr": go.edge_attr.numpy(),
                    f"{tag}:cfg": np.array([int(inference), -1 if top_k is None else top_k, int(recip), -1 if mfd == "max" else mfd], np.int64)})
        if inference:
            rec[f"{tag}:reid_emb_dists"] = go.reid_emb_dists.numpy()
        print("g14", tag, "nodes", n, "edges", go.edge_index.shape[1])
    np.savez_compressed(os.path.join(GOLD, "g14_construct_graph.npz"), **rec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="g0,g1,g4,g5,g6,g7,g8,g2,g3,g11,g12,g10,g9")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    os.makedirs(GOLD, exist_ok=True)
    mpn = import_reference()
    only = set(args.only.split(","))
    if "g0" in only: gen_g0(mpn)
    if "g1" in only: gen_g1(mpn)
    if "g4" in only: gen_g4(mpn)
    if "g5" in only: gen_g5(mpn)
    if "g6" in only: gen_g6(mpn)
    if "g7" in only: gen_g7()
    if "g8" in only: gen_g8()
    if "g2" in only: gen_cfg(mpn, "A", "g2_cfgA")
    if "g3" in only: gen_cfg(mpn, "B", "g3_cfgB", sample=4096)
    if "g10" in only: gen_g10()
    if "g9" in only: gen_g9()
    if "g11" in only: gen_g11(mpn)
    if "g12" in only: gen_g12(mpn)
    if "g13" in only: gen_g13(mpn)
    if "g14" in only: gen_g14()
    if "g15" in only: gen_g15(mpn)
    if "g16" in only: gen_g16()


if __name__ == "__main__":
    main()
