"""Host mirror of the reference's graph utilities (``src/mot_neural_solver/utils/graph.py``) and of
``MOTGraph.construct_graph_object`` (``data/mot_graph.py:283-317``) over the C ABI -- SURVEY.md section 8 rows f-3/f-4.

Same function names and argument meaning as the reference; tensors live on the HIP device (there is no CPU fallback).
``det_df`` may be a pandas DataFrame or any mapping with the columns the reference reads (``frame``, ``bb_height``,
``bb_width``, ``feet_x``, ``feet_y``)."""
import ctypes as C

import numpy as np
import torch

from . import capi
from .capi import MpnhipError, check, ptr, stream_ptr


def _dev(device=None):
    if not torch.cuda.is_available():
        raise MpnhipError("mpntrackseg_amd.graph needs a HIP device; there is no CPU fallback")
    return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


def _col(det_df, name, dtype, device):
    v = det_df[name]
    v = v.values if hasattr(v, "values") else v
    if isinstance(v, torch.Tensor):
        return v.to(device=device, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(v))).to(device=device, dtype=dtype)


@capi.on_tensor_device
def compact(flags):
    """ids (int32, ascending) of the set flags and their number (one host read: the consumers are sized by it)."""
    lib = capi.load()
    n = flags.numel()
    ids = torch.empty(max(n, 1), dtype=torch.int32, device=flags.device)
    count = torch.empty(1, dtype=torch.int32, device=flags.device)
    ws = capi.workspace(lib.mpnhip_compact_workspace_bytes(n), flags.device, "compact")
    check(lib.mpnhip_compact(ptr(flags), n, ptr(ids), ptr(count), ptr(ws), ws.numel(), stream_ptr()), "mpnhip_compact")
    k = int(count.item())
    return ids[:k], k


@capi.on_tensor_device
def gather_rows(src, ids):
    lib = capi.load()
    src = capi.f32c(src)
    src2 = src.view(src.shape[0], -1)
    out = torch.empty((ids.numel(), src2.shape[1]), dtype=torch.float32, device=src.device)
    check(lib.mpnhip_gather_rows(ptr(src2), src2.shape[1], ptr(ids), ids.numel(), src2.shape[1], ptr(out), stream_ptr()),
          "mpnhip_gather_rows")
    return out


@capi.on_tensor_device
def gather_edges(edge_index, ids, node_begin):
    lib = capi.load()
    out = torch.empty((2, ids.numel()), dtype=torch.int64, device=edge_index.device)
    check(lib.mpnhip_gather_edges(ptr(edge_index), edge_index.shape[1], ptr(ids), ids.numel(), int(node_begin), ptr(out),
                                  stream_ptr()), "mpnhip_gather_edges")
    return out


@capi.on_tensor_device
def get_time_valid_conn_ixs(frame_num, max_frame_dist, use_cuda=True, return_undirected=True):
    """utils/graph.py:6-37.  ``frame_num``: int tensor [N]; ``max_frame_dist``: int or ``'max'``.
    Returns int64 ``[2, num_pairs]`` with row < col, on the device (the reference moves it back to the CPU)."""
    assert isinstance(max_frame_dist, (int, np.integer)) or max_frame_dist == 'max'
    if not return_undirected:
        raise MpnhipError("get_time_valid_conn_ixs: only return_undirected=True is implemented (the only use in the reference)")
    lib = capi.load()
    dev = frame_num.device if isinstance(frame_num, torch.Tensor) and frame_num.is_cuda else _dev()
    frames = torch.as_tensor(frame_num).to(device=dev, dtype=torch.int64).contiguous().view(-1)
    n = frames.numel()
    maxd = -1 if max_frame_dist == 'max' else int(max_frame_dist)
    offsets = torch.empty(n + 1, dtype=torch.int64, device=dev)
    nbytes = lib.mpnhip_time_valid_conn_workspace_bytes(n)
    ws = capi.workspace(nbytes, dev, "graph_build")
    check(lib.mpnhip_time_valid_conn_count(ptr(frames), n, maxd, ptr(offsets), ptr(ws), ws.numel(), stream_ptr()),
          "mpnhip_time_valid_conn_count")
    n_pairs = int(offsets[n].item())  # the one host read: the result has to be allocated
    out = torch.empty((2, n_pairs), dtype=torch.int64, device=dev)
    check(lib.mpnhip_time_valid_conn_fill(ptr(frames), n, maxd, ptr(offsets), n_pairs, ptr(out), stream_ptr()),
          "mpnhip_time_valid_conn_fill")
    return out


@capi.on_tensor_device
def get_knn_mask(pwise_dist, edge_ixs, num_nodes, top_k_nns, use_cuda=True, reciprocal_k_nns=False, symmetric_edges=True):
    """utils/graph.py:40-87.  Returns a bool tensor [num_edges]: True = keep."""
    lib = capi.load()
    capi.require_device(edge_ixs)
    dev = edge_ixs.device
    ei = edge_ixs.to(torch.int64).contiguous()
    d = pwise_dist.to(device=dev, dtype=torch.float32).contiguous().view(-1)
    e = ei.shape[1]
    assert d.numel() == e, "one distance per edge"
    mask = torch.empty(e, dtype=torch.uint8, device=dev)
    nbytes = lib.mpnhip_knn_mask_workspace_bytes(e, 1 if symmetric_edges else 0)
    ws = capi.workspace(nbytes, dev, "knn")
    check(lib.mpnhip_knn_mask(ptr(d), ptr(ei), int(num_nodes), e, int(top_k_nns), 1 if reciprocal_k_nns else 0,
                              1 if symmetric_edges else 0, ptr(mask), ptr(ws), ws.numel(), stream_ptr()), "mpnhip_knn_mask")
    return mask.bool()


EDGE_FEAT_NAMES = ('secs_time_dists', 'norm_feet_x_dists', 'norm_feet_y_dists', 'bb_height_dists', 'bb_width_dists')


@capi.on_tensor_device
def compute_edge_feats_dict(edge_ixs, det_df, fps, use_cuda=True):
    """utils/graph.py:90-124.  Returns the reference's dict: feature name -> tensor [num_edges]."""
    lib = capi.load()
    capi.require_device(edge_ixs)
    dev = edge_ixs.device
    ei = edge_ixs.to(torch.int64).contiguous()
    e = ei.shape[1]
    frame = _col(det_df, 'frame', torch.int64, dev)
    cols = [_col(det_df, k, torch.float32, dev) for k in ('bb_height', 'bb_width', 'feet_x', 'feet_y')]
    out = torch.empty((e, 5), dtype=torch.float32, device=dev)
    check(lib.mpnhip_edge_features(ptr(ei), e, frame.numel(), ptr(frame), C.c_float(float(fps)), ptr(cols[0]), ptr(cols[1]),
                                   ptr(cols[2]), ptr(cols[3]), ptr(out), stream_ptr()), "mpnhip_edge_features")
    return {name: out[:, i] for i, name in enumerate(EDGE_FEAT_NAMES)}


@capi.on_tensor_device
def pairwise_distance(emb, edge_ixs, eps=1e-6):
    """``F.pairwise_distance(emb[edge_ixs[0]], emb[edge_ixs[1]])`` (data/mot_graph.py:298-301) without materialising
    the two gathered [E, dim] operands.  Returns [num_edges]."""
    lib = capi.load()
    capi.require_device(emb, edge_ixs)
    emb = capi.f32c(emb)
    ei = edge_ixs.to(torch.int64).contiguous()
    e = ei.shape[1]
    out = torch.empty(e, dtype=torch.float32, device=emb.device)
    check(lib.mpnhip_pairwise_distance(ptr(emb), emb.shape[1], emb.shape[1], ptr(ei), e, C.c_float(float(eps)), ptr(out),
                                       stream_ptr()), "mpnhip_pairwise_distance")
    return out


@capi.on_tensor_device
def construct_graph(det_df, reid_embeddings, fps, max_frame_dist, edge_feats_to_use, top_k_nns=None, reciprocal_k_nns=True,
                    inference_mode=True):
    """``MOTGraph._get_edge_ixs`` + ``construct_graph_object`` (data/mot_graph.py:195-317) for precomputed embeddings.

    Returns ``dict(edge_index [2, 2P] int64, edge_attr [2P, F], reid_emb_dists [2P, 1])``: the pairs followed by the
    flipped pairs, features duplicated and NOT sign-flipped, exactly as ``mot_graph.py:311-315`` builds them.  In
    training mode (``inference_mode=False``) the kNN pruning happens here (``:206-216``), in inference per window."""
    frames = _col(det_df, 'frame', torch.int64, reid_embeddings.device)
    edge_ixs = get_time_valid_conn_ixs(frames, max_frame_dist)
    if not inference_mode and top_k_nns is not None:
        d = pairwise_distance(reid_embeddings, edge_ixs)
        keep = get_knn_mask(d, edge_ixs, frames.numel(), top_k_nns, reciprocal_k_nns=reciprocal_k_nns, symmetric_edges=False)
        kept, _ = compact(keep.to(torch.uint8))
        edge_ixs = gather_edges(edge_ixs, kept, 0)   # edge_ixs.T[k_nns_mask].T (mot_graph.py:216)
    feats = compute_edge_feats_dict(edge_ixs, det_df, fps)
    cols = [feats[name] for name in edge_feats_to_use if name in feats]
    edge_feats = torch.stack(cols).T if cols else torch.empty((edge_ixs.shape[1], 0), device=edge_ixs.device)
    emb_dists = pairwise_distance(reid_embeddings, edge_ixs).view(-1, 1)
    if 'emb_dist' in edge_feats_to_use:
        edge_feats = torch.cat((edge_feats, emb_dists), dim=1)
    return dict(edge_index=torch.cat((edge_ixs, torch.stack((edge_ixs[1], edge_ixs[0]))), dim=1),
                edge_attr=torch.cat((edge_feats, edge_feats), dim=0),
                reid_emb_dists=torch.cat((emb_dists, emb_dists)))
