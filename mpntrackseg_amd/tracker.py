"""Sliding-window inference over a whole sequence graph: host mirror of ``MPNTracker._predict_edges_and_masks`` and
``MPNTracker._evaluate_graph_in_batches`` (reference ``src/mot_neural_solver/tracker/mpn_tracker.py:96-210``) over the C
ABI -- SURVEY.md section 8 row f-3.

Every window of ``frames_per_graph`` consecutive frames becomes a sub-graph (window selection, per-window kNN pruning,
compaction: ``csrc/tracker.hip``), runs through the hot path (``mpnhip_forward``), and its edge probabilities are
added into the sequence-level accumulators on the device; the result is the per-edge average over the windows an edge
appeared in.  Windows are independent, so several can be evaluated in ONE forward as a block-diagonal graph
(``windows_per_launch``; the sub-graphs do not interact, ``tests/test_gpu_parity.py::test_batched_graphs``), and they
shard round-robin over ranks (``rank`` / ``world_size``) with one final sum of the two accumulators.

The projection of the averaged scores onto trajectories (``_project_graph_model_output`` and after: LP / greedy
rounding, pandas bookkeeping) stays in the reference; so does the mask head's per-node averaging (``:191-192``),
which consumes ``MOTMPNet.forward``'s ``mask_predictions`` unchanged."""
import numpy as np
import torch

from . import capi
from .capi import MpnhipError, check, ptr, stream_ptr
from .graph import get_knn_mask, compact as _compact, gather_rows as _gather_rows, gather_edges as _gather_edges


@capi.on_tensor_device
def window_subgraph(edge_index, edge_attr, reid_emb_dists, node_begin, node_end, top_k_nns, reciprocal_k_nns, node_offset=0):
    """Edges of the window [node_begin, node_end) after kNN pruning (mpn_tracker.py:171-178 and :107-112).
    Returns ``(sub_edge_index [2, K] (local ids + node_offset), sub_edge_attr [K, F], window_ids [W] int32, kept_ids [K] int32)``."""
    lib = capi.load()
    E = edge_index.shape[1]
    flags = torch.empty(max(E, 1), dtype=torch.uint8, device=edge_index.device)[:E]
    check(lib.mpnhip_window_flags(ptr(edge_index), E, int(node_begin), int(node_end), ptr(flags), stream_ptr()),
          "mpnhip_window_flags")
    win_ids, n_win = _compact(flags)
    sub_ei = _gather_edges(edge_index, win_ids, node_begin)
    sub_dist = _gather_rows(reid_emb_dists.view(-1, 1), win_ids)
    keep = get_knn_mask(sub_dist, sub_ei, node_end - node_begin, top_k_nns, reciprocal_k_nns=reciprocal_k_nns,
                        symmetric_edges=True)
    kept_ids, _ = _compact(keep.to(torch.uint8))
    sub_ei_k = _gather_edges(sub_ei, kept_ids, -int(node_offset))
    sub_attr = _gather_rows(_gather_rows(edge_attr, win_ids), kept_ids)
    return sub_ei_k, sub_attr, win_ids, kept_ids


def frame_windows(frame_num_per_node, frames_per_graph):
    """Node ranges of the sliding windows (mpn_tracker.py:166-169): detections are ordered by frame, window w spans
    the w-th .. (w + frames_per_graph - 1)-th distinct frame."""
    f = np.asarray(frame_num_per_node.cpu() if isinstance(frame_num_per_node, torch.Tensor) else frame_num_per_node)
    if f.size and (np.diff(f) < 0).any():
        raise MpnhipError("detections must be ordered by frame (MOTGraph sorts them, data/mot_graph.py:145)")
    all_frames = np.unique(f)
    out = []
    for start, end in zip(all_frames, all_frames[frames_per_graph - 1:]):
        out.append((int(np.searchsorted(f, start, side="left")), int(np.searchsorted(f, end, side="right"))))
    return out


@torch.no_grad()
@capi.on_tensor_device
def evaluate_graph_in_batches(model, x, edge_index, edge_attr, reid_emb_dists, frame_num_per_node, frames_per_graph,
                              top_k_nns, reciprocal_k_nns=True, set_pruned_edges_to_inactive=False, windows_per_launch=1,
                              rank=0, world_size=1, reduce_fn=None):
    """``_evaluate_graph_in_batches`` for the edge scores: returns ``final_edge_preds`` [num_edges] of the full graph.

    ``x`` [N, node_in_dim] are the pooled node inputs of the WHOLE sequence (detections ordered by frame), ``edge_index``
    / ``edge_attr`` / ``reid_emb_dists`` its symmetric edge list as ``graph.construct_graph`` builds it.
    ``reduce_fn(tensor)`` sums a tensor over ranks in place (e.g. ``torch.distributed.all_reduce``) when the windows
    are sharded (``rank``, ``world_size``)."""
    capi.require_device(x, edge_index, edge_attr, reid_emb_dists)
    # the weights do not change during one sequence: pack their images once for all its windows
    with model.frozen_weights():
        return _evaluate_windows(model, x, edge_index, edge_attr, reid_emb_dists, frame_num_per_node, frames_per_graph, top_k_nns,
                                 reciprocal_k_nns, set_pruned_edges_to_inactive, windows_per_launch, rank, world_size, reduce_fn)


def _evaluate_windows(model, x, edge_index, edge_attr, reid_emb_dists, frame_num_per_node, frames_per_graph, top_k_nns,
                      reciprocal_k_nns, set_pruned_edges_to_inactive, windows_per_launch, rank, world_size, reduce_fn):
    lib = capi.load()
    edge_index = edge_index.to(torch.int64).contiguous()
    x = capi.f32c(x)
    E = edge_index.shape[1]
    overall_preds = torch.zeros(max(E, 1), dtype=torch.float32, device=x.device)[:E]
    overall_num = torch.zeros(max(E, 1), dtype=torch.float32, device=x.device)[:E]
    windows = frame_windows(frame_num_per_node, frames_per_graph)[rank::world_size]
    L = max(int(model.num_enc_steps), 1)
    for g0 in range(0, len(windows), max(int(windows_per_launch), 1)):
        group = windows[g0:g0 + max(int(windows_per_launch), 1)]
        parts, node_off = [], 0
        for (n0, n1) in group:
            ei_k, attr_k, win_ids, kept_ids = window_subgraph(edge_index, edge_attr, reid_emb_dists, n0, n1, top_k_nns,
                                                              reciprocal_k_nns, node_offset=node_off)
            parts.append((ei_k, attr_k, win_ids, kept_ids, n0, n1))
            node_off += n1 - n0
        if len(parts) == 1:
            ei_b, attr_b, x_b = parts[0][0], parts[0][1], x[parts[0][4]:parts[0][5]]
        else:
            ei_b = torch.cat([p[0] for p in parts], dim=1)
            attr_b = torch.cat([p[1] for p in parts], dim=0)
            x_b = torch.cat([x[p[4]:p[5]] for p in parts], dim=0)
        if ei_b.shape[1] > 0:
            # (the window's indices were built here, inside [0, n): no error-flag read-back, the host keeps running ahead)
            logits = model.hot_path(x_b, ei_b, attr_b, validate=False)[L - 1]  # classified_edges[-1] (mpn_tracker.py:132)
        else:
            logits = torch.empty(0, dtype=torch.float32, device=x.device)
        e_off = 0
        for (ei_k, attr_k, win_ids, kept_ids, n0, n1) in parts:
            k = kept_ids.numel()
            lg = logits[e_off:e_off + k]
            check(lib.mpnhip_window_accumulate(ptr(lg) if k else None, ptr(kept_ids) if k else None, k, ptr(win_ids),
                                               win_ids.numel(), 1 if set_pruned_edges_to_inactive else 0, ptr(overall_preds),
                                               ptr(overall_num), stream_ptr()), "mpnhip_window_accumulate")
            e_off += k
    if reduce_fn is not None and world_size > 1:
        reduce_fn(overall_preds)
        reduce_fn(overall_num)
    final = torch.empty_like(overall_preds)
    check(lib.mpnhip_average_preds(ptr(overall_preds), ptr(overall_num), E, ptr(final), stream_ptr()), "mpnhip_average_preds")
    return final
