"""CPU restatement of the reference's graph utilities and sliding-window evaluation -- TEST INFRASTRUCTURE ONLY
(imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product path).

Follows, line by line where it matters:
  get_time_valid_conn_ixs     /root/reference/src/mot_neural_solver/utils/graph.py:6-37
  get_knn_mask                utils/graph.py:40-87
  compute_edge_feats_dict     utils/graph.py:90-124
  construct_graph             data/mot_graph.py:195-218 (_get_edge_ixs) and :283-317 (construct_graph_object)
  evaluate_graph_in_batches   tracker/mpn_tracker.py:96-141 (_predict_edges_and_masks) and :143-198

PIN: the three utils/graph.py functions are pinned by tests/golden/g7_graph_utils.npz, and evaluate_graph_in_batches by
tests/golden/g10_windows.npz -- the reference's own MPNTracker._evaluate_graph_in_batches / _predict_edges_and_masks run in
the authoring container on a synthetic sequence with the reference model (tools/make_golden.py gen_g10: placeholder modules
for the packages mpn_tracker.py imports for its OTHER methods, the hard-coded 'cuda' device redirected to the CPU).
construct_graph is the composition of the pinned functions with the tensor concatenations of mot_graph.py:311-315 (the
MOTGraph class itself needs the dataset classes and is not importable here)."""
import numpy as np
import torch
import torch.nn.functional as F


def get_time_valid_conn_ixs(frame_num, max_frame_dist):
    frame_num = torch.as_tensor(frame_num).long()
    d = torch.abs(frame_num.reshape(-1, 1) - frame_num.reshape(1, -1))          # graph.py:23
    cond = d > 0                                                                  # :24
    if max_frame_dist != 'max':
        cond = cond & (d <= max_frame_dist)                                       # :27
    row, col = torch.where(cond)                                                  # :29
    m = row < col                                                                 # :33
    return torch.stack((row[m], col[m]))


def get_knn_mask(pwise_dist, edge_ixs, num_nodes, top_k_nns, reciprocal_k_nns=False, symmetric_edges=True):
    dist_mat = torch.full((num_nodes, num_nodes), float('inf'))                  # :57-58
    dist_mat[edge_ixs[0], edge_ixs[1]] = pwise_dist.view(-1).float()             # :59
    if not symmetric_edges:
        dist_mat[edge_ixs[1], edge_ixs[0]] = pwise_dist.view(-1).float()         # :61
    order = torch.argsort(dist_mat, dim=1, descending=False, stable=True)        # :66 (ties: see csrc/tracker.hip)
    ranking = torch.zeros_like(dist_mat).long()
    rows = torch.arange(num_nodes).view(-1, 1).expand(num_nodes, num_nodes)
    cols = torch.arange(num_nodes).view(1, -1).expand(num_nodes, num_nodes)
    ranking[rows.reshape(-1), order.reshape(-1)] = cols.reshape(-1)              # :70-71
    in_k = ranking < top_k_nns                                                    # :74
    in_k = (in_k & in_k.T) if reciprocal_k_nns else (in_k | in_k.T)              # :75-79
    return in_k[edge_ixs[0], edge_ixs[1]]                                         # :85


EDGE_FEAT_NAMES = ('secs_time_dists', 'norm_feet_x_dists', 'norm_feet_y_dists', 'bb_height_dists', 'bb_width_dists')


def compute_edge_feats_dict(edge_ixs, det, fps):
    row, col = edge_ixs
    t = torch.as_tensor(np.asarray(det['frame'])).float() / fps                  # :107
    h = torch.as_tensor(np.asarray(det['bb_height'])).float()
    w = torch.as_tensor(np.asarray(det['bb_width'])).float()
    fx = torch.as_tensor(np.asarray(det['feet_x'])).float()
    fy = torch.as_tensor(np.asarray(det['feet_y'])).float()
    mean_h = (h[row] + h[col]) / 2                                                # :115
    return {'secs_time_dists': t[col] - t[row],
            'norm_feet_x_dists': (fx[col] - fx[row]) / mean_h,
            'norm_feet_y_dists': (fy[col] - fy[row]) / mean_h,
            'bb_height_dists': torch.log(h[col] / h[row]),
            'bb_width_dists': torch.log(w[col] / w[row])}


def pairwise_distance(emb, edge_ixs):
    out = []
    for i in range(0, edge_ixs.shape[1], 50000):                                  # mot_graph.py:298-301
        out.append(F.pairwise_distance(emb[edge_ixs[0][i:i + 50000]], emb[edge_ixs[1][i:i + 50000]]).view(-1, 1))
    return torch.cat(out, dim=0) if out else torch.empty((0, 1))


def construct_graph(det, reid_embeddings, fps, max_frame_dist, edge_feats_to_use, top_k_nns=None, reciprocal_k_nns=True,
                    inference_mode=True):
    frames = torch.as_tensor(np.asarray(det['frame'])).long()
    edge_ixs = get_time_valid_conn_ixs(frames, max_frame_dist)                   # mot_graph.py:206
    if not inference_mode and top_k_nns is not None:                             # :210-216
        d = F.pairwise_distance(reid_embeddings[edge_ixs[0]], reid_embeddings[edge_ixs[1]])
        keep = get_knn_mask(d, edge_ixs, frames.numel(), top_k_nns, reciprocal_k_nns, symmetric_edges=False)
        edge_ixs = edge_ixs.T[keep].T
    feats = compute_edge_feats_dict(edge_ixs, det, fps)                          # :291-293
    cols = [feats[n] for n in edge_feats_to_use if n in feats]                   # :294
    edge_feats = torch.stack(cols).T
    emb_dists = pairwise_distance(reid_embeddings, edge_ixs)
    if 'emb_dist' in edge_feats_to_use:                                          # :304-305
        edge_feats = torch.cat((edge_feats, emb_dists), dim=1)
    return dict(edge_index=torch.cat((edge_ixs, torch.stack((edge_ixs[1], edge_ixs[0]))), dim=1),   # :312
                edge_attr=torch.cat((edge_feats, edge_feats), dim=0),                                  # :311
                reid_emb_dists=torch.cat((emb_dists, emb_dists)))                                      # :315


def evaluate_graph_in_batches(forward_fn, x, edge_index, edge_attr, reid_emb_dists, frame_num_per_node, frames_per_graph,
                              top_k_nns, reciprocal_k_nns=True, set_pruned_edges_to_inactive=False):
    """forward_fn(x_sub, edge_index_sub, edge_attr_sub) -> logits of the LAST classified step, [E_sub]."""
    frame_num_per_node = torch.as_tensor(np.asarray(frame_num_per_node)).long()
    all_frames = np.unique(frame_num_per_node.numpy())
    node_names = torch.arange(x.shape[0])
    E = edge_index.shape[1]
    overall_edge_preds = torch.zeros(E)
    overall_num_preds = torch.zeros(E)
    for start_frame, end_frame in zip(all_frames, all_frames[frames_per_graph - 1:]):          # mpn_tracker.py:166
        nodes_mask = (start_frame <= frame_num_per_node) & (frame_num_per_node <= end_frame)   # :170
        edges_mask = nodes_mask[edge_index[0]] & nodes_mask[edge_index[1]]                       # :171-172
        sub_ei = edge_index.T[edges_mask].T - node_names[nodes_mask][0]                          # :178
        sub_attr = edge_attr[edges_mask]
        sub_dist = reid_emb_dists[edges_mask]
        knn_mask = get_knn_mask(sub_dist, sub_ei, int(nodes_mask.sum()), top_k_nns, reciprocal_k_nns, True)   # :107-110
        sub_ei_k, sub_attr_k = sub_ei.T[knn_mask].T, sub_attr[knn_mask]                          # :111-112
        if sub_ei_k.shape[1] > 0:
            pruned = torch.sigmoid(forward_fn(x[nodes_mask], sub_ei_k, sub_attr_k).view(-1))    # :132
        else:
            pruned = torch.zeros(0)
        edge_preds = torch.zeros(knn_mask.shape[0])
        edge_preds[knn_mask] = pruned                                                            # :135-136
        pred_mask = torch.ones_like(knn_mask) if set_pruned_edges_to_inactive else knn_mask     # :138-141
        overall_edge_preds[edges_mask] += edge_preds                                             # :188
        idx = torch.where(edges_mask)[0][pred_mask]
        overall_num_preds[idx] += 1                                                              # :190
    final = overall_edge_preds / overall_num_preds                                               # :195
    final[torch.isnan(final)] = 0                                                                # :196
    return final
