"""CPU ORACLE for the two callers' steps right after the hot path.  TEST INFRASTRUCTURE ONLY.

PIN: ``tests/golden/g9_loss_metrics.npz`` holds the outputs of the reference's own ``MOTNeuralSolver._compute_loss`` (value and
autograd gradient), ``compute_perform_metrics`` and ``compute_constr_satisfaction_rate``, produced in the authoring container by
``tools/make_golden.py gen_g9`` (the modules are imported with empty placeholders for the packages their OTHER functions
need: pytorch_lightning, motmetrics, the evaluation kits); ``tests/test_oracle_golden.py`` checks these restatements against
it.  Each function cites the reference lines it follows (paths relative to /root/reference/src/mot_neural_solver/).
"""
import torch
import torch.nn.functional as F


def tracking_loss(classified_edges, edge_labels, weight=1.0):
    """MOTNeuralSolver._compute_loss, tracking term only (pl_module/pl_module.py:88-107)."""
    positive_vals = edge_labels.sum()
    if positive_vals:
        pos_weight = (edge_labels.shape[0] - positive_vals) / positive_vals            # :92-93
    else:
        pos_weight = torch.zeros(1)                                                     # :95-96
    loss = 0
    for step in range(len(classified_edges)):                                            # :100-104
        cls_loss = F.binary_cross_entropy_with_logits(classified_edges[step].view(-1), edge_labels.view(-1),
                                                      pos_weight=pos_weight)
        loss = loss + weight * cls_loss
    return loss


def tracking_loss_graphs(classified_edges, edge_labels, edge_ptr, weight=1.0):
    """accumulate_grad_batches graphs (configs/tracking_cfg.yaml:3-4) as ONE block-diagonal batch: the reference evaluates
    ``_compute_loss`` per graph (pl_module.py:88-107: the graph's own pos_weight and mean) and the accumulated backward passes are
    averaged -- the mean over the graphs of ``tracking_loss`` on each graph's slice [edge_ptr[g], edge_ptr[g+1]) of the edges.
    Pinned by tests/golden/g16_loss_graphs.npz (tools/make_golden.py gen_g16: the reference's _compute_loss per graph)."""
    K = len(edge_ptr) - 1
    total = 0
    for g in range(K):
        a, b = int(edge_ptr[g]), int(edge_ptr[g + 1])
        total = total + tracking_loss([c.view(-1)[a:b] for c in classified_edges], edge_labels.view(-1)[a:b], weight)
    return total / K


def fast_compute_class_metric(test_preds, test_sols):
    """utils/evaluation.py:340-366."""
    TP = ((test_sols == 1) & (test_preds == 1)).sum().float()
    FP = ((test_sols == 0) & (test_preds == 1)).sum().float()
    TN = ((test_sols == 0) & (test_preds == 0)).sum().float()
    FN = ((test_sols == 1) & (test_preds == 0)).sum().float()
    accuracy = (TP + TN) / (TP + FP + TN + FN)
    recall = TP / (TP + FN) if TP + FN > 0 else torch.tensor(0)
    precision = TP / (TP + FP) if TP + FP > 0 else torch.tensor(0)
    return {"accuracy": float(accuracy), "recall": float(recall), "precision": float(precision)}


def compute_constr_satisfaction_rate(edge_index, num_nodes, edges_out):
    """utils/evaluation.py:370-414 with undirected_edges=True (scatter_add = torch_scatter 2.0.4's)."""
    srt, _ = edge_index.t().sort(dim=1)                                                  # :391-392
    srt = srt.t()
    flow_out = torch.zeros(num_nodes).scatter_add_(0, srt[0], edges_out) / 2.0           # :399
    flow_in = torch.zeros(num_nodes).scatter_add_(0, srt[1], edges_out) / 2.0            # :400
    violated = ((flow_out > 1).sum() + (flow_in > 1).sum()).float()                      # :404-408
    num_constraints = len(srt[0].unique()) + len(srt[1].unique())                        # :409-410
    return float(1 - violated / num_constraints)


def compute_perform_metrics(classified_edges, edge_index, edge_labels, num_nodes):
    """utils/evaluation.py:416-437."""
    edges_out = (classified_edges[-1].view(-1) > 0).float()                              # :428-429
    m = fast_compute_class_metric(edges_out, edge_labels)
    m["constr_sr"] = compute_constr_satisfaction_rate(edge_index, num_nodes, edges_out)
    return m
