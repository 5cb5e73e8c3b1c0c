"""Host mirror of the reference's tracking loss (``pl_module/pl_module.py:88-107``) and per-step metrics
(``utils/evaluation.py:416-437``) over the native kernels in ``csrc/loss.hip`` -- the steps right after the
hot path; the loss gradient it returns is the seed of ``mpnhip_backward``."""
import ctypes as C

import torch

from . import capi


def edge_graph_ids(batch, edge_index):
    """Graph id of every edge of a block-diagonal batch (torch_geometric ``Batch.batch`` is per NODE): int32 [E]."""
    return batch.to(torch.int32)[edge_index[0]].contiguous()


def tracking_loss_and_grad(logits, edge_labels, first_step=0, weight=1.0, edge_graph=None, n_graphs=1):
    """logits [L, E] (all steps), edge_labels [E] -> (loss_vec [1 + L] on device: total then per step,
    grad_logits [L, E]).  No host synchronisation.  ``edge_graph`` (int32 [E], with ``n_graphs``): the edges belong to the graphs of
    one block-diagonal batch -- per-graph pos_weight and mean, averaged over the graphs (``mpnhip_tracking_loss_graphs``: the
    reference's accumulate_grad_batches executed in space)."""
    capi.require_device(logits, edge_labels)
    lib = capi.load()
    lg = capi.f32c(logits)
    y = capi.f32c(edge_labels).view(-1)
    L, E = lg.shape
    loss = torch.empty(1 + L, dtype=torch.float32, device=lg.device)
    grad = torch.empty_like(lg)
    if edge_graph is not None:
        capi.require_device(edge_graph)
        eg = edge_graph.to(torch.int32).contiguous().view(-1)
        if eg.numel() != E:
            raise capi.MpnhipError("edge_graph must name the graph of each of the %d edges" % E)
        with torch.cuda.device(lg.device):
            ws = capi.workspace(lib.mpnhip_tracking_loss_graphs_workspace_bytes(L, E, int(n_graphs)), lg.device, "loss")
            capi.check(lib.mpnhip_tracking_loss_graphs(capi.ptr(lg), capi.ptr(y), capi.ptr(eg), int(n_graphs), L, E, int(first_step),
                                                       float(weight), capi.ptr(loss), capi.ptr(grad), capi.ptr(ws), ws.numel(),
                                                       capi.stream_ptr()), "mpnhip_tracking_loss_graphs")
        return loss, grad
    with torch.cuda.device(lg.device):
        ws = capi.workspace(lib.mpnhip_tracking_loss_workspace_bytes(L, E), lg.device, "loss")
        capi.check(lib.mpnhip_tracking_loss(capi.ptr(lg), capi.ptr(y), L, E, int(first_step), float(weight), capi.ptr(loss),
                                            capi.ptr(grad), capi.ptr(ws), ws.numel(), capi.stream_ptr()), "mpnhip_tracking_loss")
    return loss, grad


class _TrackingLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, edge_labels, first_step, weight):
        loss, grad = tracking_loss_and_grad(logits.detach(), edge_labels, first_step, weight)
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


def tracking_loss(classified_edges, edge_labels, weight=1.0):
    """``_compute_loss``'s tracking term for the reference's ``outputs['classified_edges']`` list."""
    lg = torch.stack([t.view(-1) for t in classified_edges])
    return _TrackingLoss.apply(lg, edge_labels, 0, weight)


@capi.on_tensor_device
def compute_perform_metrics(graph_out, graph_obj):
    """utils/evaluation.py:416-437: {'accuracy','recall','precision','constr_sr'} of the last step's logits."""
    from .mpn import _prepared
    lib = capi.load()
    lg = capi.f32c(graph_out['classified_edges'][-1].detach().view(-1))
    y = capi.f32c(graph_obj.edge_labels).view(-1)
    capi.require_device(lg, y, graph_obj.edge_index)
    n = int(graph_obj.num_nodes) if hasattr(graph_obj, "num_nodes") and graph_obj.num_nodes is not None else int(graph_obj.x.shape[0])
    g = _prepared(graph_obj.edge_index, n, graph_obj)
    counts = torch.empty(8, dtype=torch.int32, device=lg.device)
    with torch.cuda.device(lg.device):
        capi.check(lib.mpnhip_step_metrics(capi.ptr(g.buf), g.N, g.E, capi.ptr(lg), capi.ptr(y), capi.ptr(counts),
                                           capi.stream_ptr()), "mpnhip_step_metrics")
    tp, fp, tn, fn, vo, vi, co, ci = [float(v) for v in counts.tolist()]  # the reference's .item() syncs, once
    tot = tp + fp + tn + fn
    return {"accuracy": (tp + tn) / tot if tot else float("nan"),
            "recall": tp / (tp + fn) if tp + fn > 0 else 0.0,
            "precision": tp / (tp + fp) if tp + fp > 0 else 0.0,
            "constr_sr": 1.0 - (vo + vi) / (co + ci) if co + ci > 0 else float("nan")}
