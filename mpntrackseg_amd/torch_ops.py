"""``torch.ops.mpnhip.*``: the hot path's operators registered with the PyTorch dispatcher (``csrc/torch_ops.cpp``,
``TORCH_LIBRARY(mpnhip, ...)`` over the C ABI of ``include/mpnhip.h``) -- SURVEY.md section 8b's form of the boundary.

    graph_prep(edge_index, n_nodes, full=True) -> graph            mpnhip_graph_prep          (mpn.py:85-93)
    forward(graph, x, edge_attr, weights, spec, save=0, workspace=None, weights_prepacked=False) -> (logits, workspace)
                                                                    mpnhip_forward             (mpn.py:349-392)
    backward(graph, x, edge_attr, grad_logits, fwd_workspace, weights, spec, need_grad_x, need_grad_edge_attr) -> [grads..., gx, gea]
    meta_layer(graph, x, edge_attr, weights, spec) -> (x', e')     mpnhip_meta_layer_forward  (mpn.py:33-54)
    segment_reduce(src, row, x_size, agg) -> out                   mpnhip_segment_reduce      (mpn.py:266-273)

``model_spec(model)`` turns a ``MOTMPNet`` into the (spec, weights) pair the ops take.  ``MOTMPNet.hot_path`` and the autograd
function of the training path call through these ops when the shim library is built (it is part of ``make``); the ctypes
binding (``capi.py``) remains for everything else and as the fallback.  There is no CPU implementation: the ops are
registered for the HIP dispatch key only."""
import os

import torch

from . import capi

_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmpnhip_torch.so")
_loaded = [None]
load_error = [None]


def available():
    """True when csrc/libmpnhip_torch.so is built and loads (it registers the ops on load)."""
    if _loaded[0] is None:
        _loaded[0] = False
        if os.path.exists(_LIB) and os.environ.get("MPNHIP_NO_TORCH_OPS") is None:
            capi.load()   # libmpnhip.so first: the shim links against it
            try:
                torch.ops.load_library(_LIB)
                _loaded[0] = True
            except OSError as exc:   # built against another torch: the ctypes binding (same C ABI, same kernels) serves instead
                load_error[0] = str(exc)
    return _loaded[0]


def ops():
    if not available():
        raise capi.MpnhipError("%s is missing: build it with `make` (torch.ops.mpnhip.* is a shim over libmpnhip.so)" % _LIB)
    return torch.ops.mpnhip


def call(name, *args):
    """torch.ops.mpnhip.<name>(*args); a failure inside the library surfaces as MpnhipError like on the ctypes path."""
    try:
        return getattr(ops(), name)(*args)
    except capi.MpnhipError:
        raise
    except RuntimeError as exc:
        raise capi.MpnhipError(str(exc).split("\n")[0]) from None


def model_spec(model, n_edges=None):
    """(spec, weights) of a MOTMPNet for the ops: see csrc/torch_ops.cpp.  ``n_edges``: the graph the spec is for ('auto' precision)."""
    prec = model.operand_precision(n_edges)
    nm = model.MPNet.node_model
    lin = nm.node_model[0]
    spec = [int(lin.weight.shape[0]), 0, int(bool(model.reattach_initial_nodes)), int(bool(model.reattach_initial_edges)),
            nm.node_agg_fn.code, int(model.num_enc_steps), capi.PRECISIONS[prec]]
    weights = []
    mlps = [model.encoder.node_model, model.encoder.edge_model, model.MPNet.edge_model.edge_model, nm.flow_in_model, nm.flow_out_model,
            None, model.classifier.edge_model]
    for m in mlps:
        if m is None:
            layers = [lin]
        else:
            m.require_fast_path()
            layers = m.linears()
        spec += [len(layers), int(layers[0].weight.shape[1])] + [int(l.weight.shape[0]) for l in layers]
        if m is None or m.fast_path:
            for l in layers:
                weights += [l.weight.detach(), l.bias.detach()]
        else:   # eval-mode BatchNorm folded into the Linear layers (mlp.py: effective_linears)
            for w, b in m.effective_linears():
                weights += [w, b]
    spec[1] = int(model.MPNet.edge_model.edge_model.linears()[-1].weight.shape[0])   # de = the edge MLP's output width
    return spec, weights
