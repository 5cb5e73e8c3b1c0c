"""torch.autograd glue around the native forward / backward entry points (plumbing only: every
gradient is computed by ``mpnhip_backward``)."""
import torch

from . import capi


def avg_pool_native(x):
    """[N,C,H,W] -> [N,C] through ``mpnhip_avgpool`` (reference mpn.py:351-352)."""
    lib = capi.load()
    x = capi.f32c(x)
    n, c = x.shape[0], x.shape[1]
    hw = int(x.shape[2] * x.shape[3])
    y = torch.empty((n, c), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        capi.check(lib.mpnhip_avgpool(capi.ptr(x), n * c, hw, capi.ptr(y), capi.stream_ptr()), "mpnhip_avgpool")
    return y


def native_forward_saved(model, g, x, ea, logits, cmodel=None):
    """mpnhip_forward(save_for_backward=1) into a private workspace; returns that workspace.  ``cmodel``: a native model description
    the caller built with ``model.c_model`` and keeps alive (train.TrainStep caches it: building one costs ~50 us of host time)."""
    lib = capi.load()
    keep = []
    N, E = x.shape[0], ea.shape[0]
    m = cmodel if cmodel is not None else model.c_model(keep, n_edges=E)
    with torch.cuda.device(x.device):
        ws = torch.empty(max(lib.mpnhip_forward_workspace_bytes(m, N, E, 1), 256), dtype=torch.uint8, device=x.device)
        capi.check(lib.mpnhip_forward(m, capi.ptr(g.buf), N, E, capi.ptr(x), capi.ptr(ea), capi.ptr(logits), None, None,
                                      capi.ptr(ws), ws.numel(), 1, capi.stream_ptr()), "mpnhip_forward")
    return ws


def native_backward(model, g, x, ea, grad_logits, fwd_ws, grads, need_gx=False, need_gea=False, defer_side_join=False, cmodel=None):
    """mpnhip_backward; ``grads``: id(param) -> buffer the parameter gradient is ACCUMULATED into.
    ``defer_side_join``: MPNHIP_BWD_DEFER_SIDE_JOIN (include/mpnhip.h) -- the caller joins the side stream itself.
    ``cmodel``: a description built with ``model.c_model(keep, grads=grads, n_edges=E)`` that the caller keeps alive."""
    lib = capi.load()
    keep = []
    N, E = x.shape[0], ea.shape[0]
    m = cmodel if cmodel is not None else model.c_model(keep, grads=grads, n_edges=E)
    gx = torch.empty_like(x) if need_gx else None
    gea = torch.empty_like(ea) if need_gea else None
    gl = capi.f32c(grad_logits)
    with torch.cuda.device(x.device):
        bws = capi.workspace(lib.mpnhip_backward_workspace_bytes(m, N, E), x.device, "bwd")
        capi.check(lib.mpnhip_backward_flags(m, capi.ptr(g.buf), N, E, capi.ptr(x), capi.ptr(ea), capi.ptr(gl), None, None,
                                             capi.ptr(gx), capi.ptr(gea), capi.ptr(fwd_ws), fwd_ws.numel(), capi.ptr(bws),
                                             bws.numel(), 1 if defer_side_join else 0, capi.stream_ptr()), "mpnhip_backward")
    return gx, gea


class _HotPath(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, g, x, edge_attr, *params):
        xd = capi.f32c(x.detach())
        ead = capi.f32c(edge_attr.detach())
        from . import torch_ops
        # eval-mode BatchNorm / Dropout (mlp.py: fast_path False): the dispatcher ops take the parameters as they are, which would
        # drop the BatchNorm fold -- the ctypes path folds it in the forward (MLP.effective_linears) and refuses the backward
        from .mlp import MLP
        plain = all(m_.fast_path for m_ in model.modules() if isinstance(m_, MLP))
        ctx.via_ops = torch_ops.available() and plain
        if ctx.via_ops:
            # through the dispatcher (csrc/torch_ops.cpp): the same C-ABI calls, outputs allocated by the op
            ctx.spec, _ = torch_ops.model_spec(model, n_edges=ead.shape[0])
            with torch.cuda.device(xd.device):
                logits, ctx.fwd_ws = torch_ops.call("forward", g.buf, xd, ead, [p.detach() for p in params], ctx.spec, 1, None, False)
        else:
            L = max(int(model.num_enc_steps), 1)
            logits = torch.empty((L, ead.shape[0]), dtype=torch.float32, device=xd.device)
            ctx.fwd_ws = native_forward_saved(model, g, xd, ead, logits)
        ctx.model, ctx.g = model, g
        # through save_for_backward: autograd's version check then catches an in-place change of x / edge_attr / a weight
        # between forward and backward (the saved activations would no longer belong to them)
        ctx.save_for_backward(xd, ead, *params)
        ctx.params = params   # the Parameter objects themselves: the native model description is keyed by their id()
        return logits

    @staticmethod
    def backward(ctx, grad_logits):
        if ctx.fwd_ws is None:
            raise capi.MpnhipError("the hot path's saved activations were already consumed by a backward pass: a second "
                                   "backward through the same forward (retain_graph=True, checkpointing) is not supported -- "
                                   "run the forward again")
        saved = ctx.saved_tensors   # (raises if one of them was modified in place since the forward)
        x, ea, params = saved[0], saved[1], ctx.params
        if ctx.via_ops:
            from . import torch_ops
            with torch.cuda.device(x.device):
                out = torch_ops.call("backward", ctx.g.buf, x, ea, capi.f32c(grad_logits), ctx.fwd_ws, [p.detach() for p in params], ctx.spec,
                                     bool(ctx.needs_input_grad[2]), bool(ctx.needs_input_grad[3]))
            ctx.fwd_ws = None
            n = len(params)
            gx = out[n] if ctx.needs_input_grad[2] else None
            gea = out[n + 1] if ctx.needs_input_grad[3] else None
            return (None, None, gx, gea) + tuple(out[i] if p.requires_grad else None for i, p in enumerate(params))
        grads = {id(p): torch.zeros_like(p) for p in params}
        gx, gea = native_backward(ctx.model, ctx.g, x, ea, grad_logits, ctx.fwd_ws, grads,
                                  need_gx=ctx.needs_input_grad[2], need_gea=ctx.needs_input_grad[3])
        ctx.fwd_ws = None
        return (None, None, gx, gea) + tuple(grads[id(p)] if p.requires_grad else None for p in params)


def mpn_hot_path_autograd(model, x, edge_index, edge_attr, holder=None, validate=True):
    from .mpn import _prepared, check_hot_path_inputs
    capi.require_device(x, edge_index, edge_attr)
    g = _prepared(edge_index, x.shape[0], holder)
    params = model.hot_path_parameters()
    for p in params:
        capi.require_device(p)
    check_hot_path_inputs(model.c_model([]), g, x, edge_attr)
    out = _HotPath.apply(model, g, x, edge_attr, *params)
    if validate:
        g.raise_if_invalid()
    return out
