"""Layer-by-layer TRAINING path for models whose MLPs carry ``nn.BatchNorm1d`` / ``nn.Dropout``
(``/root/reference/src/mot_neural_solver/models/mlp.py:12-23``).

Batch statistics need every row of a layer's output before the first activation exists, so this configuration cannot run in the
fused per-step kernels; it runs module by module like the reference's own ``forward`` (``models/mpn.py:59-99,349-392``), every
arithmetic piece a HIP kernel behind the C ABI with a hand-written gradient:

* ``linear``            -- ``mpnhip_linear`` / its input gradient (the same kernel on ``W^T``) / ``mpnhip_weight_grad``;
* ``bn_relu_dropout``   -- ``mpnhip_bn_relu_dropout_forward / _backward`` (``csrc/bn_dropout.hip``);
* ``segment_reduce``    -- ``mpnhip_segment_reduce`` / ``mpnhip_segment_reduce_backward`` (``node_agg_fn``, mpn.py:266-273);
* ``gather_rows``       -- ``mpnhip_gather_rows``; its gradient is a segment sum over the same indices (no float atomics).

torch supplies ``cat``, the boolean masks and the autograd graph between these nodes.  No shipped configuration enables
BatchNorm or Dropout (``configs/tracking_cfg.yaml:150-167``); in ``eval()`` mode BatchNorm folds into the Linear layers and the
fused path runs (``MLP.effective_linears``).  There is no CPU fallback: every op raises without the HIP library / device."""
import torch
from torch import nn

from . import capi


def _ws(nbytes, device, tag):
    return capi.workspace(nbytes, device, tag)


class _Linear(torch.autograd.Function):
    """y = x W^T + b (nn.Linear, mlp.py:13)."""

    @staticmethod
    def forward(ctx, x, w, b):
        lib = capi.load()
        x, w = capi.f32c(x.detach()), capi.f32c(w.detach())
        bb = capi.f32c(b.detach()) if b is not None else None
        m, k, n = x.shape[0], x.shape[1], w.shape[0]
        y = torch.empty((m, n), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            if m:
                capi.check(lib.mpnhip_linear(capi.ptr(x), k, capi.ptr(w), capi.ptr(bb), capi.ptr(y), n, m, n, k, 0, capi.stream_ptr()),
                           "mpnhip_linear")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = capi.load()
        x, w = ctx.saved_tensors
        dy = capi.f32c(dy)
        m, k, n = x.shape[0], x.shape[1], w.shape[0]
        dx = dw = db = None
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0]:
                dx = torch.empty((m, k), dtype=torch.float32, device=x.device)
                if m:
                    wt = w.t().contiguous()   # [k, n]: dx = dy W as a Linear with weight W^T
                    capi.check(lib.mpnhip_linear(capi.ptr(dy), n, capi.ptr(wt), None, capi.ptr(dx), k, m, k, n, 0, capi.stream_ptr()),
                               "mpnhip_linear (input gradient)")
            if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
                dw = torch.zeros((n, k), dtype=torch.float32, device=x.device)   # (mpnhip_weight_grad accumulates)
                db = torch.zeros((n,), dtype=torch.float32, device=x.device)
                if m:
                    ws = _ws(lib.mpnhip_weight_grad_workspace_bytes(n, k, m, 1), x.device, "mod_wg")
                    capi.check(lib.mpnhip_weight_grad(capi.ptr(dy), capi.ptr(x), m, n, k, 1, capi.ptr(dw), capi.ptr(db), capi.ptr(ws),
                                                      ws.numel(), capi.stream_ptr()), "mpnhip_weight_grad")
        return dx, dw, (db if ctx.has_bias else None)


class _BnReluDropout(torch.autograd.Function):
    """[BatchNorm1d (batch statistics)] -> [ReLU] -> [Dropout] behind one Linear (mlp.py:14-21)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, bn, relu, p, seed):
        lib = capi.load()
        z = capi.f32c(z.detach())
        m, n = z.shape
        use_bn = bn is not None
        y = torch.empty_like(z)
        mean = torch.empty(n, dtype=torch.float32, device=z.device) if use_bn else None
        invstd = torch.empty(n, dtype=torch.float32, device=z.device) if use_bn else None
        g = capi.f32c(gamma.detach()) if gamma is not None else None
        bt = capi.f32c(beta.detach()) if beta is not None else None
        rm = rv = None
        momentum, eps = 0.0, 1e-5
        if use_bn:
            eps = float(bn.eps)
            if m <= 1:
                raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(z.shape),))
            if bn.track_running_stats and bn.running_mean is not None:
                bn.num_batches_tracked += 1                      # torch/nn/modules/batchnorm.py
                momentum = float(bn.momentum) if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                rm, rv = bn.running_mean, bn.running_var
        with torch.cuda.device(z.device):
            ws = _ws(lib.mpnhip_bn_dropout_workspace_bytes(m, n), z.device, "mod_bn") if use_bn and m else None
            capi.check(lib.mpnhip_bn_relu_dropout_forward(capi.ptr(z), m, n, int(use_bn), capi.ptr(g), capi.ptr(bt), capi.ptr(rm),
                                                          capi.ptr(rv), momentum, eps, int(bool(relu)), float(p), int(seed), capi.ptr(y),
                                                          capi.ptr(mean), capi.ptr(invstd), capi.ptr(ws),
                                                          ws.numel() if ws is not None else 0, capi.stream_ptr()),
                       "mpnhip_bn_relu_dropout_forward")
        ctx.save_for_backward(z, g, bt, mean, invstd)
        ctx.cfg = (use_bn, bool(relu), float(p), int(seed), gamma is not None, beta is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = capi.load()
        z, g, bt, mean, invstd = ctx.saved_tensors
        use_bn, relu, p, seed, has_g, has_b = ctx.cfg
        dy = capi.f32c(dy)
        m, n = z.shape
        dz = torch.empty_like(z)
        dg = torch.zeros(n, dtype=torch.float32, device=z.device) if use_bn and has_g else None
        db = torch.zeros(n, dtype=torch.float32, device=z.device) if use_bn and has_b else None
        with torch.cuda.device(z.device):
            ws = _ws(lib.mpnhip_bn_dropout_workspace_bytes(m, n), z.device, "mod_bn") if use_bn and m else None
            capi.check(lib.mpnhip_bn_relu_dropout_backward(capi.ptr(dy), capi.ptr(z), m, n, int(use_bn), capi.ptr(g), capi.ptr(bt),
                                                           capi.ptr(mean), capi.ptr(invstd), int(relu), p, seed, capi.ptr(dz),
                                                           capi.ptr(dg), capi.ptr(db), capi.ptr(ws), ws.numel() if ws is not None else 0,
                                                           capi.stream_ptr()), "mpnhip_bn_relu_dropout_backward")
        return dz, dg, db, None, None, None, None


class _SegmentReduce(torch.autograd.Function):
    """node_agg_fn(out, row, x_size) (mpn.py:266-273) with the gradient torch_scatter derives."""

    @staticmethod
    def forward(ctx, src, row, x_size, agg):
        lib = capi.load()
        src = capi.f32c(src.detach())
        row = row.contiguous().to(torch.int64)
        m = src.shape[0]
        dim = 1
        for v in src.shape[1:]:
            dim *= int(v)
        out = torch.empty((x_size,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
        argmax = torch.empty((x_size, dim), dtype=torch.int32, device=src.device) if agg == capi.AGG_CODE['max'] else None
        with torch.cuda.device(src.device):
            ws = _ws(lib.mpnhip_segment_reduce_workspace_bytes(m, x_size), src.device, "seg")
            capi.check(lib.mpnhip_segment_reduce(capi.ptr(src), capi.ptr(row), m, dim, x_size, agg, capi.ptr(out), capi.ptr(argmax),
                                                 capi.ptr(ws), ws.numel(), capi.stream_ptr()), "mpnhip_segment_reduce")
        count = torch.bincount(row, minlength=x_size).to(torch.int32) if agg == capi.AGG_CODE['mean'] else None
        ctx.save_for_backward(row, argmax, count)
        ctx.cfg = (m, dim, x_size, agg, tuple(src.shape))
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = capi.load()
        row, argmax, count = ctx.saved_tensors
        m, dim, x_size, agg, shape = ctx.cfg
        dout = capi.f32c(dout)
        dsrc = torch.empty(shape, dtype=torch.float32, device=dout.device)
        with torch.cuda.device(dout.device):
            capi.check(lib.mpnhip_segment_reduce_backward(capi.ptr(dout), capi.ptr(row), capi.ptr(argmax), capi.ptr(count), m, dim,
                                                          x_size, agg, capi.ptr(dsrc), capi.stream_ptr()),
                       "mpnhip_segment_reduce_backward")
        return dsrc, None, None, None


class _GatherRows(torch.autograd.Function):
    """x[idx] (mpn.py:68,86,92); gradient = segment sum of the gathered rows' gradients over idx (fixed order)."""

    @staticmethod
    def forward(ctx, x, idx):
        from .graph import gather_rows
        x = capi.f32c(x.detach())
        ctx.save_for_backward(idx)
        ctx.n = x.shape[0]
        return gather_rows(x, idx.to(torch.int32).contiguous())

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return _SegmentReduce.apply(capi.f32c(dy), idx, ctx.n, capi.AGG_CODE['sum']), None


def linear(x, w, b):
    return _Linear.apply(x, w, b)


def segment_reduce(src, row, x_size, agg_code):
    return _SegmentReduce.apply(src, row, int(x_size), int(agg_code))


def gather_rows(x, idx):
    return _GatherRows.apply(x, idx)


def _next_seed():
    # one 63-bit draw from torch's CPU generator per Dropout call: torch.manual_seed makes a run repeatable, no device read
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def mlp_forward(mlp, x):
    """``MLP.forward`` (mlp.py:27-28) with autograd and TRAINING-mode BatchNorm / Dropout: walks ``fc_layers`` in the reference's
    order (Linear, [BatchNorm1d], [ReLU], [Dropout])."""
    capi.require_device(x)
    mods = list(mlp.fc_layers)
    h = capi.f32c(x)
    lead = h.shape[:-1]
    h = h.reshape(-1, h.shape[-1])
    i = 0
    while i < len(mods):
        lin = mods[i]
        if not isinstance(lin, nn.Linear):
            raise capi.MpnhipError("MLP.fc_layers: expected nn.Linear at index %d, found %s" % (i, type(lin).__name__))
        i += 1
        bn = relu = drop = None
        if i < len(mods) and isinstance(mods[i], nn.BatchNorm1d):
            bn, i = mods[i], i + 1
        if i < len(mods) and isinstance(mods[i], nn.ReLU):
            relu, i = True, i + 1
        if i < len(mods) and isinstance(mods[i], nn.Dropout):
            drop, i = mods[i], i + 1
        capi.require_device(lin.weight)
        z = linear(h, lin.weight, lin.bias)
        p = float(drop.p) if (drop is not None and drop.training) else 0.0
        if p >= 1.0:
            h = torch.zeros_like(z) * z   # nn.Dropout(p=1): all zeros (and zero gradients)
            continue
        if bn is not None and not (bn.training or bn.running_mean is None):
            # eval-mode BatchNorm: the affine map of the running statistics
            s = (bn.weight if bn.affine else torch.ones_like(bn.running_var)) / torch.sqrt(bn.running_var + bn.eps)
            z = (z - bn.running_mean) * s + (bn.bias if bn.affine else 0.0)
            bn = None
        if bn is not None or relu or p > 0.0:
            h = _BnReluDropout.apply(z, bn.weight if (bn is not None and bn.affine) else None,
                                     bn.bias if (bn is not None and bn.affine) else None, bn, bool(relu), p,
                                     _next_seed() if p > 0.0 else 0)
        else:
            h = z
    return h.reshape(*lead, h.shape[-1])


def edge_model_forward(em, x, edge_index, edge_attr):
    """EdgeModel.forward (mpn.py:67-69)."""
    row, col = edge_index[0], edge_index[1]
    out = torch.cat([gather_rows(x, row), gather_rows(x, col), capi.f32c(edge_attr)], dim=1)
    return mlp_forward(em.edge_model, out)


def node_model_forward(nm, x, edge_index, edge_attr):
    """TimeAwareNodeModel.forward (mpn.py:83-99)."""
    row, col = edge_index[0], edge_index[1]
    ea = capi.f32c(edge_attr)
    flows = []
    for mask, mlp in (((row > col), nm.flow_in_model), ((row < col), nm.flow_out_model)):   # :91-96, :85-89
        ids = torch.nonzero(mask, as_tuple=False).view(-1)
        inp = torch.cat([gather_rows(x, col[ids]), gather_rows(ea, ids)], dim=1)
        flows.append(segment_reduce(mlp_forward(mlp, inp), row[ids], x.shape[0], nm.node_agg_fn.code))
    flow = torch.cat((flows[0], flows[1]), dim=1)                                            # :97
    lin = nm.node_model[0]
    z = linear(flow, lin.weight, lin.bias)                                                   # :99 (Linear + ReLU)
    return _BnReluDropout.apply(z, None, None, None, True, 0.0, 0)


def _int64_index(edge_index, holder):
    """``edge_index`` as int64 WITHOUT defeating the holder's graph-prep cache (``mpn._prepared`` keys on tensor identity): an int64
    input is returned as is; any other dtype is converted once per (tensor, version) and the copy kept on the holder, so the
    second call presents the same object again instead of a fresh one that would redo the sort and the CSR build."""
    if edge_index.dtype == torch.int64:
        return edge_index
    if holder is not None:
        c = getattr(holder, "_mpnhip_ei64", None)
        if c is not None and c[0] is edge_index and c[1] == edge_index._version:
            return c[2]
    ei = edge_index.to(torch.int64)
    if holder is not None:
        try:
            object.__setattr__(holder, "_mpnhip_ei64", (edge_index, edge_index._version, ei))
        except Exception:
            pass
    return ei


def hot_path(model, x, edge_index, edge_attr, holder=None, return_state=False):
    """Encoder -> L x (reattach, MetaLayer, classifier): logits [max(L, 1), E] (mpn.py:349-392, the tracking branch);
    ``return_state``: (logits, final node features, final edge features) like ``MOTMPNet.hot_path``."""
    capi.require_device(x, edge_index, edge_attr)
    x, ea = capi.f32c(x), capi.f32c(edge_attr)
    ei = _int64_index(edge_index, holder)
    N = x.shape[0]
    # the reference's x[row] gather raises IndexError (mpn.py:69): validated once per prepared graph through graph prep's flag
    # (cached on the holder like the fused path's), not by a min / max host read per call
    from .mpn import _prepared
    _prepared(ei, N, holder).raise_if_invalid()
    e = mlp_forward(model.encoder.edge_model, ea)                        # mpn.py:356
    h = mlp_forward(model.encoder.node_model, x)
    e0, h0 = e, h
    logits = []
    for _ in range(int(model.num_enc_steps)):
        if model.reattach_initial_edges:                                  # mpn.py:369-373
            e = torch.cat((e0, e), dim=1)
        if model.reattach_initial_nodes:
            h = torch.cat((h0, h), dim=1)
        e = edge_model_forward(model.MPNet.edge_model, h, ei, e)         # MetaLayer.forward, mpn.py:47-53
        h = node_model_forward(model.MPNet.node_model, h, ei, e)
        logits.append(mlp_forward(model.classifier.edge_model, e).view(-1))   # mpn.py:377 -> classifier
    if not logits:
        logits.append(mlp_forward(model.classifier.edge_model, e).view(-1))
    out = torch.stack(logits, dim=0)
    return (out, h, e) if return_state else out
