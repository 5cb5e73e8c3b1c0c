"""Host-side mirror of the reference's message-passing modules
(``/root/reference/src/mot_neural_solver/models/mpn.py``): same class names, constructor arguments,
attribute names and ``state_dict`` keys, so ``mot_neural_solver`` can import
``MOTMPNet`` / ``MetaLayer`` from here unchanged and reference checkpoints load -- but every
``forward`` goes through the C ABI (``include/mpnhip.h``) into hand-written gfx950 kernels.

There is no CPU fallback: tensors must live on a HIP device, and a missing ``libmpnhip.so`` raises.
"""
import contextlib
import ctypes as C

import torch
from torch import nn

from . import capi
from .cnn import CNN, MaskRCNNPredictor
from .mlp import MLP


def _prepared(edge_index, n_nodes, holder=None, full=True):
    """Graph prep (``mpnhip_graph_prep``) cached on the object that owns edge_index (the reference's
    ``Graph`` sample), validated by tensor identity and version counter.  ``full=False`` (inference forward) accepts
    or builds a prep with the primary order only; a cached forward-only prep is replaced when a full one is asked for.
    The cache attribute is a dunder name on purpose: torch_geometric's ``Data.keys`` (what ``.to()``, ``Batch.from_data_list`` and the
    reference's ``Graph._change_attrs_types`` walk, data/mot_graph.py:27-52) skips ``__x__`` names, so the sample's own protocol never
    sees it."""
    if holder is not None:
        c = getattr(holder, "__mpnhip_prep__", None)
        if c is not None and c[0] is edge_index and c[1] == edge_index._version and c[2].N == n_nodes and (c[2].full or not full):
            return c[2]
    g = capi.PreparedGraph(edge_index, n_nodes, full=full)
    if holder is not None:
        try:
            object.__setattr__(holder, "__mpnhip_prep__", (edge_index, edge_index._version, g))
        except Exception:
            pass
    return g


def check_hot_path_inputs(m, g, x, edge_attr):
    """Shape contract of the native hot path, checked where torch would raise in the reference (the C ABI sees raw pointers
    only): x [N, node_in_dim], edge_attr [E, edge_in_dim], and the prepared graph must describe exactly these N nodes / E edges
    (its buffer layout is a function of (N, E): a mismatch would send the kernels out of bounds)."""
    if x.dim() != 2 or x.shape[1] != m.enc_node.in_dim or edge_attr.dim() != 2 or edge_attr.shape[1] != m.enc_edge.in_dim:
        raise capi.MpnhipError("input feature widths do not match the encoder (node %s, edge %s; expected [N, %d] and [E, %d])"
                               % (tuple(x.shape), tuple(edge_attr.shape), m.enc_node.in_dim, m.enc_edge.in_dim))
    if g.N != x.shape[0]:
        raise capi.MpnhipError("the prepared graph has %d nodes but x has %d rows" % (g.N, x.shape[0]))
    if g.E != edge_attr.shape[0]:
        raise capi.MpnhipError("edge_index has %d edges but edge_attr has %d rows" % (g.E, edge_attr.shape[0]))
    if x.device != g.device or edge_attr.device != g.device:
        raise capi.MpnhipError("x / edge_attr / edge_index live on different devices")


def _wants_modular(module, *inputs):
    """Operator-level calls: autograd requested, or BatchNorm / Dropout MLPs in training mode -> ``modular.py``."""
    if torch.is_grad_enabled() and (any(t.requires_grad for t in inputs) or any(p.requires_grad for p in module.parameters())):
        return True
    return any(isinstance(m_, MLP) and m_.training and not m_.fast_path for m_ in module.modules())


class NodeAggFn:
    """The reference's ``node_agg_fn`` lambdas (mpn.py:266-273): ``fn(out, row, x_size)``."""

    def __init__(self, name):
        assert name.lower() in ('mean', 'max', 'sum'), "node_agg_fn can only be 'max', 'mean' or 'sum'."
        self.name = name

    @property
    def code(self):
        return capi.AGG_CODE[self.name]

    def __call__(self, out, row, x_size):
        capi.require_device(out, row)
        if torch.is_grad_enabled() and out.requires_grad:
            from .modular import segment_reduce
            return segment_reduce(out, row, x_size, self.code)
        lib = capi.load()
        src = capi.f32c(out)
        row = row.contiguous().to(torch.int64)
        m = src.shape[0]
        dim = 1
        for v in src.shape[1:]:
            dim *= int(v)
        res = torch.empty((x_size,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
        with torch.cuda.device(src.device):
            ws = capi.workspace(lib.mpnhip_segment_reduce_workspace_bytes(m, x_size), src.device, "seg")
            capi.check(lib.mpnhip_segment_reduce(capi.ptr(src), capi.ptr(row), m, dim, x_size, self.code, capi.ptr(res),
                                                 None, capi.ptr(ws), ws.numel(), capi.stream_ptr()),
                       "mpnhip_segment_reduce")
        return res


class EdgeModel(nn.Module):
    """mpn.py:59-69."""

    def __init__(self, edge_model):
        super(EdgeModel, self).__init__()
        self.edge_model = edge_model

    def forward(self, node_feats, edge_index, edge_attr):
        """mpn.py:67-69, operator level (inference): native row gathers, the reference's cat, the native MLP.
        ``MetaLayer.forward`` / ``MOTMPNet.forward`` evaluate the same module fused (project-then-gather)."""
        from .graph import gather_rows
        capi.require_device(node_feats, edge_index, edge_attr)
        if _wants_modular(self, node_feats, edge_attr):
            from .modular import edge_model_forward
            return edge_model_forward(self, capi.f32c(node_feats), edge_index.to(torch.int64), edge_attr)
        row, col = edge_index[0].to(torch.int32).contiguous(), edge_index[1].to(torch.int32).contiguous()
        out = torch.cat([gather_rows(node_feats, row), gather_rows(node_feats, col), capi.f32c(edge_attr)], dim=1)
        return self.edge_model(out)


class TimeAwareNodeModel(nn.Module):
    """mpn.py:71-99."""

    def __init__(self, flow_in_model, flow_out_model, node_model, node_agg_fn):
        super(TimeAwareNodeModel, self).__init__()
        self.flow_in_model = flow_in_model
        self.flow_out_model = flow_out_model
        self.node_model = node_model
        self.node_agg_fn = node_agg_fn

    def forward(self, x, edge_index, edge_attr):
        """mpn.py:83-99, operator level (inference): the boolean-mask selections become native compactions, the
        gathers / MLPs / ``node_agg_fn`` / Linear native calls.  ``MetaLayer.forward`` evaluates the same module fused."""
        from .graph import compact, gather_rows
        capi.require_device(x, edge_index, edge_attr)
        if _wants_modular(self, x, edge_attr):
            from .modular import node_model_forward
            return node_model_forward(self, capi.f32c(x), edge_index.to(torch.int64), edge_attr)
        with torch.cuda.device(x.device):
            return self._forward(x, edge_index, edge_attr, compact, gather_rows)

    def _forward(self, x, edge_index, edge_attr, compact, gather_rows):
        row, col = edge_index
        ea = capi.f32c(edge_attr)
        flows = []
        for mask, mlp in (((row > col), self.flow_in_model), ((row < col), self.flow_out_model)):   # :91-96, :85-89
            ids, _ = compact(mask.to(torch.uint8).contiguous())
            sel_col = col.to(torch.int32)[ids.long()].contiguous()
            sel_row = row[ids.long()]
            inp = torch.cat([gather_rows(x, sel_col), gather_rows(ea, ids)], dim=1)
            flows.append(self.node_agg_fn(mlp(inp), sel_row, x.size(0)))
        flow = torch.cat((flows[0], flows[1]), dim=1)                                                # :97
        lin = self.node_model[0]                                                                      # Linear + ReLU, :99
        out = torch.empty((flow.shape[0], lin.out_features), dtype=torch.float32, device=flow.device)
        lib = capi.load()
        capi.check(lib.mpnhip_linear(capi.ptr(flow), flow.shape[1], capi.ptr(lin.weight), capi.ptr(lin.bias), capi.ptr(out),
                                     out.shape[1], flow.shape[0], lin.out_features, lin.in_features, 1, capi.stream_ptr()),
                   "mpnhip_linear")
        return out


class MetaLayer(nn.Module):
    """mpn.py:11-57: ``forward(x, edge_index, edge_attr) -> (x, edge_attr)``: edge update, then node
    update on the updated edges.  One fused native call (``mpnhip_meta_layer_forward``)."""

    def __init__(self, edge_model=None, node_model=None):
        super(MetaLayer, self).__init__()
        self.edge_model = edge_model
        self.node_model = node_model
        self.reset_parameters()

    def reset_parameters(self):
        for item in [self.node_model, self.edge_model]:
            if hasattr(item, 'reset_parameters'):
                item.reset_parameters()

    def core_struct(self, keep, agg_name=None, grads=None):
        """Fill the MetaLayer part of an ``mpnhip_model``."""
        if not isinstance(self.edge_model, EdgeModel) or not isinstance(self.node_model, TimeAwareNodeModel):
            raise capi.MpnhipError("MetaLayer needs an EdgeModel and a TimeAwareNodeModel (as MOTMPNet builds them)")
        nm = self.node_model
        m = capi.Model()
        m.edge = self.edge_model.edge_model.c_struct(keep, grads)
        m.flow_in = nm.flow_in_model.c_struct(keep, grads)
        m.flow_out = nm.flow_out_model.c_struct(keep, grads)
        lin = nm.node_model[0]
        capi.require_device(lin.weight)
        capi.fill_mlp(m.node, [(lin.weight.detach(), lin.bias.detach())], keep=keep)
        if grads is not None:
            m.node.grad_weight[0] = grads[id(lin.weight)].data_ptr()
            m.node.grad_bias[0] = grads[id(lin.bias)].data_ptr()
        m.dn = int(lin.weight.shape[0])
        m.de = int(m.edge.out_dims[m.edge.n_layers - 1])
        agg = nm.node_agg_fn
        m.agg = agg.code if isinstance(agg, NodeAggFn) else capi.AGG_CODE[agg_name or 'sum']
        return m

    def forward(self, x, edge_index, edge_attr):
        capi.require_device(x, edge_index, edge_attr)
        if _wants_modular(self, x, edge_attr):
            # autograd at operator level (and BatchNorm / Dropout in training mode): module by module, mpn.py:47-53
            edge_attr = self.edge_model(x, edge_index, edge_attr)
            return self.node_model(x, edge_index, edge_attr), edge_attr
        lib = capi.load()
        keep = []
        m = self.core_struct(keep)
        x = capi.f32c(x)
        e = capi.f32c(edge_attr)
        N, E = x.shape[0], e.shape[0]
        # reattach factors follow from the input widths (mpn.py:276-285)
        if x.shape[1] not in (m.dn, 2 * m.dn) or e.shape[1] not in (m.de, 2 * m.de):
            raise capi.MpnhipError("MetaLayer input widths do not match the edge / flow MLP dims")
        m.reattach_nodes = int(x.shape[1] == 2 * m.dn)
        m.reattach_edges = int(e.shape[1] == 2 * m.de)
        g = _prepared(edge_index, N)
        x_new = torch.empty((N, m.dn), dtype=torch.float32, device=x.device)
        e_new = torch.empty((E, m.de), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            ws = capi.workspace(lib.mpnhip_meta_layer_workspace_bytes(m, N, E), x.device, "meta")
            capi.check(lib.mpnhip_meta_layer_forward(m, capi.ptr(g.buf), N, E, capi.ptr(x), capi.ptr(e), capi.ptr(x_new),
                                                     capi.ptr(e_new), capi.ptr(ws), ws.numel(), capi.stream_ptr()),
                       "mpnhip_meta_layer_forward")
        return x_new, e_new

    def __repr__(self):
        return '{}(edge_model={}, node_model={})'.format(self.__class__.__name__, self.edge_model, self.node_model)


class MLPGraphIndependent(nn.Module):
    """mpn.py:139-178: independent node / edge MLPs (encoder, classifier)."""

    def __init__(self, edge_in_dim=None, node_in_dim=None, edge_out_dim=None, node_out_dim=None,
                 node_dims=None, edge_dims=None, dropout_p=None, use_batchnorm=None):
        super(MLPGraphIndependent, self).__init__()
        if node_in_dim is not None:
            self.node_model = MLP(input_dim=node_in_dim, fc_dims=list(node_dims) + [node_out_dim],
                                  dropout_p=dropout_p, use_batchnorm=use_batchnorm)
        else:
            self.node_model = None
        if edge_in_dim is not None:
            self.edge_model = MLP(input_dim=edge_in_dim, fc_dims=list(edge_dims) + [edge_out_dim],
                                  dropout_p=dropout_p, use_batchnorm=use_batchnorm)
        else:
            self.edge_model = None

    def forward(self, edge_feats=None, nodes_feats=None):
        out_node_feats = self.node_model(nodes_feats) if self.node_model is not None else nodes_feats
        out_edge_feats = self.edge_model(edge_feats) if self.edge_model is not None else edge_feats
        return out_edge_feats, out_node_feats


class _AttentionAggregate(torch.autograd.Function):
    """flow_in, flow_out of TimeAwareAttentionModel.forward (mpn.py:117-134) through
    ``mpnhip_attention_aggregate`` and its hand-written backward."""

    @staticmethod
    def forward(ctx, g, x, logits):
        lib = capi.load()
        xc = capi.f32c(x.detach())
        lg = capi.f32c(logits.detach()).view(-1)
        n = xc.shape[0]
        feat = int(xc[0].numel()) if n else 0
        flow_in = torch.empty_like(xc)
        flow_out = torch.empty_like(xc)
        wts = torch.empty(max(g.E, 1), dtype=torch.float32, device=xc.device)
        with torch.cuda.device(xc.device):
            capi.check(lib.mpnhip_attention_aggregate(capi.ptr(g.buf), g.N, g.E, capi.ptr(xc), feat, capi.ptr(lg),
                                                      capi.ptr(flow_in), capi.ptr(flow_out), capi.ptr(wts), capi.stream_ptr()),
                       "mpnhip_attention_aggregate")
        ctx.g, ctx.feat = g, feat
        ctx.save_for_backward(xc, wts)
        ctx.logit_shape = logits.shape
        return flow_in, flow_out

    @staticmethod
    def backward(ctx, d_in, d_out):
        lib = capi.load()
        xc, wts = ctx.saved_tensors
        g = ctx.g
        d_in, d_out = capi.f32c(d_in), capi.f32c(d_out)
        gx = torch.empty_like(xc) if ctx.needs_input_grad[1] else None
        gl = torch.zeros(max(g.E, 1), dtype=torch.float32, device=xc.device) if ctx.needs_input_grad[2] else None
        dw = torch.empty(max(g.E, 1), dtype=torch.float32, device=xc.device)
        with torch.cuda.device(xc.device):
            capi.check(lib.mpnhip_attention_aggregate_backward(capi.ptr(g.buf), g.N, g.E, capi.ptr(xc), ctx.feat, capi.ptr(wts),
                                                               capi.ptr(d_in), capi.ptr(d_out), capi.ptr(gx), 0, capi.ptr(gl),
                                                               capi.ptr(dw), capi.stream_ptr()),
                       "mpnhip_attention_aggregate_backward")
        return None, gx, (gl[:g.E].view(ctx.logit_shape) if gl is not None else None)


class TimeAwareAttentionModel(nn.Module):
    """mpn.py:102-137.  Like the reference, only ``node_model`` is kept (the two attention MLPs passed to the
    constructor are never registered there either, mpn.py:106-109)."""

    def __init__(self, node_model, flow_in_attention_model=None, flow_out_attention_model=None):
        super(TimeAwareAttentionModel, self).__init__()
        self.node_model = node_model

    def aggregate(self, x, edge_index, dec_edge_feats, holder=None):
        capi.require_device(x, edge_index, dec_edge_feats)
        g = _prepared(edge_index, x.shape[0], holder)
        flow_in, flow_out = _AttentionAggregate.apply(g, x, dec_edge_feats)
        flow = torch.cat((x, flow_in, flow_out), dim=1)          # mpn.py:136
        return self.node_model(flow)

    def forward(self, x, edge_index, edge_attr, cls_net):
        dec_edge_feats, _ = cls_net(edge_attr)                    # mpn.py:114
        return self.aggregate(x, edge_index, dec_edge_feats), dec_edge_feats


class MaskModel(nn.Module):
    """mpn.py:180-206 (stock convolutions / LayerNorm)."""

    def __init__(self, mask_model_params):
        super(MaskModel, self).__init__()
        self.feature_encoder = CNN(**mask_model_params['feature_encoder_feats_dict'])
        self.layer_norm = nn.LayerNorm([64, 14, 14])
        self.mask_head = CNN(**mask_model_params['mask_head_feats_dict'])
        self.mask_predictor = MaskRCNNPredictor(**mask_model_params['mask_predictor_feats_dict'])

    def forward(self, feature_embeds, node_embeds):
        x = torch.cat((self.feature_encoder(feature_embeds), node_embeds), dim=1)
        return self.mask_predictor(self.mask_head(self.layer_norm(x)))


class MOTMPNet(nn.Module):
    """mpn.py:209-394.  ``MOTMPNet(model_params, bb_encoder=None)``; ``forward(data)`` returns
    ``{'classified_edges': [Tensor[E,1]] * num_class_steps, 'mask_predictions': [...]}``.

    The encoder -> message passing -> classifier loop (the hot path) is ONE native call.  The x_ext /
    attention / mask branch of the reference (mpn.py:102-137,180-206) never feeds back into the edge logits
    (SURVEY.md section 3.3), so it runs AFTER the hot path, step by step, on the per-step logits: its neighbour
    aggregation is native (``mpnhip_attention_aggregate``), its convolutions are stock PyTorch-ROCm modules.
    It is built only when ``model_params`` carries the mask-branch dicts (as configs/tracking_cfg.yaml does) and
    evaluated only when ``data.x_ext`` is present; otherwise ``mask_predictions`` is an empty list.
    """

    def __init__(self, model_params, bb_encoder=None):
        super(MOTMPNet, self).__init__()
        self.node_cnn = bb_encoder
        self.model_params = model_params
        encoder_feats_dict = model_params['encoder_feats_dict']
        classifier_feats_dict = model_params['classifier_feats_dict']
        self.encoder = MLPGraphIndependent(**encoder_feats_dict)
        self.classifier = MLPGraphIndependent(**classifier_feats_dict)
        self.has_mask_branch = all(k in model_params for k in ('node_ext_encoder_feats_dict', 'mask_model_feats_dict',
                                                              'node_ext_model_feats_dict'))
        if self.has_mask_branch:
            self.node_ext_encoder = CNN(**model_params['node_ext_encoder_feats_dict'])
            self.mask_predictor = MaskModel(model_params['mask_model_feats_dict'])
        self.MPNet = self._build_core_MPNet(model_params=model_params, encoder_feats_dict=encoder_feats_dict)
        if self.has_mask_branch:
            self.MPAttentionNet = self._build_attention_MPNet(model_params=model_params)
        self.num_enc_steps = model_params['num_enc_steps']
        self.num_class_steps = model_params['num_class_steps']
        self.last_logits = None  # [max(L,1), E]: classifier output of every step (the mask branch's input)
        # inference only: keep the packed weight images between calls (see frozen_weights()); off by default, because an
        # in-place write through ``p.data`` / a raw pointer does not move ``p._version`` and would go unnoticed
        self.keep_packed_weights = False

    def _build_core_MPNet(self, model_params, encoder_feats_dict):
        """mpn.py:254-317."""
        node_agg_fn = model_params['node_agg_fn']
        assert node_agg_fn.lower() in ('mean', 'max', 'sum'), "node_agg_fn can only be 'max', 'mean' or 'sum'."
        node_agg_fn = NodeAggFn(node_agg_fn)
        self.reattach_initial_nodes = model_params['reattach_initial_nodes']
        self.reattach_initial_edges = model_params['reattach_initial_edges']
        self.edge_factor = 2 if self.reattach_initial_edges else 1
        self.node_factor = 2 if self.reattach_initial_nodes else 1
        edge_model_in_dim = self.node_factor * 2 * encoder_feats_dict['node_out_dim'] + \
            self.edge_factor * encoder_feats_dict['edge_out_dim']
        node_model_in_dim = self.node_factor * encoder_feats_dict['node_out_dim'] + encoder_feats_dict['edge_out_dim']
        edge_model_feats_dict = model_params['edge_model_feats_dict']
        node_model_feats_dict = model_params['node_model_feats_dict']
        edge_model = MLP(input_dim=edge_model_in_dim, fc_dims=edge_model_feats_dict['dims'],
                         dropout_p=edge_model_feats_dict['dropout_p'],
                         use_batchnorm=edge_model_feats_dict['use_batchnorm'])
        flow_in_model = MLP(input_dim=node_model_in_dim, fc_dims=node_model_feats_dict['dims'],
                            dropout_p=node_model_feats_dict['dropout_p'],
                            use_batchnorm=node_model_feats_dict['use_batchnorm'])
        flow_out_model = MLP(input_dim=node_model_in_dim, fc_dims=node_model_feats_dict['dims'],
                             dropout_p=node_model_feats_dict['dropout_p'],
                             use_batchnorm=node_model_feats_dict['use_batchnorm'])
        node_model = nn.Sequential(*[nn.Linear(2 * encoder_feats_dict['node_out_dim'],
                                               encoder_feats_dict['node_out_dim']), nn.ReLU(inplace=True)])
        return MetaLayer(edge_model=EdgeModel(edge_model=edge_model),
                         node_model=TimeAwareNodeModel(flow_in_model=flow_in_model, flow_out_model=flow_out_model,
                                                       node_model=node_model, node_agg_fn=node_agg_fn))

    def _build_attention_MPNet(self, model_params):
        """mpn.py:319-331 (the two attention MLPs the reference constructs there are never used nor registered)."""
        node_ext_model_feats_dict = model_params['node_ext_model_feats_dict']
        node_ext_model_in_dim = 3 * model_params['node_ext_encoder_feats_dict']['dims'][-1] * self.node_factor
        node_ext_model = CNN(input_dim=node_ext_model_in_dim, **node_ext_model_feats_dict)
        return TimeAwareAttentionModel(node_model=node_ext_model)

    # ------------------------------------------------------------------ native model description
    def hot_path_parameters(self):
        """Parameters of the hot path in ``state_dict`` order (what the native backward fills)."""
        mods = [self.encoder.node_model, self.encoder.edge_model, self.MPNet.edge_model.edge_model,
                self.MPNet.node_model.flow_in_model, self.MPNet.node_model.flow_out_model]
        out = []
        for m in mods:
            for l in m.linears():
                out += [l.weight, l.bias]
        lin = self.MPNet.node_model.node_model[0]
        out += [lin.weight, lin.bias]
        for l in self.classifier.edge_model.linears():
            out += [l.weight, l.bias]
        return out

    def _hot_path_mlps(self):
        return [self.encoder.node_model, self.encoder.edge_model, self.MPNet.edge_model.edge_model,
                self.MPNet.node_model.flow_in_model, self.MPNet.node_model.flow_out_model, self.classifier.edge_model]

    def c_model(self, keep, grads=None, n_edges=None):
        if self.encoder.node_model is None or self.encoder.edge_model is None or self.classifier.edge_model is None:
            raise capi.MpnhipError("MOTMPNet needs node and edge encoders and an edge classifier")
        m = self.MPNet.core_struct(keep, grads=grads)
        m.reattach_nodes = int(bool(self.reattach_initial_nodes))
        m.reattach_edges = int(bool(self.reattach_initial_edges))
        m.num_enc_steps = int(self.num_enc_steps)
        # operand precision of the Linear products (include/mpnhip.h MPNHIP_PREC_*): 'fp32' (fp32 MFMAs), 'fp32_split'
        # (fp32 results from three-piece bf16 operands in the fused chain kernels: same accuracy, fewer MFMA cycles) or
        # 'bf16' (operands rounded to bf16, fp32 accumulation -- BASELINE.json's "bf16 MLP GEMMs" mode; inference and training)
        m.precision = capi.PRECISIONS[self.operand_precision(n_edges)]
        m.enc_node = self.encoder.node_model.c_struct(keep, grads)
        m.enc_edge = self.encoder.edge_model.c_struct(keep, grads)
        m.classifier = self.classifier.edge_model.c_struct(keep, grads)
        return m

    # Operand precision of the Linear products: 'auto' (default), 'fp32', 'fp32_split', 'bf16' (see c_model / include/mpnhip.h)
    gemm_precision = 'auto'

    AUTO_SPLIT_MIN_HIDDEN = 256     # first hidden width of the edge MLP from which 'auto' means 'fp32_split' whatever the graph
    AUTO_SPLIT_MIN_EDGES = 32768    # ... and the graph size from which it does at narrower widths

    def operand_precision(self, n_edges=None):
        """``gemm_precision`` with 'auto' resolved for a graph of ``n_edges`` edges (None: unknown).  'auto' = 'fp32_split' where
        the fused chain kernels have enough MFMA work for the cheaper matrix instruction to show -- the 128-d class of BASELINE.json's
        configs[1] at any size (cfg-B training step 6.6 -> 5.7 ms, inference 2.05 -> 1.53 ms), the reference's 32-d widths from
        ~32k edges (cfg-C stand-in, 77.8k edges: 2.15 -> 2.07 ms / 0.47 -> 0.43 ms) -- and 'fp32' (fp32 MFMAs) on small graphs at
        narrow widths, where a launch is one wave per SIMD and the operand splitting sits on its critical path (cfg-D stand-in
        forward, 14.4k edges: 0.112 ms against 0.143) -- as 'fp32_wgsplit': fp32 MFMAs in the forward and the activation-gradient
        chain, the batched row-panel kernel (split operands) for the weight gradients, which is where a small graph's training step
        spends its launches.  Logits and every gradient are as close to a float64 oracle in one mode as
        in the other.  The forward and the backward of one call see the same edge count, hence one mode."""
        prec = getattr(self, 'gemm_precision', 'auto')
        if prec == 'auto':
            try:
                he = int(self.MPNet.edge_model.edge_model.linears()[0].weight.shape[0])
            except Exception:
                he = 0
            wide = he >= self.AUTO_SPLIT_MIN_HIDDEN
            big = n_edges is not None and int(n_edges) >= self.AUTO_SPLIT_MIN_EDGES
            prec = 'fp32_split' if (wide or big) else 'fp32_wgsplit'
        if prec not in capi.PRECISIONS:
            raise capi.MpnhipError("gemm_precision must be 'auto' or one of %s, not %r" % (sorted(capi.PRECISIONS), prec))
        return prec

    @contextlib.contextmanager
    def frozen_weights(self):
        """``with model.frozen_weights():`` -- the caller vouches that no hot-path weight changes inside the block (e.g. one
        sequence of sliding-window inference, ``tracker.evaluate_graph_in_batches``).  Inference calls inside it pack the
        weight images (about 20 small launches) once and reuse them (``mpnhip_model.weights_prepacked``); outside such a
        block every call packs again, which is always safe.  Changes torch can see (``p._version``, ``load_state_dict``,
        the native Adam step) still invalidate the images inside the block; writes through ``p.data`` / raw pointers do
        NOT -- call ``invalidate_packed_weights()`` after those, or do them outside the block."""
        old = self.keep_packed_weights
        self.keep_packed_weights = True
        try:
            yield self
        finally:
            self.keep_packed_weights = old
            if not old:
                self.invalidate_packed_weights()

    def invalidate_packed_weights(self):
        """Forget every packed weight image (the next inference call packs again)."""
        capi._packed_state.clear()

    def _hp_params(self):
        """hot_path_parameters(), built once (the module structure does not change after __init__)."""
        c = getattr(self, '_hp_params_cache', None)
        if c is None:
            c = self._hp_params_cache = self.hot_path_parameters()
        return c

    def _packed_key(self, precision_code):
        """What the packed weight images depend on: precision, native-update epoch, every weight's address and version."""
        return (precision_code, capi._weights_epoch[0]) + tuple((p_.data_ptr(), p_._version) for p_ in self._hp_params())

    def _uid(self):
        # (a token that is never reused: id() of a collected model can come back, together with recycled parameter
        # addresses and equal version counts, for a model with other weights)
        if getattr(self, '_mpnhip_uid', None) is None:
            capi._model_uid[0] += 1
            self._mpnhip_uid = capi._model_uid[0]
        return self._mpnhip_uid

    def hot_path(self, x, edge_index, edge_attr, holder=None, return_state=False, validate=True):
        """Encoder + L message-passing steps + per-step classifier: logits [max(L,1), E].  ``validate``: raise IndexError
        like the reference when edge_index leaves [0, N) (read once per prepared graph, after the launch; callers that build
        the indices themselves -- the sliding-window driver -- skip it)."""
        capi.require_device(x, edge_index, edge_attr)
        slow = [m_ for m_ in self._hot_path_mlps() if not m_.fast_path]
        if slow and (any(m_.training for m_ in slow) or (torch.is_grad_enabled() and (
                x.requires_grad or edge_attr.requires_grad or any(p.requires_grad for p in self._hp_params())))):
            # BatchNorm / Dropout (mlp.py:14,20) in TRAINING mode -- batch statistics rule the fused kernels out -- or their
            # gradients in eval mode (BatchNorm as the affine map of its running statistics): layer by layer
            from . import modular
            return modular.hot_path(self, x, edge_index, edge_attr, holder=holder, return_state=return_state)
        if torch.is_grad_enabled() and (x.requires_grad or edge_attr.requires_grad or
                                        any(p.requires_grad for p in self._hp_params())):
            from .autograd import mpn_hot_path_autograd
            return mpn_hot_path_autograd(self, x, edge_index, edge_attr, holder, validate=validate)
        from . import torch_ops
        if torch_ops.available() and not return_state:
            return self._hot_path_ops(x, edge_index, edge_attr, holder, validate, torch_ops)
        lib = capi.load()
        keep = []
        x = capi.f32c(x)
        ea = capi.f32c(edge_attr)
        N, E = x.shape[0], ea.shape[0]
        m = self.c_model(keep, n_edges=E)
        g = _prepared(edge_index, N, holder, full=False)   # inference: the primary order is all mpnhip_forward reads
        check_hot_path_inputs(m, g, x, ea)
        L = max(int(self.num_enc_steps), 1)
        logits = torch.empty((L, E), dtype=torch.float32, device=x.device)
        x_out = torch.empty((N, m.dn), dtype=torch.float32, device=x.device) if return_state else None
        e_out = torch.empty((E, m.de), dtype=torch.float32, device=x.device) if return_state else None
        with torch.cuda.device(x.device):
            ws = capi.workspace(lib.mpnhip_forward_workspace_bytes(m, N, E, 0), x.device, "fwd")
            # inside ``frozen_weights()`` the packed weight images at the head of the workspace survive between calls: skip
            # re-packing them while the buffer, the model and every weight (address, torch version, native-update epoch) are
            # unchanged
            state = (self._uid(), self._packed_key(m.precision)) if self.keep_packed_weights else None
            m.weights_prepacked = 1 if state is not None and capi._packed_state.get(ws.data_ptr()) == state else 0
            capi._packed_state.pop(ws.data_ptr(), None)
            capi.check(lib.mpnhip_forward(m, capi.ptr(g.buf), N, E, capi.ptr(x), capi.ptr(ea), capi.ptr(logits),
                                          capi.ptr(x_out), capi.ptr(e_out), capi.ptr(ws), ws.numel(), 0,
                                          capi.stream_ptr()), "mpnhip_forward")
            if state is not None:
                capi._packed_state[ws.data_ptr()] = state
        if validate:
            g.raise_if_invalid()
        if return_state:
            return logits, x_out, e_out
        return logits

    def _hot_path_ops(self, x, edge_index, edge_attr, holder, validate, torch_ops):
        """The inference hot path through the dispatcher: ``torch.ops.mpnhip.forward`` (csrc/torch_ops.cpp) -- the same C-ABI call;
        the op owns the output allocation and the workspace (kept here per stream so that packed weight images can survive
        between calls inside ``frozen_weights()``), and the model crosses as the cached (spec, weights) pair."""
        c = getattr(self, '_ops_cache', None)
        ptrs = tuple(p_.data_ptr() for p_ in self._hp_params())
        prec = self.operand_precision(edge_attr.shape[0])
        folded = not all(m_.fast_path for m_ in self.modules() if isinstance(m_, MLP))   # BatchNorm: fold afresh every call
        if c is None or c[0] != ptrs or c[1] != prec or c[2] != int(self.num_enc_steps) or folded:
            spec, weights = torch_ops.model_spec(self, n_edges=edge_attr.shape[0])
            c = self._ops_cache = (ptrs, prec, int(self.num_enc_steps), spec, weights, spec[8], spec[9 + spec[7] + 1])
        spec, weights = c[3], c[4]
        x = capi.f32c(x)
        ea = capi.f32c(edge_attr)
        g = _prepared(edge_index, x.shape[0], holder, full=False)
        if g.E != ea.shape[0] or x.dim() != 2 or ea.dim() != 2 or x.shape[1] != c[5] or ea.shape[1] != c[6] or x.device != g.device \
                or ea.device != g.device:
            check_hot_path_inputs(self.c_model([]), g, x, ea)   # (raises with the detailed message)
        with torch.cuda.device(x.device):
            skey = int(torch.cuda.current_stream().cuda_stream)
            wss = self.__dict__.setdefault('_ops_ws', {})
            ws, ws_state = wss.get(skey, (None, None))
            state = (self._uid(), self._packed_key(spec[6])) if self.keep_packed_weights else None
            prepacked = state is not None and ws is not None and ws_state == state and capi._packed_state.get(ws.data_ptr()) == state
            logits, ws_out = torch_ops.call("forward", g.buf, x, ea, weights, spec, 0, ws, prepacked)
            if state is not None:
                capi._packed_state[ws_out.data_ptr()] = state
            else:
                capi._packed_state.pop(ws_out.data_ptr(), None)
            wss[skey] = (ws_out, state)
        if validate:
            g.raise_if_invalid()
        return logits

    def forward(self, data):
        """mpn.py:333-394 (hot path; see the class docstring for the mask branch)."""
        x, edge_index, edge_attr = data.x, data.edge_index, data.edge_attr
        if x.dim() == 4:
            # global_avgpool + view (mpn.py:351-352)
            x = avg_pool(x)
        logits = self.hot_path(x, edge_index, edge_attr, holder=data)
        self.last_logits = logits
        E = logits.shape[1]
        L, k = int(self.num_enc_steps), int(self.num_class_steps)
        x_ext = getattr(data, 'x_ext', None)
        mask_branch = self.has_mask_branch and x_ext is not None
        outputs_dict = {'classified_edges': [], 'mask_predictions': []}
        if mask_branch:
            latent_node_ext_feats = self.node_ext_encoder(x_ext)                       # mpn.py:356
            initial_node_ext_feats = latent_node_ext_feats
        first_class_step = L - k + 1
        for step in range(1, L + 1):
            if mask_branch:
                if self.reattach_initial_nodes:                                         # mpn.py:373
                    latent_node_ext_feats = torch.cat((initial_node_ext_feats, latent_node_ext_feats), dim=1)
                latent_node_ext_feats = self.MPAttentionNet.aggregate(latent_node_ext_feats, edge_index,
                                                                      logits[step - 1], holder=data)   # mpn.py:377
            if step >= first_class_step:
                outputs_dict['classified_edges'].append(logits[step - 1].view(E, 1))
                if mask_branch:
                    outputs_dict['mask_predictions'].append(self.mask_predictor(x_ext, latent_node_ext_feats))
        if L == 0:
            outputs_dict['classified_edges'].append(logits[0].view(E, 1))
            if mask_branch:
                outputs_dict['mask_predictions'].append(self.mask_predictor(x_ext, latent_node_ext_feats))
        return outputs_dict


def avg_pool(x):
    """nn.AdaptiveAvgPool2d((1,1)) + view (mpn.py:252,351-352): [N,C,H,W] -> [N,C]."""
    capi.require_device(x)
    from .autograd import avg_pool_native
    return avg_pool_native(x)
