"""CPU ORACLE for the MPNTrackSeg message-passing hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU restatement of the reference algorithm (same op order as the
reference: cat -> Linear -> ReLU, boolean-mask split, scatter).  It is *not* part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the timed CPU baseline.  ``mpntrackseg_amd`` never imports it.

Parity pin: the reference has no tests or golden vectors for this path, so the
pin is the reference itself, imported in the authoring container by ``tools/make_golden.py``
(with a ``torch_scatter`` shim) to write ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks this restatement against those fixtures.  On the build container the restatement agrees
with the imported reference bit-for-bit (same torch CPU kernels, same op order).

Third-party arithmetic restated here: torch-scatter 2.0.4 (``environment.yml:146``), which is not
vendored in /root/reference -- ``scatter_add`` = zeros.scatter_add_, ``scatter_mean`` =
scatter_add / clamp(count, min=1), ``scatter_max`` fills empty segments with 0.

Each function cites the reference lines it follows (paths relative to
/root/reference/src/mot_neural_solver/).
"""
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- torch_scatter 2.0.4
def scatter_add(src, index, dim_size):
    """torch_scatter.scatter_add(src, index, dim=0, dim_size) (call site models/mpn.py:273)."""
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    if src.numel():
        out.scatter_add_(0, index.view(-1, *([1] * (src.dim() - 1))).expand_as(src), src)
    return out


def scatter_mean(src, index, dim_size):
    """torch_scatter.scatter_mean (call site models/mpn.py:267): sum / clamp(count, min=1)."""
    out = scatter_add(src, index, dim_size)
    cnt = torch.zeros(dim_size, dtype=src.dtype)
    if index.numel():
        cnt.scatter_add_(0, index, torch.ones(index.shape[0], dtype=src.dtype))
    return out / cnt.clamp(min=1).view(-1, *([1] * (src.dim() - 1)))


def scatter_max(src, index, dim_size):
    """torch_scatter.scatter_max(...)[0] (call site models/mpn.py:270): empty segments -> 0."""
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    if src.numel():
        idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
        out.scatter_reduce_(0, idx, src, reduce="amax", include_self=False)
    return out


AGG = {"sum": scatter_add, "mean": scatter_mean, "max": scatter_max}


# ----------------------------------------------------------------------------- models/mlp.py:4-28
# Operand precision of every Linear product (mpnhip_model.precision): "fp32" = the reference's arithmetic; "bf16" =
# activations and weights rounded to bfloat16 (round to nearest even) as they enter the product, fp32 accumulation and
# fp32 bias -- the restatement of BASELINE.json's "bf16 MLP GEMMs" configuration ("compare
# against the CPU restatement with inputs/weights rounded to bf16 and fp32 accumulation").  Set with `precision(...)`.
_PRECISION = ["fp32"]


class precision:
    """``with oracle.precision("bf16"): ...`` -- scoped switch of the Linear operand precision."""
    def __init__(self, p):
        assert p in ("fp32", "bf16")
        self.p = p

    def __enter__(self):
        self.old = _PRECISION[0]
        _PRECISION[0] = self.p

    def __exit__(self, *a):
        _PRECISION[0] = self.old


def linear(x, w, b):
    if _PRECISION[0] == "bf16":
        x, w = x.bfloat16().float(), w.bfloat16().float()
    return F.linear(x, w, b)


def mlp(x, W, prefix):
    """MLP.forward (models/mlp.py:27-28) for dropout_p=0, use_batchnorm=False: Linear (+ReLU
    unless the layer's out-dim is 1, mlp.py:17).  ``W`` maps state_dict keys to tensors."""
    i = 0
    while f"{prefix}.fc_layers.{i}.weight" in W:
        w, b = W[f"{prefix}.fc_layers.{i}.weight"], W[f"{prefix}.fc_layers.{i}.bias"]
        x = linear(x, w, b)
        if w.shape[0] != 1:
            x = torch.relu(x)
            i += 2
        else:
            i += 1
    return x


# ----------------------------------------------------------------------------- models/mpn.py:59-69
def edge_model(x, edge_index, e, W):
    """EdgeModel.forward (models/mpn.py:67-69)."""
    row, col = edge_index
    return mlp(torch.cat([x[row], x[col], e], dim=1), W, "MPNet.edge_model.edge_model")


# ----------------------------------------------------------------------------- models/mpn.py:71-99
def node_model(x, edge_index, e, W, agg):
    """TimeAwareNodeModel.forward (models/mpn.py:83-99)."""
    row, col = edge_index
    n = x.size(0)
    out_mask = row < col                                                     # :85
    out_in = torch.cat([x[col[out_mask]], e[out_mask]], dim=1)               # :86-87
    flow_out = AGG[agg](mlp(out_in, W, "MPNet.node_model.flow_out_model"), row[out_mask], n)   # :88-89
    in_mask = row > col                                                      # :91
    in_in = torch.cat([x[col[in_mask]], e[in_mask]], dim=1)                  # :92-93
    flow_in = AGG[agg](mlp(in_in, W, "MPNet.node_model.flow_in_model"), row[in_mask], n)       # :94-96
    flow = torch.cat((flow_in, flow_out), dim=1)                             # :97
    return torch.relu(linear(flow, W["MPNet.node_model.node_model.0.weight"],
                               W["MPNet.node_model.node_model.0.bias"]))    # :99, :309-310


def meta_layer(x, edge_index, e, W, agg):
    """MetaLayer.forward (models/mpn.py:33-54): edge update, then node update on the NEW edges."""
    e = edge_model(x, edge_index, e, W)
    x = node_model(x, edge_index, e, W, agg)
    return x, e


def classify(e, W):
    """classifier MLPGraphIndependent -> edge MLP only (models/mpn.py:114, :164-178, :238)."""
    return mlp(e, W, "classifier.edge_model")


# ----------------------------------------------------------------------------- models/mpn.py:333-394
def forward(params, W, x, edge_index, edge_attr, return_state=False):
    """MOTMPNet.forward restricted to the hot path (models/mpn.py:349-392 minus the x_ext lines).

    x: [N, node_in_dim] (already pooled) or [N, C, H, W] (pooled here, :351-352).
    Returns the reference's ``classified_edges`` list (k = num_class_steps tensors [E,1]); with
    return_state also every step's logits and the final latent node / edge features."""
    agg = params["node_agg_fn"]
    L, k = params["num_enc_steps"], params["num_class_steps"]
    if x.dim() == 4:
        x = x.mean(dim=(2, 3))                                               # :351-352
    e = mlp(edge_attr, W, "encoder.edge_model")                              # :355
    x = mlp(x, W, "encoder.node_model")
    e0, x0 = e, x                                                            # :358-359
    first_class_step = L - k + 1                                             # :364
    classified, all_logits = [], []
    for step in range(1, L + 1):                                             # :366
        if params["reattach_initial_edges"]:
            e = torch.cat((e0, e), dim=1)                                    # :370
        if params["reattach_initial_nodes"]:
            x = torch.cat((x0, x), dim=1)                                    # :372
        x, e = meta_layer(x, edge_index, e, W, agg)                          # :376
        dec = classify(e, W)                                                 # :377 -> :114
        all_logits.append(dec)
        if step >= first_class_step:                                         # :379-381
            classified.append(dec)
    if L == 0:                                                               # :387-389
        dec = classify(e, W)
        classified.append(dec)
        all_logits.append(dec)
    if return_state:
        return classified, all_logits, x, e
    return classified


def to_tensors(W, dtype=torch.float32, requires_grad=False):
    out = {}
    for k, v in W.items():
        t = torch.as_tensor(v).to(dtype).clone()
        t.requires_grad_(requires_grad)
        out[k] = t
    return out
