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


# ----------------------------------------------------------------------------- decision pinning (parity tests)
# ReLU and arg-max make the path piecewise linear: its gradient jumps where a pre-activation crosses zero (or two messages
# tie), so two fp32 evaluations that differ by rounding noise can legitimately sit on different branches and disagree in the
# gradient by far more than rounding noise.  The tests therefore take the DECISIONS from the implementation under test:
#   compare: run the oracle normally and count, per site, the units whose decision differs from the given one together with
#            how far from the boundary they are (|z| / rms(z)): a correct forward disagrees only on knife-edge units;
#   impose : relu(z) := z * given_mask, max := gather at the given arg-max -- the oracle then differentiates exactly the branch
#            the implementation took, and the gradients must agree to rounding noise.
# Sites: "enc_e.<i>", "enc_n.<i>", "s<step>.edge.<i>", "s<step>.flow.<i>" (both directions, full [E, w] arrays),
# "s<step>.cls.<i>", "s<step>.node", and "s<step>.argmax" ([N, 2 dn] original edge ids, -1 = empty; [flow_in | flow_out]).
class Decisions:
    # "record" (tests only): run normally and KEEP every site's own ReLU decisions (bit-packed) together with the pre-activations
    # that lie within NEAR x rms of zero -- everything a later compare_recorded(given) needs, so that one free-running float64
    # forward serves the comparison of several HIP runs (the fp32 / fp32_split / fp32_wgsplit variants of one test case)
    NEAR = 1e-3

    def __init__(self, given, mode):
        assert mode in ("compare", "impose", "record")
        self.given, self.mode = given, mode
        self.units = 0
        self.mismatches = 0
        self.worst_margin = 0.0      # largest |z| / rms(z) over the mismatching units
        self.per_site = {}
        self.recorded = []           # record mode: (site, rows, packed own decisions, shape, rms, near indices, near |z|)

    def relu(self, z, site, rows=None):
        if self.mode == "record":
            import numpy as np
            zd = z.detach()
            scale = float(zd.double().pow(2).mean().sqrt())
            flat = zd.reshape(-1)
            near = torch.nonzero(flat.abs() < self.NEAR * max(scale, 1e-300)).view(-1)
            self.recorded.append((site, None if rows is None else rows.clone(), np.packbits((zd > 0).numpy().reshape(-1)), tuple(zd.shape),
                                  scale, near.numpy(), flat[near].abs().double().numpy()))
            return torch.relu(z)
        mask = self.given[site]
        if rows is not None:
            mask = mask[rows]
        if self.mode == "impose":
            return z * mask.to(z.dtype)
        own = z > 0
        bad = own != mask
        nbad = int(bad.sum())
        self.units += z.numel()
        if nbad:
            scale = float(z.detach().double().pow(2).mean().sqrt())
            margin = float(z.detach()[bad].abs().max()) / max(scale, 1e-300)
            self.mismatches += nbad
            self.worst_margin = max(self.worst_margin, margin)
            self.per_site[site] = self.per_site.get(site, 0) + nbad
        return torch.relu(z)

    def compare_recorded(self, given):
        """The statistics a "compare" run with `given` would have produced, from a "record" run's data (ReLU sites)."""
        import numpy as np
        out = Decisions(given, "compare")
        for site, rows, packed, shape, scale, near_idx, near_abs in self.recorded:
            mask = given[site]
            if rows is not None:
                mask = mask[rows]
            n = int(np.prod(shape))
            own = np.unpackbits(packed, count=n).astype(bool)
            bad = np.nonzero(own != mask.numpy().reshape(-1))[0]
            out.units += n
            if bad.size:
                if len(near_idx):
                    pos = np.minimum(np.searchsorted(near_idx, bad), len(near_idx) - 1)
                    found = near_idx[pos] == bad
                else:
                    pos, found = np.zeros(bad.shape, np.int64), np.zeros(bad.shape, bool)
                # a differing unit outside the recorded band is at least NEAR x rms from zero: far beyond any accepted margin
                margin = 0.0 if bool(np.all(found)) else self.NEAR
                if bool(np.any(found)):
                    margin = max(margin, float(near_abs[pos[found]].max()) / max(scale, 1e-300))
                out.mismatches += int(bad.size)
                out.worst_margin = max(out.worst_margin, margin)
                out.per_site[site] = out.per_site.get(site, 0) + int(bad.size)
        return out


_DECISIONS = [None]


class decisions:
    """``with oracle.decisions(Decisions(given, mode)) as d: forward(...)``"""
    def __init__(self, d):
        self.d = d

    def __enter__(self):
        self.old = _DECISIONS[0]
        _DECISIONS[0] = self.d
        return self.d

    def __exit__(self, *a):
        _DECISIONS[0] = self.old


def relu(z, site=None, rows=None):
    d = _DECISIONS[0]
    if d is None or site is None:
        return torch.relu(z)
    return d.relu(z, site, rows)


def linear(x, w, b):
    if _PRECISION[0] == "bf16":
        x, w = x.bfloat16().float(), w.bfloat16().float()
    return F.linear(x, w, b)


def mlp(x, W, prefix, site=None, rows=None):
    """MLP.forward (models/mlp.py:27-28) for dropout_p=0, use_batchnorm=False: Linear (+ReLU
    unless the layer's out-dim is 1, mlp.py:17).  ``W`` maps state_dict keys to tensors.
    ``site`` / ``rows``: decision-pinning label of this module and the subset of edges it runs on (tests only)."""
    i, layer = 0, 0
    while f"{prefix}.fc_layers.{i}.weight" in W:
        w, b = W[f"{prefix}.fc_layers.{i}.weight"], W[f"{prefix}.fc_layers.{i}.bias"]
        x = linear(x, w, b)
        if w.shape[0] != 1:
            x = relu(x, None if site is None else f"{site}.{layer}", rows)
            i += 2
        else:
            i += 1
        layer += 1
    return x


# ----------------------------------------------------------------------------- models/mpn.py:59-69
def edge_model(x, edge_index, e, W, site=None):
    """EdgeModel.forward (models/mpn.py:67-69)."""
    row, col = edge_index
    return mlp(torch.cat([x[row], x[col], e], dim=1), W, "MPNet.edge_model.edge_model", None if site is None else site + ".edge")


def _aggregate(agg, msg, rows_mask, row, n, site, half):
    """node_agg_fn on one direction's messages; with imposed decisions the max becomes a gather at the given arg-max."""
    d = _DECISIONS[0]
    if agg != "max" or d is None or site is None:
        return AGG[agg](msg, row[rows_mask], n)
    dn = msg.shape[1]
    arg = d.given[site + ".argmax"][:, half * dn:(half + 1) * dn]        # [N, dn] original edge ids, -1 = empty segment
    full = torch.zeros((rows_mask.shape[0], dn), dtype=msg.dtype).index_put((torch.nonzero(rows_mask).view(-1),), msg)
    picked = full.gather(0, arg.clamp(min=0)) * (arg >= 0).to(msg.dtype)
    if d.mode == "impose":
        return picked
    own = AGG[agg](msg, row[rows_mask], n)
    bad = own != picked.detach()
    d.units += own.numel()
    if bool(bad.any()):
        scale = float(own.detach().double().pow(2).mean().sqrt())
        d.mismatches += int(bad.sum())
        d.worst_margin = max(d.worst_margin, float((own.detach() - picked.detach())[bad].abs().max()) / max(scale, 1e-300))
        d.per_site[site + ".argmax"] = d.per_site.get(site + ".argmax", 0) + int(bad.sum())
    return own


# ----------------------------------------------------------------------------- models/mpn.py:71-99
def node_model(x, edge_index, e, W, agg, site=None):
    """TimeAwareNodeModel.forward (models/mpn.py:83-99)."""
    row, col = edge_index
    n = x.size(0)
    fsite = None if site is None else site + ".flow"
    out_mask = row < col                                                     # :85
    out_in = torch.cat([x[col[out_mask]], e[out_mask]], dim=1)               # :86-87
    flow_out = _aggregate(agg, mlp(out_in, W, "MPNet.node_model.flow_out_model", fsite, out_mask), out_mask, row, n, site, 1)   # :88-89
    in_mask = row > col                                                      # :91
    in_in = torch.cat([x[col[in_mask]], e[in_mask]], dim=1)                  # :92-93
    flow_in = _aggregate(agg, mlp(in_in, W, "MPNet.node_model.flow_in_model", fsite, in_mask), in_mask, row, n, site, 0)        # :94-96
    flow = torch.cat((flow_in, flow_out), dim=1)                             # :97
    return relu(linear(flow, W["MPNet.node_model.node_model.0.weight"],
                       W["MPNet.node_model.node_model.0.bias"]), None if site is None else site + ".node")   # :99, :309-310


def meta_layer(x, edge_index, e, W, agg, site=None):
    """MetaLayer.forward (models/mpn.py:33-54): edge update, then node update on the NEW edges."""
    e = edge_model(x, edge_index, e, W, site)
    x = node_model(x, edge_index, e, W, agg, site)
    return x, e


def classify(e, W, site=None):
    """classifier MLPGraphIndependent -> edge MLP only (models/mpn.py:114, :164-178, :238)."""
    return mlp(e, W, "classifier.edge_model", None if site is None else site + ".cls")


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
    e = mlp(edge_attr, W, "encoder.edge_model", "enc_e")                     # :355
    x = mlp(x, W, "encoder.node_model", "enc_n")
    e0, x0 = e, x                                                            # :358-359
    first_class_step = L - k + 1                                             # :364
    classified, all_logits = [], []
    for step in range(1, L + 1):                                             # :366
        if params["reattach_initial_edges"]:
            e = torch.cat((e0, e), dim=1)                                    # :370
        if params["reattach_initial_nodes"]:
            x = torch.cat((x0, x), dim=1)                                    # :372
        x, e = meta_layer(x, edge_index, e, W, agg, f"s{step}")              # :376
        dec = classify(e, W, f"s{step}")                                     # :377 -> :114
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
