"""Shared comparison helpers of the parity tests (no GPU code here).

Gradient criteria of an UNPINNED comparison (HIP fp32 against another fp32 / float64 evaluation of the same network):
  strict : max |d| <= tol * max |ref| over the tensor (tol = 2e-4: fp32 re-association over up to 50k-term sums) -- holds on
           the small graphs and wherever no ReLU / arg-max decision sits within rounding noise of its boundary;
  robust : rows within tol >= 99.9 % and relative L2 < 1e-3 -- the bound VERDICT r01 asked for.  Measured in round 2
           (profiles/r02/grad_seed_sweep.txt): it is NOT attainable in general, not even by the fp32 oracle against the float64
           oracle -- on 2 of 6 cfg-A graphs with sum aggregation the fp32 ORACLE's gradients are off by 5e-4 ... 7e-3 relative
           L2 in every tensor, because one pre-activation on a dominant path sits within fp32 noise of zero and the
           piecewise-linear function is differentiated on the neighbouring branch.  Every weight-gradient row sums over all
           edges, so a single such decision moves ALL rows.
  loose  : relative L2 < 1e-2: the sanity bound kept for unpinned comparisons at sizes where such decisions occur.
The sharp statement at those sizes is the decision-pinned one (tests/pinned.py, tests/test_gpu_pinned.py): decisions agree
up to knife-edge units AND, on the branch taken, every gradient agrees to 2e-5.
All helpers print their statistics so that a pass can be audited (pytest -s / on failure).
"""
import numpy as np

GTOL = 2e-4
ROW_FRACTION = 0.999
REL_L2 = 1e-3
LOOSE_REL_L2 = 1e-2


def nerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if b.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-30))


def grad_stats(a, b, tol=GTOL):
    """(max error / max |ref|, relative L2 error, fraction of rows whose max error is within tol * max |ref|, rows)"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if b.size == 0:
        return 0.0, 0.0, 1.0, 0
    scale = max(float(np.abs(b).max()), 1e-30)
    d = np.abs(a - b)
    rows = d.reshape(d.shape[0], -1).max(axis=1) if d.ndim >= 2 else d.reshape(-1)
    ok = float((rows <= tol * scale).mean())
    rel_l2 = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
    return float(d.max() / scale), rel_l2, ok, int(rows.size)


def grad_close(a, b, tol=GTOL, robust=False, name="", log=None):
    """robust: False = strict, True = strict or (rows, rel-L2) bound, 'loose' = strict or rel-L2 < 1e-2"""
    mx, l2, frac, rows = grad_stats(a, b, tol)
    msg = "%s: max %.3g  rel_l2 %.3g  rows within %.0e: %.4f%% of %d" % (name, mx, l2, tol, 100.0 * frac, rows)
    if log is not None:
        log.append(msg)
    good = mx <= tol or (robust is True and frac >= ROW_FRACTION and l2 < REL_L2) or (robust == "loose" and l2 < LOOSE_REL_L2)
    return good, msg


def logits_close_abs(got, ref, tol=1e-4):
    """mean / max aggregation: absolute 1e-4."""
    d = float(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).max()) if np.size(ref) else 0.0
    return d <= tol, d


def logits_close_per_element(got, ref, tol=1e-4):
    """sum aggregation with O(1) logits: |d_e| <= tol * max(1, |ref_e|) for EVERY edge (not relative to the step's maximum)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    if ref.size == 0:
        return True, 0.0
    q = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
    return bool((q <= tol).all()), float(q.max())


def fixture_param_view(a, z, key):
    """The part of gradient `a` that fixture `z` stores under G:key (whole tensor, or its first 20,000 elements)."""
    ref = z["G:" + key]
    a = np.asarray(a)
    if ref.shape == a.shape:
        return a, ref
    return a.reshape(-1)[: ref.size], ref


def check_grads_against_fixture(z, pg, gx, gea, tol=GTOL, robust=False):
    """pg {state_dict key: gradient}, gx, gea against a g11 / g12 style fixture.  Returns (list of failures, log lines)."""
    log, bad = [], []
    for k in [f[2:] for f in z.files if f.startswith("G:")]:
        a, ref = fixture_param_view(pg[k], z, k)
        good, msg = grad_close(a, ref, tol, robust, k, log)
        # whole-tensor norm (covers the elements a sampled fixture does not store)
        n = float(np.sqrt((np.asarray(pg[k], np.float64) ** 2).sum()))
        nref = float(z["Gn:" + k])
        if abs(n - nref) > (1e-2 if robust == "loose" else 1e-3) * max(nref, 1e-30):
            good = False
            msg += "  NORM %.6g vs %.6g" % (n, nref)
        if not good:
            bad.append(msg)
    ref_x = z["grad_x"]
    good, msg = grad_close(np.asarray(gx)[: ref_x.shape[0]], ref_x, tol, robust, "grad_x", log)
    rn = np.sqrt((np.asarray(gx, np.float64) ** 2).sum(1))
    good2, msg2 = grad_close(rn, z["grad_x_rownorm"], 1e-3, robust, "grad_x row norms", log)
    if robust == "loose":
        good2 = True
    if not good:
        bad.append(msg)
    if not good2:
        bad.append(msg2)
    good, msg = grad_close(np.asarray(gea)[z["edge_ids"]], z["grad_edge_attr"], tol, robust, "grad_edge_attr", log)
    if not good:
        bad.append(msg)
    n = float(np.sqrt((np.asarray(gea, np.float64) ** 2).sum()))
    if abs(n - float(z["grad_edge_attr_norm"])) > (1e-2 if robust == "loose" else 1e-3) * max(float(z["grad_edge_attr_norm"]), 1e-30):
        bad.append("grad_edge_attr NORM %.6g vs %.6g" % (n, float(z["grad_edge_attr_norm"])))
    return bad, log
