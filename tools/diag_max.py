import sys, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from mpntrackseg_amd import synth
import test_gpu_backward as T
c = synth.CONFIGS["A"]
for agg in ("max",):
    params = synth.model_params(c["d"], c["L"], agg)
    W = synth.make_weights(params, seed=7)
    g = synth.make_graph(c["N"], c["E"], seed=1)
    r = synth.normal(11, (c["L"], c["E"]))
    model = T.make_model(params, W)
    lo, gx, gea, pg = T.native_grads(model, g["x"], g["edge_index"], g["edge_attr"], r)
    lr, rx, rea, rpg = T.oracle_grads(params, W, g["x"], g["edge_index"], g["edge_attr"], r)
    def stat(name, a, b):
        a, b = a.astype(np.float64), b.astype(np.float64)
        sc = max(np.abs(b).max(), 1e-6)
        d = np.abs(a - b)
        print(name, "max", d.max() / sc, "frac_ok", (d <= 2e-4 * sc).mean(), "rel_l2", np.linalg.norm(a - b) / np.linalg.norm(b), "nbad", int((d > 2e-4 * sc).sum()))
    stat("logits", lo, lr)
    stat("gx", gx, rx)
    stat("gea", gea, rea)
    for k in W:
        stat(k, pg[k], rpg[k])
    # rows of gx that differ
    d = np.abs(gx - rx).max(1)
    print("bad node rows:", np.nonzero(d > 1e-3 * np.abs(rx).max())[0][:20])
