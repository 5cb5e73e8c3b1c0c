"""Native tracking loss / step metrics against the CPU restatement of
pl_module.py:88-107 and utils/evaluation.py:340-437 (oracle/loss_oracle.py) and the fixtures the REFERENCE functions themselves produced (tests/golden/g9_loss_metrics.npz)."""
import numpy as np
import pytest
import torch

from mpntrackseg_amd import synth
from mpntrackseg_amd.loss import compute_perform_metrics, tracking_loss, tracking_loss_and_grad
from oracle import loss_oracle as LO

pytestmark = pytest.mark.gpu
dev = lambda: torch.device("cuda:0")


@pytest.mark.parametrize("E,L,frac", [(1, 1, 1.0), (1000, 3, 0.2), (50000, 12, 0.02), (777, 4, 0.0)])
def test_tracking_loss_and_grad(E, L, frac):
    logits = torch.from_numpy(synth.normal(3, (L, E), std=3.0))
    labels = torch.from_numpy((synth.uniform01(4, E) < frac).astype(np.float32))
    lg = logits.clone().requires_grad_(True)
    ref = LO.tracking_loss([lg[s].view(E, 1) for s in range(L)], labels, weight=1.0)
    ref.backward()
    loss, grad = tracking_loss_and_grad(logits.to(dev()), labels.to(dev()), 0, 1.0)
    assert abs(float(loss[0]) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert float((grad.cpu() - lg.grad).abs().max()) <= 1e-6 * max(1.0, float(lg.grad.abs().max()))
    # autograd wrapper over the reference's list-of-[E,1] output
    lgd = logits.to(dev()).requires_grad_(True)
    out = tracking_loss([lgd[s].view(E, 1) for s in range(L)], labels.to(dev()))
    out.backward()
    assert float((lgd.grad.cpu() - lg.grad).abs().max()) <= 1e-6 * max(1.0, float(lg.grad.abs().max()))


def test_unclassified_steps_get_zero_gradient():
    E, L = 500, 6
    logits = torch.from_numpy(synth.normal(5, (L, E)))
    labels = torch.from_numpy((synth.uniform01(6, E) < 0.3).astype(np.float32))
    loss, grad = tracking_loss_and_grad(logits.to(dev()), labels.to(dev()), 2, 1.0)
    ref = LO.tracking_loss([logits[s].view(E, 1) for s in range(2, L)], labels)
    assert abs(float(loss[0]) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert float(grad[:2].abs().max()) == 0.0 and float(grad[2:].abs().max()) > 0.0


def test_step_metrics_match_reference_formulas():
    g = synth.batch_graphs([synth.make_graph(60, 400, T=6, seed=s, node_in_dim=4) for s in (1, 2)])
    ei = g["edge_index"].copy()
    ei[:, 3] = [7, 7]  # a self loop
    E, N = ei.shape[1], 120
    logit = torch.from_numpy(synth.normal(8, (E, 1)))
    labels = torch.from_numpy((synth.uniform01(9, E) < 0.25).astype(np.float32))

    class G:
        pass
    go = G()
    go.edge_index = torch.from_numpy(ei).to(dev())
    go.edge_labels = labels.to(dev())
    go.num_nodes = N
    got = compute_perform_metrics({"classified_edges": [logit.to(dev())]}, go)
    ref = LO.compute_perform_metrics([logit], torch.from_numpy(ei), labels, N)
    for k in ("accuracy", "recall", "precision", "constr_sr"):
        assert abs(got[k] - ref[k]) < 1e-6, k


def test_flat_adam_matches_torch_adam():
    """mpnhip_adam_step over a FlatBucket == torch.optim.Adam (pl_module.py:76-77 with configs/tracking_cfg.yaml:6-10)."""
    import torch
    from mpntrackseg_amd.train import FlatAdam, FlatBucket
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    shapes = [(37, 5), (128,), (16, 16), (1,)]
    for wd in (0.0, 1e-4):
        ref = [torch.randn(s, device=dev).requires_grad_(True) for s in shapes]
        mine = [r.detach().clone().requires_grad_(True) for r in ref]
        opt = torch.optim.Adam(ref, lr=1e-3, weight_decay=wd)
        bucket = FlatBucket(mine)
        fopt = FlatAdam(bucket, lr=1e-3, weight_decay=wd)
        for it in range(6):
            grads = [torch.randn(s, device=dev) * (1 + it) for s in shapes]
            for r, m, g in zip(ref, mine, grads):
                r.grad = g.clone()
                bucket.views[id(m)].copy_(g)
            opt.step()
            fopt.step()
            for r, m in zip(ref, mine):
                assert float((r.detach() - m.detach()).abs().max()) <= 2e-6 * max(1.0, float(r.detach().abs().max()))


def test_a_skipped_guarded_adam_step_does_not_count():
    """Guarded step with the device-side flag set: nothing changes and the call is NOT counted as an optimizer step -- the next
    good step is torch.optim.Adam's FIRST step (bias corrections of t = 1), the one after it the second."""
    import torch
    from mpntrackseg_amd.train import FlatAdam, FlatBucket
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    shapes = [(33, 7), (64,), (3, 3)]
    ref = [torch.randn(s, device=dev).requires_grad_(True) for s in shapes]
    mine = [r.detach().clone().requires_grad_(True) for r in ref]
    opt = torch.optim.Adam(ref, lr=1e-2, weight_decay=1e-4)
    bucket = FlatBucket(mine)
    fopt = FlatAdam(bucket, lr=1e-2, weight_decay=1e-4)
    before = bucket.flat_params.clone()
    for skip in (1.0, 1.0, 0.0, 0.0, 1.0, 0.0):
        grads = [torch.randn(s, device=dev) for s in shapes]
        for r, m, g in zip(ref, mine, grads):
            r.grad = g.clone()
            bucket.views[id(m)].copy_(g)
        bucket.flat[bucket.n:].fill_(skip)
        if skip == 0.0:
            opt.step()
        fopt.step(guarded=True)
        if skip and fopt.applied_steps == 0:
            assert torch.equal(before, bucket.flat_params)
        for r, m in zip(ref, mine):
            assert float((r.detach() - m.detach()).abs().max()) <= 2e-6 * max(1.0, float(r.detach().abs().max()))
    assert fopt.t == 6 and fopt.applied_steps == 3


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_tracking_loss_against_reference_compute_loss(golden, tag):
    """csrc/loss.hip against MOTNeuralSolver._compute_loss of the reference itself (tests/golden/g9_loss_metrics.npz: loss value
    and its autograd gradient w.r.t. every classified step; incl. the no-positive-label and the single-edge cases)."""
    z = golden("g9_loss_metrics.npz")
    logits = torch.from_numpy(z[f"{tag}:logits"]).to(dev())
    labels = torch.from_numpy(z[f"{tag}:labels"]).to(dev())
    w = float(z[f"{tag}:weight"])
    loss, grad = tracking_loss_and_grad(logits, labels, 0, w)
    ref = float(z[f"{tag}:loss"])
    assert abs(float(loss[0]) - ref) <= 1e-5 * max(1.0, abs(ref))
    assert float(np.abs(grad.cpu().numpy() - z[f"{tag}:grad"]).max()) <= 1e-6 * max(1.0, float(np.abs(z[f"{tag}:grad"]).max()))
    lg = logits.clone().requires_grad_(True)
    k, E = lg.shape
    out = tracking_loss([lg[s].view(E, 1) for s in range(k)], labels, weight=w)
    out.backward()
    assert abs(float(out) - ref) <= 1e-5 * max(1.0, abs(ref))
    assert float(np.abs(lg.grad.cpu().numpy() - z[f"{tag}:grad"]).max()) <= 1e-6 * max(1.0, float(np.abs(z[f"{tag}:grad"]).max()))


@pytest.mark.parametrize("tag", ["m1", "m2", "m3"])
def test_step_metrics_against_reference(golden, tag):
    """mpnhip_step_metrics against the reference's compute_perform_metrics / compute_constr_satisfaction_rate."""
    z = golden("g9_loss_metrics.npz")

    class G:
        pass
    go = G()
    go.edge_index = torch.from_numpy(z["edge_index"]).to(dev())
    go.edge_labels = torch.from_numpy(z[f"{tag}:labels"]).to(dev())
    go.num_nodes = 120
    got = compute_perform_metrics({"classified_edges": [torch.from_numpy(z[f"{tag}:logit"]).to(dev())]}, go)
    want = z[f"{tag}:metrics"]
    for i, k in enumerate(("accuracy", "recall", "precision", "constr_sr")):
        assert abs(got[k] - float(want[i])) < 1e-6, k


@pytest.mark.parametrize("tag", ["g3", "g8"])
def test_graph_batched_loss_against_the_reference_per_graph_loss(golden, tag):
    """mpnhip_tracking_loss_graphs against the reference's _compute_loss evaluated graph by graph and averaged (g16 fixture:
    3 graphs of different sizes incl. one without a positive label; 8 graphs = the shipped accumulate_grad_batches)."""
    z = golden("g16_loss_graphs.npz")
    logits = torch.from_numpy(z[f"{tag}:logits"]).to(dev())
    labels = torch.from_numpy(z[f"{tag}:labels"]).to(dev())
    ptr = z[f"{tag}:edge_ptr"]
    K = len(ptr) - 1
    eg = torch.from_numpy(np.repeat(np.arange(K, dtype=np.int32), np.diff(ptr))).to(dev())
    loss, grad = tracking_loss_and_grad(logits, labels, 0, float(z[f"{tag}:weight"]), edge_graph=eg, n_graphs=K)
    ref = float(z[f"{tag}:loss"])
    assert abs(float(loss[0]) - ref) <= 1e-5 * max(1.0, abs(ref))
    assert float(np.abs(grad.cpu().numpy() - z[f"{tag}:grad"]).max()) <= 1e-6 * max(1.0, float(np.abs(z[f"{tag}:grad"]).max()))
    # one graph: the plain loss
    l1, g1 = tracking_loss_and_grad(logits, labels, 0, 0.75)
    l2, g2 = tracking_loss_and_grad(logits, labels, 0, 0.75, edge_graph=torch.zeros_like(eg), n_graphs=1)
    assert abs(float(l1[0]) - float(l2[0])) <= 1e-6 * max(1.0, abs(float(l1[0]))) and float((g1 - g2).abs().max()) <= 1e-7 * float(g1.abs().max())


def test_train_step_over_a_batch_of_graphs_is_the_mean_of_the_single_graph_steps():
    """TrainStep on the block-diagonal batch of the 8 cfg-D graphs (configs[3]: one KITTIMOTS-like graph per accumulate_grad_batches
    micro-step) with the per-graph loss: ONE forward / backward whose gradient equals the mean of the 8 single-graph gradients
    (relative L2 per tensor <= 1e-5: the sub-graphs do not interact; only fp32 summation order differs), i.e. the reference's
    optimizer step (configs/tracking_cfg.yaml:3-4) from one launch sequence."""
    from mpntrackseg_amd import synth
    from mpntrackseg_amd.mpn import MOTMPNet
    from mpntrackseg_amd.train import TrainStep
    c = synth.CONFIGS["D"]
    graphs = [synth.make_knn_graph(seed=1 + r, node_in_dim=64, **c["knn"]) for r in range(8)]
    params = synth.model_params(c["d"], c["L"], "sum", num_class_steps=3, node_in_dim=64)
    W = synth.make_weights(params, seed=7, gain=0.5)

    def fresh():
        m = MOTMPNet(params)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
        return m.to(dev()).train()

    def labels_of(g, seed):
        return torch.from_numpy((synth.uniform01(seed, g["edge_index"].shape[1]) < 0.15).astype(np.float32)).to(dev())

    singles = []
    for i, g in enumerate(graphs):
        st = TrainStep(fresh())
        st(*(torch.from_numpy(g[k]).to(dev()) for k in ("x", "edge_index", "edge_attr")), labels=labels_of(g, 30 + i), optimizer_step=False)
        torch.cuda.synchronize()
        singles.append(st.bucket.flat.double().cpu().numpy().copy())
    want = sum(singles) / 8
    b = synth.batch_graphs(graphs)
    st = TrainStep(fresh())
    lab = torch.cat([labels_of(g, 30 + i) for i, g in enumerate(graphs)])
    st(*(torch.from_numpy(b[k]).to(dev()) for k in ("x", "edge_index", "edge_attr")), labels=lab, optimizer_step=False,
       edge_graph=torch.from_numpy(b["edge_graph"]).to(dev()), n_graphs=8)
    torch.cuda.synchronize()
    got = st.bucket.flat.double().cpu().numpy()
    bad = []
    for p_ in st.bucket.params:
        o, n = st.bucket.offsets[id(p_)], p_.numel()
        ref = want[o:o + n]
        if np.linalg.norm(ref) > 0:
            err = float(np.linalg.norm(got[o:o + n] - ref) / np.linalg.norm(ref))
            if err > 1e-5:
                bad.append((tuple(p_.shape), err))
    assert not bad, bad
