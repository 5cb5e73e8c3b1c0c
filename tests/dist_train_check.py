"""Child program of tests/test_gpu_distributed.py (launched with torch.distributed.run, 2 ranks, gloo, both on cuda:0):
one data-parallel TrainStep -- each rank its own graph -- and the checks that make it the reference's
accumulate_grad_batches averaging (configs/tracking_cfg.yaml:4): the averaged gradient equals the mean of the two
single-rank gradients, and both ranks hold identical parameters after the Adam step."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
import torch.distributed as dist

from mpntrackseg_amd import capi, synth
from mpntrackseg_amd.mpn import MOTMPNet
from mpntrackseg_amd.train import TrainStep


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    what = sys.argv[3] if len(sys.argv) > 3 else "average"
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    if what == "nccl1":
        # ONE rank, backend nccl (= RCCL): the communicator, stream and async-work plumbing of the data-parallel step on real RCCL
        dist.init_process_group("nccl", device_id=dev)
        return nccl_one_rank_check(L, d, dev)
    dist.init_process_group("gloo")
    if what == "overlap":
        return overlap_check(rank, world, L, d, dev)
    if what == "badgraph":
        return bad_graph_check(rank, world, L, d, dev)
    params = synth.model_params(d, L, "sum", node_in_dim=64)
    W = synth.make_weights(params, seed=7, gain=0.6)

    def fresh():
        m = MOTMPNet(params)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
        return m.to(dev).train()

    graphs = [synth.make_graph(300, 2400 + 200 * r, seed=50 + r, node_in_dim=64) for r in range(world)]

    def tensors(g):
        return [torch.from_numpy(g[k]).to(dev) for k in ("x", "edge_index", "edge_attr")]

    # single-rank gradients of every graph (no collective, no optimizer step)
    singles = []
    for g in graphs:
        st = TrainStep(fresh(), world_size=1)
        st(*tensors(g), optimizer_step=False)
        torch.cuda.synchronize()
        singles.append(st.bucket.flat.double().cpu().numpy().copy())
    want = sum(singles) / world
    # the data-parallel step
    model = fresh()
    step = TrainStep(model, world_size=world, lr=1e-3)
    capi.path_counters(reset=True)
    step(*tensors(graphs[rank]), optimizer_step=True)
    torch.cuda.synchronize()
    got = step.bucket.flat.double().cpu().numpy()
    err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))
    uses_side = bool(capi.load().mpnhip_backward_uses_side_stream(model.c_model([])))
    # parameters after Adam: identical on both ranks (same averaged gradient, same update kernel)
    mine = step.bucket.flat_params.detach().clone()
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    same = all(torch.equal(gathered[0], t) for t in gathered[1:])
    moved = not np.array_equal(mine[:step.bucket.n].cpu().numpy(), np.concatenate([np.pad(W[k].ravel(), (0, (-W[k].size) % 4)) for k in W]).astype(np.float32))
    print("RANK %d err %.3e side_stream %d same_params %d moved %d" % (rank, err, uses_side, same, moved), flush=True)
    ok = err < 2e-5 and same and moved and uses_side == (L >= 4)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 3)


def overlap_check(rank, world, L, d, dev):
    """The message-passing bucket's all-reduce is reached on the library's side stream BEFORE the caller's stream has finished the
    encoder's backward (MPNHIP_BWD_DEFER_SIDE_JOIN): event timestamps of train.TrainStep.allreduce_buckets.  The node encoder
    reads 2048-d inputs here, as in the reference, so its backward is a real piece of work."""
    params = synth.model_params(d, L, "sum")
    W = synth.make_weights(params, seed=7, gain=0.6)
    m = MOTMPNet(params)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
    m = m.to(dev).train()
    g = synth.make_graph(3000, 24000, seed=60 + rank)
    x, ei, ea = (torch.from_numpy(g[k]).to(dev) for k in ("x", "edge_index", "edge_attr"))
    step = TrainStep(m, world_size=world, lr=1e-4)
    leads = []
    for _ in range(4):
        step(x, ei, ea)
        torch.cuda.synchronize()
        leads.append(step.ev_mp_ready.elapsed_time(step.ev_main_done) * 1e3)   # us by which the side stream got there first
    print("RANK %d overlap lead_us %s" % (rank, " ".join("%.0f" % v for v in leads)), flush=True)
    ok = max(leads[1:]) > 0.0
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 4)


def nccl_one_rank_check(L, d, dev):
    """TrainStep(force_collectives=True) on a 1-rank RCCL group: the backward leaves its side stream un-joined
    (MPNHIP_BWD_DEFER_SIDE_JOIN), the message-passing bucket's all-reduce is enqueued on that stream through an ExternalStream with
    async_op=True, the encoder's bucket on the caller's stream, both are waited for, mpnhip_side_stream_join, / 1, guarded Adam --
    and must give the plain single-rank step: the same gradients (a 1-rank sum, / 1) and parameters after Adam, to summation noise."""
    params = synth.model_params(d, L, "sum", node_in_dim=64)
    W = synth.make_weights(params, seed=7, gain=0.6)

    def fresh():
        m = MOTMPNet(params)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
        return m.to(dev).train()

    g = synth.make_graph(3000, 24000, seed=61, node_in_dim=64)
    t = [torch.from_numpy(g[k]).to(dev) for k in ("x", "edge_index", "edge_attr")]
    plain = TrainStep(fresh(), world_size=1, lr=1e-3)
    coll = TrainStep(fresh(), world_size=1, lr=1e-3, force_collectives=True)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    ok = True
    for it in range(3):
        plain(*t)
        coll(*t)
        torch.cuda.synchronize()
        # (not bitwise: with the deferred join the encoder's weight-gradient products stay on the caller's stream instead of riding in
        # the tail batch -- other row chunks, another fp32 summation order)
        gp, gc = plain.bucket.flat[:plain.bucket.n].double(), coll.bucket.flat[:coll.bucket.n].double()
        pp, pc = plain.bucket.flat_params.double(), coll.bucket.flat_params.double()
        dg = float((gp - gc).abs().max() / gp.abs().max().clamp(min=1e-30))
        dp = float((pp - pc).abs().max() / pp.abs().max().clamp(min=1e-30))
        same_g, same_p = dg < 1e-5, dp < 1e-5
        print("NCCL1 step %d grad diff %.2e param diff %.2e" % (it, dg, dp), flush=True)
        lead = coll.ev_mp_ready.elapsed_time(coll.ev_main_done) * 1e3 if L >= 4 else float("nan")
        print("NCCL1 step %d same_grads %d same_params %d side_lead_us %.0f skipped %d" % (it, same_g, same_p, lead, coll.opt.t - coll.opt.applied_steps),
              flush=True)
        ok = ok and same_g and same_p and coll.opt.applied_steps == it + 1
    moved = not torch.equal(coll.bucket.flat_params, TrainStep(fresh(), world_size=1).bucket.flat_params)
    # an invalid graph: the flag rides through the RCCL all-reduce, the guarded Adam skips on the device, then IndexError
    bad = t[1].clone()
    bad[1, 5] = 3000
    before = coll.bucket.flat_params.clone()
    raised = False
    try:
        coll(t[0], bad, t[2])
    except IndexError:
        raised = True
    torch.cuda.synchronize()
    ok = ok and moved and raised and bool(torch.equal(before, coll.bucket.flat_params)) and coll.opt.applied_steps == 3
    print("NCCL1 ok %d moved %d raised %d" % (ok, moved, raised), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 6)


def bad_graph_check(rank, world, L, d, dev):
    """One rank's edge_index leaves [0, N): that rank raises IndexError like the reference's gather (mpn.py:69) AFTER taking part
    in the step's collectives, and NO rank applies the optimizer step (the flag rides in the bucket's spare element; guarded Adam)."""
    params = synth.model_params(d, L, "sum", node_in_dim=64)
    W = synth.make_weights(params, seed=7, gain=0.6)
    m = MOTMPNet(params)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
    m = m.to(dev).train()
    g = synth.make_graph(300, 2400, seed=70 + rank, node_in_dim=64)
    ei = g["edge_index"].copy()
    if rank == 1:
        ei[1, 7] = 300   # one past the last node
    x, eit, ea = torch.from_numpy(g["x"]).to(dev), torch.from_numpy(ei).to(dev), torch.from_numpy(g["edge_attr"]).to(dev)
    step = TrainStep(m, world_size=world, lr=1e-2)
    before = step.bucket.flat_params.detach().clone()

    class Holder:
        pass
    holder = Holder()   # (the prepared graph is cached here: the SECOND call finds it validated already and must raise again)
    raised = []
    for _ in range(2):
        try:
            step(x, eit, ea, holder=holder)
            raised.append(False)
        except IndexError:
            raised.append(True)
    torch.cuda.synchronize()
    unchanged = bool(torch.equal(before, step.bucket.flat_params))
    # both calls were skipped on the device on BOTH ranks: no optimizer step counted (the bias corrections of the next good
    # step are those of step 1)
    applied = step.opt.applied_steps
    print("RANK %d badgraph raised %s unchanged %d applied %d" % (rank, raised, unchanged, applied), flush=True)
    ok = unchanged and raised == [rank == 1] * 2 and applied == 0 and step.opt.t == 2
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 5)


if __name__ == "__main__":
    main()
