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
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo")
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
    moved = not np.array_equal(mine.cpu().numpy(), np.concatenate([np.pad(W[k].ravel(), (0, (-W[k].size) % 4)) for k in W]).astype(np.float32))
    print("RANK %d err %.3e side_stream %d same_params %d moved %d" % (rank, err, uses_side, same, moved), flush=True)
    ok = err < 2e-5 and same and moved and uses_side == (L >= 4)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 3)


if __name__ == "__main__":
    main()
