"""The N>1 path on CPU with gloo, world_size 2: graphs shard round-robin over ranks (no data-path
collective), and the only exchange -- the all-reduce(sum)/W of the flat gradient bucket -- reproduces the
reference's accumulate_grad_batches averaging (configs/tracking_cfg.yaml:4)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mpntrackseg_amd import synth
from mpntrackseg_amd.train import FlatBucket, allreduce_mean_, shard_indices


def test_shard_indices_partition():
    for n, w in [(8, 8), (8, 2), (10, 4), (3, 8), (0, 2)]:
        parts = [shard_indices(n, r, w) for r in range(w)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_flat_bucket_views_alias_param_grads():
    lin = [torch.nn.Linear(5, 3), torch.nn.Linear(3, 1)]
    params = [p for l in lin for p in l.parameters()]
    b = FlatBucket(params)
    # every parameter starts on a 16-byte boundary of the flat buffers (sizes rounded up to 4 floats; the padding stays zero)
    # ... plus four spare elements: flat[n] carries a rank's invalid-graph flag through the gradient all-reduce (train.TrainStep)
    assert b.n == sum((p.numel() + 3) // 4 * 4 for p in params) and b.flat.numel() == b.n + 4
    assert all(p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0 for p in params)
    assert all(p.data_ptr() >= b.flat_params.data_ptr() for p in params)
    b.views[id(params[0])].fill_(2.0)
    assert params[0].grad is not None and float(params[0].grad.sum()) == 2.0 * params[0].numel()
    assert float(b.flat.sum()) == 2.0 * params[0].numel()
    b.zero_()
    assert float(params[0].grad.abs().sum()) == 0.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank holds the same weights and its own shard of 5 graphs' (synthetic) gradients
        n = 1000
        mine = shard_indices(5, rank, world)
        flat = torch.zeros(n)
        for gi in mine:
            flat += torch.from_numpy(synth.normal(100 + gi, (n,)))
        flat /= max(len(mine), 1)          # local mean over the rank's graphs
        allreduce_mean_(flat, world)
        out[rank] = flat.numpy().copy()
    finally:
        dist.destroy_process_group()


def test_gloo_allreduce_mean_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert np.array_equal(a, b)  # all ranks end with identical averaged gradients
    g = [synth.normal(100 + i, (1000,)) for i in range(5)]
    r0 = (g[0] + g[2] + g[4]) / 3
    r1 = (g[1] + g[3]) / 2
    assert np.allclose(a, (r0 + r1) / 2, atol=1e-6)
