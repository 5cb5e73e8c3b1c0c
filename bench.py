#!/usr/bin/env python3
"""bench.py -- edges/ms of the MPN hot path on synthetic tracking graphs (BASELINE.json metric).

One "step" = one full pass of the hot path (encoder + L message-passing steps + per-step classifier,
and -- in training mode -- the hand-written backward and, for N>1 ranks, the RCCL all-reduce of the
flat gradient bucket) over one synthetic graph per GPU, inputs and weights resident in HBM.
Workload at N=1: BASELINE.json configs[1] = cfg-B (5,000 nodes / 50,000 directed edges / 128-d / 12
steps, fp32).  Each rank owns one graph (graphs shard by sequence; weak scaling); `value` is
sum_ranks(E) * K / max_rank(time).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode fwd|train] [--config B] [--agg sum]
N>1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np
import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="B")
    ap.add_argument("--agg", default="sum", help="node_agg_fn (reference default: sum, configs/tracking_cfg.yaml:135)")
    ap.add_argument("--mode", default="auto", choices=["auto", "fwd", "train"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def cpu_baseline(params, W, g, budget_s=15.0):
    """The CPU oracle (torch-CPU restatement of the reference path) timed on this box's host cores:
    forward over the same graph, as many runs as fit ~budget_s (at least 2, first one discarded)."""
    from oracle import mpn_oracle as O
    nthreads = os.cpu_count() or 1
    torch.set_num_threads(nthreads)
    Wt = O.to_tensors(W)
    x, ei, ea = (torch.from_numpy(g[k]) for k in ("x", "edge_index", "edge_attr"))
    times = []
    t_start = time.time()
    with torch.no_grad():
        while len(times) < 2 or (time.time() - t_start < budget_s and len(times) < 20):
            t0 = time.perf_counter()
            O.forward(params, Wt, x, ei, ea)
            times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:]))
    E = ei.shape[1]
    return {"value": E / (t * 1e3), "unit": "edges/ms", "cores": nthreads, "kind": "port",
            "sample": "oracle/mpn_oracle.py forward (torch %s CPU, %d threads) on the same graph, median of %d runs, %.0f ms each"
                      % (torch.__version__, nthreads, len(times) - 1, t * 1e3)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    from mpntrackseg_amd import capi, synth
    from mpntrackseg_amd.mpn import MOTMPNet

    c = synth.CONFIGS[args.config]
    params = synth.model_params(c["d"], c["L"], args.agg)
    W = synth.make_weights(params, seed=7)
    g = synth.make_graph(c["N"], c["E"], seed=1 + rank)  # one graph (sequence) per rank
    model = MOTMPNet(params)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
    model = model.to(dev)
    x = torch.from_numpy(g["x"]).to(dev)
    ei = torch.from_numpy(g["edge_index"]).to(dev)
    ea = torch.from_numpy(g["edge_attr"]).to(dev)
    E, N = c["E"], c["N"]

    from mpntrackseg_amd import train as mtrain
    have_bwd = mtrain.backward_available()
    mode = args.mode
    if mode == "auto":
        mode = "train" if have_bwd else "fwd"
    if mode == "train" and not have_bwd:
        raise SystemExit("--mode train needs mpnhip_backward")

    class Holder:
        pass
    holder = Holder()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    # graph prep is done once per graph (cached on the holder) and reported separately
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    from mpntrackseg_amd.mpn import _prepared
    _prepared(ei, N, holder)
    torch.cuda.synchronize()
    prep_ms = (time.perf_counter() - t0) * 1e3

    if mode == "fwd":
        model.eval()

        def step():
            with torch.no_grad():
                return model.hot_path(x, ei, ea, holder=holder)
    else:
        stepper = mtrain.TrainStep(model, world_size=world)

        def step():
            return stepper(x, ei, ea, holder=holder)

    for _ in range(args.warmup):
        step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    value = world * E / ms_per_step

    out = {
        "metric": "edges/ms (MPN %s) on synthetic tracking graph" % ("forward+backward" if mode == "train" else "forward"),
        "value": value, "unit": "edges/ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "cfg-%s: %d nodes / %d directed edges / %d-d feats / %d MP steps, node_agg_fn=%s, "
                               "%s, one graph per GPU" % (args.config, N, E, c["d"], c["L"], args.agg,
                                                          "training step (fwd+bwd%s)" % ("+RCCL grad all-reduce" if world > 1 else "")
                                                          if mode == "train" else "inference forward"),
                   "nodes": N, "edges": E, "feat_dim": c["d"], "mp_steps": c["L"], "agg": args.agg, "mode": mode,
                   "parallelism": "graphs sharded 1 per GPU (dp%d)" % world},
        "graph_prep_ms": prep_ms,
        "edge_steps_per_ms": value * c["L"],
    }

    if rank == 0 and not args.no_roofline:
        out.update(measure_rooflines(capi, synth, model, c, args, dev, x, ei, ea, holder, N, E))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(params, W, g)
    if mode == "train":
        # forward-only rate beside the training rate (the north-star target is quoted on forward)
        model.eval()
        with torch.no_grad():
            for _ in range(3):
                model.hot_path(x, ei, ea, holder=holder)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                model.hot_path(x, ei, ea, holder=holder)
            torch.cuda.synchronize()
        out["forward_edges_per_ms"] = E / ((time.perf_counter() - t0) * 1e3 / 20)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def measure_rooflines(capi, synth, model, c, args, dev, x, ei, ea, holder, N, E):
    """Live HIP-event timing (on the launch stream) of the two kernels that bound the path:
    the fp32 MFMA GEMM at the edge-MLP layer-1 shape (dominant by time) and the aggregation kernel."""
    lib = capi.load()
    d = c["d"]
    dn, de, he = d, d // 2, 5 * d // 2
    res = {}
    # --- dominant kernel: edge MLP first layer  [E, 2de] x [2de, he]  (+bias, ReLU), fp32 MFMA
    K, Nn = 2 * de, he
    a = torch.from_numpy(synth.normal(3, (E, K))).to(dev)
    w = torch.from_numpy(synth.normal(3, (Nn, K), stream=1, std=(2.0 / K) ** 0.5)).to(dev)
    b = torch.zeros(Nn, device=dev)
    y = torch.empty((E, Nn), device=dev)
    us = ctypes.c_float(0)
    capi.check(lib.mpnhip_time_linear(capi.ptr(a), capi.ptr(w), capi.ptr(b), capi.ptr(y), E, Nn, K, 50,
                                      ctypes.byref(us), capi.stream_ptr()), "time_linear")
    flops = 2.0 * E * K * Nn
    res["roofline"] = {"bound": "mfma", "kernel": "gemm_kernel (edge MLP layer 1: [%d,%d]x[%d,%d] fp32 MFMA 32x32x2)" % (E, K, K, Nn),
                       "achieved": flops / (us.value * 1e-6) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                       "frac": flops / (us.value * 1e-6) / 1e12 / 157.3, "traffic": None, "avg_us": us.value}
    # --- aggregation kernel: M (dn*4 + 4) + N dn 4 bytes per direction, both directions
    from mpntrackseg_amd.mpn import _prepared
    g = _prepared(ei, N, holder)
    msg = torch.from_numpy(np.maximum(synth.normal(4, (E, dn)), 0)).to(dev)
    out = torch.empty((N, 2 * dn), device=dev)
    capi.check(lib.mpnhip_time_aggregate(capi.ptr(g.buf), N, E, capi.ptr(msg), dn, capi.AGG_CODE[args.agg], capi.ptr(out),
                                         100, ctypes.byref(us), capi.stream_ptr()), "time_aggregate")
    bytes_agg = E * (dn * 4 + 4) + 2 * N * dn * 4 + (2 * N + 1) * 4
    res["roofline_aggregation"] = {"bound": "hbm", "kernel": "k_segment_reduce (both directions, %d messages x %d-d)" % (E, dn),
                                   "achieved": bytes_agg / (us.value * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                   "frac": bytes_agg / (us.value * 1e-6) / 1e9 / 8000.0, "traffic": None,
                                   "avg_us": us.value, "algorithmic_bytes": bytes_agg}
    return res


if __name__ == "__main__":
    main()
