#!/usr/bin/env python3
"""bench.py -- edges/ms of the MPN hot path on synthetic tracking graphs (BASELINE.json metric).

One "step" = one full pass of the hot path (encoder + L message-passing steps + per-step classifier,
and -- in training mode -- the hand-written backward and, for N>1 ranks, the RCCL all-reduce of the
flat gradient bucket) over one synthetic graph per GPU, inputs and weights resident in HBM.
Workload at N=1: BASELINE.json configs[1] = cfg-B (5,000 nodes / 50,000 directed edges / 128-d / 12
steps, fp32).  Each rank owns one graph (graphs shard by sequence; weak scaling); `value` is
sum_ranks(E) * K / max_rank(time).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode fwd|train] [--config B] [--agg sum]
N>1: either the driver's launcher form  python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...  (ranks read
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env), or plain  python bench.py --gpus N ...  -- with no WORLD_SIZE in the env the
parent starts the N ranks itself as fresh child processes (spawn_ranks) before it touches the GPU, and rank 0 prints the ONE line.
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
    ap.add_argument("--graph-file", default=None,
                    help="a precomputed detection graph (mpntrackseg_amd/graphfile.py .npz, e.g. MOTS20-02) instead of the "
                         "synthetic one; --config then only selects the model dims and step count")
    ap.add_argument("--agg", default="sum", help="node_agg_fn (reference default: sum, configs/tracking_cfg.yaml:135)")
    ap.add_argument("--mode", default="auto", choices=["auto", "fwd", "train"])
    ap.add_argument("--no-split-line", action="store_true", help="skip the extra measurement in the other fp32 mode (fp32 MFMAs / split)")
    ap.add_argument("--precision", default="auto", choices=["auto", "fp32", "fp32_split", "fp32_wgsplit", "bf16"],
                    help="operand precision of the Linear products; bf16 (bf16 operands, fp32 accumulate: inference and training) is "
                         "BASELINE.json configs[4]'s mode, never the headline configuration")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1: nccl (= RCCL over xGMI, one rank per GPU); gloo lets several ranks "
                         "share ONE GPU -- a functional run of the N > 1 path where only one GPU is available, not a scaling number")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1 only: create a 1-rank process group on --backend and issue the training step's collectives anyway "
                         "(TrainStep(force_collectives=True)): the RCCL code path on the one GPU that is there; adds allreduce_ms")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-forward-rate", action="store_true",
                    help="training mode: skip the forward-only rate measured beside the step (profiling runs: the kernel trace then holds training steps only)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the short labelled measurements of the other BASELINE.json configurations appended to the default line")
    return ap.parse_args()


def usable_cores():
    """Host cores this process may really use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(params, W, g, train, budget_s=20.0):
    """The CPU oracle (torch-CPU restatement of the reference path: same ops as the reference's PyTorch CPU
    path) timed on this box's host cores on the SAME graph: forward, or forward+backward with the bench's loss
    gradient when `train`.  Bounded sample: first run discarded, then runs until ~budget_s (2..10 runs)."""
    from oracle import mpn_oracle as O
    nthreads = usable_cores()
    torch.set_num_threads(nthreads)
    # bounded sample: graphs whose full pass would take minutes on the host (cfg-E: 12.7 TFLOP per forward) are timed with ONE and
    # with TWO message-passing steps; every step costs the same, so t(L) = t_fixed + L t_step with t_step = t(2) - t(1) and
    # t_fixed = t(1) - t_step (encoders, loss, the encoder's backward: counted once, not per step)
    L_full = int(params["num_enc_steps"])
    E_ = int(g["edge_index"].shape[1])
    d_ = int(params["encoder_feats_dict"]["node_out_dim"])
    extrapolate = L_full > 2 and E_ * d_ * d_ * L_full > 1e11
    params_full = params
    if extrapolate:
        params = dict(params_full, num_enc_steps=2, num_class_steps=min(int(params_full["num_class_steps"]), 2))
        params_one = dict(params_full, num_enc_steps=1, num_class_steps=1)
    Wt = O.to_tensors(W, requires_grad=train)
    x, ei, ea = (torch.from_numpy(g[k]) for k in ("x", "edge_index", "edge_attr"))
    E = ei.shape[1]
    labels = (torch.arange(E) % 7 == 0).float()

    def run(prm):
        if not train:
            with torch.no_grad():
                O.forward(prm, Wt, x, ei, ea)
            return
        _, logits, _, _ = O.forward(prm, Wt, x, ei, ea, return_state=True)
        lg = torch.stack([l.view(-1) for l in logits])
        pos = labels.sum()
        pw = (E - pos) / pos.clamp(min=1)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(lg, labels.expand_as(lg), pos_weight=pw, reduction="sum") / E
        torch.autograd.grad(loss, list(Wt.values()), allow_unused=True)

    def sample(prm, budget):
        times = []
        t_start = time.time()
        while len(times) < 3 and (len(times) < 2 or time.time() - t_start < budget):
            t0 = time.perf_counter()
            run(prm)
            times.append(time.perf_counter() - t0)
        while time.time() - t_start < budget and len(times) < 11:
            t0 = time.perf_counter()
            run(prm)
            times.append(time.perf_counter() - t0)
        return float(np.median(times[1:])), len(times) - 1

    note = ""
    if extrapolate:
        t2, n2 = sample(params, budget_s * 0.6)
        t1, n1 = sample(params_one, budget_s * 0.4)
        t_step = max(t2 - t1, 0.0)
        t_fixed = max(t1 - t_step, 0.0)
        t = t_fixed + L_full * t_step
        nruns = n1 + n2
        note = ("; timed with 1 and 2 of the %d message-passing steps (%.0f / %.0f ms) and extrapolated as t_fixed + L t_step = %.0f + %d x %.0f ms"
                % (L_full, t1 * 1e3, t2 * 1e3, t_fixed * 1e3, L_full, t_step * 1e3))
    else:
        t, nruns = sample(params, budget_s)
    return {"value": E / (t * 1e3), "unit": "edges/ms", "cores": nthreads, "kind": "port",
            "sample": "oracle/mpn_oracle.py %s (torch %s CPU ops, %d threads) on the same cfg graph, median of %d runs after 1 warm-up, %.0f ms each%s"
                      % ("forward+backward" if train else "forward", torch.__version__, nthreads, nruns, t * 1e3, note)}


def measure_case(cfg_name, mode, precision, steps, warmup, dev, agg="sum", seed=1, env=None, profile=True, n_batch=1):
    """One short, self-contained measurement of another BASELINE.json configuration (own model, own graph): ms per step,
    edges/ms and the rooflines of its profiled kernels.  Used for the labelled objects appended to the default line; never
    touches the headline's `value`."""
    import types
    from mpntrackseg_amd import capi, synth
    from mpntrackseg_amd import train as mtrain
    from mpntrackseg_amd.mpn import MOTMPNet, _prepared
    old_env = {}
    for k, v in (env or {}).items():
        old_env[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        c = dict(synth.CONFIGS[cfg_name])
        params = synth.model_params(c["d"], c["L"], agg)
        W = synth.make_weights(params, seed=7)
        if n_batch > 1:
            # n_batch graphs (the seeds the ranks of a data-parallel run would take) as ONE block-diagonal batch with the reference's
            # per-graph loss: accumulate_grad_batches optimizer micro-steps (configs/tracking_cfg.yaml:3-4) in one launch sequence
            gs = [synth.make_knn_graph(seed=seed + i, **c["knn"]) if c.get("knn") else synth.make_graph(c["N"], c["E"], seed=seed + i) for i in range(n_batch)]
            g = synth.batch_graphs(gs)
            c["E"], c["N"] = int(g["edge_index"].shape[1]), int(g["x"].shape[0])
        elif c.get("knn"):
            g = synth.make_knn_graph(seed=seed, **c["knn"])
            c["E"] = int(g["edge_index"].shape[1])
        else:
            g = synth.make_graph(c["N"], c["E"], seed=seed)
        model = MOTMPNet(params)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
        model = model.to(dev)
        x, ei, ea = (torch.from_numpy(g[k]).to(dev) for k in ("x", "edge_index", "edge_attr"))
        E, N = c["E"], c["N"]
        model.gemm_precision = precision
        prec = model.operand_precision(E)

        class Holder:
            pass
        holder = Holder()
        _prepared(ei, N, holder)
        if mode == "fwd":
            model.eval()
            model.keep_packed_weights = True

            def step():
                with torch.no_grad():
                    return model.hot_path(x, ei, ea, holder=holder)
        else:
            stepper = mtrain.TrainStep(model, world_size=1)
            eg = torch.from_numpy(g["edge_graph"]).to(dev) if n_batch > 1 else None

            def step():
                return stepper(x, ei, ea, holder=holder, edge_graph=eg, n_graphs=n_batch)
        lib = capi.load()
        t_warm = time.perf_counter()
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        # (these short cases follow the headline's CPU leg: at least 0.2 s of untimed steps, so that a 0.1 ms step is not timed
        # while the clocks are still coming back up -- seen once: cfg-D forward 0.139 ms in the line against 0.099 alone)
        while time.perf_counter() - t_warm < 0.2:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
        if profile:
            lib.mpnhip_profile_enable(5)   # coprime with the launches per step (4 / 7 / 12 chain, 3 weight-gradient)
        capi.path_counters(reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        counts = capi.path_counters(reset=True)
        out = {"workload": "cfg-%s%s: %d nodes / %d directed edges / %d-d feats / %d MP steps, node_agg_fn=%s, %s" % (
                   cfg_name, " x %d graphs as one block-diagonal batch, per-graph loss" % n_batch if n_batch > 1 else "", N, E, c["d"], c["L"], agg,
                   "training step (fwd+bwd+Adam)" if mode == "train" else "inference forward"),
               "precision": prec, "ms_per_step": ms, "value": E / ms, "unit": "edges/ms", "steps": steps, "warmup": warmup}
        if profile:
            extra = {}
            for name, kind in (("chain_bwd", 2), ("weight_grad", 3)):
                u, n, w = ctypes.c_float(0), ctypes.c_int(0), ctypes.c_double(0)
                capi.check(lib.mpnhip_profile_read_kind(kind, ctypes.byref(u), ctypes.byref(n), ctypes.byref(w)), "profile_read_kind")
                extra[name] = (u.value, n.value, w.value)
            gu, gc, au, ac, eu = ctypes.c_float(0), ctypes.c_int(0), ctypes.c_float(0), ctypes.c_int(0), ctypes.c_float(0)
            capi.check(lib.mpnhip_profile_read(ctypes.byref(gu), ctypes.byref(gc), ctypes.byref(au), ctypes.byref(ac), ctypes.byref(eu)), "profile_read")
            lib.mpnhip_profile_enable(0)
            prof = (gu.value, gc.value, au.value, ac.value, eu.value, extra, {k: v / float(steps) for k, v in counts.items()})
            keep = []
            chain = int(lib.mpnhip_edge_chain_active(model.c_model(keep, n_edges=E)))
            ns = types.SimpleNamespace(precision=prec, config=cfg_name, agg=agg, steps=steps)
            rf = rooflines(prof, c, ns, N, E, chain, mode)
            for k, v in rf.items():
                out[k] = {kk: vv for kk, vv in v.items() if kk not in ("what", "empty_event_pair_us", "naive_flop_equivalent_tflops")}
        return out
    finally:
        for k, v in old_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        torch.cuda.empty_cache()


def extras(dev, budget_s=90.0):
    """Short labelled measurements of the other BASELINE.json configurations: configs[4] (cfg-E, bf16-operand
    forward, plus its aggregation kernel as a separate launch -- the HBM-streaming figure), the configs[2] / configs[3] stand-ins
    (cfg-C / cfg-D training step, cfg-D forward).  Each is its own model and graph; a failure or the time budget drops the rest."""
    plan = [("cfgD_fwd", ("D", "fwd", "auto", 200, 20), None),
            ("cfgD_train", ("D", "train", "auto", 40, 10), None),
            ("cfgD_train_x8", ("D", "train", "auto", 30, 8), {"__n_batch": "8"}),
            ("cfgC_train", ("C", "train", "auto", 30, 8), None),
            ("cfgC_fwd", ("C", "fwd", "auto", 60, 10), None),
            ("cfgE_bf16_fwd", ("E", "fwd", "bf16", 5, 2), None),
            ("cfgE_bf16_train", ("E", "train", "bf16", 5, 1), None),   # (5 steps = 15 weight-gradient launches: every 5th samples each of the three sizes once)
            ("cfgE_bf16_fwd_unfused_aggregation", ("E", "fwd", "bf16", 3, 1), {"MPNHIP_NO_AGG_FUSION": "1"}),
            # node_agg_fn = mean / max (north_star: "scatter-mean/max neighbour aggregation"; reference models/mpn.py:266-273 selects the
            # lambda by config) on the headline workload, beside the shipped default `sum` the headline itself runs
            ("cfgB_train_mean", ("B", "train", "auto", 12, 4), {"__agg": "mean"}),
            ("cfgB_train_max", ("B", "train", "auto", 12, 4), {"__agg": "max"}),
            ("cfgB_fwd_mean", ("B", "fwd", "auto", 20, 5), {"__agg": "mean"}),
            ("cfgB_fwd_max", ("B", "fwd", "auto", 20, 5), {"__agg": "max"})]
    res, t_start, cache = {}, time.time(), {}
    for name, (cfg, mode, prec, steps, warm), env in plan:
        if time.time() - t_start > budget_s:
            res[name] = {"skipped": "time budget of the extra measurements (%.0f s) used up" % budget_s}
            continue
        try:
            nb, agg = 1, "sum"
            if env and "__n_batch" in env:
                nb, env = int(env["__n_batch"]), None
            if env and "__agg" in env:
                agg, env = env["__agg"], None
            res[name] = measure_case(cfg, mode, prec, steps, warm, dev, agg=agg, env=env, n_batch=nb)
            if nb > 1 and "cfgD_train" in res and "value" in res["cfgD_train"]:
                res[name]["edges_per_ms_over_one_graph_per_step"] = res[name]["value"] / res["cfgD_train"]["value"]
            if env:
                res[name]["env"] = env
                # only the separately launched aggregation kernel's line is of interest here
                res[name] = {k: v for k, v in res[name].items() if k in ("workload", "precision", "ms_per_step", "env", "roofline_aggregation")}
        except Exception as exc:
            res[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    res["seconds"] = time.time() - t_start
    return res


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes of this one (never by replacing it;
    the parent makes no HIP call -- torch.cuda.device_count() does not initialise the device), each with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in its env: the same shape torch.distributed.run gives them
    (reference: Trainer(gpus=1, accumulate_grad_batches=8), scripts/train.py:65-77 -- eight graphs per optimizer step, here in space).
    Rank 0's stdout (the ONE JSON line) is this process's stdout.  A rank that fails takes the others down; exit code = the worst."""
    import socket
    import subprocess
    n = int(args.gpus)
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and n > ndev:
        raise SystemExit("bench.py --gpus %d --backend nccl: only %d HIP device(s) visible; RCCL needs one GPU per rank "
                         "(--backend gloo shares devices: a functional run, not a scaling number)" % (n, ndev))
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    worst, alive = 0, list(procs)
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0:
                worst = worst or rc
                for q in alive:      # (exact PIDs of the children started above)
                    q.terminate()
    sys.exit(worst if 0 <= worst < 256 else 1)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and args.backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit("bench.py: %d ranks on backend nccl with %d HIP device(s) visible -- one GPU per rank, or --backend gloo"
                         % (world, torch.cuda.device_count()))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_pg = args.force_collectives and world == 1
    if world > 1 or force_pg:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # a launcher (torch.distributed.run) always sets it; a lone --force-collectives run picks a free port, so that two
            # bench / pytest processes on one host do not collide on a fixed one
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    from mpntrackseg_amd import capi, synth
    from mpntrackseg_amd.mpn import MOTMPNet

    c = synth.CONFIGS[args.config]
    params = synth.model_params(c["d"], c["L"], args.agg)
    W = synth.make_weights(params, seed=7)
    if args.graph_file:
        from mpntrackseg_amd import graphfile
        z = graphfile.load_graph(args.graph_file)
        g = {k: z[k].numpy() for k in ("x", "edge_index", "edge_attr")}
        c = dict(c, N=int(g["x"].shape[0]), E=int(g["edge_index"].shape[1]))
        params = synth.model_params(c["d"], c["L"], args.agg, node_in_dim=int(g["x"].shape[1]), edge_in_dim=int(g["edge_attr"].shape[1]))
        W = synth.make_weights(params, seed=7)
    elif c.get("knn"):
        g = synth.make_knn_graph(seed=1 + rank, **c["knn"])
        c = dict(c, E=int(g["edge_index"].shape[1]))
    else:
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
        mode = "train" if have_bwd and args.precision != "bf16" else "fwd"   # (bf16 operands: the forward is BASELINE.json's configs[4] line; --mode train runs its training step)
    model.gemm_precision = args.precision
    # 'auto' (the library's default, MOTMPNet.operand_precision): fp32 results from three-piece bf16 operands where the fused chain
    # kernels have the MFMA work for it (cfg-B / cfg-E widths; the reference's widths from ~32k edges: cfg-C), fp32 MFMAs on small graphs (cfg-D)
    args.precision = model.operand_precision(E)
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
    # what a loop that sees a NEW graph every step pays per step (the first call above includes allocations and
    # rocPRIM's one-time set-up): steady-state prep of the same edge list, not cached
    for _ in range(2):
        capi.PreparedGraph(ei, N)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        capi.PreparedGraph(ei, N)
    torch.cuda.synchronize()
    prep_steady_ms = (time.perf_counter() - t0) * 1e3 / 10

    if mode == "fwd":
        model.eval()
        model.keep_packed_weights = True   # inference with fixed weights (MOTMPNet.frozen_weights): packed images are reused

        def step():
            with torch.no_grad():
                return model.hot_path(x, ei, ea, holder=holder)
    else:
        stepper = mtrain.TrainStep(model, world_size=world, force_collectives=force_pg)

        def step():
            return stepper(x, ei, ea, holder=holder)

    lib = capi.load()
    for _ in range(args.warmup):
        step()
    profiled = rank == 0 and not args.no_roofline
    if profiled:
        # HIP events attached to every 7th launch of each profiled kernel kind (7: coprime with the 12 chain launches and the 3
        # differently-sized weight-gradient launches of a step, so every position of a step is sampled); large graphs (cfg-E: ~10
        # steps, 30 weight-gradient launches) every 5th -- six samples, two of each launch size (with 7: five samples, unbalanced:
        # avg_us 5.9 - 6.7 ms in the line against 4.8 ms in the rocprofv3 trace)
        lib.mpnhip_profile_enable(5 if E >= 200000 else 7)
    capi.path_counters(reset=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = None
    counts = capi.path_counters(reset=True)
    if profiled:
        # the two kernels of the backward first (mpnhip_profile_read resets every kind)
        extra = {}
        for name, kind in (("chain_bwd", 2), ("weight_grad", 3)):
            u, n, w = ctypes.c_float(0), ctypes.c_int(0), ctypes.c_double(0)
            capi.check(lib.mpnhip_profile_read_kind(kind, ctypes.byref(u), ctypes.byref(n), ctypes.byref(w)), "profile_read_kind")
            extra[name] = (u.value, n.value, w.value)
        gu, gc, au, ac, eu = ctypes.c_float(0), ctypes.c_int(0), ctypes.c_float(0), ctypes.c_int(0), ctypes.c_float(0)
        capi.check(lib.mpnhip_profile_read(ctypes.byref(gu), ctypes.byref(gc), ctypes.byref(au), ctypes.byref(ac),
                                           ctypes.byref(eu)), "profile_read")
        lib.mpnhip_profile_enable(0)
        prof = (gu.value, gc.value, au.value, ac.value, eu.value, extra, {k: v / float(args.steps) for k, v in counts.items()})
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    # whole-job units = the SUM of the ranks' edge counts: the kNN configurations (cfg-C / cfg-D) build a different graph per rank
    # (seed = 1 + rank), so rank 0's count times the world size would be off by the spread of E
    edges_total, edges_per_rank = E, [E]
    if world > 1:
        te = torch.zeros(world, device=dev, dtype=torch.float64)
        te[rank] = float(E)
        dist.all_reduce(te, op=dist.ReduceOp.SUM)
        edges_per_rank = [int(round(v)) for v in te.tolist()]
        edges_total = sum(edges_per_rank)
    value = edges_total / ms_per_step

    out = {
        "metric": "edges/ms (MPN %s) on %s tracking graph" % ("forward+backward" if mode == "train" else "forward",
                                                                 "the supplied" if args.graph_file else "synthetic"),
        "value": value, "unit": "edges/ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32", "fp32_split": "f32 (fused chain kernels: every f32 operand as the exact sum of three bf16 pieces, six "
                                                 "v_mfma_f32_32x32x16_bf16 products per multiply, f32 accumulate -- nothing rounded to bf16; "
                                                 "weight gradients: the same operand form in the row-panel kernel)",
                  "fp32_wgsplit": "f32 (fp32 MFMAs in the forward and the activation-gradient chain; weight gradients by the batched row-panel "
                                  "kernel on three-piece bf16 operands, f32 accumulate -- nothing rounded to bf16)",
                  "bf16": "bf16 operands, f32 accumulate"}[args.precision],
        "data": "synthetic" if not args.graph_file else "graph file %s (random-init weights)" % os.path.basename(args.graph_file),
        "config": {"workload": "cfg-%s: %d nodes / %d directed edges / %d-d feats / %d MP steps, node_agg_fn=%s, "
                               "%s, one graph per GPU" % (args.config, N, E, c["d"], c["L"], args.agg,
                                                          "training step (fwd+bwd%s)" % ("+RCCL grad all-reduce" if world > 1 else "")
                                                          if mode == "train" else "inference forward"),
                   "nodes": N, "edges": E, "edges_all_ranks": edges_total, "edges_per_rank": edges_per_rank, "feat_dim": c["d"], "mp_steps": c["L"], "agg": args.agg, "mode": mode,
                   "parallelism": ("1-rank %s process group, collectives issued (functional run of the data-parallel code path)"
                                   % ("RCCL (nccl)" if args.backend == "nccl" else args.backend)) if force_pg else
                                  "graphs sharded 1 per GPU (dp%d)" % world if args.backend == "nccl" or world == 1 else
                                  "dp%d over gloo, %d ranks per GPU (functional run of the N > 1 path, not a scaling number)"
                                  % (world, (world + torch.cuda.device_count() - 1) // torch.cuda.device_count())},
        "graph_prep_ms": prep_ms, "graph_prep_steady_ms": prep_steady_ms,
        "edge_steps_per_ms": value * c["L"],
    }

    if prof is not None:
        keep = []
        chain = int(lib.mpnhip_edge_chain_active(model.c_model(keep, n_edges=E)))   # 1: fp32 / split chain kernels, 2: bf16-operand chain
        out.update(rooflines(prof, c, args, N, E, chain, mode))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(params, W, g, mode == "train")
    def forward_rate():
        # forward-only rate beside the training rate (the north-star target is quoted on forward)
        model.eval()
        # (inference: the weights do not change between the calls -- their packed images are kept, MOTMPNet.frozen_weights)
        with torch.no_grad(), model.frozen_weights():
            for _ in range(3):
                model.hot_path(x, ei, ea, holder=holder)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                model.hot_path(x, ei, ea, holder=holder)
            torch.cuda.synchronize()
        model.train(mode == "train")
        return E / ((time.perf_counter() - t0) * 1e3 / 20)

    if mode == "train" and not args.no_forward_rate:
        out["forward_edges_per_ms"] = forward_rate()
    if world == 1 and args.precision in ("fp32", "fp32_split", "fp32_wgsplit") and not args.no_split_line:
        other = "fp32_split" if args.precision == "fp32" else "fp32"
        try:
            # the same step in the OTHER fp32 mode, reported beside the headline: fp32 MFMAs
            # (v_mfma_f32_32x32x2_f32 everywhere) when the headline runs the split chain kernels, and the other way round
            model.gemm_precision = other
            for _ in range(max(args.warmup, 5)):   # (first launches of the split kernel variants load their code objects)
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t0) * 1e3 / args.steps
            key = "fp32_split" if other == "fp32_split" else "fp32_mfma"
            out[key] = {"value": E / ms2, "unit": "edges/ms", "ms_per_step": ms2, "steps": args.steps,
                        "what": "same workload, mpnhip_model.precision = %s (logits and gradients of both modes are equally close to a "
                                "float64 oracle: tests/test_gpu_split.py, tests/test_gpu_pinned.py)"
                                % ("MPNHIP_PREC_FP32_SPLIT" if other == "fp32_split" else "MPNHIP_PREC_FP32: fp32 MFMAs everywhere")}
            if mode == "train":
                out[key]["forward_edges_per_ms"] = forward_rate()
            model.gemm_precision = args.precision
        except Exception as exc:   # the headline line above must survive a failure of the extra measurement
            model.gemm_precision = args.precision
            out["fp32_split" if other == "fp32_split" else "fp32_mfma"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if (world > 1 or force_pg) and mode == "train":
        # SURVEY.md section 8e "Reporting": the gradient all-reduce alone -- S bytes of the flat fp32 bucket, all-reduce(sum) as
        # one collective, HIP events around 20 calls -- and the bus bandwidth 2 (n - 1) / n x S / t it corresponds to (ring
        # all-reduce over xGMI: 7 links x ~153 GB/s per GPU; at these sizes, 1.2 - 19 MB, the collective is latency-bound)
        import torch.distributed as dist
        try:
            flat = stepper.bucket.flat
            keepg = flat.clone()
            for _ in range(3):
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            barrier()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            e1.record()
            torch.cuda.synchronize()
            ar_ms = e0.elapsed_time(e1) / 20
            flat.copy_(keepg)
            t = torch.tensor([ar_ms], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ar_ms = float(t.item())
            S = flat.numel() * 4
            out["allreduce_ms"] = ar_ms
            out["allreduce_bytes"] = S
            out["bus_gbs"] = 2.0 * (world - 1) / world * S / (ar_ms * 1e-3) / 1e9
            out["allreduce_what"] = ("all-reduce(sum) of the flat fp32 gradient bucket alone (%d bytes, %s backend), mean of 20 calls, max over ranks; "
                                     "in the step it runs as two buckets overlapped with the backward (train.TrainStep.allreduce_buckets)"
                                     % (S, args.backend))
        except Exception as exc:
            out["allreduce_ms"] = None
            out["allreduce_error"] = "%s: %s" % (type(exc).__name__, exc)
    if rank == 0 and world == 1 and not args.no_extras and args.config == "B" and not args.graph_file and args.mode == "auto":
        # the other BASELINE.json configurations, short and clearly labelled (never part of `value`)
        try:
            del x, ei, ea
            torch.cuda.empty_cache()
            out["other_configs"] = extras(dev)
        except Exception as exc:
            out["other_configs"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if rank == 0:
        print(json.dumps(ordered_line(out)))
    if world > 1 or force_pg:
        import torch.distributed as dist
        dist.destroy_process_group()


TRAFFIC_SOURCE = ("committed builder-run rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command (profiles/rNN/pmc_summary*.json, "
                  "tools/pmc_summary.py), not this run")


def pmc_traffic(kernel_key, cfg_name, precision="fp32", mode="fwd"):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/rNN/pmc_summary*.json, made by
    tools/pmc_summary.py with the MI355X guide's corrections; newest round first), or None.  cfg-B passes: the default training
    command; cfg-E (bf16): the forward command, and from round 4 the training command (pmc_summary_cfgE_train.json)."""
    rounds = ("r06", "r05", "r04", "r03", "r02", "r01")
    if cfg_name == "E" and precision == "bf16":
        name = "pmc_summary_cfgE_train.json" if mode == "train" else "pmc_summary_cfgE.json"
        for rnd in rounds:
            pth = os.path.join(REPO, "profiles", rnd, name)
            try:
                v = json.load(open(pth)).get(kernel_key, {}).get("hbm_bytes_per_launch")
            except Exception:
                v = None
            if v is not None:
                return v
        return None
    if cfg_name != "B" or precision == "bf16":
        return None
    for rnd in rounds:
        path = os.path.join(REPO, "profiles", rnd, "pmc_summary.json" if precision in ("fp32", "fp32_wgsplit") else "pmc_summary_split.json")
        try:
            v = json.load(open(path)).get(kernel_key, {}).get("hbm_bytes_per_launch")
        except Exception:
            v = None
        if v is not None:
            return v
    return None


def pmc_counters(kernel_key, cfg_name, precision):
    """SQ / TCC counter figures of the kernels the headline runs, from the committed rocprofv3 --pmc passes of the default command
    (tools/pmc_cfgB.sh -> profiles/rNN/pmc_cfgB_split.json: one counter group per pass, kernel-trace only): MFMA-busy fraction
    (SQ_VALU_MFMA_BUSY_CYCLES / (1,024 SIMDs x GRBM_GUI_ACTIVE / 8)), SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, L2 hit rate.  cfg-B in the
    split mode only (that is what the passes ran); None otherwise."""
    if cfg_name != "B" or precision != "fp32_split":
        return None
    for rnd in ("r06",):
        try:
            v = json.load(open(os.path.join(REPO, "profiles", rnd, "pmc_cfgB_split.json"))).get(kernel_key)
        except Exception:
            v = None
        if v:
            out = {k: round(float(v[k]), 4) for k in ("mfma_busy_frac", "wait_inst_any_over_wave_cycles", "l2_hit_rate", "lds_bank_conflict_over_active_lds") if k in v}
            out["source"] = "profiles/%s/pmc_cfgB_split.txt (committed rocprofv3 --pmc passes of the default command, not this run)" % rnd
            return out
    return None


def rooflines(prof, c, args, N, E, chain, mode="fwd"):
    """Roofline fractions from the in-stream HIP-event timings taken over the timed region (events attached to the dispatches
    on the stream each kernel is launched on).  `roofline` is the kernel with the largest time per step IN THE MEASURED MODE:
    inference = the fused forward chain; training = whichever of forward chain / backward chain / weight-gradient products
    takes most of a step (round 2: the weight-gradient product kernel); the others are reported beside it."""
    gemm_us, gemm_n, agg_us, agg_n, empty_us, extra, per_step = prof
    # which fused chain kernel the timed region actually launched (path counters): 2 = bf16-operand chain, 1 = fp32 / split chain,
    # 0 = none (the unfused GEMM path) -- mpnhip_edge_chain_active() only says what the model's shapes allow
    if per_step.get("edge_chain_fwd_bf16", 0) > 0:
        chain = 2
    elif per_step.get("edge_chain_fwd", 0) + per_step.get("edge_chain_fwd_split", 0) > 0:
        chain = 1
    else:
        chain = 0
    # avg_us: HIP events attached to the kernel's own dispatch on the launch stream (hipExtLaunchKernelGGL start / stop
    # events): the dispatch's begin -> end, what rocprofv3's kernel trace reports too (profiles/).  `empty_event_pair_us`
    # is what a plain record pair with nothing between costs on this box -- the overhead the attached events avoid.
    d = c["d"]
    dn, de, he = d, d // 2, 5 * d // 2
    res = {}
    hn, hc = 7 * d // 4, d // 4
    if gemm_n and chain == 2:
        # bf16-operand chain (edge_chain_bf16.hip): all of [e0 | e] is contracted in the kernel (no Q0 hoist: at these widths the
        # re-read of an [E, he] fp32 table costs more than the MFMAs it saves)
        macs = 2 * de * he + he * de + de * hc + hc + de * hn + hn * dn
        flops = 2.0 * E * macs
        # what the kernel must move at least -- both first-layer inputs and both outputs once, the
        # per-node projection table once, the edge indices, the logits
        # (the kernel aggregates the messages itself -- they never reach HBM -- unless MPNHIP_NO_AGG_FUSION is set: then M is written)
        fused_agg = not os.environ.get("MPNHIP_NO_AGG_FUSION")
        alg_bytes = E * (2 * de * 4 + de * 4 + 12 + 4) + N * (2 * he + 2 * hn) * 4 + (N * 2 * dn * 4 if fused_agg else E * dn * 4)
        mask_words = sum(((w + 31) // 32 + 1) // 2 for w in (he, de, hc, hn, dn))
        if mode == "train":
            # the SAVE variant also writes, per edge, the bf16 rows of H1 / HF / HC and of e' and the ReLU decision bits (8 bytes per
            # mask word: 64 lanes x 4 bytes per 32-edge wave tile) -- every one of them is read again by the backward
            alg_bytes += E * ((he + hn + hc + de) * 2 + mask_words * 8)
        ach = alg_bytes / (gemm_us * 1e-6) / 1e9
        traffic = pmc_traffic("edge_chain_bf16_save" if mode == "train" else "edge_chain_bf16", args.config, args.precision, mode)
        res["roofline"] = {"bound": "hbm",
                           "kernel": "edge_chain_bf16_kernel<%d,%d,%d,%d,%d> (%s" % ((he + 31) // 32, (de + 31) // 32, (hn + 31) // 32, (dn + 31) // 32, (hc + 31) // 32,
                                                                                   "two 4-wave blocks per CU" if d >= 256 else "one 8-wave block per CU")
                                     + (", SAVE variant" if mode == "train" else "") + "): fused edge MLP + classifier + flow MLPs of one MP step, bf16 operands / "
                                     "fp32 accumulate (v_mfma_f32_32x32x16_bf16), hidden layers N-tiled in registers, node_agg_fn in the kernel; bound by the per-edge "
                                     "gathers of the projection table (%d B per edge from a %d MB table, uniformly random columns)"
                                     % ((2 * he + hn) * 4, N * (2 * he + 2 * hn) * 4 // 1000000),
                           "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic,
                           "algorithmic_bytes": alg_bytes, "avg_us": gemm_us, "empty_event_pair_us": empty_us, "launches": gemm_n,
                           "algorithmic_flops": flops, "mfma_tflops": flops / (gemm_us * 1e-6) / 1e12,
                           "mfma_frac_of_bf16_peak": flops / (gemm_us * 1e-6) / 1e12 / 2516.6,
                           "ms_per_step": gemm_us * c["L"] / 1e3}
        if traffic:
            res["roofline"]["traffic_over_algorithmic"] = traffic / alg_bytes
            res["roofline"]["traffic_rate_gbs"] = traffic / (gemm_us * 1e-6) / 1e9
    elif gemm_n and chain:
        # fused per-edge chain (edge MLP e-part + classifier + flow MLP e-part): MACs per edge, DESIGN.md section 4
        # the re-attached e0's share of the first layer is computed once per forward (Q0, one GEMM) when de >= 32: the
        # kernel then contracts de, not 2 de, columns there -- EXECUTED flops are what the MFMA fraction is quoted on
        k1 = de if (de >= 32 and c["L"] > 1 and args.precision != "fp32_split") else 2 * de   # (no Q0 hoist in the split mode)
        macs = k1 * he + he * de + de * hc + hc + de * hn + hn * dn
        flops = 2.0 * E * macs
        ach = flops / (gemm_us * 1e-6) / 1e12
        split = args.precision == "fp32_split"
        if split:
            # six bf16 products per fp32 multiply-add: EXECUTED flops against the dense bf16 MFMA peak
            ach *= 6.0
        peak = 2516.6 if split else 157.3
        res["roofline"] = {"bound": "mfma", "kernel": "edge_chain_kernel<%s>: fused edge MLP + classifier + flow MLPs of one MP step, "
                                                      "%s, %d edges x %d MACs" % (
                                                          {128: "10,2,7,4", 64: "5,1,4,2", 32: "3,1,2,1"}.get(d, "?"),
                                                          "six v_mfma_f32_32x32x16_bf16 products per fp32 MAC (three-piece split operands)"
                                                          if split else "fp32 v_mfma_f32_32x32x2_f32", E, macs),
                           "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                           "traffic": pmc_traffic("edge_chain", args.config, args.precision), "avg_us": gemm_us,
                           "empty_event_pair_us": empty_us, "launches": gemm_n, "algorithmic_flops": flops,
                           # what the reference executes for the same per-edge modules (gathered
                           # [x_row | x_col | e] and [x_col | e'] inputs multiplied per edge), for comparability
                           "naive_flop_equivalent_tflops": 2.0 * E * ((4 * dn + 2 * de) * he + he * de + (2 * dn + de) * hn + hn * dn
                                                                      + de * hc + hc) / (gemm_us * 1e-6) / 1e12}
        # compulsory bytes of one fused message-passing step (every distinct input read once, every
        # output written once) -- what `traffic` (PMC, the chain kernel alone) is to be read against
        alg_bytes = E * (2 * de * 4 + de * 4 + 8 + 4) + N * (2 * dn * 4 + dn * 4)
        if mode == "train":
            # the training launch also keeps, per edge, the fp32 rows of H1 / HF / HC for the weight-gradient products and the ReLU
            # decisions of all five activation blocks as bits (8 bytes per mask word: 64 lanes x 4 bytes per 32-edge wave tile); the
            # messages M [E, dn] are written for the node side either way -- every one of them is read again by the backward
            mask_words = sum(((w + 31) // 32 + 1) // 2 for w in (he, de, hn, dn)) + 1
            alg_bytes += E * ((he + hn + hc) * 4 + mask_words * 8)
        alg_bytes += E * dn * 4          # (M: the chain kernel's own output, aggregated by node_chain_kernel)
        res["roofline"]["algorithmic_bytes"] = alg_bytes
        if res["roofline"]["traffic"]:
            res["roofline"]["traffic_over_algorithmic"] = res["roofline"]["traffic"] / alg_bytes
        res["roofline"]["ms_per_step"] = gemm_us * c["L"] / 1e3
    elif gemm_n:
        K, Nn = 2 * de, he
        flops = 2.0 * E * K * Nn  # algorithmic: E rows x [e0|e] (2 de) x he outputs
        ach = flops / (gemm_us * 1e-6) / 1e12
        bf16 = args.precision == "bf16"
        pk = 2516.6 if bf16 else 157.3   # dense bf16 MFMA peak for the bf16-operand mode
        res["roofline"] = {"bound": "mfma", "kernel": "gemm_kernel (B K-contiguous): edge-MLP layer 1 [%d,%d]x[%d,%d] %s, gather-add epilogue"
                                                      % (E, K, K, Nn, "bf16 operands / fp32 accumulate, v_mfma_f32_32x32x16_bf16" if bf16
                                                         else "fp32, v_mfma_f32_32x32x2_f32"),
                           "achieved": ach, "peak": pk, "unit": "TFLOP/s", "frac": ach / pk,
                           "traffic": pmc_traffic("gemm_edge_l1", args.config), "avg_us": gemm_us,
                           "empty_event_pair_us": empty_us, "launches": gemm_n,
                           "algorithmic_flops": flops, "ms_per_step": gemm_us * c["L"] / 1e3}
    if agg_n and per_step.get("node_chain", 0) > 0:
        # node_agg_fn runs inside node_chain_kernel (csrc/node_chain.hip: aggregate -> node update -> next step's projections): what that
        # launch must move at least -- the messages once, the CSR offsets, the projection table in (P0) and out (P') for all steps but
        # the last, x' out, and in training the aggregate rows kept for the backward
        L_ = max(int(c["L"]), 1)
        pw = 2 * he + 2 * hn
        bytes_nc = E * dn * 4 + (2 * N + 1) * 4 + N * dn * 4 + (N * 2 * dn * 4 if mode == "train" else 0) + 2.0 * N * pw * 4 * (L_ - 1) / L_
        ach = bytes_nc / (agg_us * 1e-6) / 1e9
        res["roofline_aggregation"] = {"bound": "hbm", "kernel": "node_chain_kernel (node_agg_fn=%s over %d messages x %d-d + node update + next projections [%d x %d], one launch)"
                                                                 % (args.agg, E, dn, N, pw),
                                       "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": None, "avg_us": agg_us,
                                       "launches": agg_n, "algorithmic_bytes": bytes_nc, "ms_per_step": agg_us * L_ / 1e3,
                                       "fused_into": "node_chain_kernel"}
    elif agg_n:
        # M (dn s + 4) + N dn s per direction (+ CSR offsets), both directions in one launch
        bytes_agg = E * (dn * 4 + 4) + 2 * N * dn * 4 + (2 * N + 1) * 4
        ach = bytes_agg / (agg_us * 1e-6) / 1e9
        res["roofline_aggregation"] = {"bound": "hbm", "kernel": "k_aggregate (both directions, %d messages x %d-d, %s)" % (E, dn, args.agg),
                                       "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                                       "traffic": pmc_traffic("k_aggregate", args.config, args.precision), "avg_us": agg_us,
                                       "empty_event_pair_us": empty_us, "launches": agg_n,
                                       "algorithmic_bytes": bytes_agg}
    if mode == "train" and "roofline" in res:
        split = args.precision == "fp32_split"
        peak = 157.3
        res["roofline_fwd_chain"] = res.pop("roofline")
        cand = {"roofline_fwd_chain": res["roofline_fwd_chain"]["ms_per_step"]}
        bu, bn, _ = extra.get("chain_bwd", (0.0, 0, 0.0))
        if bn and chain == 2 and per_step.get("edge_chain_bwd_bf16", 0) > 0:
            # edge_chain_bf16_bwd_kernel: what it must move at least per launch -- in: the gradient w.r.t. e_s (fp32), the ReLU decision
            # bits, the logit gradient and the row index per edge, the aggregate's gradient once per node; out: the five dZ blocks as
            # bf16 rows and the gradient w.r.t. e_{s-1} (fp32)
            mask_words = sum(((w + 31) // 32 + 1) // 2 for w in (he, de, hc, hn, dn))
            bytes_b = E * (de * 4 + mask_words * 8 + 4 + 8 + (dn + hn + hc + de + he) * 2 + de * 4) + N * 2 * dn * 4
            macs_b = hn * dn + de * hn + hc * de + hc + he * de + de * he      # B2 .. B6 (the re-attached e0's share of B6 is hoisted)
            fl = 2.0 * E * macs_b
            ach = bytes_b / (bu * 1e-6) / 1e9
            res["roofline_bwd_chain"] = {"bound": "hbm", "kernel": "edge_chain_bf16_bwd_kernel: fused activation-gradient chain of one MP step, bf16 operands, "
                                                                   "bf16 dZ rows out; %d edges x %d MACs" % (E, macs_b),
                                         "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "avg_us": bu, "launches": bn,
                                         "traffic": pmc_traffic("edge_chain_bf16_bwd", args.config, args.precision, "train"),
                                         "algorithmic_bytes": bytes_b, "algorithmic_flops": fl,
                                         "mfma_frac_of_bf16_peak": fl / (bu * 1e-6) / 1e12 / 2516.6, "ms_per_step": bu * c["L"] / 1e3}
            cand["roofline_bwd_chain"] = res["roofline_bwd_chain"]["ms_per_step"]
        elif bn and chain:
            ke = 2 * de
            macs_b = hn * dn + de * hn + hc * de + hc + he * de + ke * he      # B2 .. B6 of the backward chain (csrc/edge_chain.hip)
            fl = 2.0 * E * macs_b
            ach = fl / (bu * 1e-6) / 1e12 * (6.0 if split else 1.0)
            pk = 2516.6 if split else peak
            res["roofline_bwd_chain"] = {"bound": "mfma", "kernel": "edge_chain_bwd_kernel: fused activation-gradient chain of one MP step, %d edges x %d MACs" % (E, macs_b),
                                         "achieved": ach, "peak": pk, "unit": "TFLOP/s", "frac": ach / pk, "avg_us": bu, "launches": bn,
                                         "traffic": pmc_traffic("edge_chain_bwd", args.config, args.precision), "algorithmic_flops": fl,
                                         "ms_per_step": bu * c["L"] / 1e3}
            cand["roofline_bwd_chain"] = res["roofline_bwd_chain"]["ms_per_step"]
        tu, tn_, tw = extra.get("weight_grad", (0.0, 0, 0.0))
        if tn_ and per_step.get("gemm_tn_panel", 0.0) > 0:
            # MPNHIP_PREC_FP32_SPLIT: the row-panel kernel (csrc/wgrad_panel.hip): every operand row fetched once, six bf16 piece
            # products per fp32 multiply-add -- bound by the operand stream.  The profile hook's `work` is the launch's operand
            # bytes (rows x (n_out + k_in) x 4 over all jobs of the launch); launches differ (groups of 5 / 4 / 3 steps, the tail
            # batch), so achieved = sum of bytes / sum of durations over the sampled launches.
            launches = per_step.get("wgrad_panel_launches", 0.0) or float(tn_) / max(args.steps, 1)
            # round 5, bf16 rows at the 256-d widths: the profiled launch of a batch is the one-pass LDS-DMA kernel (csrc/wgrad_rows16.hip),
            # its work figure the bytes of ITS jobs; the batch's remaining jobs (narrow / [1 x k] shapes) follow in the row-panel launch
            rows16 = per_step.get("wgrad_rows16_launches", 0.0)
            if rows16 > 0:
                launches = rows16
            ach = tw / (tu * 1e-6) / 1e9
            # the same launches' products as fp32-equivalent flops (12 steps x edge-level + node-level + hoisted + encoder)
            res["roofline_weight_grad"] = {"bound": "hbm", "kernel": ("wgrad_rows16_kernel: dW += dZ^T H over bf16 rows for the 256-d products of a group of steps in one "
                                                                      "launch (one block per CU owns a row chunk and the whole output: every operand row fetched once; "
                                                                      "LDS-DMA ring, ds_read_b64_tr_b16 operands, v_mfma_f32_32x32x16_bf16), side stream") if rows16 > 0 else
                                                                     ("wgrad_panel_narrow_kernel (64 x 64 tiles, four blocks per CU)" if per_step.get("wgrad_panel_narrow_launches", 0.0) > 0
                                                                      else "wgrad_panel_kernel") + ": dW += dZ^T H for all products of a group of steps in one launch "
                                                                     "(row-panel blocks, %s, ds_read_b64_tr_b16 operands, v_mfma_f32_32x32x16_bf16), side stream"
                                                                     % ("bf16 source rows as stored by the chain kernels, one product per k block" if chain == 2 else
                                                                        "three-piece bf16 operands split in the loader"),
                                           "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "avg_us": tu, "launches": tn_,
                                           "traffic": pmc_traffic("wgrad_rows16" if rows16 > 0 else "wgrad_panel", args.config, args.precision, "train"),
                                           "algorithmic_bytes": tw, "launches_per_step": launches, "ms_per_step": tu * launches / 1e3}
            if res["roofline_weight_grad"]["traffic"]:
                res["roofline_weight_grad"]["traffic_over_algorithmic"] = res["roofline_weight_grad"]["traffic"] / tw
                if rows16 > 0:
                    # the PMC figure is the average over ALL launches of a step (two six-step groups and the tail's one product), the
                    # sampled `algorithmic_bytes` the average of the few launches that carried events: compare the PMC average with
                    # the operand bytes of a step's bf16-row jobs / launches per step instead (edge L1 / L2, flow L1 / L2 over L steps,
                    # + the hoisted e0 share's one product)
                    # (late round 5: + the node-level products -- per-node projections [pw x dn], node update [dn x 2 dn] over L steps,
                    # the hoisted x0 share; the node encoder's wide layers are left out of the figure: they depend on the input width)
                    pw = 2 * he + 2 * hn
                    step_bytes = c["L"] * E * 2.0 * ((he + de) + (de + he) + (hn + de) + (dn + hn)) + E * 2.0 * (he + de)
                    if per_step.get("wgrad_rows16", 0.0) > 5.5 * launches:   # (13 edge-level jobs per step over 3 launches; 20 with the node level)
                        step_bytes += c["L"] * N * 2.0 * ((pw + dn) + (dn + 2 * dn)) + N * 2.0 * (pw + dn)
                    avg = step_bytes / launches
                    res["roofline_weight_grad"]["algorithmic_bytes_avg_per_launch"] = avg
                    res["roofline_weight_grad"]["traffic_over_algorithmic"] = res["roofline_weight_grad"]["traffic"] / avg
            cand["roofline_weight_grad"] = res["roofline_weight_grad"]["ms_per_step"]
        elif tn_:
            per = per_step.get("gemm_tn_mfma", 0.0)
            ach = tw / (tu * 1e-6) / 1e12
            res["roofline_weight_grad"] = {"bound": "mfma", "kernel": "gemm_tn_kernel: dW += dZ^T H over the steps of a group (fp32 v_mfma_f32_32x32x2_f32), "
                                                                      "%.1f launches per step, side stream" % per,
                                           "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "avg_us": tu, "launches": tn_,
                                           "traffic": pmc_traffic("gemm_tn", args.config, args.precision),
                                           "algorithmic_flops": tw, "launches_per_step": per, "ms_per_step": tu * per / 1e3}
            cand["roofline_weight_grad"] = res["roofline_weight_grad"]["ms_per_step"]
        top = max(cand, key=cand.get)
        res["roofline"] = dict(res[top], dominant_of={k: round(v, 3) for k, v in cand.items()},
                               what="the kernel with the largest time per training step (ms_per_step = avg_us x launches per step); "
                                    "also listed under '%s'" % top)
    if "roofline_aggregation" not in res and per_step.get("node_chain", 0) > 0:
        # no separate aggregation launch in this mode: node_agg_fn runs inside node_chain_kernel (aggregate -> node update -> next
        # projections, csrc/node_chain.hip); the HBM-streaming aggregation figure is other_configs.cfgE_*_unfused_aggregation
        res["roofline_aggregation"] = {"fused_into": "node_chain_kernel", "launches": 0}
    elif "roofline_aggregation" not in res and chain == 2 and agg_n == 0:
        res["roofline_aggregation"] = {"fused_into": "edge_chain_bf16_kernel", "launches": 0}
    for v in res.values():
        if isinstance(v, dict) and "traffic" in v:
            v["traffic_source"] = TRAFFIC_SOURCE if v["traffic"] else None
    if mode == "train":
        for key, kern in (("roofline_fwd_chain", "edge_chain_kernel"), ("roofline_bwd_chain", "edge_chain_bwd_kernel"), ("roofline_weight_grad", "wgrad_panel_kernel")):
            ctr = pmc_counters(kern, args.config, args.precision)
            if ctr and isinstance(res.get(key), dict):
                res[key]["counters"] = ctr
                res[key]["mfma_busy_frac"] = ctr.get("mfma_busy_frac")
        if isinstance(res.get("roofline"), dict) and res["roofline"].get("dominant_of"):
            top = max(res["roofline"]["dominant_of"], key=res["roofline"]["dominant_of"].get)
            if isinstance(res.get(top), dict) and "counters" in res[top]:
                res["roofline"]["counters"] = res[top]["counters"]
                res["roofline"]["mfma_busy_frac"] = res[top].get("mfma_busy_frac")
        ctr = pmc_counters("node_chain_kernel", args.config, args.precision)
        if ctr and isinstance(res.get("roofline_aggregation"), dict):
            res["roofline_aggregation"]["counters"] = ctr
    return res


def ordered_line(out):
    """The JSON line with the long descriptive objects FIRST and the contract keys plus the compact figures LAST, so that a
    2,000-character tail of the line carries the numbers.  Nothing is dropped: the full per-kernel
    roofline objects stay under `details`."""
    def short(v, n=60):
        return v if not isinstance(v, str) or len(v) <= n else v[:n - 3] + "..."

    def compact_roofline(r):
        if not isinstance(r, dict):
            return r
        keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "avg_us", "launches_per_step",
                "ms_per_step", "algorithmic_bytes", "algorithmic_flops", "traffic_over_algorithmic", "mfma_frac_of_bf16_peak",
                "dominant_of", "fused_into", "mfma_busy_frac")
        c = {k: r[k] for k in keep if k in r}
        if "kernel" in c:
            c["kernel"] = short(c["kernel"].split(":")[0].split(" (")[0], 60)
        if c.get("traffic_source"):
            c["traffic_source"] = "committed rocprofv3 --pmc pass (profiles/), not this run"
        return c

    details = {}
    for k in ("roofline_fwd_chain", "roofline_bwd_chain", "roofline_weight_grad", "roofline_aggregation"):
        if k in out:
            details[k] = out[k]
    if "roofline" in out:
        details["roofline_full"] = out["roofline"]
    if isinstance(out.get("dtype"), str) and len(out["dtype"]) > 40:
        details["dtype_note"] = out["dtype"]
    for k in ("fp32_mfma", "fp32_split", "allreduce_what"):
        if k in out:
            details[k] = out[k]
    line = {}
    if "other_configs" in out:
        oc = {}
        for name, v in out["other_configs"].items():
            if isinstance(v, dict):
                v = dict(v)
                for kk in list(v):
                    if kk.startswith("roofline") and isinstance(v[kk], dict):
                        v[kk] = compact_roofline(v[kk])
                if "workload" in v:
                    v["workload"] = short(v["workload"], 110)
            oc[name] = v
        line["other_configs"] = oc
    line["details"] = details
    # ---- the contract keys and the compact figures: the tail of the line --------------------------------------------------
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline"):
        line[k] = out[k]
    line["dtype"] = out["dtype"].split(" (")[0] if isinstance(out.get("dtype"), str) else out.get("dtype")
    line["data"] = out["data"]
    line["config"] = out["config"]
    for k in ("graph_prep_ms", "graph_prep_steady_ms", "edge_steps_per_ms", "allreduce_ms", "allreduce_bytes", "bus_gbs", "allreduce_error"):
        if k in out:
            line[k] = out[k]
    if "cpu_baseline" in out:
        cb = dict(out["cpu_baseline"])
        cb["sample"] = short(cb.get("sample", ""), 200)
        line["cpu_baseline"] = cb
    if "roofline" in out:
        line["roofline"] = compact_roofline(out["roofline"])
    summary = {}
    if "forward_edges_per_ms" in out:
        summary["forward_edges_per_ms"] = round(out["forward_edges_per_ms"], 1)
    for k in ("fp32_mfma", "fp32_split"):
        if isinstance(out.get(k), dict) and "value" in out[k]:
            summary[k + "_edges_per_ms"] = round(out[k]["value"], 1)
            if "forward_edges_per_ms" in out[k]:
                summary[k + "_forward_edges_per_ms"] = round(out[k]["forward_edges_per_ms"], 1)
    for k in ("roofline_fwd_chain", "roofline_bwd_chain", "roofline_weight_grad", "roofline_aggregation"):
        r = out.get(k)
        if isinstance(r, dict) and "frac" in r:
            summary[k.replace("roofline_", "") + "_frac"] = round(r["frac"], 4)
            if "ms_per_step" in r:
                summary[k.replace("roofline_", "") + "_ms_per_step"] = round(r["ms_per_step"], 3)
            if r.get("traffic_over_algorithmic"):
                summary[k.replace("roofline_", "") + "_traffic_over_algorithmic"] = round(r["traffic_over_algorithmic"], 2)
            if r.get("mfma_busy_frac") is not None:
                summary[k.replace("roofline_", "") + "_mfma_busy_frac"] = r["mfma_busy_frac"]
        elif isinstance(r, dict) and "fused_into" in r:
            summary[k.replace("roofline_", "") + "_fused_into"] = r["fused_into"]
    for name, v in (out.get("other_configs") or {}).items():
        if isinstance(v, dict) and "ms_per_step" in v:
            summary[name + "_ms"] = round(v["ms_per_step"], 4)
            for kk in ("roofline", "roofline_weight_grad", "roofline_aggregation", "roofline_fwd_chain", "roofline_bwd_chain"):
                if isinstance(v.get(kk), dict) and "frac" in v[kk]:
                    summary[name + "_" + kk.replace("roofline_", "").replace("roofline", "top") + "_frac"] = round(v[kk]["frac"], 4)
        elif isinstance(v, dict) and ("error" in v or "skipped" in v):
            summary[name] = short(v.get("error") or v.get("skipped"), 80)
    line["summary"] = summary
    return line


if __name__ == "__main__":
    main()
