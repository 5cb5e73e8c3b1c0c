"""The N > 1 path EXECUTED on the GPU box (one MI355X is leased there, so two ranks share it over gloo): TrainStep(world_size=2)
with the bucketed, side-stream-overlapped gradient all-reduce (mpntrackseg_amd/train.py), and bench.py's N > 1 branch.
The ranks are fresh child processes started by torch.distributed.run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(script_args, nproc=2, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port())] + script_args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("L,d", [(6, 32), (2, 32), (5, 128)])
def test_two_rank_train_step_averages_gradients_and_keeps_ranks_in_step(L, d):
    """L >= 4: the backward forks its weight-gradient groups to the side stream and the message-passing bucket's all-reduce is
    ordered behind it (MPNHIP_BWD_DEFER_SIDE_JOIN); L = 2: no fork, one all-reduce."""
    r = run_ranks([os.path.join("tests", "dist_train_check.py"), str(L), str(d)])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    import re
    assert len(re.findall(r"RANK \d err", r.stdout)) == 2, r.stdout[-2000:]   # (the two ranks' lines may interleave)


def test_message_passing_bucket_collective_overlaps_the_encoder_backward():
    """the side stream reaches the message-passing bucket's all-reduce before the caller's stream has finished
    the encoder's backward (event timestamps; the best of three steps must show a positive lead)."""
    r = run_ranks([os.path.join("tests", "dist_train_check.py"), "6", "64", "overlap"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "overlap lead_us" in r.stdout


def test_invalid_graph_on_one_rank_stops_every_ranks_optimizer_step():
    r = run_ranks([os.path.join("tests", "dist_train_check.py"), "6", "32", "badgraph"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    # (two steps on the same cached prepared graph: the offending rank raises BOTH times, nobody steps, no step is counted)
    assert "RANK 1 badgraph raised [True, True] unchanged 1 applied 0" in r.stdout and \
        "RANK 0 badgraph raised [False, False] unchanged 1 applied 0" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("L,d", [(6, 32), (5, 128)])
def test_one_rank_rccl_group_runs_the_data_parallel_step(L, d):
    """no second GPU is leased, so RCCL sees ONE rank -- but the real communicator: TrainStep's bucketed,
    side-stream-ordered asynchronous all-reduces (ExternalStream + async_op + mpnhip_side_stream_join) and the guarded Adam on
    backend "nccl", equal to the plain single-rank step; an invalid graph's flag through the RCCL all-reduce."""
    r = run_ranks([os.path.join("tests", "dist_train_check.py"), str(L), str(d), "nccl1"], nproc=1)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "NCCL1 ok 1" in r.stdout and r.stdout.count("same_grads 1 same_params 1") == 3, r.stdout[-2000:]


def test_bench_one_rank_rccl_group_reports_the_allreduce():
    """bench.py --gpus 1 --backend nccl --force-collectives: a 1-rank RCCL group, the step's collectives issued, allreduce_ms printed."""
    r = run_ranks(["bench.py", "--gpus", "1", "--backend", "nccl", "--force-collectives", "--config", "D", "--steps", "5", "--warmup", "3",
                   "--no-cpu-baseline", "--no-split-line", "--no-extras"], nproc=1)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["allreduce_ms"] > 0 and d["allreduce_bytes"] > 0, d
    assert "rccl" in d["config"]["parallelism"].lower() or "nccl" in d["config"]["parallelism"].lower(), d["config"]


@pytest.mark.parametrize("cfg", ["A", "D"])
def test_bench_two_ranks_on_one_gpu(cfg):
    """bench.py's N > 1 branch; cfg-D is the configuration BASELINE.json names for the 8-GPU run (one KITTIMOTS-like graph per rank)."""
    r = run_ranks(["bench.py", "--gpus", "2", "--backend", "gloo", "--config", cfg, "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                   "--no-split-line"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["mode"] == "train"
    # SURVEY.md section 8e "Reporting": the all-reduce alone and the bus bandwidth it corresponds to
    assert d["allreduce_ms"] > 0 and d["allreduce_bytes"] > 0 and d["bus_gbs"] > 0, d


def test_bench_eight_ranks_on_one_gpu_reports_the_sum_of_the_ranks_edges():
    """the exact layout of BASELINE.json configs[3] -- eight ranks, one KITTIMOTS-like graph each (cfg-D: every
    rank builds its OWN kNN graph, seed 1 + rank, so the edge counts differ) -- on the one leased GPU over gloo (functional only).
    `value` must be the sum of the ranks' edge counts over the step time, not rank 0's count times eight."""
    r = run_ranks(["bench.py", "--gpus", "8", "--backend", "gloo", "--config", "D", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                   "--no-split-line", "--no-extras"], nproc=8, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    per_rank = d["config"]["edges_per_rank"]
    assert d["n_gpus"] == 8 and len(per_rank) == 8 and all(e > 0 for e in per_rank), d["config"]
    assert len(set(per_rank)) > 1, per_rank                       # the ranks' graphs really differ
    assert d["config"]["edges_all_ranks"] == sum(per_rank)
    assert abs(d["value"] - sum(per_rank) / d["ms_per_step"]) <= 1e-6 * d["value"], (d["value"], per_rank, d["ms_per_step"])
    assert d["scaling"] == "weak" and d["allreduce_ms"] > 0


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2 ...` with NO launcher and no WORLD_SIZE in the env -- the parent starts the two
    ranks as child processes (bench.spawn_ranks) and rank 0 prints the one line with n_gpus = 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--config", "D", "--steps", "3", "--warmup", "2"],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["config"]["edges_per_rank"]) == 2 and d["scaling"] == "weak" and d["value"] > 0, d
    assert d["allreduce_ms"] > 0 and d["bus_gbs"] > 0 and d["config"]["mode"] == "train", d

