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


def test_bench_two_ranks_on_one_gpu():
    r = run_ranks(["bench.py", "--gpus", "2", "--backend", "gloo", "--config", "A", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                   "--no-split-line"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["mode"] == "train"
