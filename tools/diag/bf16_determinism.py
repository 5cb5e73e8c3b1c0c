"""Run-to-run determinism of the bf16-operand training step (fused kernels): the same forward + backward twice, every gradient
compared bit for bit, under the A-B switches of the row stores (MPNHIP_CHAIN_BF16_DEBUG_SKIP).  python tools/diag/bf16_determinism.py [d]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch

from mpntrackseg_amd import synth
from mpntrackseg_amd.mpn import MOTMPNet
from pinned import hip_run

d = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
g = synth.make_graph(6000, 90000, seed=17, node_in_dim=64)
params = synth.model_params(d, 2, "sum", node_in_dim=64)
W = synth.make_weights(params, seed=5, gain=0.7)
model = MOTMPNet(params)
model.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
model = model.to(dev).train()
model.gemm_precision = "bf16"
r = synth.normal(13, (2, g["edge_index"].shape[1]))
for env in ("0", "8", "16", "24", "4", "0"):
    os.environ["MPNHIP_CHAIN_BF16_DEBUG_SKIP"] = env
    runs = [hip_run(model, g, r, dev) for _ in range(3)]
    bad = sorted(k for k in runs[0][1] if not all(np.array_equal(runs[0][1][k], o[1][k]) for o in runs[1:]))
    lg = all(np.array_equal(runs[0][0], o[0]) for o in runs[1:])
    print("DEBUG_SKIP=%s logits_equal %d differing gradient tensors %d %s" % (env, lg, len(bad), bad[:4]), flush=True)
