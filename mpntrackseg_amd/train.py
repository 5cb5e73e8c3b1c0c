"""Data-parallel training step: one graph per GPU, flat gradient bucket, one RCCL all-reduce.

The reference trains on one GPU with ``accumulate_grad_batches: 8`` (configs/tracking_cfg.yaml:4,
scripts/train.py:76); 8 GPUs x 1 graph with gradient averaging is the same optimisation step executed
in space instead of time."""
from . import capi


def backward_available():
    """True when libmpnhip.so carries the hand-written backward."""
    lib = capi.load()
    m = capi.Model()
    return lib.mpnhip_backward_workspace_bytes(m, 0, 0) != 0 or getattr(lib, "_has_bwd", False)


def shard_indices(n_items, rank, world_size):
    """Round-robin shard of sample (graph / sequence) ids: rank r takes r, r+W, r+2W, ..."""
    return list(range(rank, n_items, world_size))


class TrainStep:
    def __init__(self, model, world_size=1):
        raise capi.MpnhipError("TrainStep needs mpnhip_backward")
