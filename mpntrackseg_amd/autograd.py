"""torch.autograd glue around the native forward / backward entry points (plumbing only)."""
import torch

from . import capi


def avg_pool_native(x):
    """[N,C,H,W] -> [N,C] through ``mpnhip_avgpool`` (reference mpn.py:351-352)."""
    lib = capi.load()
    x = capi.f32c(x)
    n, c = x.shape[0], x.shape[1]
    hw = int(x.shape[2] * x.shape[3])
    y = torch.empty((n, c), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        capi.check(lib.mpnhip_avgpool(capi.ptr(x), n * c, hw, capi.ptr(y), capi.stream_ptr()), "mpnhip_avgpool")
    return y


def mpn_hot_path_autograd(model, x, edge_index, edge_attr, holder=None):
    raise capi.MpnhipError("training through the native path needs mpnhip_backward, which this build does not have yet")
