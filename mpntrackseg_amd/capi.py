"""ctypes binding of the C ABI in ``include/mpnhip.h`` (``mpntrackseg_amd/csrc/libmpnhip.so``).

This is the only place the Python host code touches native code.  PyTorch is used for device
memory (``tensor.data_ptr()``) and the current HIP stream -- plumbing; every numeric op of the hot
path runs in the hand-written HIP kernels behind these entry points.  There is NO CPU fallback:
if the library cannot be loaded, or a tensor is not on a HIP device, the call raises.
"""
import ctypes as C
import os

import torch

MAX_LAYERS = 8
AGG_CODE = {"sum": 0, "mean": 1, "max": 2}

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmpnhip.so")
_lib = None


class MpnhipError(RuntimeError):
    pass


# mpnhip_model.precision (include/mpnhip.h)
PRECISIONS = {'fp32': 0, 'bf16': 1, 'fp32_split': 2}


class Mlp(C.Structure):
    _fields_ = [
        ("n_layers", C.c_int),
        ("in_dim", C.c_int),
        ("out_dims", C.c_int * MAX_LAYERS),
        ("weight", C.c_void_p * MAX_LAYERS),
        ("bias", C.c_void_p * MAX_LAYERS),
        ("grad_weight", C.c_void_p * MAX_LAYERS),
        ("grad_bias", C.c_void_p * MAX_LAYERS),
    ]


class Model(C.Structure):
    _fields_ = [
        ("dn", C.c_int), ("de", C.c_int), ("reattach_nodes", C.c_int), ("reattach_edges", C.c_int),
        ("agg", C.c_int), ("num_enc_steps", C.c_int),
        ("enc_node", Mlp), ("enc_edge", Mlp), ("edge", Mlp), ("flow_in", Mlp), ("flow_out", Mlp),
        ("node", Mlp), ("classifier", Mlp), ("precision", C.c_int), ("weights_prepacked", C.c_int),
    ]


# name -> (restype, argtypes); mirrors include/mpnhip.h one to one (tests/test_capi_symbols.py
# checks that every function the header declares is listed here and exported by the library)
_P, _I, _L, _Z = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
SIGNATURES = {
    "mpnhip_version": (C.c_char_p, []),
    "mpnhip_last_error": (C.c_char_p, []),
    "mpnhip_graph_bytes": (_Z, [_I, _L]),
    "mpnhip_graph_prep_workspace_bytes": (_Z, [_I, _L]),
    "mpnhip_graph_prep": (_I, [_P, _I, _L, _P, _Z, _P, _Z, _P]),
    "mpnhip_graph_prep_forward": (_I, [_P, _I, _L, _P, _Z, _P, _Z, _P]),
    "mpnhip_graph_status": (_I, [_P, _I, _L, C.POINTER(C.c_int32), _P]),
    "mpnhip_forward_workspace_bytes": (_Z, [C.POINTER(Model), _I, _L, _I]),
    "mpnhip_forward": (_I, [C.POINTER(Model), _P, _I, _L, _P, _P, _P, _P, _P, _P, _Z, _I, _P]),
    "mpnhip_backward_workspace_bytes": (_Z, [C.POINTER(Model), _I, _L]),
    "mpnhip_backward": (_I, [C.POINTER(Model), _P, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P, _Z, _P]),
    "mpnhip_meta_layer_workspace_bytes": (_Z, [C.POINTER(Model), _I, _L]),
    "mpnhip_meta_layer_forward": (_I, [C.POINTER(Model), _P, _I, _L, _P, _P, _P, _P, _P, _Z, _P]),
    "mpnhip_segment_reduce_workspace_bytes": (_Z, [_L, _I]),
    "mpnhip_segment_reduce": (_I, [_P, _P, _L, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "mpnhip_linear": (_I, [_P, _L, _P, _P, _P, _L, _L, _I, _I, _I, _P]),
    "mpnhip_mlp_workspace_bytes": (_Z, [C.POINTER(Mlp), _L]),
    "mpnhip_mlp_forward": (_I, [C.POINTER(Mlp), _P, _P, _L, _P, _Z, _P]),
    "mpnhip_tracking_loss_workspace_bytes": (_Z, [_I, _L]),
    "mpnhip_tracking_loss": (_I, [_P, _P, _I, _L, _I, C.c_float, _P, _P, _P, _Z, _P]),
    "mpnhip_step_metrics": (_I, [_P, _I, _L, _P, _P, _P, _P]),
    "mpnhip_attention_aggregate": (_I, [_P, _I, _L, _P, _L, _P, _P, _P, _P, _P]),
    "mpnhip_attention_aggregate_backward": (_I, [_P, _I, _L, _P, _L, _P, _P, _P, _P, _I, _P, _P, _P]),
    "mpnhip_avgpool": (_I, [_P, _L, _I, _P, _P]),
    "mpnhip_adam_step": (_I, [_P, _P, _P, _P, _L, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _I, _P]),
    "mpnhip_time_valid_conn_workspace_bytes": (_Z, [_I]),
    "mpnhip_time_valid_conn_count": (_I, [_P, _I, _L, _P, _P, _Z, _P]),
    "mpnhip_time_valid_conn_fill": (_I, [_P, _I, _L, _P, _L, _P, _P]),
    "mpnhip_edge_features": (_I, [_P, _L, _I, _P, C.c_float, _P, _P, _P, _P, _P, _P]),
    "mpnhip_pairwise_distance": (_I, [_P, _L, _I, _P, _L, C.c_float, _P, _P]),
    "mpnhip_embedding_keep": (_I, [_P, _L, _L, _P, _L, _P, _P]),
    "mpnhip_embedding_check": (_I, [_P, _L, _P, _L, _P, _P, _P]),
    "mpnhip_knn_mask_workspace_bytes": (_Z, [_L, _I]),
    "mpnhip_knn_mask": (_I, [_P, _P, _I, _L, _I, _I, _I, _P, _P, _Z, _P]),
    "mpnhip_window_flags": (_I, [_P, _L, _L, _L, _P, _P]),
    "mpnhip_compact_workspace_bytes": (_Z, [_L]),
    "mpnhip_compact": (_I, [_P, _L, _P, _P, _P, _Z, _P]),
    "mpnhip_gather_rows": (_I, [_P, _L, _P, _L, _I, _P, _P]),
    "mpnhip_gather_edges": (_I, [_P, _L, _P, _L, _L, _P, _P]),
    "mpnhip_window_accumulate": (_I, [_P, _P, _L, _P, _L, _I, _P, _P, _P]),
    "mpnhip_average_preds": (_I, [_P, _P, _L, _P, _P]),
    "mpnhip_profile_enable": (_I, [_I]),
    "mpnhip_edge_chain_active": (_I, [C.POINTER(Model)]),
    "mpnhip_profile_read": (_I, [C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_int),
                                 C.POINTER(C.c_float)]),
    "mpnhip_time_aggregate": (_I, [_P, _I, _L, _P, _I, _I, _P, _I, C.POINTER(C.c_float), _P]),
    "mpnhip_time_linear": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, C.POINTER(C.c_float), _P]),
}


def lib_path():
    return _LIB_PATH


def load():
    """Load libmpnhip.so (built by ``make`` / ``__graft_entry__.build()``); raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise MpnhipError(
                f"{_LIB_PATH} is missing: build it with `make` (or `python -c 'import __graft_entry__ as g; "
                "g.build()'`). mpntrackseg_amd has no CPU fallback.")
        lib = C.CDLL(_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().mpnhip_last_error().decode("utf-8", "replace")
        raise MpnhipError(f"{what} failed (code {rc}): {msg}")


def require_device(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise MpnhipError("mpntrackseg_amd runs on a HIP device only (tensor on %s); there is no CPU fallback"
                              % t.device)


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def f32c(t):
    """contiguous float32 view/copy of a tensor (no-op for the expected layout)"""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_ws_cache = {}
# weight images left at the head of a forward workspace: buffer address -> (model id, weight key); see MOTMPNet.hot_path
_packed_state = {}
# bumped by every native in-place parameter update (train.FlatAdam writes through raw pointers: no torch version bump)
_weights_epoch = [0]
_model_uid = [0]  # source of MOTMPNet._mpnhip_uid tokens


def workspace(nbytes, device, tag="ws"):
    """Grow-only per-(device, tag) scratch buffer from torch's caching allocator."""
    key = (str(device), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
        _packed_state.clear()  # a new buffer (possibly at a recycled address) holds no weight images
    return buf


def fill_mlp(dst, linears, with_grads=False, keep=None):
    """Fill an ``Mlp`` struct from a list of nn.Linear-like (weight [out,in], bias [out])."""
    assert 1 <= len(linears) <= MAX_LAYERS, "MLP depth not supported"
    dst.n_layers = len(linears)
    dst.in_dim = int(linears[0][0].shape[1])
    for i, (w, b) in enumerate(linears):
        assert w.dtype == torch.float32 and w.is_contiguous() and b.is_contiguous()
        dst.out_dims[i] = int(w.shape[0])
        dst.weight[i] = w.data_ptr()
        dst.bias[i] = b.data_ptr()
        dst.grad_weight[i] = None
        dst.grad_bias[i] = None
        if keep is not None:
            keep.extend([w, b])
    return dst


class PreparedGraph:
    """Device-side sort of an edge_index (see ``mpnhip_graph_prep``)."""

    def __init__(self, edge_index, n_nodes, validate=False, full=True):
        """``full=False``: the primary order only (inference forward); anything that differentiates or evaluates the step
        metrics needs ``full=True`` (``self.full`` records which one this is)."""
        require_device(edge_index)
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise MpnhipError("edge_index must be int64 [2, E] (reference data/mot_graph.py:312)")
        lib = load()
        ei = edge_index.contiguous()
        self.N = int(n_nodes)
        self.E = int(ei.shape[1])
        self.device = ei.device
        self.full = bool(full)
        nb = lib.mpnhip_graph_bytes(self.N, self.E)
        self.buf = torch.empty(max(nb, 256), dtype=torch.uint8, device=ei.device)
        wsb = lib.mpnhip_graph_prep_workspace_bytes(self.N, self.E)
        ws = workspace(wsb, ei.device, "prep")
        with torch.cuda.device(ei.device):
            fn = lib.mpnhip_graph_prep if full else lib.mpnhip_graph_prep_forward
            check(fn(ptr(ei), self.N, self.E, ptr(self.buf), self.buf.numel(), ptr(ws), ws.numel(), stream_ptr()),
                  "mpnhip_graph_prep")
        if validate:
            st = self.status()
            if st[0] != 0:
                raise MpnhipError("edge_index has entries outside [0, N)")

    def status(self):
        arr = (C.c_int32 * 4)()
        with torch.cuda.device(self.device):
            check(load().mpnhip_graph_status(ptr(self.buf), self.N, self.E, arr, stream_ptr()), "mpnhip_graph_status")
        return list(arr)
