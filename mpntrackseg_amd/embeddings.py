"""Precomputed-embedding files of the reference: ``utils/rgb.py:150-188``
(``load_precomputed_embeddings``) on the device, plus writers for the same on-disk format.

Format (written by the reference's ``seq_processor.py:395-562``): one ``<frame>.pt`` per frame under
``<seq_path>/processed_data/<embeddings_dir>/``; '1D' files are ``[n, 1 + D]`` with the detection id in column 0, '3D' files
``[n, 1 + C, H, W]`` with the id broadcast into channel 0.  Reading the files is host work (``torch.load``); everything after
that -- the id filter, the order check, dropping the id column / channel, and optionally the spatial mean the model would
apply first anyway (``mpn.py:351-352``) -- runs on the MI355X through the C ABI.  No CPU fallback."""
import os.path as osp

import numpy as np
import torch

from . import capi
from .capi import MpnhipError, check, ptr, stream_ptr
from .graph import compact, gather_rows


def _col(det_df, name):
    v = det_df[name]
    return np.asarray(v.values if hasattr(v, "values") else v)


def load_precomputed_embeddings(det_df, seq_info_dict, embeddings_dir, use_cuda=True, embedding_dim='1D', pooled=False):
    """Same arguments and result as the reference function; ``det_df`` needs the columns ``frame`` and ``detection_id``.
    ``pooled=True`` ('3D' only) returns the spatial mean ``[N, C]`` instead of ``[N, C, H, W]`` -- what ``MOTMPNet.forward``
    computes from it first; the graph file of ``graphfile.py`` stores that form (32x fewer bytes)."""
    assert embedding_dim in ['1D', '3D'], "Embedding dimension is not valid!"
    if not use_cuda:
        raise MpnhipError("load_precomputed_embeddings: this build selects the embeddings on the device (use_cuda=True)")
    lib = capi.load()
    dev = torch.device("cuda", torch.cuda.current_device())   # (no tensor argument: the caller's current device, like the reference's .cuda())
    path = osp.join(seq_info_dict['seq_path'], 'processed_data', embeddings_dir)
    frames = sorted(np.unique(_col(det_df, 'frame')).tolist())
    stored = torch.cat([torch.load(osp.join(path, f"{int(f)}.pt")) for f in frames], dim=0).float().contiguous()
    if (embedding_dim == '1D') != (stored.dim() == 2):
        raise MpnhipError("load_precomputed_embeddings: files are not %s embeddings (shape %s)" % (embedding_dim, tuple(stored.shape)))
    n = stored.shape[0]
    ld = stored[0].numel() if n else 1
    skip = 1 if embedding_dim == '1D' else stored.shape[2] * stored.shape[3]     # id column / id channel
    stored = stored.to(dev)
    det_ids = np.asarray(_col(det_df, 'detection_id'), dtype=np.int64)
    if det_ids.size and (det_ids.min() < 0 or det_ids.max() >= 2 ** 24):
        raise MpnhipError("load_precomputed_embeddings: detection ids must lie in [0, 2^24) (they are stored as float32)")
    ids = torch.from_numpy(det_ids.astype(np.int32)).to(dev)
    ids_sorted = torch.from_numpy(np.sort(det_ids).astype(np.int32)).to(dev)
    keep = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
    check(lib.mpnhip_embedding_keep(ptr(stored), ld, n, ptr(ids_sorted), ids_sorted.numel(), ptr(keep), stream_ptr()),
          "mpnhip_embedding_keep")
    rows, k = compact(keep[:n]) if n else (torch.empty(0, dtype=torch.int32, device=dev), 0)
    mism = torch.empty(1, dtype=torch.int32, device=dev)
    if k == ids.numel():
        check(lib.mpnhip_embedding_check(ptr(stored), ld, ptr(rows), k, ptr(ids), ptr(mism), stream_ptr()), "mpnhip_embedding_check")
    if k != ids.numel() or int(mism.item()) != 0:
        # the reference's assertion text (rgb.py:177)
        raise AssertionError("Problems loading embeddings. Indices between query and stored embeddings do not match. "
                             "BOTH SHOULD BE SORTED!")
    # rows without the id column / channel: a gather of ld - skip contiguous floats per kept row
    flat = stored.view(n, ld) if n else stored.view(0, ld)
    out = torch.empty((k, ld - skip), dtype=torch.float32, device=dev)
    if k:
        check(lib.mpnhip_gather_rows(flat.data_ptr() + 4 * skip, ld, ptr(rows), k, ld - skip, ptr(out), stream_ptr()),
              "mpnhip_gather_rows")
    if embedding_dim == '1D':
        return out
    c, h, w = stored.shape[1] - 1, stored.shape[2], stored.shape[3]
    out = out.view(k, c, h, w)
    if not pooled:
        return out
    pooled_out = torch.empty((k, c), dtype=torch.float32, device=dev)
    if k:
        check(lib.mpnhip_avgpool(ptr(out), k * c, h * w, ptr(pooled_out), stream_ptr()), "mpnhip_avgpool")
    return pooled_out


def write_frame_embeddings(seq_path, embeddings_dir, frame, detection_id, embeddings):
    """Writes ``<frame>.pt`` files in the reference's format (id in column / channel 0).  ``frame`` / ``detection_id``: [n] ints
    (rows grouped by frame, ids ascending inside a frame, as ``seq_processor.py`` stores them); ``embeddings``: [n, D] or
    [n, C, H, W]."""
    import os
    d = osp.join(seq_path, 'processed_data', embeddings_dir)
    os.makedirs(d, exist_ok=True)
    frame = np.asarray(frame)
    ids = torch.as_tensor(np.asarray(detection_id), dtype=torch.float32)
    emb = torch.as_tensor(embeddings, dtype=torch.float32).cpu()
    for f in np.unique(frame):
        sel = torch.from_numpy(np.nonzero(frame == f)[0])
        e = emb[sel]
        if e.dim() == 2:
            t = torch.cat([ids[sel].view(-1, 1), e], dim=1)
        else:
            idc = ids[sel].view(-1, 1, 1, 1).expand(-1, 1, e.shape[2], e.shape[3])
            t = torch.cat([idc, e], dim=1)
        torch.save(t, osp.join(d, f"{int(f)}.pt"))
