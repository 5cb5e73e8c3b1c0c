"""Graph file format for precomputed detection graphs (SURVEY.md section 8d cfg-C, row f-4): one ``.npz`` per sequence
graph, so that a real MOTS20 / KITTIMOTS graph built by the reference's data pipeline (``seq_processor.py`` embeddings +
``MOTGraph.construct_graph_object``) can be fed to this path -- and to ``bench.py --graph-file`` -- without the dataset
classes.

Arrays (the attributes ``MOTMPNet.forward`` reads, ``mpn.py:349``, plus the training / inference extras):
    x            float32 [N, C]          node inputs ALREADY avg-pooled (``mpn.py:351-352``): the reference stores
                                         [N, 2048, 8, 4] ReID maps per detection (``utils/rgb.py:150-188``); pooling once at
                                         write time cuts the file and the host-to-device copy 32x
    edge_index   int64   [2, E]          ``[pairs | flipped pairs]`` (``data/mot_graph.py:312``)
    edge_attr    float32 [E, F]
    edge_labels  float32 [E]             optional (training)
    reid_emb_dists float32 [E]           optional (inference: per-window kNN pruning)
    frame        int64   [N]             optional (sliding windows)
``save_graph`` accepts 4-D ``x`` and pools it (on the HIP device through ``mpnhip_avgpool`` when it is a device tensor,
with numpy otherwise: writing files is host-side IO, not the product path)."""
import numpy as np
import torch

REQUIRED = ("x", "edge_index", "edge_attr")
OPTIONAL = ("edge_labels", "reid_emb_dists", "frame")


def _np(v):
    return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)


def save_graph(path, x, edge_index, edge_attr, edge_labels=None, reid_emb_dists=None, frame=None):
    if isinstance(x, torch.Tensor) and x.dim() == 4 and x.is_cuda:
        from .mpn import avg_pool
        x = avg_pool(x)
    x = _np(x)
    if x.ndim == 4:
        x = x.mean(axis=(2, 3))
    ei = _np(edge_index).astype(np.int64)
    ea = _np(edge_attr).astype(np.float32)
    if x.ndim != 2 or ei.ndim != 2 or ei.shape[0] != 2 or ea.ndim != 2 or ea.shape[0] != ei.shape[1]:
        raise ValueError("save_graph: x [N, C] (or [N, C, H, W]), edge_index [2, E], edge_attr [E, F] expected")
    if ei.size and (ei.min() < 0 or ei.max() >= x.shape[0]):
        raise ValueError("save_graph: edge_index refers to a node outside [0, N)")
    out = {"x": x.astype(np.float32), "edge_index": ei, "edge_attr": ea}
    for name, v, dt in (("edge_labels", edge_labels, np.float32), ("reid_emb_dists", reid_emb_dists, np.float32),
                        ("frame", frame, np.int64)):
        if v is not None:
            a = _np(v).astype(dt).reshape(-1)
            if a.shape[0] != (x.shape[0] if name == "frame" else ei.shape[1]):
                raise ValueError("save_graph: %s has the wrong length" % name)
            out[name] = a
    np.savez_compressed(path, **out)


def load_graph(path, device=None):
    """Returns a dict of tensors (on ``device`` when given): the REQUIRED arrays and whichever OPTIONAL ones the file has."""
    z = np.load(path)
    missing = [k for k in REQUIRED if k not in z.files]
    if missing:
        raise ValueError("%s: not a graph file (missing %s)" % (path, ", ".join(missing)))
    out = {}
    for k in REQUIRED + OPTIONAL:
        if k in z.files:
            t = torch.from_numpy(z[k])
            out[k] = t.to(device) if device is not None else t
    n, e = out["x"].shape[0], out["edge_index"].shape[1]
    if out["x"].dim() != 2 or out["edge_index"].shape[0] != 2 or out["edge_attr"].shape[0] != e:
        raise ValueError("%s: inconsistent array shapes" % path)
    if e and (int(out["edge_index"].min()) < 0 or int(out["edge_index"].max()) >= n):
        raise ValueError("%s: edge_index refers to a node outside [0, N)" % path)
    return out
