"""Stand-in for the container types the reference's callers hand to ``MOTMPNet.forward`` (test infrastructure only).

``Graph`` (reference data/mot_graph.py:21-83) subclasses ``torch_geometric.data.Data``; the training loop takes mini-batches out of a
``torch_geometric.data.DataLoader`` (pl_module/pl_module.py:6,54), i.e. ``Batch`` objects.  torch_geometric (pinned 1.5.0,
environment.yml:149) is an un-vendored third-party dependency that is not installed in this image, so the part of its published
attribute protocol the path can observe is restated here: attributes live in ``__dict__``; ``keys`` lists the non-None ones whose name is
not ``__dunder__``; ``obj[key]``; ``apply`` / ``to`` walk ``keys`` and touch tensors (recursing into lists / tuples / dicts);
``Batch.from_data_list`` concatenates every key along ``__cat_dim__`` (-1 for names containing "index" / "face", else 0), shifts
index-like keys by the running node count (``__inc__``) and adds the ``batch`` vector.  ``Graph`` adds the reference's bulk type-change
helpers; note that its ``to(device)`` returns None (mot_graph.py:71-72)."""
import re

import numpy as np
import torch


class Data:
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, pos=None, **kwargs):
        self.x, self.edge_index, self.edge_attr, self.y, self.pos = x, edge_index, edge_attr, y, pos
        for k, v in kwargs.items():
            self[k] = v

    def __getitem__(self, key):
        return getattr(self, key, None)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    @property
    def keys(self):
        return [k for k in self.__dict__ if self[k] is not None and not (k[:2] == "__" and k[-2:] == "__")]

    def __contains__(self, key):
        return key in self.keys

    def __iter__(self):
        for k in sorted(self.keys):
            yield k, self[k]

    @property
    def num_nodes(self):
        if getattr(self, "__num_nodes__", None) is not None:
            return self.__dict__["__num_nodes__"]
        for k in ("x", "pos"):
            if self[k] is not None:
                return self[k].size(0)
        return int(self.edge_index.max()) + 1 if self.edge_index is not None and self.edge_index.numel() else 0

    def __cat_dim__(self, key, value):
        return -1 if re.search("(index|face)", key) else 0

    def __inc__(self, key, value):
        return self.num_nodes if re.search("(index|face)", key) else 0

    def _apply_item(self, item, fn):
        if torch.is_tensor(item):
            return fn(item)
        if isinstance(item, (tuple, list)):
            return [self._apply_item(v, fn) for v in item]
        if isinstance(item, dict):
            return {k: self._apply_item(v, fn) for k, v in item.items()}
        return item

    def apply(self, fn, *keys):
        for k in (keys or self.keys):
            if self[k] is not None:
                self[k] = self._apply_item(self[k], fn)
        return self

    def to(self, device, *keys):
        return self.apply(lambda t: t.to(device), *keys)


class Graph(Data):
    """The reference's sample type (data/mot_graph.py:21-83): Data + bulk type changes over a fixed list of attribute names."""
    DATA_ATTRS = ("x", "x_ext", "edge_attr", "edge_index", "mask_attr", "node_names", "edge_labels", "edge_preds", "reid_emb_dists")

    def _change(self, fn):
        for name in self.DATA_ATTRS:
            v = getattr(self, name, None)
            if v is not None:
                setattr(self, name, fn(v))

    def float(self):
        self._change(lambda t: t.float())
        return self

    def cpu(self):
        self._change(lambda t: t.cpu())
        return self

    def cuda(self):
        self._change(lambda t: t.cuda())
        return self

    def to(self, device):               # (returns None in the reference)
        self._change(lambda t: t.to(device))

    def numpy(self):
        self._change(lambda t: t if isinstance(t, np.ndarray) else t.detach().cpu().numpy())
        return self


class Batch(Data):
    @staticmethod
    def from_data_list(data_list):
        keys = sorted(set().union(*[set(d.keys) for d in data_list]))
        assert "batch" not in keys
        out = Batch()
        out.__dict__["__data_class__"] = data_list[0].__class__
        cols = {k: [] for k in keys}
        cols["batch"] = []
        shift = {k: 0 for k in keys}
        for i, d in enumerate(data_list):
            for k in d.keys:
                item = d[k]
                if torch.is_tensor(item) and item.dtype != torch.bool:
                    item = item + shift[k] if shift[k] else item
                cols[k].append(item)
                shift[k] += d.__inc__(k, d[k])
            cols["batch"].append(torch.full((d.num_nodes,), i, dtype=torch.long))
        for k in list(cols):
            v = cols[k][0]
            if torch.is_tensor(v):
                out[k] = torch.cat(cols[k], dim=data_list[0].__cat_dim__(k, v))
            elif isinstance(v, (int, float)):
                out[k] = torch.tensor(cols[k])
            else:
                out[k] = cols[k]
        return out

    @property
    def num_graphs(self):
        return int(self.batch[-1]) + 1
