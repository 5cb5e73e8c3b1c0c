"""Stock-torch evaluation of the mirror's modules composed as the reference's forward composes them (mpn.py:59-99,349-392): the
reference of tests/test_gpu_modular.py, itself pinned to the reference's own output by tests/test_oracle_golden.py (g15)."""
import torch


def scatter(src, idx, n, agg):
    """torch_scatter's scatter_add / scatter_mean / scatter_max values (empty segments 0), with autograd."""
    out = torch.zeros((n, src.shape[1]), dtype=src.dtype)
    ix = idx.view(-1, 1).expand(-1, src.shape[1])
    if agg == "sum":
        return out.scatter_add(0, ix, src)
    if agg == "mean":
        cnt = torch.bincount(idx, minlength=n).clamp(min=1).to(src.dtype).view(-1, 1)
        return out.scatter_add(0, ix, src) / cnt
    return out.scatter_reduce(0, ix, src, "amax", include_self=False)


def ref_forward(m, x, ei, ea, agg):
    """mpn.py:349-392 (tracking branch) over the stock modules of a CPU float64 copy of the mirror."""
    row, col = ei
    e = m.encoder.edge_model.fc_layers(ea)
    h = m.encoder.node_model.fc_layers(x)
    e0, h0 = e, h
    logits = []
    nm = m.MPNet.node_model
    for _ in range(int(m.num_enc_steps)):
        if m.reattach_initial_edges:
            e = torch.cat((e0, e), dim=1)
        if m.reattach_initial_nodes:
            h = torch.cat((h0, h), dim=1)
        e = m.MPNet.edge_model.edge_model.fc_layers(torch.cat([h[row], h[col], e], dim=1))       # mpn.py:67-69
        fi, fo = row > col, row < col                                                             # mpn.py:85-96
        flow_out = scatter(nm.flow_out_model.fc_layers(torch.cat([h[col[fo]], e[fo]], dim=1)), row[fo], h.shape[0], agg)
        flow_in = scatter(nm.flow_in_model.fc_layers(torch.cat([h[col[fi]], e[fi]], dim=1)), row[fi], h.shape[0], agg)
        h = nm.node_model(torch.cat((flow_in, flow_out), dim=1))                                  # mpn.py:97-99
        logits.append(m.classifier.edge_model.fc_layers(e).view(-1))
    return torch.stack(logits)
