"""Host-side mirror of the reference's ``MLP`` (``/root/reference/src/mot_neural_solver/models/mlp.py:4-28``).

Same constructor arguments, same ``fc_layers`` Sequential (so ``state_dict`` keys are identical and
reference checkpoints load), but ``forward`` runs the layers through the HIP kernels behind the C ABI
(``mpnhip_mlp_forward``) instead of ``nn.Linear``'s aten addmm.
"""
import torch
from torch import nn

from . import capi


class MLP(nn.Module):
    def __init__(self, input_dim, fc_dims, dropout_p=0.4, use_batchnorm=False):
        super(MLP, self).__init__()
        assert isinstance(fc_dims, (list, tuple)), 'fc_dims must be either a list or a tuple, but got {}'.format(
            type(fc_dims))
        layers = []
        for dim in fc_dims:
            layers.append(nn.Linear(input_dim, dim))
            if use_batchnorm and dim != 1:
                layers.append(nn.BatchNorm1d(dim))
            if dim != 1:
                layers.append(nn.ReLU(inplace=True))
            if dropout_p != 0 and dim != 1:
                layers.append(nn.Dropout(p=dropout_p))
            input_dim = dim
        self.fc_layers = nn.Sequential(*layers)
        self.input_dim_ = None
        self.fast_path = not use_batchnorm and dropout_p == 0

    def linears(self):
        return [m for m in self.fc_layers if isinstance(m, nn.Linear)]

    def require_fast_path(self):
        """BatchNorm1d / Dropout (mlp.py:14,20; never enabled in the shipped configs, tracking_cfg.yaml:150-167): in eval mode
        Dropout is the identity and BatchNorm is an affine map of the Linear's output that folds into its weight and bias
        (``effective_linears``), so inference runs on the fused kernels.  TRAINING with them -- batch statistics, random masks
        and their gradients -- runs layer by layer (``modular.py``); the FUSED path refuses such a module in training mode."""
        if not self.fast_path and self.training:
            raise capi.MpnhipError(
                "BatchNorm / Dropout in training mode cannot run in the fused kernels (batch statistics need every row of a layer "
                "first): MOTMPNet.forward / MLP.forward take the layer-by-layer path (mpntrackseg_amd/modular.py) for it")

    def effective_linears(self):
        """[(weight, bias)] of the Linear layers as the kernels see them: eval-mode BatchNorm1d folded in
        (y = gamma (W x + b - mean) / sqrt(var + eps) + beta  =>  W' = s W, b' = s (b - mean) + beta, s = gamma / sqrt(var + eps))."""
        out, mods = [], list(self.fc_layers)
        for i, m in enumerate(mods):
            if not isinstance(m, nn.Linear):
                continue
            w, b = m.weight.detach(), m.bias.detach()
            bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d) else None
            if bn is not None:
                s = (bn.weight.detach() if bn.affine else torch.ones_like(bn.running_var)) / torch.sqrt(bn.running_var + bn.eps)
                beta = bn.bias.detach() if bn.affine else torch.zeros_like(bn.running_mean)
                w, b = (w * s.view(-1, 1)).contiguous(), ((b - bn.running_mean) * s + beta).contiguous()
            out.append((w, b))
        return out

    def c_struct(self, keep, grads=None):
        """``mpnhip_mlp`` for this module; ``grads`` maps id(param) -> gradient buffer (same shape)."""
        self.require_fast_path()
        if grads is not None and not self.fast_path:
            raise capi.MpnhipError("gradients of an MLP with BatchNorm / Dropout are not covered by the HIP path")
        s = capi.Mlp()
        lin = self.effective_linears()
        for w, b in lin:
            capi.require_device(w, b)
        capi.fill_mlp(s, lin, keep=keep)
        if grads is not None:
            for i, l in enumerate(self.linears()):
                s.grad_weight[i] = grads[id(l.weight)].data_ptr()
                s.grad_bias[i] = grads[id(l.bias)].data_ptr()
        return s

    def forward(self, input):
        """models/mlp.py:27-28.  Inference: all layers in one native call; with autograd (or BatchNorm / Dropout in training
        mode): ``modular.mlp_forward``.  (``MOTMPNet.forward`` evaluates its MLPs fused and owns its own backward.)"""
        capi.require_device(input)
        needs_grad = torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad or (self.training and not self.fast_path):
            # autograd, and training-mode BatchNorm / Dropout: layer by layer, every layer a HIP op with a hand-written gradient
            from .modular import mlp_forward
            return mlp_forward(self, input)
        lib = capi.load()
        x = capi.f32c(input)
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        keep = []
        s = self.c_struct(keep)
        if x2.shape[1] != s.in_dim:
            raise capi.MpnhipError(f"MLP input dim {x2.shape[1]} != {s.in_dim}")
        m = x2.shape[0]
        y = torch.empty((m, s.out_dims[s.n_layers - 1]), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            ws = capi.workspace(lib.mpnhip_mlp_workspace_bytes(s, m), x.device, "mlp")
            capi.check(lib.mpnhip_mlp_forward(s, capi.ptr(x2), capi.ptr(y), m, capi.ptr(ws), ws.numel(),
                                              capi.stream_ptr()), "mpnhip_mlp_forward")
        return y.reshape(*lead, y.shape[-1])
