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
        if not self.fast_path:
            raise capi.MpnhipError(
                "BatchNorm / Dropout inside the MPN MLPs is not covered by the HIP path "
                "(all shipped reference configs use use_batchnorm=False, dropout_p=0: configs/tracking_cfg.yaml:150-167)")

    def c_struct(self, keep, grads=None):
        """``mpnhip_mlp`` for this module; ``grads`` maps id(param) -> gradient buffer (same shape)."""
        self.require_fast_path()
        s = capi.Mlp()
        lin = [(l.weight.detach(), l.bias.detach()) for l in self.linears()]
        for w, b in lin:
            capi.require_device(w, b)
        capi.fill_mlp(s, lin, keep=keep)
        if grads is not None:
            for i, l in enumerate(self.linears()):
                s.grad_weight[i] = grads[id(l.weight)].data_ptr()
                s.grad_bias[i] = grads[id(l.bias)].data_ptr()
        return s

    def forward(self, input):
        """models/mlp.py:27-28.  Inference-only at operator level (the fused MOTMPNet path owns autograd)."""
        self.require_fast_path()
        capi.require_device(input)
        if torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise capi.MpnhipError("operator-level MLP.forward has no autograd; use MOTMPNet.forward for training "
                                   "or wrap the call in torch.no_grad()")
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
