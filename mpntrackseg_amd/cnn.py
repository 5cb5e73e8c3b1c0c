"""Host-side mirror of the reference's conv stacks of the mask branch
(``/root/reference/src/mot_neural_solver/models/cnn.py``): same constructor arguments and the same
``layers`` Sequential indices, so the reference's ``state_dict`` keys load.  These stay stock PyTorch-ROCm
(MIOpen) modules -- SURVEY.md section 2.2 leaves the convolutions to the vendor library; the native part of the
mask branch is the attention aggregation (``mpnhip_attention_aggregate``)."""
from torch import nn


def _check_lists(**kw):
    for name, v in kw.items():
        assert isinstance(v, (list, tuple)), '%s must be either a list or a tuple, but got %s' % (name, type(v))
    lens = {len(v) for v in kw.values()}
    assert len(lens) == 1, 'Number of elements mismatch between dims, kernel_sizes and strides'


class CNN(nn.Module):
    """cnn.py:4-44: Conv2d (+BatchNorm2d) + ReLU (+Dropout2d) per entry of ``dims``."""

    def __init__(self, input_dim, dims, kernel_sizes, strides, paddings, dropout_p=0.4, use_batchnorm=False):
        super(CNN, self).__init__()
        _check_lists(dims=dims, kernel_sizes=kernel_sizes, strides=strides, paddings=paddings)
        mods = []
        c_in = input_dim
        for c_out, k, st, pad in zip(dims, kernel_sizes, strides, paddings):
            mods.append(nn.Conv2d(c_in, c_out, kernel_size=k, stride=st, padding=pad))
            if use_batchnorm and c_out != 1:
                mods.append(nn.BatchNorm2d(c_out))
            if c_out != 0:
                mods.append(nn.ReLU(inplace=True))
            if dropout_p != 0 and c_out != 1:
                mods.append(nn.Dropout2d(p=dropout_p))
            c_in = c_out
        self.layers = nn.Sequential(*mods)

    def forward(self, input):
        return self.layers(input)


class MaskRCNNPredictor(nn.Module):
    """cnn.py:47-84: (transposed) convolutions with a ReLU after every layer but the last."""

    def __init__(self, input_dim, dims, kernel_sizes, strides, paddings, transposed):
        super(MaskRCNNPredictor, self).__init__()
        _check_lists(dims=dims, kernel_sizes=kernel_sizes, strides=strides, paddings=paddings)
        mods = []
        c_in = input_dim
        n = len(dims)
        for i, (c_out, k, st, pad) in enumerate(zip(dims, kernel_sizes, strides, paddings)):
            conv = nn.ConvTranspose2d if transposed[i] else nn.Conv2d
            mods.append(conv(c_in, c_out, kernel_size=k, stride=st, padding=pad))
            if i < n - 1:
                mods.append(nn.ReLU(inplace=True))
            c_in = c_out
        self.layers = nn.Sequential(*mods)

    def forward(self, input):
        return self.layers(input)
