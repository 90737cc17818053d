"""Conv stacks with the module/parameter naming of rlpyt/models/conv2d.py:8-117
(``conv.<i>`` inside ``Conv2dModel``; ``conv`` + ``head`` inside ``Conv2dHeadModel``), so
state dicts interchange with the reference."""
import torch

from .mlp import MlpModel
from .utils import conv2d_output_shape


class Conv2dModel(torch.nn.Module):
    def __init__(self, in_channels, channels, kernel_sizes, strides, paddings=None,
                 nonlinearity=torch.nn.ReLU, use_maxpool=False, head_sizes=None):
        super().__init__()
        paddings = [0] * len(channels) if paddings is None else paddings
        assert len(channels) == len(kernel_sizes) == len(strides) == len(paddings)
        ins = [in_channels] + list(channels[:-1])
        pool_strides = strides if use_maxpool else [1] * len(strides)
        conv_strides = [1] * len(strides) if use_maxpool else strides
        seq = []
        for ic, oc, k, s, p, ms in zip(ins, channels, kernel_sizes, conv_strides, paddings,
                                       pool_strides):
            seq += [torch.nn.Conv2d(ic, oc, kernel_size=k, stride=s, padding=p), nonlinearity()]
            if ms > 1:
                seq.append(torch.nn.MaxPool2d(ms))
        self.conv = torch.nn.Sequential(*seq)

    def forward(self, input):
        return self.conv(input)

    def conv_out_size(self, h, w, c=None):
        for m in self.conv.children():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.MaxPool2d)):
                h, w = conv2d_output_shape(h, w, m.kernel_size, m.stride, m.padding)
            if isinstance(m, torch.nn.Conv2d):
                c = m.out_channels
        return h * w * c


class Conv2dHeadModel(torch.nn.Module):
    def __init__(self, image_shape, channels, kernel_sizes, strides, hidden_sizes,
                 output_size=None, paddings=None, nonlinearity=torch.nn.ReLU,
                 use_maxpool=False):
        super().__init__()
        c, h, w = image_shape
        self.conv = Conv2dModel(c, channels, kernel_sizes, strides, paddings=paddings,
                                nonlinearity=nonlinearity, use_maxpool=use_maxpool)
        n = self.conv.conv_out_size(h, w)
        if hidden_sizes or output_size:
            self.head = MlpModel(n, hidden_sizes, output_size=output_size,
                                 nonlinearity=nonlinearity)
            self._output_size = self.head.output_size
        else:
            self.head = lambda x: x
            self._output_size = n

    def forward(self, input):
        return self.head(self.conv(input).reshape(input.shape[0], -1))

    @property
    def output_size(self):
        return self._output_size
