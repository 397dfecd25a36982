"""Parameter-holding layers.  They subclass the torch.nn classes so that state-dict keys, isinstance checks
(deeplab.py:75-99 selects LR groups by isinstance(m, nn.Conv2d / BatchNorm)) and checkpoint loading behave
exactly like the reference, but their forward runs the HIP kernels.  Conv weights are stored channels_last
([Cout][KH][KW][Cin] in memory = the K-contiguous GEMM operand) while keeping the logical OIHW shape."""
import torch
import torch.nn as nn

from .. import functional as Fz
from .. import ops


class Conv2d(nn.Conv2d):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert self.groups == 1 and self.padding_mode == "zeros"
        assert self.stride[0] == self.stride[1] and self.dilation[0] == self.dilation[1] and self.padding[0] == self.padding[1]

    def to_channels_last_(self):
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)
        return self

    def forward_nhwc(self, x, bn=None, residual=None, act=Fz.ACT_NONE, out=None, pass_through=False,
                     input_has_one_consumer=False, dropout=None, next_conv=None, out_dtype=None, lane=None, lane_forked=False):
        """dropout: the nn.Dropout module that follows bn + act in the reference's Sequential (fused where possible).
        next_conv: the conv that is the only consumer of this layer's output (it may apply this layer's BN + ReLU itself)."""
        return Fz.conv_bn_act(x, self.weight, bn=bn, bias=self.bias, residual=residual, stride=self.stride[0],
                              pad=self.padding[0], dil=self.dilation[0], act=act, out=out, pass_through=pass_through,
                              input_has_one_consumer=input_has_one_consumer,
                              dropout=None if dropout is None else (dropout.p, dropout.training), next_conv=next_conv,
                              out_dtype=out_dtype, lane=lane, lane_forked=lane_forked)

    def forward(self, x):  # logical NCHW in / out
        return ops.nchw(self.forward_nhwc(ops.nhwc(x)))


class BatchNorm2d(nn.BatchNorm2d):
    def forward_nhwc(self, x, act=Fz.ACT_NONE):
        return Fz.bn_act(x, self, act)

    def forward(self, x):
        if x.dim() != 4:
            raise ValueError(f"expected 4D input (got {x.dim()}D input)")
        return ops.nchw(self.forward_nhwc(ops.nhwc(x)))


class ReLU(nn.ReLU):
    def forward(self, x):
        if x.dim() == 4:
            return ops.nchw(Fz.relu(ops.nhwc(x)))
        return Fz.relu(x)


class Dropout(nn.Dropout):
    def forward_nhwc(self, x):
        return Fz.dropout(x, self.p, self.training)

    def forward(self, x):
        if x.dim() == 4:
            return ops.nchw(self.forward_nhwc(ops.nhwc(x)))
        return Fz.dropout(x, self.p, self.training)


def to_channels_last_(module):
    for m in module.modules():
        if isinstance(m, Conv2d):
            m.to_channels_last_()
    return module
