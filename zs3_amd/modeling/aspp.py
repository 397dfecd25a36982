"""ASPP head (zs3/modeling/aspp.py:8-133) on the HIP kernels.  The five branches write straight into
channel slices of one [N,h,w,1280] buffer (no torch.cat pass); the pooled branch is a row-GEMM on
[N,2048] followed by a broadcast."""
import torch
import torch.nn as nn

from .. import functional as Fz
from .. import ops
from .layers import BatchNorm2d, Conv2d, Dropout, to_channels_last_


def _kaiming_all(module):
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


class _ASPPModule(nn.Module):
    def __init__(self, inplanes, planes, kernel_size, padding, dilation, BatchNorm):
        super().__init__()
        self.atrous_conv = Conv2d(inplanes, planes, kernel_size=kernel_size, stride=1, padding=padding,
                                  dilation=dilation, bias=False)
        self.bn = BatchNorm(planes)
        self.relu = nn.ReLU()
        self._init_weight()

    def forward_nhwc(self, x, out=None):
        return self.atrous_conv.forward_nhwc(x, self.bn, act=Fz.ACT_RELU, out=out)

    def forward(self, x):
        return ops.nchw(self.forward_nhwc(ops.nhwc(x)))

    def _init_weight(self):
        _kaiming_all(self)


class ASPP(nn.Module):
    def __init__(self, output_stride, BatchNorm, global_avg_pool_bn=True):
        super().__init__()
        inplanes = 2048
        rates = {16: [1, 6, 12, 18], 8: [1, 12, 24, 36]}.get(output_stride)
        if rates is None:
            raise NotImplementedError
        self.aspp1 = _ASPPModule(inplanes, 256, 1, padding=0, dilation=rates[0], BatchNorm=BatchNorm)
        self.aspp2 = _ASPPModule(inplanes, 256, 3, padding=rates[1], dilation=rates[1], BatchNorm=BatchNorm)
        self.aspp3 = _ASPPModule(inplanes, 256, 3, padding=rates[2], dilation=rates[2], BatchNorm=BatchNorm)
        self.aspp4 = _ASPPModule(inplanes, 256, 3, padding=rates[3], dilation=rates[3], BatchNorm=BatchNorm)
        pool = [nn.AdaptiveAvgPool2d((1, 1)), Conv2d(inplanes, 256, 1, stride=1, bias=False)]
        if global_avg_pool_bn:
            pool.append(BatchNorm(256))
        pool.append(nn.ReLU())
        self.global_avg_pool = nn.Sequential(*pool)
        self.conv1 = Conv2d(1280, 256, 1, bias=False)
        self.bn1 = BatchNorm(256)
        self.relu = nn.ReLU()
        self.dropout = Dropout(0.5)
        self._init_weight()
        to_channels_last_(self)

    def forward_nhwc(self, x):
        n, h, w, _ = x.shape
        cat = torch.empty((n, h, w, 1280), dtype=torch.float32, device=x.device)
        parts = [br.forward_nhwc(x, out=cat[..., 256 * i:256 * (i + 1)])
                 for i, br in enumerate((self.aspp1, self.aspp2, self.aspp3, self.aspp4))]
        pooled = Fz.global_avg_pool(x)                                         # [N,1,1,2048]
        bn = self.global_avg_pool[2] if isinstance(self.global_avg_pool[2], nn.BatchNorm2d) else None
        pooled = self.global_avg_pool[1].forward_nhwc(pooled, bn, act=Fz.ACT_RELU)  # [N,1,1,256]
        parts.append(Fz.broadcast_to(pooled, (h, w), out=cat[..., 1024:1280]))
        y = self.conv1.forward_nhwc(Fz.cat_slices(cat, parts), self.bn1, act=Fz.ACT_RELU)
        return self.dropout.forward_nhwc(y)

    def forward(self, x):
        return ops.nchw(self.forward_nhwc(ops.nhwc(x)))

    def _init_weight(self):
        _kaiming_all(self)


def build_aspp(output_stride, BatchNorm, global_avg_pool_bn=True):
    return ASPP(output_stride, BatchNorm, global_avg_pool_bn)
