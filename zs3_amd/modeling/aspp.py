"""ASPP head (zs3/modeling/aspp.py:8-133) on the HIP kernels.  The five branches write straight into
channel slices of one [N,h,w,1280] buffer (no torch.cat pass); the pooled branch is a row-GEMM on
[N,2048] followed by a broadcast."""
import torch
import torch.nn as nn

from .. import functional as Fz
from .. import ops
from .layers import Conv2d, Dropout, to_channels_last_


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

    def forward_nhwc(self, x, out=None, lane=None, lane_forked=False):
        return self.atrous_conv.forward_nhwc(x, self.bn, act=Fz.ACT_RELU, out=out, lane=lane, lane_forked=lane_forked)

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

    # The five branches are independent and, at output stride 16, each of them launches fewer workgroups than the chip has CUs
    # (33x33 maps): the two heaviest atrous branches run on LANES (functional.lane_streams: side streams entered inside the fused
    # layer, forward and backward, with recorded waits) next to the others on the main stream, all writing into disjoint channel
    # slices of one buffer.  Worth 3 ms of the 44 ms step (same-box A/B, round 6: 44.1 against 47.1 ms).
    def forward_nhwc(self, x):
        n, h, w, _ = x.shape
        cat = torch.empty((n, h, w, 1280), dtype=ops.ACT_DTYPE, device=x.device)
        bn = self.global_avg_pool[2] if isinstance(self.global_avg_pool[2], nn.BatchNorm2d) else None

        xs = Fz.fork(x, 5)    # five consumers: their gradients are added in one launch, not pairwise

        def branch(i, lane=None):
            br = (self.aspp1, self.aspp2, self.aspp3, self.aspp4)[i]
            return br.forward_nhwc(xs[i], out=cat[..., 256 * i:256 * (i + 1)], lane=lane, lane_forked=lane is not None)

        def pooled():
            p = self.global_avg_pool[1].forward_nhwc(Fz.global_avg_pool(xs[4]), bn, act=Fz.ACT_RELU)   # [N,1,1,256]
            return Fz.broadcast_to(p, (h, w), out=cat[..., 1024:1280])

        concurrent = x.is_cuda and Fz.ASPP_CONCURRENT and (Fz.CAPTURE_SIDE_STREAMS or not torch.cuda.is_current_stream_capturing())
        l1, l2 = Fz.lane_streams(x.device, 2) if concurrent else (None, None)
        # Order: the lanes start from the fork point (x is ready), so the host order of the five launches does not matter to the
        # forward pass; it does to the backward pass -- autograd runs the nodes created LAST first, and the two heavy atrous
        # branches on the lanes have to be under way before the main stream works through its own three (round 6 trace: with the
        # lane branches created first the main stream finished its branches and then sat 1.4 ms waiting for the lanes to start and
        # finish).
        Fz.lanes_fork(l1, l2)
        parts = [None] * 5
        if not Fz.ASPP_LANES_LAST:      # (A/B switch: the round-6 mid-round order, lane branches created first)
            parts[2] = branch(2, l1)
            parts[3] = branch(3, l2)
        parts[4] = pooled()
        parts[0] = branch(0)
        parts[1] = branch(1)
        if Fz.ASPP_LANES_LAST:
            parts[2] = branch(2, l1)
            parts[3] = branch(3, l2)
        Fz.lanes_join()       # the projection below reads all five slices on the main stream
        return self.conv1.forward_nhwc(Fz.cat_slices(cat, parts), self.bn1, act=Fz.ACT_RELU, dropout=self.dropout)

    def forward(self, x):
        return ops.nchw(self.forward_nhwc(ops.nhwc(x)))

    def _init_weight(self):
        _kaiming_all(self)


def build_aspp(output_stride, BatchNorm, global_avg_pool_bn=True):
    return ASPP(output_stride, BatchNorm, global_avg_pool_bn)
