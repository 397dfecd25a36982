"""Dilated ResNet-101 backbone on the HIP kernels.  Interface, state-dict keys and init follow
zs3/modeling/backbone/resnet.py:9-242; the arithmetic runs as fused conv+BN(+residual)+ReLU layers."""
import math

import torch
import torch.nn as nn

from ... import functional as Fz
from ... import ops
from ..layers import BatchNorm2d, Conv2d, to_channels_last_

_STAGES = {
    16: ((64, 3, 1, (1, 1, 1)), (128, 4, 2, (1,) * 4), (256, 23, 2, (1,) * 23), (512, 3, 1, (2, 4, 8))),
    8: ((64, 3, 1, (1, 1, 1)), (128, 4, 2, (1,) * 4), (256, 23, 1, (2,) * 23), (512, 3, 1, (4, 8, 16))),
}


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, BatchNorm=None):
        super().__init__()
        BatchNorm = BatchNorm or BatchNorm2d
        self.conv1 = Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = BatchNorm(planes)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=stride, dilation=dilation, padding=dilation, bias=False)
        self.bn2 = BatchNorm(planes)
        self.conv3 = Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = BatchNorm(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward_nhwc(self, x):
        # conv1's node also hands the input through (an alias of x) as the tensor the skip path consumes, so that the skip
        # path's gradient comes back to conv1 and is accumulated inside its dgrad epilogue -- no separate add kernel in
        # backward for the identity blocks, and none for the projection blocks either (the downsample conv's dgrad is the
        # incoming skip gradient there).  Being the only consumer of x this way, conv1's dgrad also sums the previous
        # block's bn3 backward statistics.
        # bn1 -> relu -> conv2 and bn2 -> relu -> conv3 (resnet.py:39-46): where the consuming conv runs on kernels whose producer
        # waves convert the operand, it applies the BatchNorm scale / shift and the ReLU itself and the activation is never stored
        # (`next_conv`; functional.conv_bn_act decides per layer geometry)
        y, skip = self.conv1.forward_nhwc(x, self.bn1, act=Fz.ACT_RELU, pass_through=True, input_has_one_consumer=True,
                                          next_conv=self.conv2)
        if self.downsample is not None:
            skip = self.downsample[0].forward_nhwc(skip, self.downsample[1])
        y = self.conv2.forward_nhwc(y, self.bn2, act=Fz.ACT_RELU, input_has_one_consumer=True, next_conv=self.conv3)
        return self.conv3.forward_nhwc(y, self.bn3, residual=skip, act=Fz.ACT_RELU,   # bn3 + add + relu in one pass
                                       input_has_one_consumer=True)

    def forward(self, x):
        return ops.nchw(self.forward_nhwc(ops.nhwc(x)))


class ResNet(nn.Module):
    def __init__(self, block, layers, output_stride, BatchNorm, pretrained=True, imagenet_pretrained_path=""):
        super().__init__()
        if output_stride not in _STAGES:
            raise NotImplementedError
        self.inplanes = 64
        self.conv1 = Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for si, ((planes, _, stride, dils), nblk) in enumerate(zip(_STAGES[output_stride], layers), start=1):
            blocks = []
            for bi in range(nblk):
                s = stride if bi == 0 else 1
                down = None
                if bi == 0 and (s != 1 or self.inplanes != planes * block.expansion):
                    down = nn.Sequential(Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=s, bias=False),
                                         BatchNorm(planes * block.expansion))
                blocks.append(block(self.inplanes, planes, s, dils[bi], down, BatchNorm))
                self.inplanes = planes * block.expansion
            setattr(self, f"layer{si}", nn.Sequential(*blocks))
        self._init_weight()
        if pretrained:
            self._load_pretrained_model(imagenet_pretrained_path)
        to_channels_last_(self)

    # -- stem: 7x7/s2 conv as a 7x1 conv over 32-float (8 pixels x 4 channels) windows of a padded NHWC4 image
    def _stem(self, image):
        from ..._lib import I, P, check, lib, stream
        n, c, h, w = image.shape
        assert c == 3, "the stem expects a 3-channel NCHW image"
        image = image.contiguous().float()
        ho, wo = ops.conv_out_size(h, 7, 2, 3, 1), ops.conv_out_size(w, 7, 2, 3, 1)
        wp = max(w + 7, 2 * (wo - 1) + 8)
        xp = torch.empty((n, h, wp, 4), dtype=torch.float32, device=image.device)
        check(lib().zs3_nchw3_to_nhwc4(P(image), P(xp), I(n), I(h), I(w), I(wp), I(3), stream()), "zs3_nchw3_to_nhwc4")
        w_eff = _StemWeight.apply(self.conv1.weight)     # [64, 32, 7, 1]: 7 taps x (8 pixels x 4 channels), zero padded
        geom = dict(ho=ho, wo=wo, cin_pad=32, cin_valid=32, kh=7, kw=1, stride=2, pad_h=3, pad_w=0, dil=1, ncols=64)

        def wgrad(dy, x):
            return ops.conv2d_wgrad(dy, x, 64, 32, 7, 1, 2, 3, 0, 1, ci_read=32).permute(0, 3, 1, 2)

        return Fz.conv_bn_act(xp, w_eff, bn=self.bn1, act=Fz.ACT_RELU, stride=2, geom=geom, wgrad=wgrad)

    def forward_nhwc(self, image):
        x = self._stem(image)
        x = Fz.max_pool(x, 3, 2, 1)
        for blk in self.layer1:
            x = blk.forward_nhwc(x)
        x, low = Fz.fork(x, 2)   # layer1's output feeds layer2 and the decoder (deeplab.py:41-42): gradients added by zs3_sum_n
        for layer in (self.layer2, self.layer3, self.layer4):
            for blk in layer:
                x = blk.forward_nhwc(x)
        return x, low

    def forward(self, input):
        x, low = self.forward_nhwc(input)
        return ops.nchw(x), ops.nchw(low)

    def _init_weight(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _load_pretrained_model(self, imagenet_pretrained_path):
        pretrain_dict = torch.load(imagenet_pretrained_path)["state_dict"]
        state_dict = self.state_dict()
        state_dict.update({k[7:]: v for k, v in pretrain_dict.items() if k[7:] in state_dict})
        self.load_state_dict(state_dict)


class _StemWeight(torch.autograd.Function):
    """[64, 3, 7, 7] -> [64, 32, 7, 1]: row (o, kh) of the channels_last weight, 7 x 3 floats, becomes 8 x 4 floats (one zero pixel, one
    zero channel per pixel) -- F.pad(w.permute(0, 2, 3, 1), (0, 1, 0, 1)).reshape(64, 7, 1, 32).permute(0, 3, 1, 2), and its backward,
    as two launches of the library (zs3_repack_pad) so that a recorded plan carries them."""

    @staticmethod
    def forward(ctx, weight):
        from ..._lib import I, P, check, lib, stream
        co, ci, kh, kw = weight.shape
        wl = weight.detach().permute(0, 2, 3, 1).contiguous()          # a view for channels_last parameters
        out = torch.empty((co, kh, 1, 32), dtype=torch.float32, device=weight.device)
        check(lib().zs3_repack_pad(P(wl), co * kh, I(kw), I(ci), P(out), I(8), I(4), I(0), stream()), "zs3_repack_pad")
        ctx.geom = (co, ci, kh, kw)
        ctx.param = weight if isinstance(weight, torch.nn.Parameter) else None
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dw):
        from ..._lib import I, P, check, lib, stream
        co, ci, kh, kw = ctx.geom
        d = dw.permute(0, 2, 3, 1).contiguous()                         # [64, 7, 1, 32]
        buf = Fz.grad_buffer(ctx.param) if ctx.param is not None and ctx.param.is_contiguous(memory_format=torch.channels_last) else None
        out = buf.view(co, kh, kw, ci) if buf is not None and buf.numel() == co * kh * kw * ci else \
            torch.empty((co, kh, kw, ci), dtype=torch.float32, device=dw.device)      # (the data-parallel bucket's slice, when there is one)
        check(lib().zs3_repack_pad(P(d), co * kh, I(kw), I(ci), P(out), I(8), I(4), I(1), stream()), "zs3_repack_pad")
        return out.permute(0, 3, 1, 2)                                  # logical OIHW, channels_last memory like the parameter


def ResNet101(output_stride, BatchNorm, pretrained=True, imagenet_pretrained_path=""):
    return ResNet(Bottleneck, [3, 4, 23, 3], output_stride, BatchNorm, pretrained=pretrained,
                  imagenet_pretrained_path=imagenet_pretrained_path)
