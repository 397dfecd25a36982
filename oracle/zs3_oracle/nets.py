"""Oracle DeepLabv3+ (dilated ResNet-101 + ASPP + decoder), plain torch fp32, table driven.

State-dict keys, constructor RNG consumption and forward arithmetic follow the reference:
  zs3/modeling/backbone/resnet.py:9-242, zs3/modeling/aspp.py:8-133,
  zs3/modeling/decoder.py:8-87, zs3/modeling/deeplab.py:10-99.
Test infrastructure only (see package docstring).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

# (planes, n_blocks, stride, dilations per block) at output stride 16 -- resnet.py:68-71,84-115,236
_STAGES_OS16 = (
    (64, 3, 1, (1, 1, 1)),
    (128, 4, 2, (1,) * 4),
    (256, 23, 2, (1,) * 23),
    (512, 3, 1, (2, 4, 8)),  # multi-grid unit [1,2,4] x dilation 2 -- resnet.py:67,108-115,167-182
)
_STAGES_OS8 = (
    (64, 3, 1, (1, 1, 1)),
    (128, 4, 2, (1,) * 4),
    (256, 23, 1, (2,) * 23),
    (512, 3, 1, (4, 8, 16)),
)


def _bilinear(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


class Bottleneck(nn.Module):
    """1x1 -> 3x3(stride, dilation, pad=dilation) -> 1x1(x4) + residual (resnet.py:9-53)."""

    def __init__(self, cin, planes, stride, dilation, project):
        super().__init__()
        # Registration order = reference (conv1,bn1,conv2,bn2,conv3,bn3,downsample): it fixes the
        # order of the re-initialisation pass below (resnet.py:199-209).
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, 4 * planes, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(4 * planes)
        self.downsample = project

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        skip = x if self.downsample is None else self.downsample(x)
        return F.relu(y + skip)


class ResNet101Dilated(nn.Module):
    def __init__(self, output_stride=16):
        super().__init__()
        stages = {16: _STAGES_OS16, 8: _STAGES_OS8}.get(output_stride)
        if stages is None:
            raise NotImplementedError
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for si, (planes, nblk, stride, dils) in enumerate(stages, start=1):
            blocks = []
            for bi in range(nblk):
                s = stride if bi == 0 else 1
                proj = None
                if bi == 0 and (s != 1 or cin != 4 * planes):
                    # the projection is created BEFORE the block's own convs (resnet.py:121-141)
                    proj = nn.Sequential(nn.Conv2d(cin, 4 * planes, 1, stride=s, bias=False), nn.BatchNorm2d(4 * planes))
                blocks.append(Bottleneck(cin, planes, s, dils[bi], proj))
                cin = 4 * planes
            setattr(self, f"layer{si}", nn.Sequential(*blocks))
        # He-normal with fan = kh*kw*Cout, BN affine = (1, 0): resnet.py:199-209
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.max_pool2d(x, 3, stride=2, padding=1)
        low = self.layer1(x)
        x = self.layer4(self.layer3(self.layer2(low)))
        return x, low


def _kaiming_all(module):
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


class _Branch(nn.Module):
    def __init__(self, cin, cout, k, dilation):
        super().__init__()
        self.atrous_conv = nn.Conv2d(cin, cout, k, padding=0 if k == 1 else dilation, dilation=dilation, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        _kaiming_all(self)  # first init pass, aspp.py:23,31-40

    def forward(self, x):
        return F.relu(self.bn(self.atrous_conv(x)))


class ASPP(nn.Module):
    def __init__(self, output_stride=16, global_avg_pool_bn=True):
        super().__init__()
        rates = {16: (1, 6, 12, 18), 8: (1, 12, 24, 36)}.get(output_stride)
        if rates is None:
            raise NotImplementedError
        self.aspp1 = _Branch(2048, 256, 1, rates[0])
        self.aspp2 = _Branch(2048, 256, 3, rates[1])
        self.aspp3 = _Branch(2048, 256, 3, rates[2])
        self.aspp4 = _Branch(2048, 256, 3, rates[3])
        pool = [nn.AdaptiveAvgPool2d((1, 1)), nn.Conv2d(2048, 256, 1, bias=False)]
        if global_avg_pool_bn:
            pool.append(nn.BatchNorm2d(256))
        pool.append(nn.ReLU())
        self.global_avg_pool = nn.Sequential(*pool)
        self.conv1 = nn.Conv2d(1280, 256, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(256)
        self.dropout = nn.Dropout(0.5)
        _kaiming_all(self)  # second pass over every conv, aspp.py:101,118-129

    def forward(self, x):
        hw = x.shape[2:]
        parts = [self.aspp1(x), self.aspp2(x), self.aspp3(x), self.aspp4(x)]
        parts.append(_bilinear(self.global_avg_pool(x), hw))
        y = F.relu(self.bn1(self.conv1(torch.cat(parts, 1))))
        return self.dropout(y)


class Decoder(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.conv1 = nn.Conv2d(256, 48, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(48)
        self.last_conv = nn.Sequential(
            nn.Conv2d(304, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256), nn.ReLU(), nn.Dropout(0.5),
            nn.Conv2d(256, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256), nn.ReLU(), nn.Dropout(0.1),
        )
        self.pred_conv = nn.Conv2d(256, num_classes, 1)
        _kaiming_all(self)  # decoder.py:74-83 (pred_conv.bias keeps its default init)

    def _merge(self, x, low):
        low = F.relu(self.bn1(self.conv1(low)))
        return torch.cat((_bilinear(x, low.shape[2:]), low), 1)

    def forward_before_class_prediction(self, x, low):
        return self.last_conv(self._merge(x, low))

    def forward_before_last_conv_finetune(self, x, low):
        return self.last_conv[:4](self._merge(x, low))

    def forward_class_last_conv_finetune(self, x):
        return self.last_conv[4:](x)

    def forward_class_prediction(self, x):
        return self.pred_conv(x)

    def forward(self, x, low):
        return self.pred_conv(self.forward_before_class_prediction(x, low))


class DeepLab(nn.Module):
    """deeplab.py:10-99.  ``sync_bn`` only selects the BN class in the reference; on one device the
    vendored SyncBN falls back to F.batch_norm (sync_batchnorm/batchnorm.py:48-58), so the oracle
    always uses nn.BatchNorm2d.  ``pretrained`` loads an ImageNet checkpoint (resnet.py:211-226)."""

    def __init__(self, output_stride=16, num_classes=21, sync_bn=True, freeze_bn=False, pretrained=True,
                 global_avg_pool_bn=True, imagenet_pretrained_path=""):
        super().__init__()
        self.backbone = ResNet101Dilated(output_stride)
        if pretrained:
            ck = torch.load(imagenet_pretrained_path)["state_dict"]
            own = self.backbone.state_dict()
            own.update({k[7:]: v for k, v in ck.items() if k[7:] in own})
            self.backbone.load_state_dict(own)
        self.aspp = ASPP(output_stride, global_avg_pool_bn)
        self.decoder = Decoder(num_classes)
        if freeze_bn:
            self.freeze_bn()

    def forward(self, x):
        return self.forward_class_prediction(self.forward_before_class_prediction(x), x.shape[2:])

    def forward_before_class_prediction(self, x):
        top, low = self.backbone(x)
        return self.decoder.forward_before_class_prediction(self.aspp(top), low)

    def forward_class_prediction(self, feat, input_size):
        return _bilinear(self.decoder.forward_class_prediction(feat), tuple(input_size))

    def forward_before_last_conv_finetune(self, x):
        top, low = self.backbone(x)
        return self.decoder.forward_before_last_conv_finetune(self.aspp(top), low)

    def forward_class_last_conv_finetune(self, x):
        return self.decoder.forward_class_last_conv_finetune(x)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def _conv_bn_params(self, roots):
        for root in roots:
            for _, m in root.named_modules():
                if isinstance(m, (nn.Conv2d, nn.BatchNorm2d)):
                    for p in m.parameters():
                        if p.requires_grad:
                            yield p

    def get_1x_lr_params(self):
        return self._conv_bn_params([self.backbone])

    def get_10x_lr_params(self):
        return self._conv_bn_params([self.aspp, self.decoder])
