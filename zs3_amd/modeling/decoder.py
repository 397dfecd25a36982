"""DeepLabv3+ decoder (zs3/modeling/decoder.py:8-87) on the HIP kernels."""
import torch
import torch.nn as nn

from .. import functional as Fz
from .. import ops
from .aspp import _kaiming_all
from .layers import Conv2d, Dropout, to_channels_last_


class Decoder(nn.Module):
    def __init__(self, num_classes, BatchNorm):
        super().__init__()
        self.conv1 = Conv2d(256, 48, 1, bias=False)
        self.bn1 = BatchNorm(48)
        self.relu = nn.ReLU()
        self.last_conv = nn.Sequential(
            Conv2d(304, 256, kernel_size=3, stride=1, padding=1, bias=False), BatchNorm(256), nn.ReLU(), Dropout(0.5),
            Conv2d(256, 256, kernel_size=3, stride=1, padding=1, bias=False), BatchNorm(256), nn.ReLU(), Dropout(0.1),
        )
        self.pred_conv = Conv2d(256, num_classes, kernel_size=1, stride=1)
        self._init_weight()
        to_channels_last_(self)

    # ---- NHWC internals
    def _merge(self, x, low):
        n, h, w, _ = low.shape
        cat = torch.empty((n, h, w, 304), dtype=ops.ACT_DTYPE, device=x.device)
        up = Fz.bilinear(x, (h, w), out=cat[..., :256])
        lo = self.conv1.forward_nhwc(low, self.bn1, act=Fz.ACT_RELU, out=cat[..., 256:304])
        return Fz.cat_slices(cat, [up, lo])

    def _head(self, x, first=True, second=True):
        lc = self.last_conv
        if first:     # conv + BN + ReLU + Dropout(0.5): one fused layer
            x = lc[0].forward_nhwc(x, lc[1], act=Fz.ACT_RELU, dropout=lc[3])
        if second:    # ... + Dropout(0.1)
            x = lc[4].forward_nhwc(x, lc[5], act=Fz.ACT_RELU, dropout=lc[7])
        return x

    def features_nhwc(self, x, low):
        return self._head(self._merge(x, low))

    def predict_nhwc(self, feat):
        # the class scores stay fp32 whatever the activation storage: they feed the resize to image size and the loss
        return self.pred_conv.forward_nhwc(feat, out_dtype=torch.float32)

    # ---- reference interface (logical NCHW)
    def forward(self, x, low_level_feat):
        return ops.nchw(self.predict_nhwc(self.features_nhwc(ops.nhwc(x), ops.nhwc(low_level_feat))))

    def forward_before_class_prediction(self, x, low_level_feat):
        return ops.nchw(self.features_nhwc(ops.nhwc(x), ops.nhwc(low_level_feat)))

    def forward_before_last_conv_finetune(self, x, low_level_feat):
        return ops.nchw(self._head(self._merge(ops.nhwc(x), ops.nhwc(low_level_feat)), second=False))

    def forward_class_prediction(self, x):
        return ops.nchw(self.predict_nhwc(ops.nhwc(x)))

    def forward_class_last_conv_finetune(self, x):
        return ops.nchw(self._head(ops.nhwc(x), first=False))

    def _init_weight(self):
        _kaiming_all(self)


def build_decoder(num_classes, BatchNorm):
    return Decoder(num_classes, BatchNorm)
