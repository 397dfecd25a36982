"""DeepLab -- drop-in for zs3.modeling.deeplab.DeepLab (deeplab.py:10-99): same constructor, attributes,
forward variants, LR-group generators and state-dict keys; the arithmetic runs on libzs3hip.so.

Tensors cross this API as logical NCHW; results are channels_last in memory (NHWC is the native
layout of the kernels), which every torch consumer accepts."""
import torch.nn as nn

from .. import functional as Fz
from .. import ops
from .aspp import build_aspp
from .backbone import build_backbone
from .decoder import build_decoder
from .layers import BatchNorm2d
from .sync_batchnorm.batchnorm import SynchronizedBatchNorm2d


class DeepLab(nn.Module):
    def __init__(self, output_stride=16, num_classes=21, sync_bn=True, freeze_bn=False, pretrained=True,
                 global_avg_pool_bn=True, imagenet_pretrained_path=""):
        super().__init__()
        BatchNorm = SynchronizedBatchNorm2d if sync_bn else BatchNorm2d
        self.backbone = build_backbone(output_stride, BatchNorm, pretrained=pretrained,
                                       imagenet_pretrained_path=imagenet_pretrained_path)
        self.aspp = build_aspp(output_stride, BatchNorm, global_avg_pool_bn)
        self.decoder = build_decoder(num_classes, BatchNorm)
        if freeze_bn:
            self.freeze_bn()

    # ---- NHWC internals
    def _features(self, image):
        x, low = self.backbone.forward_nhwc(image)
        return self.aspp.forward_nhwc(x), low

    def _upsample(self, logits, size):
        return ops.nchw(Fz.bilinear(logits, size))

    # ---- reference interface
    def forward(self, input):
        x, low = self._features(input)
        return self._upsample(self.decoder.predict_nhwc(self.decoder.features_nhwc(x, low)), input.shape[2:])

    def forward_before_class_prediction(self, input):
        x, low = self._features(input)
        return ops.nchw(self.decoder.features_nhwc(x, low))

    def forward_class_prediction(self, x, input_size):
        return self._upsample(self.decoder.predict_nhwc(ops.nhwc(x)), input_size)

    def forward_before_last_conv_finetune(self, input):
        x, low = self._features(input)
        return ops.nchw(self.decoder._head(self.decoder._merge(x, low), second=False))

    def forward_class_last_conv_finetune(self, x):
        return self.decoder.forward_class_last_conv_finetune(x)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def _group(self, roots):
        for root in roots:
            for _, m in root.named_modules():
                if isinstance(m, (nn.Conv2d, nn.BatchNorm2d)):
                    for p in m.parameters():
                        if p.requires_grad:
                            yield p

    def get_1x_lr_params(self):
        return self._group([self.backbone])

    def get_10x_lr_params(self):
        return self._group([self.aspp, self.decoder])
