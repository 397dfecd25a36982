"""Backbone factory (zs3/modeling/backbone/__init__.py): only the dilated ResNet-101 exists in ZS3."""
from .resnet import ResNet101


def build_backbone(output_stride, BatchNorm, pretrained=False, imagenet_pretrained_path=""):
    kwargs = {"pretrained": pretrained, "imagenet_pretrained_path": imagenet_pretrained_path}
    return ResNet101(output_stride, BatchNorm, **kwargs)
