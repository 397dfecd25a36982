from . import resnet


def build_backbone(output_stride, BatchNorm, pretrained=False, imagenet_pretrained_path=""):
    """zs3/modeling/backbone/__init__.py:4-12"""
    return resnet.ResNet101(output_stride, BatchNorm, pretrained=pretrained,
                            imagenet_pretrained_path=imagenet_pretrained_path)
