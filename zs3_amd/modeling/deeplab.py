"""DeepLabv3+ on the MI355X kernels, interface-compatible with zs3.modeling.deeplab.DeepLab (deeplab.py:10-99).

Kept from the reference so that its scripts, checkpoints and optimiser wiring work unchanged: the constructor
signature, the attributes `backbone` / `aspp` / `decoder`, the five forward variants, `freeze_bn`, the two
LR-group generators (Conv/BN parameters of the backbone vs. of ASPP + decoder, in `named_modules()` order) and
the 680 / 675 state-dict keys.  Everything numeric runs in libzs3hip.so; tensors cross this API as logical NCHW
(channels_last in memory, which is the kernels' native NHWC)."""
import torch
import torch.nn as nn

from .. import functional as Fz
from .. import ops
from . import aspp as _aspp
from . import decoder as _decoder
from .backbone import build_backbone
from .layers import BatchNorm2d
from .sync_batchnorm.batchnorm import SynchronizedBatchNorm2d

_LR_GROUPS = {"1x": ("backbone",), "10x": ("aspp", "decoder")}


class DeepLab(nn.Module):
    def __init__(self, output_stride=16, num_classes=21, sync_bn=True, freeze_bn=False, pretrained=True,
                 global_avg_pool_bn=True, imagenet_pretrained_path=""):
        super().__init__()
        norm = SynchronizedBatchNorm2d if sync_bn else BatchNorm2d
        # construction order fixes the RNG stream of the default initialisers: backbone, ASPP, decoder
        self.backbone = build_backbone(output_stride, norm, pretrained=pretrained,
                                       imagenet_pretrained_path=imagenet_pretrained_path)
        self.aspp = _aspp.build_aspp(output_stride, norm, global_avg_pool_bn)
        self.decoder = _decoder.build_decoder(num_classes, norm)
        if freeze_bn:
            self.freeze_bn()

    def __getstate__(self):
        """copy.deepcopy / torch.save of the whole module (saver.py:24-67 saves state dicts, but scripts also deep-copy models):
        the data-parallel arming of this process -- GradSync with its bucket tensors, hooks and process group -- stays behind;
        a copy arms itself at its own first training forward."""
        state = self.__dict__.copy()
        state.pop("_zs3_grad_sync", None)
        state.pop("_zs3_broadcast_done", None)
        return state

    # ------------------------------------------------------------------ NHWC pipeline pieces
    def _encode(self, image):
        top, low = self.backbone.forward_nhwc(image)
        return self.aspp.forward_nhwc(top), low

    def _logits_to_image(self, logits_nhwc, size):
        out = ops.nchw(Fz.bilinear(logits_nhwc, size))   # align_corners=True resize of deeplab.py:44,55
        return out

    # ------------------------------------------------------------------ the reference's forward variants
    def forward(self, input):
        if self.training and torch.is_grad_enabled():
            # data parallelism by construction ("DDP must live inside the build's modules"): the supervised scripts
            # run this forward in training mode on every rank (base_trainer.py:17), so under torch.distributed with more than
            # one rank the first such call arms the gradient all-reduce -- no GradSync line in the training script.  The
            # GMMN / GCN-context steps never come here with gradients enabled (they train pred_conv through
            # forward_class_prediction and exchange its gradients themselves).
            from .. import parallel
            if getattr(self, "_zs3_grad_sync", None) is None:
                parallel.ensure_data_parallel(self, broadcast=not getattr(self, "_zs3_broadcast_done", False))
        context, low = self._encode(input)
        features = self.decoder.features_nhwc(context, low)
        return self._logits_to_image(self.decoder.predict_nhwc(features), input.shape[2:])

    def forward_before_class_prediction(self, input):
        context, low = self._encode(input)
        return ops.nchw(self.decoder.features_nhwc(context, low))

    def forward_class_prediction(self, x, input_size):
        return self._logits_to_image(self.decoder.predict_nhwc(ops.nhwc(x)), input_size)

    def forward_before_last_conv_finetune(self, input):
        context, low = self._encode(input)
        return ops.nchw(self.decoder._head(self.decoder._merge(context, low), second=False))

    def forward_class_last_conv_finetune(self, x):
        return self.decoder.forward_class_last_conv_finetune(x)

    # ------------------------------------------------------------------ training utilities
    def freeze_bn(self):
        """Put every BatchNorm (plain or synchronised: the latter subclasses the former) into eval mode."""
        for module in self.modules():
            if isinstance(module, nn.BatchNorm2d):
                module.eval()

    def _lr_group(self, which):
        for attr in _LR_GROUPS[which]:
            for _, module in getattr(self, attr).named_modules():
                if isinstance(module, (nn.Conv2d, nn.BatchNorm2d)):
                    yield from (p for p in module.parameters() if p.requires_grad)

    def get_1x_lr_params(self):
        return self._lr_group("1x")

    def get_10x_lr_params(self):
        return self._lr_group("10x")
