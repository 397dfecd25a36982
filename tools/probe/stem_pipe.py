"""probe: the stem's forward conv (7x1 windows, stride 2, 64x64 tiles) after another conv launch, saved for an A/B of kernel builds"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn as nn
from zs3_amd import ops
from zs3_amd.modeling.backbone.resnet import ResNet101
dev = torch.device("cuda:0")
torch.manual_seed(5)
net = ResNet101(16, nn.BatchNorm2d, pretrained=False).to(dev).train()
g = torch.Generator().manual_seed(11)
image = torch.randn(1, 3, 513, 513, generator=g).to(dev)
if len(sys.argv) > 2 and sys.argv[2] == "warm":
    x = torch.randn(2, 33, 33, 256, device=dev)
    wp = ops.prep_weight(torch.randn(21, 256, 1, 1, device=dev))
    ops.conv2d_fwd(x, wp)
    junk = [torch.full((1 << 24,), float("nan"), device=dev) for _ in range(8)]
    del junk
with torch.no_grad():
    out = net._stem(image)
torch.cuda.synchronize()
torch.save(out.cpu(), sys.argv[1])
print(sys.argv[1], float(out.abs().sum()), bool(torch.isnan(out).any()))
