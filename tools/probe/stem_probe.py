import sys, copy, torch, torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from zs3_amd.modeling.backbone.resnet import ResNet101
dev = torch.device('cuda:0')
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()
for hw in (65, 129, 257, 513):
    torch.manual_seed(5)
    net = ResNet101(16, nn.BatchNorm2d, pretrained=False)
    w64, bn64 = net.conv1.weight.detach().double().clone().requires_grad_(True), copy.deepcopy(net.bn1).double()
    net = net.to(dev).train()
    g = torch.Generator().manual_seed(11)
    image = torch.randn(1, 3, hw, hw, generator=g)
    pre = F.conv2d(image.double(), w64, stride=2, padding=3); pre.retain_grad()
    ref = F.relu(bn64.train()(pre))
    up = torch.randn(ref.shape, generator=g)
    ref.backward(up.double())
    out = net._stem(image.to(dev))
    out.backward(up.to(dev).permute(0, 2, 3, 1).contiguous())
    torch.cuda.synchronize()
    # fp32 torch reference of the same thing
    w32 = net.conv1.weight.detach().float().cpu().clone().requires_grad_(True); bn32 = copy.deepcopy(bn64).float()
    torch.manual_seed(5)
    net2 = ResNet101(16, nn.BatchNorm2d, pretrained=False); bn32 = copy.deepcopy(net2.bn1)
    r32 = F.relu(bn32.train()(F.conv2d(image, w32, stride=2, padding=3))); r32.backward(up)
    print(hw, 'out', rel(out.permute(0,3,1,2), ref), 'dW', rel(net.conv1.weight.grad, w64.grad), 'torch-fp32 dW', rel(w32.grad, w64.grad),
          'dgamma', rel(net.bn1.weight.grad, bn64.weight.grad), 'dbeta', rel(net.bn1.bias.grad, bn64.bias.grad),
          '|dW|max', w64.grad.abs().max().item())
