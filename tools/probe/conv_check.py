import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from zs3_amd import ops
torch.backends.cudnn.allow_tf32 = False
dev = torch.device("cuda:0")
def rel(a, b): return ((a.double()-b.double()).abs().max() / b.double().abs().max()).item()
cases = [  # N,H,W,Cin,Cout,k,stride,dil
 (2, 33, 33, 256, 256, 3, 1, 1), (2, 33, 33, 512, 512, 3, 1, 4), (2, 65, 65, 128, 128, 3, 2, 1), (3, 17, 19, 64, 256, 1, 1, 1),
 (2, 65, 65, 256, 512, 1, 2, 1), (2, 33, 33, 2048, 256, 3, 1, 18), (1, 129, 129, 320, 256, 3, 1, 1), (2, 9, 9, 256, 48, 1, 1, 1),
 (2, 40, 40, 256, 21, 1, 1, 1), (16, 33, 33, 1024, 256, 1, 1, 1)]
for (n,h,w,ci,co,k,s,d) in cases:
    g = torch.Generator(device="cpu").manual_seed(n*h+ci)
    x = torch.randn(n, ci, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, k, k, generator=g) / (ci*k*k) ** 0.5).to(dev)
    pad = d * (k // 2)
    ref = F.conv2d(x.double().cpu(), wt.double().cpu(), stride=s, padding=pad, dilation=d)
    wp = ops.prep_weight(wt)
    for prec in (3, 1):
        for cfg in (0, 1, 4):
            y, st = ops.conv2d_fwd(ops.nhwc(x), wp, s, pad, d, prec=prec, tile_cfg=cfg, want_stats=True)
            torch.cuda.synchronize()
            e = rel(ops.nchw(y).cpu(), ref)
            ssum = st[:, 0].double().sum(0).cpu(); ssq = st[:, 1].double().sum(0).cpu()
            es = ((ssum - ref.sum((0,2,3))).abs().max() / ref.sum((0,2,3)).abs().max()).item()
            eq = ((ssq - (ref**2).sum((0,2,3))).abs().max() / (ref**2).sum((0,2,3)).abs().max()).item()
            print(f"fwd {n,h,w,ci,co,k,s,d} prec{prec} cfg{cfg}: rel {e:.2e} stat {es:.1e} {eq:.1e}")
    ygpu = F.conv2d(x, wt, stride=s, padding=pad, dilation=d)
    print("   torch-gpu fp32 rel", f"{rel(ygpu.cpu(), ref):.2e}")
    # dgrad
    dy = torch.randn(ref.shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    xr = x.double().cpu().requires_grad_(True)
    F.conv2d(xr, wt.double().cpu(), stride=s, padding=pad, dilation=d).backward(dy.double().cpu())
    if co % 8 == 0:
        dx = ops.conv2d_dgrad(ops.nhwc(dy), wp, (h, w), s, pad, d)
        torch.cuda.synchronize()
        print(f"   dgrad rel {rel(ops.nchw(dx).cpu(), xr.grad):.2e}")
# timing of the heavy shapes at B=16
def bench(n,h,w,ci,co,k,s,d,prec=3,cfg=0,iters=10):
    x = torch.randn(n,h,w,ci,device=dev); wt = torch.randn(co,ci,k,k,device=dev)*0.02
    wp = ops.prep_weight(wt); pad = d*(k//2)
    for _ in range(3): ops.conv2d_fwd(x, wp, s, pad, d, prec=prec, tile_cfg=cfg)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(iters): y,_ = ops.conv2d_fwd(x, wp, s, pad, d, prec=prec, tile_cfg=cfg)
    torch.cuda.synchronize(); dt=(time.time()-t)/iters
    fl = 2.0*y.numel()*ci*k*k
    print(f"bench {n,h,w,ci,co,k,s,d} prec{prec} cfg{cfg}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TF (algorithmic)")
for cfg in (0,1,2,3,4):
    bench(16,129,129,256,256,3,1,1,cfg=cfg)
for cfg in (0,1,3):
    bench(16,33,33,1024,256,1,1,1,cfg=cfg); bench(16,33,33,256,1024,1,1,1,cfg=cfg); bench(16,33,33,256,256,3,1,1,cfg=cfg); bench(16,33,33,2048,256,3,1,12,cfg=cfg)
bench(16,129,129,256,256,3,1,1,prec=1,cfg=1)
