#!/usr/bin/env python3
"""Which device work of one supervised training step does NOT come from libzs3hip.so?  (A recorded plan -- zs3_amd/plan.py --
replays library launches only: anything the tensor library launches inside the step is either hoisted out or turned into a
library call.)  torch.profiler over one eager step: every kernel / memcpy / memset whose name is not one of ours, with the
operator that launched it.  Usage: python tools/probe/aten_in_step.py [--size 513 --batch 16 --dtype fp32|bf16]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=513)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--ddp", action="store_true", help="the N > 1 code path with a one-rank RCCL group (SyncBN, GradSync, global CE)")
    args = ap.parse_args()
    if args.ddp:
        import torch.distributed as dist
        import zs3_amd.parallel as par
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        par.FORCE_COLLECTIVES = True
    from zs3_amd import functional as Fz
    from zs3_amd import ops
    from zs3_amd.modeling.deeplab import DeepLab
    from zs3_amd.optim import SGD
    from zs3_amd.plan import StepPlan
    from zs3_amd.utils.loss import SegmentationLosses
    from zs3_amd.utils.synthetic import make_batch
    dev = torch.device("cuda:0")
    if args.dtype == "bf16":
        ops.set_storage(torch.bfloat16)
    torch.manual_seed(1)
    model = DeepLab(num_classes=21, pretrained=False, sync_bn=args.ddp).to(dev).train()
    groups = [{"params": model.get_1x_lr_params(), "lr": 0.007}, {"params": model.get_10x_lr_params(), "lr": 0.07}]
    opt = SGD(groups, momentum=0.9, weight_decay=5e-4)
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    b = make_batch(args.batch, args.size, 21, [10, 14], seed=1, device=dev)
    step = StepPlan(model, crit, opt, enabled=False)
    for _ in range(3):
        step(b["image"], b["label"])
    torch.cuda.synchronize()
    Fz.PLAN_RECORDING = True        # the step as a recording sees it (ASPP on one stream, keep-alive instead of record_stream)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step(b["image"], b["label"])
        torch.cuda.synchronize()
    Fz.PLAN_RECORDING = False
    Fz._plan_keep.clear()
    ours = ("zs3", "conv_", "bn_", "colstats", "affine_act", "sgd_multi", "prep_", "maxpool", "bilinear", "ce_", "sum_n", "nchw3",
            "group_colsum", "colsum_kernel", "pad_rows", "strip_reduce", "wgrad", "slab")
    foreign = {}
    total = 0
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CUDA:
            continue
        total += 1
        name = ev.name
        if any(k in name for k in ours):
            continue
        foreign.setdefault(name, [0, 0.0])
        foreign[name][0] += 1
        foreign[name][1] += ev.device_time
    print(f"{total} device activities in one step; not from libzs3hip.so:")
    for name, (n, us) in sorted(foreign.items(), key=lambda kv: -kv[1][0]):
        print(f"  {n:5d} x  {us:9.1f} us  {name[:150]}")
    # call sites of the fills / copies (python stacks of the launching operators)
    sites = {}
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CPU and ev.name in ("aten::zero_", "aten::fill_", "aten::copy_", "aten::_foreach_copy_", "aten::clone", "aten::zeros") and ev.stack:
            key = (ev.name, tuple(f for f in ev.stack if "zs3_amd" in f or "bench" in f)[:3])
            sites[key] = sites.get(key, 0) + 1
    for (name, stack), n in sorted(sites.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {n:5d} x {name}  <- {' <- '.join(stack)}")
    # who launched them: CPU-side operators that have such a kernel as a child
    print("launching operators (aten::*) with device time:")
    for ev in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:60]:
        if ev.key.startswith("aten::") and ev.device_time_total > 0:
            print(f"  {ev.count:5d} x  {ev.device_time_total:9.1f} us  {ev.key}")


if __name__ == "__main__":
    main()
