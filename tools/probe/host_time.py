"""Is the supervised step host-bound?  Times the host side of forward / loss / backward / optimizer WITHOUT syncs over a few
steps (how long python needs to enqueue a step) next to the synced wall time per step, and the queue depth proxy: how long
torch.cuda.synchronize() blocks after the host has finished enqueuing N steps."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd.modeling.deeplab import DeepLab
from zs3_amd.optim import SGD
from zs3_amd.utils.loss import SegmentationLosses
from zs3_amd.utils.synthetic import make_batch

import os
from zs3_amd import ops
if os.environ.get("ZS3_STORAGE") == "bf16":
    ops.set_storage(torch.bfloat16)
dev = torch.device("cuda:0")
torch.manual_seed(1)
model = DeepLab(num_classes=21, pretrained=False).to(dev).train()
groups = [{"params": model.get_1x_lr_params(), "lr": 0.007}, {"params": model.get_10x_lr_params(), "lr": 0.07}]
opt = SGD(groups, momentum=0.9, weight_decay=5e-4)
crit = SegmentationLosses(cuda=True).build_loss("ce")
b = make_batch(16, 513, 21, [10, 14], seed=1, device=dev)
image, label = b["image"], b["label"]


def step(t):
    t0 = time.perf_counter()
    opt.zero_grad()
    out = model(image)
    t1 = time.perf_counter()
    loss = crit(out, label)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    t.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))


for _ in range(4):
    step([])
torch.cuda.synchronize()
n = 10
t = []
w0 = time.perf_counter()
for _ in range(n):
    step(t)
h = time.perf_counter() - w0
torch.cuda.synchronize()
w = time.perf_counter() - w0
f, l, bw, o = (sum(x[i] for x in t) / n * 1e3 for i in range(4))
print(f"host enqueue per step: forward {f:.2f} ms, loss {l:.2f}, backward {bw:.2f}, optimizer {o:.2f}  -> {h / n * 1e3:.2f} ms; "
      f"wall per step {w / n * 1e3:.2f} ms; final sync waited {(w - h) * 1e3:.2f} ms for the queue to drain")

# ---- round 6: the same step as a recorded launch plan (zs3_amd/plan.py): host time of one replay call
from zs3_amd.plan import StepPlan
plan_step = StepPlan(model, crit, opt)
for _ in range(5):
    plan_step(image, label)
torch.cuda.synchronize()
assert plan_step.replays >= 2, (plan_step.eager_calls, plan_step.recordings, plan_step.replays)
w0 = time.perf_counter()
for _ in range(n):
    plan_step(image, label)
h = time.perf_counter() - w0
torch.cuda.synchronize()
w = time.perf_counter() - w0
print(f"recorded plan ({plan_step.recorded_ops} launches): host enqueue per step {h / n * 1e3:.2f} ms; wall per step {w / n * 1e3:.2f} ms; "
      f"final sync waited {(w - h) * 1e3:.2f} ms for the queue to drain")
# how fast can the host issue the plan when the GPU is not the limit?  (the queue is empty at the start; 3 steps fit the queues)
torch.cuda.synchronize()
w0 = time.perf_counter()
plan_step(image, label)
h1 = time.perf_counter() - w0
torch.cuda.synchronize()
print(f"one replay into an empty queue: {h1 * 1e3:.2f} ms of host time")
