"""Where and when do the workgroups of the persistent pointwise kernel run?  (-DZS3_CONV_TIMING build: every workgroup reports its
start / end cycle and HW_ID / XCC_ID)  Answers: are two 128-register, 66 KB workgroups really co-resident on a CU?"""
import ctypes, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from zs3_amd import ops
from zs3_amd._lib import lib
dev = torch.device("cuda:0")
dbg = torch.zeros(64 + 4 * 1024, dtype=torch.int64, device=dev)
fn = lib().zs3_conv_pw_timing
fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
fn(dbg.data_ptr())
h, ci, co = 33, 256, 1024
x = torch.randn(16, h, h, ci, device=dev); wt = torch.randn(co, ci, 1, 1, device=dev) * 0.02
wp = ops.prep_weight(wt, f16_forward=True)
for rep in range(3):
    dbg.zero_(); dbg[0] = 0
    y, st = ops.conv2d_fwd(x, wp, 1, 0, 1, want_stats=True, tile_cfg=52)
torch.cuda.synchronize()
t = dbg.cpu()[64:64 + 4 * 512].view(512, 4)
t = t[t[:, 1] > 0]
print("workgroups that reported:", t.shape[0])
t0 = int(t[:, 0].min())
per_cu = collections.defaultdict(list)
for b in range(t.shape[0]):
    hw, xcc = int(t[b, 2]), int(t[b, 3]) & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    per_cu[(xcc, se, sh, cu)].append((int(t[b, 0]) - t0, int(t[b, 1]) - t0, b))
print("distinct CUs:", len(per_cu))
overl = 0
for k, v in sorted(per_cu.items())[:12]:
    print(k, [(a, e, b) for a, e, b in sorted(v)])
for v in per_cu.values():
    v.sort()
    for i in range(len(v) - 1):
        if v[i + 1][0] < v[i][1]:
            overl += 1
print("CUs-pairs of workgroups overlapping in time on one CU:", overl, "; workgroups per CU histogram:", collections.Counter(len(v) for v in per_cu.values()))
starts = sorted(int(a) - t0 for a in t[:, 0])
print("start cycles (sorted, every 32nd):", starts[::32])
