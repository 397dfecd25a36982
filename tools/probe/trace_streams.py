"""Per-queue view of a rocprofv3 --kernel-trace CSV for ONE training step (the last complete one): for every HIP queue the
busy time, the idle time between its kernels and the largest idle gaps with the kernels around them; then the union busy time
of the device.  Answers "is the dependent chain waiting for the host, or for the GPU?"."""
import csv, sys, re, collections


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n).split("(")[0][:48]


rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in rows))
# steps are delimited by the optimizer's multi-tensor launch
marks = [i for i, k in enumerate(ks) if k[2].startswith(sys.argv[2] if len(sys.argv) > 2 else "sgd_multi_kernel")]
if len(marks) < 3:
    sys.exit("fewer than 3 step markers in the trace")
i0, i1 = marks[-2] + 1, marks[-1] + 1
step = ks[i0:i1]
t0, t1 = step[0][0], max(k[1] for k in step)
print(f"step: {len(step)} launches, {(t1 - t0) / 1e6:.2f} ms")
byq = collections.defaultdict(list)
for k in step:
    byq[k[3]].append(k)
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _, _ in lst)
    gaps = [(lst[i + 1][0] - lst[i][1], lst[i][2], lst[i + 1][2]) for i in range(len(lst) - 1)]
    idle = sum(max(0, g[0]) for g in gaps)
    print(f"queue {q}: {len(lst)} launches, busy {busy / 1e6:.2f} ms, idle between launches {idle / 1e6:.2f} ms "
          f"(first start +{(lst[0][0] - t0) / 1e6:.2f} ms, last end +{(lst[-1][1] - t0) / 1e6:.2f} ms)")
    hist = collections.Counter()
    for g, _, _ in gaps:
        hist["<2us" if g < 2000 else "2-10us" if g < 10000 else "10-50us" if g < 50000 else ">50us"] += 1
    print("   gap histogram:", dict(hist))
    order = sorted(range(len(gaps)), key=lambda i: -gaps[i][0])[:6]
    for i in order:
        ctx_before = " > ".join(k[2][:28] for k in lst[max(0, i - 2):i + 1])
        ctx_after = " > ".join(k[2][:28] for k in lst[i + 1:i + 4])
        print(f"   {gaps[i][0] / 1e3:8.1f} us idle at +{(lst[i][1] - t0) / 1e6:6.2f} ms:  {ctx_before}  ||  {ctx_after}")
ev = sorted([(s, 1) for s, e, _, _ in step] + [(e, -1) for s, e, _, _ in step])
depth, last, union = 0, t0, 0
for t, d in ev:
    if depth > 0:
        union += t - last
    depth += d
    last = t
print(f"device busy (union over queues): {union / 1e6:.2f} ms of {(t1 - t0) / 1e6:.2f} ms")
