"""Host time stamps inside GMMNStep.__call__ over a bench run (ZS3_GMMN_TICKS=1): average host milliseconds per section of the
step in steady state.   ZS3_GMMN_TICKS=1 python tools/probe/gmmn_ticks.py <bench args>"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["ZS3_GMMN_TICKS"] = "1"
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
finally:
    import torch
    import zs3_amd.gmmn_trainer as g
    torch.cuda.synchronize()
    t = g._TICKS
    steps, cur = [], []
    for rec in t:
        if rec[0] == "take" and cur:
            steps.append(cur)
            cur = []
        cur.append(rec)
    steps = steps[len(steps) // 2:]
    host, gpu = {}, {}
    for a, b in zip(steps[:-1], steps[1:]):
        for tag, ts, ev in a + [("next-take", b[0][1], b[0][2])]:
            host.setdefault(tag, []).append((ts - a[0][1]) * 1e3)
            gpu.setdefault(tag, []).append(a[0][2].elapsed_time(ev))
    print("point            host ms   GPU ms   (after the step's first point; GPU: when the stream reached it)", file=sys.stderr)
    for k in host:
        print(f"{k:14s} {sum(host[k]) / len(host[k]):8.2f} {sum(gpu[k]) / len(gpu[k]):8.2f}", file=sys.stderr)
