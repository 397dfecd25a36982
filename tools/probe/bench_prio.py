"""bench.py with the whole step on a HIGH-priority stream (its own hardware-queue pool), lanes / weight gradient at normal priority:
   python tools/probe/bench_prio.py <bench args>"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
torch.cuda.set_device(0)
s = torch.cuda.Stream(priority=-1)
torch.cuda.set_stream(s)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
