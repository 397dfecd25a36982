"""bench.py with the persistent pointwise kernel's resident workgroups per launch set first (zs3_conv_pw_set_wgs; default 256, the
128-row tiles launch twice that): python tools/probe/bench_pw_wgs.py 192 -- <bench args>"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sep = sys.argv.index("--")
n = int(sys.argv[1])
from zs3_amd._lib import lib
print("previous:", lib().zs3_conv_pw_set_wgs(n), "->", n, file=sys.stderr)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[sep + 1:]
runpy.run_path(sys.argv[0], run_name="__main__")
