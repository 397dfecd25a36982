"""bench.py with module-level flags of zs3_amd.ops / functional set first (the decided switches are constants since round 5):
   python tools/probe/bench_flags.py ops.PW16=True functional.DEFER_BN_APPLY=False -- --dtype bf16 --steps 20 ..."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sep = sys.argv.index("--") if "--" in sys.argv else len(sys.argv)
flags, rest = sys.argv[1:sep], sys.argv[sep + 1:]
import zs3_amd.ops, zs3_amd.functional, zs3_amd.gmmn_trainer   # noqa: E401
for f in flags:
    name, val = f.split("=", 1)
    mod, attr = name.rsplit(".", 1)
    m = sys.modules["zs3_amd." + mod]
    assert hasattr(m, attr), name
    setattr(m, attr, eval(val))
sys.argv = [os.path.join(ROOT, "bench.py")] + rest
runpy.run_path(sys.argv[0], run_name="__main__")
