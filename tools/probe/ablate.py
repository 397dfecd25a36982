import sys, os, time, subprocess
here = os.path.dirname(os.path.abspath(__file__))
code = '''
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath("%s")))))
import torch
from zs3_amd import ops
dev = torch.device("cuda:0")
for (h, ci, co, k) in ((129,256,256,3),(33,1024,256,1),(33,256,1024,1)):
    x = torch.randn(16, h, h, ci, device=dev); wt = torch.randn(co, ci, k, k, device=dev) * 0.02
    wp = ops.prep_weight(wt)
    for _ in range(3): ops.conv2d_fwd(x, wp, 1, k // 2, 1, tile_cfg=11)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): ops.conv2d_fwd(x, wp, 1, k // 2, 1, tile_cfg=11)
    torch.cuda.synchronize(); print("  %%d^2 %%d->%%d k%%d: %%.1f us" %% (h, ci, co, k, (time.perf_counter()-t)/10*1e6))
''' % os.path.join(here, "x.py")
for ab, name in ((0, "full"), (1, "no MFMA"), (2, "no ds_read+MFMA"), (4, "no cvt+ds_write"), (8, "no global loads"), (12, "no loads, no stores (LDS read+MFMA only)"), (14, "barriers only"), (6, "global loads only")):
    print(f"ablate={ab} ({name})", flush=True)
    subprocess.run([sys.executable, "-c", code], env={**os.environ, "ZS3_ABLATE": str(ab)})
