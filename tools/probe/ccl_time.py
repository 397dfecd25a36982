import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from zs3_amd.gcn_context import construct_adj_mat
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
coarse = rng.randint(0, 8, size=(9, 9))
seg = torch.from_numpy(np.kron(coarse, np.ones((15, 15), dtype=np.int64))[:129, :129].copy()).to(dev)
emb = torch.randn(300, 129, 129, device=dev); feat = torch.randn(256, 129, 129, device=dev)
cg = construct_adj_mat(seg, emb, feat); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): cg = construct_adj_mat(seg, emb, feat)
torch.cuda.synchronize()
print(f"cluster graph of a 129x129 label map ({cg.num_clusters} clusters), incl. seed gathers and the size read-back: {(time.perf_counter()-t0)/10*1e3:.3f} ms")
