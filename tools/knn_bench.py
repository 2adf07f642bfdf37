import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from fisher_rast import synthetic, ops
dev = torch.device("cuda:0")
for P in (100_000, 500_000, 2_000_000):
    pts = synthetic.room_shell(P, 2)["means3D"].to(dev)
    out = ops.knn_dist2(pts); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): out = ops.knn_dist2(pts)
    torch.cuda.synchronize()
    print(f"knn P={P}: {(time.perf_counter()-t)/3*1e3:.2f} ms  mean dist2 {float(out.mean()):.3e}")
