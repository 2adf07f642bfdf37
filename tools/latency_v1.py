"""Single-view latency of the drop-in calls the tester makes once per path step (compute_Hessian, gaussian.py:1503-1570):
wall time per call and, under rocprofv3 --kernel-trace, the kernels behind it.  usage: python3 tools/latency_v1.py [n_calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fisher-nerf-customized_amd"))
import numpy as np, torch
import models.gaussian_slam as mgs
from fisher_rast import synthetic
dev = torch.device("cuda:0")
P, W, H = 500_000, 256, 256
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
slam = mgs.GaussianSLAM(params={k: v.to(dev) for k, v in synthetic.room_shell(P, 2).items()}, intrinsics=synthetic.intrinsics(W, H), width=W, height=H, device=dev)
w2cs = synthetic.invert_rigid(synthetic.candidate_poses(n, 2)).to(dev)
slam.compute_Hessian(w2cs[0], return_points=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for w in w2cs:
    h = slam.compute_Hessian(w, return_points=True)
    float(h[0, 0])
dt = (time.perf_counter() - t0) / n
print(f"compute_Hessian, one view, result read on the host: {dt * 1e3:.3f} ms per call")
