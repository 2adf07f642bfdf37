"""Target of the rocprofv3 --pmc passes: a few launches of the 64-view scorer on the bench workload (configs[1]).
usage: python3 tools/pmc_target.py [columns]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fisher-nerf-customized_amd"))
import numpy as np, torch
from fisher_rast import synthetic
from fisher_rast.ops import FisherScorer
from models.SLAM.utils.recon_helpers import setup_camera
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
P, V, W, H = 500_000, 64, 256, 256
act = synthetic.activate(synthetic.room_shell(P, 2))
cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
sc = FisherScorer(cam, *(act[k].to(dev) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")), columns=C)
w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, 2)).to(dev)
kf = synthetic.invert_rigid(synthetic.candidate_poses(16, 102)).to(dev)
Ht = torch.zeros((P, C), device=dev)
sc.run(kf, out_H=Ht)
Hi = torch.reciprocal(Ht + 0.1)
sc.run(w2c, H_inv=Hi)
for _ in range(3):
    r = sc.launch(w2c, H_inv=Hi)
torch.cuda.synchronize()
print("scores[:4]", r["scores"][:4].cpu().tolist())
