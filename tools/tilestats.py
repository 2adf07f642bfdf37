import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/fisher-nerf-customized_amd")
import numpy as np, torch
from fisher_rast import synthetic
from fisher_rast.ops import FisherScorer
from models.SLAM.utils.recon_helpers import setup_camera
dev = torch.device("cuda:0")
P, V, W, H = 500_000, 64, 256, 256
act = synthetic.activate(synthetic.room_shell(P, 2))
cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
sc = FisherScorer(cam, *(act[k].to(dev) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")))
w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, 2)).to(dev)
Hi = torch.rand((P, 4), device=dev)
r = sc.launch(w2c, H_inv=Hi)
torch.cuda.synchronize()
print("status", r["status"].cpu().tolist())
nr = r["num_rendered"].cpu().numpy(); print("num_rendered min/mean/max", nr.min(), nr.mean(), nr.max())
# tile counts live at a known offset of the workspace: recompute layout by reading back through ctypes is overkill; use status[2]
