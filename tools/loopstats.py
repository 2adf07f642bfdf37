import sys, os, subprocess, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/fisher-nerf-customized_amd")
import numpy as np, torch
mode = int(os.environ.get("FR_DEBUG_MODE", "0"))
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
r = sc.run(w2c, H_inv=Hi)
print("mode", mode, "sum over 64 views", float(r["scores"].double().sum()), "tile instances", int(r["num_rendered"].sum()))
