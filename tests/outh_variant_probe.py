"""Helper of test_gpu_outh_variants.py (not a test): per-view diagonals of a small synthetic scene through fr_fisher_views'
out_H mode, written to the .npy named on the command line.  The parent sets FR_DEBUG_MODE (read once per process by the
library) to pick the kernel generation."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd"), os.path.dirname(os.path.abspath(__file__))):
    sys.path.insert(0, p)
import numpy as np
import torch

import __graft_entry__ as entry

entry.build()
from fisher_rast import synthetic
from fisher_rast.ops import FisherScorer
from models.SLAM.utils.recon_helpers import setup_camera

P, V, W, H, seed = (int(a) for a in sys.argv[2:7])
dev = torch.device("cuda:0")
act = {k: v.to(dev) for k, v in synthetic.activate(synthetic.room_shell(P, seed=seed)).items()}
cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, seed=seed)).to(dev)
sc = FisherScorer(cam, act["means3D"], act["rgb_colors"], act["rotations"], act["opacities"], act["scales"])
out = torch.zeros((V, P, 4), device=dev)
sc.run(w2c, out_H=out, out_H_per_view=True)
np.save(sys.argv[1], out.cpu().numpy())
# ... and the scores of the same views with fixed weights (compared bit for bit between the record layouts)
H_inv = (torch.rand((P, 4), generator=torch.Generator().manual_seed(3)) * 2.0 + 0.05).to(dev)
np.save(sys.argv[1].replace(".npy", "_scores.npy"), sc.run(w2c, H_inv=H_inv)["scores"].cpu().numpy())
