#!/usr/bin/env python3
"""out_H modes of fr_fisher_views on the benchmark scene: H_train over 16 keyframes (accumulated), 64 views with a diagonal
per view, and one view.  (-DFR_AB rig builds only, FISHER_RAST_SO=tools/_build/ab_rig.so) FR_DEBUG_MODE=9 in the environment keeps the second-generation two-pass kernel (A/B)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from fisher_rast import synthetic
from fisher_rast.ops import FisherScorer
from models.SLAM.utils.recon_helpers import setup_camera

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 4
W = H = 256
act = {k: v.to(dev) for k, v in synthetic.activate(synthetic.room_shell(P, 2)).items()}
cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
sc = FisherScorer(cam, act["means3D"], act["rgb_colors"], act["rotations"], act["opacities"], act["scales"], columns=C)
kf = synthetic.invert_rigid(synthetic.candidate_poses(16, 102)).to(dev)
views = synthetic.invert_rigid(synthetic.candidate_poses(64, 2)).to(dev)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


mode = os.environ.get("FR_DEBUG_MODE", "-")
print(f"FR_DEBUG_MODE={mode} P={P} C={C}")
allv = synthetic.invert_rigid(synthetic.candidate_poses(64, 2)).to(dev)
Ht = torch.zeros((P, C), device=dev)
for V in (1, 2, 4, 8, 16, 64):
    print(f"  {V:3d} views accumulated : {timed(lambda: sc.run(allv[:V], out_H=Ht), 50 if V < 8 else 10):.3f} ms")
Hv = torch.zeros((64, P, C), device=dev)
print(f"   64 views, one diagonal per view : {timed(lambda: sc.run(allv, out_H=Hv, out_H_per_view=True)):.3f} ms")
print(f"  checksum {float(Hv.sum()):.6e}")
