#!/usr/bin/env python3
"""POp-GS "simple diag" pose evaluation (gaussian_object.py:1619-1662): K random probes per pose.  Times the reference's
route (one forward + K generic power-2 backward passes per pose through the autograd rasteriser) against the fused route
(all poses x probes as views of fr_fisher_views with per-view upstream-gradient images)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from fisher_rast import synthetic
import models.gaussian_slam as mgs

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
V = int(sys.argv[2]) if len(sys.argv) > 2 else 16
K = 4
W = H = 256
params = {k: v.to(dev) for k, v in synthetic.room_shell(P, 2).items()}
slam = mgs.GaussianObjectSLAM(params=params, intrinsics=synthetic.intrinsics(W, H), width=W, height=H, device=dev)
for kf in synthetic.invert_rigid(synthetic.candidate_poses(4, 102)):
    slam.add_keyframe(kf.to(dev))
poses = [p.to(dev) for p in synthetic.candidate_poses(V, 2)]
for fused in (True, False):
    n = V if fused else min(V, 4)
    slam.pose_eval_popgs(poses[:2], K=K, fused=fused)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s, _ = slam.pose_eval_popgs(poses[:n], K=K, fused=fused)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"pose_eval_popgs P={P} K={K} {'fused' if fused else 'autograd route'}: {n} poses in {dt*1e3:.1f} ms -> {dt/n*1e3:.2f} ms per pose (incl. H_train over 4 keyframes)")
