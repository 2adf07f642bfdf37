#!/usr/bin/env python3
"""Planner-side kernels alone (for rocprofv3): N occupancy updates and frontier builds on the synthetic room."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from fisher_rast import synthetic
from oracle.occupancy_frontier import room_depth          # input generator only
from planning import AstarPlanner

dev = torch.device("cuda:0")
W = H = 256
K = synthetic.intrinsics(W, H)
poses = synthetic.candidate_poses(8, 202).numpy().astype(np.float32)
pts = synthetic.room_shell(200_000, 2)["means3D"].to(dev)
pl = AstarPlanner(device=dev, cell_size=0.05, frontier_select_method="combined")
pl.init(torch.eye(4), torch.from_numpy(np.asarray(K, dtype=np.float32)))
depths = [torch.from_numpy(room_depth(p, W, H, K)).to(dev) for p in poses]
c2ws = [torch.from_numpy(p).to(dev) for p in poses]
for rep in range(3):
    for t, (d, c) in enumerate(zip(depths, c2ws)):
        pl.update_occ_map(d, c, t)
    fr, free = pl.build_frontiers(pts)
torch.cuda.synchronize()
print("free cells", int(free.sum()), "frontier cells", 0 if fr is None else len(fr))
