#!/usr/bin/env python3
"""H_inv AND out_H in one fr_fisher_views call (fisher_rast/path_eval.py: per-view weights, per-view diagonals) against the two separate
launches (out_H per view, then the scores from it on the device) on the benchmark scene."""
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
V = int(sys.argv[3]) if len(sys.argv) > 3 else 32
W = H = 256
act = {k: v.to(dev) for k, v in synthetic.activate(synthetic.room_shell(P, 2)).items()}
cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
sc = FisherScorer(cam, act["means3D"], act["rgb_colors"], act["rotations"], act["opacities"], act["scales"], columns=C)
w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, 2)).to(dev)
Hv = (torch.rand((V, P, C), generator=torch.Generator().manual_seed(1)) + 0.05).to(dev)
cur = torch.zeros((V, P, C), device=dev)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def both():
    cur.zero_()
    return sc.run(w2c, H_inv=Hv, H_inv_per_view=True, out_H=cur, out_H_per_view=True)["scores"]


def split():
    cur.zero_()
    sc.run(w2c, out_H=cur, out_H_per_view=True)
    return (cur * Hv).sum(dim=(1, 2))


a = both().clone(); b = split().clone()
print("one call", a[:4].tolist(), "split", b[:4].tolist(), "zeros", int((b == 0).sum()), int((a == 0).sum()))
print(f"P={P} C={C} V={V}: one call {timed(both):.3f} ms, out_H launch + torch reduction {timed(split):.3f} ms, max rel diff {float(((a - b).abs() / b.abs()).max()):.2e}")
