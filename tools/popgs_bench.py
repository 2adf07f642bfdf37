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


def autograd_route(slam, c2ws, K):
    """The reference's calling pattern on this repository's kernels: per pose one forward of the drop-in autograd rasteriser
    (backward_power=2) and K backward passes with random upstream gradients; squares averaged (gaussian_object.py:2066-2107)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from models.SLAM.utils.slam_helpers import transformed_params2rendervar
    out = []
    for c2w in c2ws:
        w2c = torch.linalg.inv(c2w)
        pts = slam.params["means3D"]
        tp = (w2c @ torch.cat((pts, torch.ones_like(pts[:, :1])), 1).T).T[:, :3].contiguous()
        rv = {k: (v.detach().clone().requires_grad_(True) if k != "means2D" else v) for k, v in transformed_params2rendervar(slam.params, tp).items()}
        im, _, _ = GaussianRasterizer(raster_settings=slam.cam, backward_power=2)(**rv)
        acc = 0.0
        for k in range(K):
            for t in rv.values():
                if t.is_leaf:
                    t.grad = None
            im.backward(gradient=torch.randn_like(im), retain_graph=k + 1 < K)
            g = torch.cat([rv[n].grad.reshape(-1) for n in ("means3D", "opacities", "rotations", "scales")])
            acc = acc + g * g
        out.append(acc / K)
    return torch.stack(out)


for name, n, fn in (("fused", V, lambda c: slam.pose_eval_popgs(c, K=K)), ("autograd route", min(V, 4), lambda c: autograd_route(slam, c, K))):
    fn(poses[:2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(poses[:n])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"POp-GS diag estimates P={P} K={K} {name}: {n} poses in {dt*1e3:.1f} ms -> {dt/n*1e3:.2f} ms per pose")
