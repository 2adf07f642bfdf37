#!/usr/bin/env python3
"""the reference's calling pattern on MI355X -- one view at a time through the drop-in
`GaussianRasterizer` autograd API (gaussian.py:1523-1567: torch transform, fresh rendervar, forward, backward(power=2),
cat, sum(...).item()), i.e. per-view allocations and host syncs, using this repository's own kernels.
Prints views/s; this is the "1x" the batched scorer's speed-up is quoted against (no CUDA number exists)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from fisher_rast import synthetic
from models.SLAM.utils.recon_helpers import setup_camera
from diff_gaussian_rasterization import GaussianRasterizer as Renderer

dev = torch.device("cuda:0")
P, V, W, H = 500_000, int(sys.argv[1]) if len(sys.argv) > 1 else 16, 256, 256
params = {k: v.to(dev) for k, v in synthetic.room_shell(P, 2).items()}
cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
w2cs = synthetic.invert_rigid(synthetic.candidate_poses(V, 2)).to(dev)
H_inv = torch.rand((P, 4), device=dev)

def compute_Hessian(rel_w2c):
    with torch.no_grad():
        pts = params['means3D']
        pts4 = torch.cat((pts, torch.ones(pts.shape[0], 1, device=dev)), dim=1)
        tp = (rel_w2c @ pts4.T).T[:, :3]
        rot = torch.nn.functional.normalize(params['unnorm_rotations'])
        op = torch.sigmoid(params['logit_opacities'])
        sc = torch.exp(params['log_scales'])
    rv = {'means3D': tp.requires_grad_(True), 'colors_precomp': params['rgb_colors'].requires_grad_(True),
          'rotations': rot.requires_grad_(True), 'opacities': op.requires_grad_(True), 'scales': sc.requires_grad_(True),
          'means2D': torch.zeros_like(tp, requires_grad=True, device=dev) + 0}
    rv['means2D'].retain_grad()
    im, radius, _ = Renderer(raster_settings=cam, backward_power=2)(**rv)
    im.backward(gradient=torch.ones_like(im) * 1e-3)
    vis = int((radius > 0).sum().item())
    cur_H = torch.cat([tp.grad.detach().reshape(P, -1), op.grad.detach().reshape(P, -1)], dim=1)
    for k, v in rv.items():
        v.grad.fill_(0.)
    return cur_H

compute_Hessian(w2cs[0]); torch.cuda.synchronize()
t0 = time.perf_counter()
scores = []
for v in range(V):
    cur_H = compute_Hessian(w2cs[v])
    scores.append(torch.sum(cur_H * H_inv).item())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"B4 serial loop: {V} views in {dt*1e3:.1f} ms -> {V/dt:.1f} views/s ({dt/V*1e3:.2f} ms/view)")
