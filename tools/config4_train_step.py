#!/usr/bin/env python3
"""BASELINE.json configs[3]: 2M Gaussians, 512x512, one forward + power=1 backward with dL_dpix = N(0,1) (seed 44) through
the drop-in rasteriser (train-step proxy).  Reports ms per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from fisher_rast import synthetic, ops
from models.SLAM.utils.recon_helpers import setup_camera

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
W = H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
act = {k: v.to(dev) for k, v in synthetic.activate(synthetic.room_shell(P, 4)).items()}
cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
w2c = synthetic.invert_rigid(synthetic.candidate_poses(1, 4))[0].to(dev)
pts = act["means3D"]
tp = (w2c @ torch.cat((pts, torch.ones_like(pts[:, :1])), 1).T).T[:, :3].contiguous()
g = torch.Generator().manual_seed(44)
dL = torch.randn((3, H, W), generator=g).to(dev)
e = torch.Tensor([])

def step():
    R, color, radii, geom, binning, img, depth = ops.rasterize_forward(cam.bg, tp, act["rgb_colors"], act["opacities"], act["scales"], act["rotations"], 1.0, e,
                                                                   cam.viewmatrix, cam.projmatrix, cam.tanfovx, cam.tanfovy, H, W, e, 0, cam.campos, False)
    grads = ops.rasterize_backward(cam.bg, tp, radii, act["rgb_colors"], act["scales"], act["rotations"], 1.0, e, cam.viewmatrix, cam.projmatrix,
                                   cam.tanfovx, cam.tanfovy, dL, e, 0, cam.campos, geom, R, binning, img, 1)
    return R, grads

R, grads = step(); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 5
for _ in range(n):
    R, grads = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"config4: P={P} {W}x{H} tile instances {R}  forward+backward(power=1): {dt*1e3:.2f} ms/step; |dL_dmeans3D| = {float(grads[3].abs().sum()):.4e}")

# ---- the double render of the reference's get_loss (RGB, then depth / silhouette / depth^2) vs the fused pair -------------
z = tp[:, 2:3]
feats = torch.cat((z, torch.ones_like(z), z * z), 1).contiguous()
dL2 = torch.randn((3, H, W), generator=g).to(dev)
cfg_args = (P, H, W, cam.tanfovx, cam.tanfovy, 1.0, cam.bg, cam.viewmatrix, cam.projmatrix, cam.campos)


def two_renders():
    out = []
    for col, d in ((act["rgb_colors"], dL), (feats, dL2)):
        R, color, radii, geom, binning, img, depth = ops.rasterize_forward(cam.bg, tp, col, act["opacities"], act["scales"], act["rotations"], 1.0, e,
                                                                       cam.viewmatrix, cam.projmatrix, cam.tanfovx, cam.tanfovy, H, W, e, 0, cam.campos, False)
        out.append(ops.rasterize_backward(cam.bg, tp, radii, col, act["scales"], act["rotations"], 1.0, e, cam.viewmatrix, cam.projmatrix,
                                          cam.tanfovx, cam.tanfovy, d, e, 0, cam.campos, geom, R, binning, img, 1))
    return out


def fused_pair():
    R, color, radii, geom, binning, img, depth, fimg = ops.rasterize_forward(cam.bg, tp, act["rgb_colors"], act["opacities"], act["scales"], act["rotations"], 1.0, e,
                                                                         cam.viewmatrix, cam.projmatrix, cam.tanfovx, cam.tanfovy, H, W, e, 0, cam.campos, False,
                                                                         features=feats)
    return ops.rasterize_backward_pair(cam.bg, tp, radii, act["rgb_colors"], feats, act["scales"], act["rotations"], 1.0, e, cam.viewmatrix,
                                       cam.projmatrix, cam.tanfovx, cam.tanfovy, dL, dL2, cam.campos, geom, binning, img, num_rendered=R)


for name, fn in (("two renders (reference flow)", two_renders), ("fused pair", fused_pair)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    print(f"config4 get_loss renders, {name}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms/step")
