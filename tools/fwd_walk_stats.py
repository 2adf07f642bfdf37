"""tools/fwd_walk_stats.py (GPU box, rig build with -DFR_FWD_STATS: tools/build_variant.sh fwdstats -DFR_FWD_STATS, FISHER_RAST_SO set to it):
where the waves of k_render_forward_walk spend their time on one 256 x 256 view of the benchmark room -- s_memtime ticks (/ 64) in the
key stream / the chunk set-up / the walk, chunks, wave-level walk steps, 64-key windows streamed, the tile's list length."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")]
import numpy as np, torch
from fisher_rast import synthetic, ops
from models.SLAM.utils.recon_helpers import setup_camera
dev = torch.device("cuda:0")
P, W, H = 500_000, 256, 256
act = {k: v.to(dev) for k, v in synthetic.activate(synthetic.room_shell(P, 2)).items()}
cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
w2c = synthetic.invert_rigid(synthetic.candidate_poses(16, 2))[int(sys.argv[1]) if len(sys.argv) > 1 else 1].to(dev)
pts = act["means3D"]
tp = (w2c @ torch.cat((pts, torch.ones_like(pts[:, :1])), 1).T).T[:, :3].contiguous()
e = torch.Tensor([])
R, color, radii, geom, binning, img, depth = ops.rasterize_forward(cam.bg, tp, act["rgb_colors"], act["opacities"], act["scales"], act["rotations"], 1.0, e,
                                                                    cam.viewmatrix, cam.projmatrix, cam.tanfovx, cam.tanfovy, H, W, e, 0, cam.campos, False)
d = depth[0].cpu().numpy()
rows = []
for ty in range(H // 16):
    for tx in range(W // 16):
        for w in range(4):
            v = d[ty * 16 + 4 * w, tx * 16: tx * 16 + 8]
            rows.append((v[0] + v[1] + v[2],) + tuple(v[:7]) + (ty * 16 + tx, w))
rows.sort(reverse=True)
tick_us = 1.0                 # (ticks / 64 as stamped; the slowest wave's total is the kernel's duration)
print("s_memtime ticks / 64; slowest waves first:  total | stream setup walk | chunks steps windows | list length | tile strip")
for r in rows[:12]:
    print(f"{r[0] * tick_us:8.1f} | {r[1] * tick_us:7.1f} {r[2] * tick_us:7.1f} {r[3] * tick_us:7.1f} | {int(r[4]):5d} {int(r[5]):6d} {int(r[6]):5d} | {int(r[7]):6d} | {r[8]} {r[9]}")
a = np.array([r[:8] for r in rows])
print("mean over the 1024 waves: total %.1f = stream %.1f + setup %.1f + walk %.1f; chunks %.1f, steps %.1f, windows %.1f" %
      (a[:, 0].mean() * tick_us, a[:, 1].mean() * tick_us, a[:, 2].mean() * tick_us, a[:, 3].mean() * tick_us, a[:, 4].mean(), a[:, 5].mean(), a[:, 6].mean()))
