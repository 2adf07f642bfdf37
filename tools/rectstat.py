"""tools/rectstat.py (GPU box): how divergent the per-splat tile-rectangle loops of the front end are on the bench workload --
reads the packed-list front end's visible lists back and compares, per wave of 64 consecutive entries, the largest rectangle
(what a wave's loop runs) with the mean (what a perfectly balanced loop would run).  r03: mean 1.93 tiles, 0.9 % of the entries
above 4 tiles, wave loops 3.4x the balanced length (6.5 trips)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")]
import numpy as np, torch
from fisher_rast import synthetic
from fisher_rast.ops import FisherScorer
from models.SLAM.utils.recon_helpers import setup_camera
dev = torch.device("cuda:0")
P, V, W, H = 500_000, 64, 256, 256
act = synthetic.activate(synthetic.room_shell(P, 2))
cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
sc = FisherScorer(cam, *(act[k].to(dev) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")))
sc.tile_capacity = 0
w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, 2)).to(dev)
Hi = torch.rand((P, 4), generator=torch.Generator().manual_seed(1)).to(dev)
r = sc.launch(w2c, H_inv=Hi); torch.cuda.synchronize()
ws = sc._ws[0]
mr = V * sc._keys_per_view()
o = (ctypes.c_size_t * 8)()
sc.lib.fr_fisher_workspace_layout(P, W, H, V, mr, 4, o)
vis_n_off = o[7]
# G and nblk: cap = (vis_n_off rounded) / (V*nblk*16)
for G in range(1, 33):
    nblk = (P + 256 * G - 1) // (256 * G)
    if ((V * nblk * 256 * G * 16 + 255) & ~255) == vis_n_off: break
cap = 256 * G
print("G", G, "nblk", nblk, "cap", cap)
lst = ws[:V * nblk * cap * 16].view(torch.int32).view(V, nblk, cap, 4).cpu().numpy().view(np.uint32)
vn = ws[vis_n_off:vis_n_off + V * nblk * 4].view(torch.int32).view(V, nblk).cpu().numpy()
tot = 0; steps64 = 0; steps_sum = 0; areas = []
for v in range(0, V, 4):
    for b in range(nblk):
        n = vn[v, b]
        e = lst[v, b, :n]
        x0 = e[:, 2] & 0xffff; y0 = e[:, 2] >> 16; x1 = e[:, 3] & 0xffff; y1 = e[:, 3] >> 16
        a = ((x1 - x0) * (y1 - y0)).astype(np.int64)
        areas.append(a)
        for c in range(0, n, 64):
            ch = a[c:c + 64]
            steps64 += ch.max(); steps_sum += ch.sum()
a = np.concatenate(areas)
print("entries", a.size, "mean area", a.mean(), "max", a.max(), "p99", np.percentile(a, 99), "frac area>4", (a > 4).mean(), "share of keys in area>4", a[a > 4].sum() / a.sum(), "area>16 share", a[a > 16].sum() / a.sum())
print("wave steps (max per 64)", steps64, "ideal (sum/64)", steps_sum / 64, "ratio", steps64 / (steps_sum / 64))
