"""Distribution of the per-(view, tile) list lengths of the bench workload (sort tiers), read from the scorer's workspace."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fisher-nerf-customized_amd"))
import numpy as np, torch
from fisher_rast import synthetic, _lib
from fisher_rast.ops import FisherScorer
from models.SLAM.utils.recon_helpers import setup_camera
dev = torch.device("cuda:0")
P, V, W, H = 500_000, 64, 256, 256
act = synthetic.activate(synthetic.room_shell(P, 2))
cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
sc = FisherScorer(cam, *(act[k].to(dev) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")))
w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, 2)).to(dev)
Hi = torch.rand((P, 4), device=dev)
r = sc.run(w2c, H_inv=Hi)
torch.cuda.synchronize()
cap = V * sc._keys_per_view()
off = (ctypes.c_size_t * 8)()
_lib.check(_lib.load().fr_fisher_workspace_layout(P, W, H, V, cap, 4, off), "layout")
ws = sc._ws[0]
T = (W // 16) * (H // 16)
cnt = ws[off[0]:off[0] + V * T * 4].view(torch.int32).cpu().numpy().astype(np.int64)
print("tiles", cnt.size, "instances", cnt.sum(), "mean", cnt.mean(), "max", cnt.max())
for lo, hi in ((0, 0), (1, 512), (513, 1024), (1025, 2048), (2049, 4096), (4097, 8192), (8193, 16384), (16385, 1 << 30)):
    m = (cnt >= lo) & (cnt <= hi)
    print(f"n in [{lo},{hi}]: tiles {m.sum():6d}  keys {cnt[m].sum():9d}")
# sortedness of every segment (packed lists or fixed segments: the offsets say where a list starts)
toff = ws[off[1]:off[1] + V * T * 4].view(torch.int32).long()
cnt_d = torch.from_numpy(cnt).to(dev)
seg = torch.repeat_interleave(torch.arange(V * T, device=dev), cnt_d)
first = torch.cumsum(cnt_d, 0) - cnt_d
pos = toff[seg] + (torch.arange(int(cnt.sum()), device=dev) - first[seg])
keys = ws[off[2]:off[2] + cap * 8].view(torch.int64)[pos]
d = keys[1:] > keys[:-1]
same = seg[1:] == seg[:-1]
print("all segments strictly ascending:", bool((d | ~same).all()), " fixed segments of", sc.tile_capacity, "keys" if sc.tile_capacity else "(packed lists)")
