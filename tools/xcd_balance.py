"""tools/xcd_balance.py (GPU box): how evenly the XCD-aware tile map (view v -> XCD v mod 8) spreads the walk's work on the bench
workload: tile instances listed per view (sum of the tile counts), summed per XCD, against a balanced deal of the views."""
import ctypes, os, sys
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
w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, 2)).to(dev)
Hi = torch.rand((P, 4), generator=torch.Generator().manual_seed(1)).to(dev)
r = sc.launch(w2c, H_inv=Hi); torch.cuda.synchronize()
ws = sc._ws[0]
o = (ctypes.c_size_t * 8)()
sc.lib.fr_fisher_workspace_layout(P, W, H, V, V * sc._keys_per_view(), 4, o)
T = 256
cnt = ws[o[0]:o[0] + V * T * 4].view(torch.int32).view(V, T).cpu().numpy().astype(np.int64)
per_view = cnt.sum(1)
# the walk's cost per tile grows a little faster than its list (longer lists = more chunks per strip), n is a fair proxy
print("listed per view: min %d  mean %d  max %d" % (per_view.min(), per_view.mean(), per_view.max()))
cur = np.array([per_view[x::8].sum() for x in range(8)])
print("per XCD, view v -> XCD v mod 8:", cur.tolist(), " max / mean = %.3f" % (cur.max() / cur.mean()))
order = np.argsort(-per_view)
bins = np.zeros(8, np.int64)
for rnd in range(V // 8):
    vs = order[8 * rnd: 8 * rnd + 8]
    xs = np.argsort(bins)            # heaviest view of the round to the lightest XCD so far
    for v, x in zip(vs, xs):
        bins[x] += per_view[v]
print("per XCD, views dealt by weight:", bins.tolist(), " max / mean = %.3f" % (bins.max() / bins.mean()))
