"""tools/outh11_bench.py (GPU box): the out_H modes of GaussianObjectSLAM at the bench size -- 11 columns, constant gradient
(compute_Hessian / compute_H_train) and per-view gradient images (POp-GS probes) -- ms per call, front end included.
(-DFR_AB rig builds only, FISHER_RAST_SO=tools/_build/ab_rig.so) FR_DEBUG_MODE=22 keeps round 2's two-pass kernel (k_fisher_tile_v2<11>) for A/B runs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")]
import numpy as np, torch
from fisher_rast import synthetic
from fisher_rast.ops import FisherScorer
from models.SLAM.utils.recon_helpers import setup_camera
dev = torch.device("cuda:0")
P, W, H = 500_000, 256, 256
act = synthetic.activate(synthetic.room_shell(P, 2))
cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
sc = FisherScorer(cam, *(act[k].to(dev) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")), columns=11)
w2c = synthetic.invert_rigid(synthetic.candidate_poses(64, 2)).to(dev)
out = torch.zeros((P, 11), device=dev)


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


for V in (1, 16, 64):
    print(f"11 columns, constant gradient, {V:2d} views accumulated: {t(lambda: sc.launch(w2c[:V], out_H=out)):.3f} ms")
z = torch.randn((16, 3, H, W), generator=torch.Generator().manual_seed(1)).to(dev)
per = torch.zeros((16, P, 11), device=dev)
print(f"11 columns, gradient images, 16 views, one diagonal per view: {t(lambda: sc.launch(w2c[:16], out_H=per, out_H_per_view=True, dL_image=z)):.3f} ms")
print("checksum", float(out.double().sum()), float(per.double().sum()))
