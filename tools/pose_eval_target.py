"""Target of a rocprofv3 --kernel-trace run: a few GaussianSLAM.pose_eval(poses) calls on the bench workload (16 keyframes, 64 poses).
tools/timeline.py on the trace shows what one call puts on the GPU and where the host leaves it idle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")]
import torch
import models.gaussian_slam as mgs
from fisher_rast import synthetic
dev = torch.device("cuda:0")
P, V, W, H, seed = 500_000, 64, 256, 256, 2
raw = synthetic.room_shell(P, seed)
slam = mgs.GaussianSLAM(params={k: v.to(dev) for k, v in raw.items()}, intrinsics=synthetic.intrinsics(W, H), width=W, height=H, device=dev)
for kf in synthetic.invert_rigid(synthetic.candidate_poses(16, seed + 100)):
    slam.add_keyframe(kf.to(dev))
poses = [p.to(dev) for p in synthetic.candidate_poses(V, seed)]
for _ in range(3):
    slam.pose_eval(poses)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    scores, _ = slam.pose_eval(poses)
torch.cuda.synchronize()
print("pose_eval %.3f ms per call" % ((time.perf_counter() - t0) / 5 * 1e3), scores[:3].tolist())
