#!/usr/bin/env python3
"""Golden vectors produced by RUNNING the reference (imported from /root/reference in the build container only):
`compute_next_campos` (models/SLAM/utils/slam_external.py:44-65) and `matrix_to_quaternion`
(models/SLAM/utils/slam_helpers.py:102-162), the two hot-path-adjacent helpers that import with numpy + torch alone
(SURVEY.md 8c).  Inputs and outputs only; no reference source is copied.  Run from the repo root:
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_reference_vectors.py
"""
import os

import numpy as np
import torch

from models.SLAM.utils.slam_external import compute_next_campos          # reference
from models.SLAM.utils.slam_helpers import matrix_to_quaternion          # reference

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(1234)


def rand_pose():
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    r, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                  [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                  [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
    H = np.eye(4)
    H[:3, :3] = R
    H[:3, 3] = rng.uniform(-4, 4, 3)
    return H


poses = np.stack([rand_pose() for _ in range(12)])
actions = rng.integers(0, 4, size=(12, 25))          # 0 = stop/no-op, 1 forward, 2 / 3 turns
steps = [(0.065, 10.0), (0.25, 30.0)]
out = dict(poses=poses, actions=actions, steps=np.array(steps))
for si, (fs, ta) in enumerate(steps):
    traj = np.zeros((12, 25, 4, 4))
    for i in range(12):
        H = poses[i]
        for t in range(25):
            H = compute_next_campos(H, int(actions[i, t]), fs, ta)
            traj[i, t] = H
    out[f"traj_{si}"] = traj
R = torch.tensor(poses[:, :3, :3], dtype=torch.float32)
out["quat_wxyz"] = matrix_to_quaternion(R).numpy()
np.savez_compressed(os.path.join(HERE, "reference_pose_helpers.npz"), **out)
print("written", {k: v.shape for k, v in out.items()})

# ---- checkpoint format: a `params{t}.npz` written by the reference's own save_params_ckpt (common_utils.py:45-59) ----
import shutil
import tempfile

from models.SLAM.utils.common_utils import save_params_ckpt              # reference

g = torch.Generator().manual_seed(77)
N = 24
ck_params = {
    'means3D': torch.randn((N, 3), generator=g),
    'rgb_colors': torch.rand((N, 3), generator=g),
    'unnorm_rotations': torch.randn((N, 4), generator=g),
    'logit_opacities': torch.randn((N, 1), generator=g),
    'log_scales': torch.randn((N, 1), generator=g) - 3.0,
    'cam_unnorm_rots': torch.randn((1, 4, 5), generator=g),
    'cam_trans': torch.randn((1, 3, 5), generator=g),
}
tmp = tempfile.mkdtemp()
save_params_ckpt(ck_params, tmp, 7, Uncertainty=torch.rand((N,), generator=g), occ_map=np.arange(12, dtype=np.float32).reshape(3, 4))
shutil.copy(os.path.join(tmp, "params7.npz"), os.path.join(HERE, "reference_params7.npz"))
shutil.rmtree(tmp)
print("written reference_params7.npz", sorted(np.load(os.path.join(HERE, "reference_params7.npz")).keys()))
