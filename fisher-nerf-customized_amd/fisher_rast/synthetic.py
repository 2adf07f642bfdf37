"""Seeded synthetic inputs for the benchmark and the parity tests.

Everything is generated with a CPU torch.Generator so that the same seed gives the same bytes on every
machine; callers move the tensors to the device.  Parameter names follow the reference's npz/param-dict
keys (models/SLAM/gaussian.py:156-168): means3D, rgb_colors, unnorm_rotations, logit_opacities, log_scales.
"""
import math

import numpy as np
import torch


def room_shell(P: int, seed: int, kind: str = "room_shell"):
    """An axis-aligned room x,z in [-5,5] m, y in [-1.25,1.25] m (camera frame is y-down, z-forward).
    room_shell: 85 % of the means on the six faces (area-weighted, +-1 cm normal jitter), 15 % inside.
    uniform_box: all means uniform inside."""
    g = torch.Generator().manual_seed(int(seed))
    half = torch.tensor([5.0, 1.25, 5.0])
    u = torch.rand((P, 3), generator=g) * 2 - 1
    means = u * half
    if kind == "room_shell":
        n_face = int(0.85 * P)
        areas = torch.tensor([half[1] * half[2], half[1] * half[2], half[0] * half[2], half[0] * half[2],
                              half[0] * half[1], half[0] * half[1]]) * 4
        face = torch.multinomial(areas / areas.sum(), n_face, replacement=True, generator=g)
        axis = face // 2
        sign = (face % 2).float() * 2 - 1
        jitter = torch.randn((n_face,), generator=g) * 0.01
        idx = torch.arange(n_face)
        means[idx, axis] = sign * half[axis] + jitter
    elif kind != "uniform_box":
        raise ValueError(kind)
    s = torch.exp(math.log(0.02) + 0.35 * torch.randn((P, 1), generator=g)).clamp(0.003, 0.1)
    scales3 = (s * (1 + 0.1 * torch.randn((P, 3), generator=g))).clamp_min(0.001)
    params = {
        "means3D": means.float().contiguous(),
        "rgb_colors": torch.rand((P, 3), generator=g).float(),
        "unnorm_rotations": torch.randn((P, 4), generator=g).float(),
        "logit_opacities": (1.0 + 1.5 * torch.randn((P, 1), generator=g)).float(),
        "log_scales": torch.log(scales3).float(),
    }
    return params


def activate(params):
    """The render variables of gaussian.py:1529-1533 (normalised rotations, sigmoid, exp, tile to 3)."""
    rot = torch.nn.functional.normalize(params["unnorm_rotations"])
    op = torch.sigmoid(params["logit_opacities"])
    sc = torch.exp(params["log_scales"])
    if sc.shape[-1] == 1:
        sc = torch.tile(sc, (1, 3))
    return dict(means3D=params["means3D"], rgb_colors=params["rgb_colors"], rotations=rot, opacities=op, scales=sc)


def candidate_poses(V: int, seed: int, cam_height: float = 0.0):
    """c2w [V,4,4] built like AstarPlanner.generate_candidate (planning/astar.py:1406-1423):
    position x,z ~ U[-4,4], y = cam_height, yaw ~ U[0,2pi) as quaternion (cos t/2, 0, sin t/2, 0) ->
    rotation matrix -> columns 0 and 1 negated (y axis facing down)."""
    g = torch.Generator().manual_seed(int(seed))
    pos = torch.rand((V, 2), generator=g) * 8 - 4
    theta = torch.rand((V,), generator=g) * 2 * math.pi
    r = torch.cos(theta / 2)
    y = torch.sin(theta / 2)
    x = torch.zeros_like(r)
    z = torch.zeros_like(r)
    R = torch.zeros((V, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - r * z); R[:, 0, 2] = 2 * (x * z + r * y)
    R[:, 1, 0] = 2 * (x * y + r * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - r * x)
    R[:, 2, 0] = 2 * (x * z - r * y); R[:, 2, 1] = 2 * (y * z + r * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    R[:, :, 0] *= -1
    R[:, :, 1] *= -1
    c2w = torch.zeros((V, 4, 4))
    c2w[:, :3, :3] = R
    c2w[:, 0, 3] = pos[:, 0]
    c2w[:, 1, 3] = cam_height
    c2w[:, 2, 3] = pos[:, 1]
    c2w[:, 3, 3] = 1.0
    return c2w.float()


def invert_rigid(c2w: torch.Tensor) -> torch.Tensor:
    """fp64 inverse rounded to fp32 (the reference calls torch.linalg.inv on the device)."""
    return torch.linalg.inv(c2w.double()).float()


def intrinsics(W: int, H: int):
    """fx = fy = W/2, cx = cy = W/2 (configs/base_config.py:179-190 at 256x256) -> tanfov = 1."""
    return np.array([[W / 2.0, 0.0, W / 2.0], [0.0, H / 2.0, H / 2.0], [0.0, 0.0, 1.0]], dtype=np.float64)
