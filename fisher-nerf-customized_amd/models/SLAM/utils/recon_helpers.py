"""Camera construction with the reference's signature (models/SLAM/utils/recon_helpers.py:4-32)."""
import numpy as np
import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings as Camera


def setup_camera(w, h, k, w2c, near=0.01, far=100, device="cuda"):
    """Returns GaussianRasterizationSettings whose matrices are stored transposed (column-major in memory):
    viewmatrix = w2c^T [1,4,4], projmatrix = (opengl_proj @ w2c)^T [1,4,4]; bg = 0, sh_degree = 0."""
    fx, fy, cx, cy = k[0][0], k[1][1], k[0][2], k[1][2]
    w2c = torch.as_tensor(np.asarray(w2c), dtype=torch.float32, device=device)
    cam_center = torch.inverse(w2c)[:3, 3]
    w2c_t = w2c.unsqueeze(0).transpose(1, 2)
    proj = torch.tensor([[2 * fx / w, 0.0, -(w - 2 * cx) / w, 0.0],
                         [0.0, 2 * fy / h, -(h - 2 * cy) / h, 0.0],
                         [0.0, 0.0, far / (far - near), -(far * near) / (far - near)],
                         [0.0, 0.0, 1.0, 0.0]], dtype=torch.float32, device=device).unsqueeze(0).transpose(1, 2)
    full_proj = w2c_t.bmm(proj)
    return Camera(image_height=h, image_width=w, tanfovx=w / (2 * fx), tanfovy=h / (2 * fy),
                  bg=torch.zeros(3, dtype=torch.float32, device=device), scale_modifier=1.0,
                  viewmatrix=w2c_t.contiguous(), projmatrix=full_proj.contiguous(), sh_degree=0,
                  campos=cam_center, prefiltered=False)
