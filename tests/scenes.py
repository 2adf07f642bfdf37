"""Shared scene builders for the tests (NumPy, seeded)."""
import numpy as np


def random_scene(P, seed, zmin=0.3, zmax=6.0, spread=1.2, scale=0.05, opacity_mean=1.0):
    """Gaussians in front of an identity camera, already in the camera frame."""
    rng = np.random.default_rng(seed)
    z = rng.uniform(zmin, zmax, P)
    means = np.stack([rng.uniform(-spread, spread, P) * z, rng.uniform(-spread, spread, P) * z, z], 1).astype(np.float32)
    scales = np.exp(rng.normal(np.log(scale), 0.4, (P, 3))).astype(np.float32)
    rot = rng.normal(size=(P, 4)).astype(np.float32)
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    op = (1 / (1 + np.exp(-rng.normal(opacity_mean, 1.5, P)))).astype(np.float32)
    col = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    return dict(means3D=means, scales=scales, rotations=rot, opacities=op, colors=col)


def intrinsics(W, H):
    return [[W / 2.0, 0, W / 2.0], [0, H / 2.0, H / 2.0], [0, 0, 1]]


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b).max() if a.size else 0.0
    s = np.abs(b).max() if b.size else 0.0
    return d / s if s > 0 else d
