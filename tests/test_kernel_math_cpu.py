"""csrc/fr_math.h (the arithmetic every HIP kernel calls) compiled with g++ and compared with the oracle on the
CPU: the forward half must be BIT-EXACT, the Jacobian half within rounding.  Catches transcription errors
without spending GPU time; the kernels' control flow is covered by the -m gpu tests."""
import ctypes

import numpy as np
import pytest

from scenes import random_scene, intrinsics, rel_err

c_f = ctypes.POINTER(ctypes.c_float)
c_i = ctypes.POINTER(ctypes.c_int32)
c_u = ctypes.POINTER(ctypes.c_uint32)
c_b = ctypes.POINTER(ctypes.c_uint8)


def P_(a, t=c_f):
    return a.ctypes.data_as(t)


def test_expf_bit_exact(oracle, harness):
    x = np.concatenate([np.linspace(-110, 90, 20001), -np.logspace(-8, 1.5, 3000), [0.0, -0.0]]).astype(np.float32)
    y = np.zeros_like(x)
    harness.h_expf(ctypes.c_int(x.size), P_(x), P_(y))
    assert np.array_equal(y.view(np.uint32), oracle.expf(x).view(np.uint32))


@pytest.mark.parametrize("W,H,seed", [(256, 256, 0), (200, 120, 1), (64, 48, 2)])
def test_preprocess_bit_exact(oracle, harness, W, H, seed):
    P = 5000
    sc = random_scene(P, seed, zmin=-0.5, zmax=8.0, spread=1.6, scale=0.05)
    # a few splats hugging the camera (appendix B.1) and far outside the frustum (B.2)
    sc["means3D"][:50, 2] = np.random.default_rng(seed).uniform(0.0005, 0.2, 50)
    sc["means3D"][50:100, 0] *= 4
    w2c = np.eye(4, dtype=np.float32)
    w2c[:3, :3] = np.array([[0.96, 0, 0.28], [0, 1, 0], [-0.28, 0, 0.96]], np.float32)
    w2c[:3, 3] = [0.3, -0.1, 0.2]
    cam = oracle.setup_camera(W, H, intrinsics(W, H), w2c)
    fwd = oracle.rasterize_forward(cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors"],
                                   scales=sc["scales"], rotations=sc["rotations"])
    cov3D = np.zeros((P, 6), np.float32)
    harness.h_cov3d(ctypes.c_int(P), P_(sc["scales"]), ctypes.c_float(1.0), P_(sc["rotations"]), P_(cov3D))
    vis = fwd["radii"] > 0
    assert np.array_equal(cov3D[vis], fwd["cov3D"][vis])
    radii = np.zeros(P, np.int32); depths = np.zeros(P, np.float32); m2d = np.zeros((P, 2), np.float32)
    conic = np.zeros((P, 3), np.float32); tiles = np.zeros(P, np.uint32); rect = np.zeros((P, 4), np.uint32)
    view = np.ascontiguousarray(cam.viewmatrix, np.float32); proj = np.ascontiguousarray(cam.projmatrix, np.float32)
    harness.h_preprocess(ctypes.c_int(P), P_(sc["means3D"]), P_(cov3D), P_(view), P_(proj), ctypes.c_int(W), ctypes.c_int(H),
                         ctypes.c_float(cam.tanfovx), ctypes.c_float(cam.tanfovy),
                         P_(radii, c_i), P_(depths), P_(m2d), P_(conic), P_(tiles, c_u), P_(rect, c_u))
    assert vis.sum() > 500
    assert np.array_equal(radii, fwd["radii"])
    assert np.array_equal(tiles, fwd["tiles_touched"])
    assert np.array_equal(depths.view(np.uint32), fwd["depths"].view(np.uint32))
    assert np.array_equal(m2d.view(np.uint32), fwd["means2D"].view(np.uint32))
    assert np.array_equal(conic.view(np.uint32), fwd["conic_opacity"][:, :3].copy().view(np.uint32))


def test_world_to_cam_bit_exact(oracle, harness):
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(1000, 3)).astype(np.float32) * 3
    w2c = np.eye(4, dtype=np.float32)
    w2c[:3, :4] = rng.normal(size=(3, 4)).astype(np.float32)
    out = np.zeros_like(pts)
    harness.h_world_to_cam(ctypes.c_int(1000), P_(np.ascontiguousarray(w2c.reshape(-1))), P_(pts), P_(out))
    assert np.array_equal(out.view(np.uint32), oracle.transform_points(w2c, pts).view(np.uint32))


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_forward_bit_exact(oracle, harness, deg):
    W = H = 64
    P = 800
    sc = random_scene(P, 7 + deg)
    rng = np.random.default_rng(deg)
    M = 16
    shs = rng.normal(scale=0.4, size=(P, M, 3)).astype(np.float32)
    cam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4))
    cam = cam._replace(sh_degree=deg, campos=np.array([0.1, -0.2, 0.05], np.float32))
    fwd = oracle.rasterize_forward(cam, sc["means3D"], sc["opacities"], shs=shs, scales=sc["scales"], rotations=sc["rotations"])
    rgb = np.zeros((P, 3), np.float32); cl = np.zeros((P, 3), np.uint8)
    harness.h_sh_to_rgb(ctypes.c_int(P), ctypes.c_int(deg), ctypes.c_int(M), P_(sc["means3D"]), P_(cam.campos), P_(shs), P_(rgb), P_(cl, c_b))
    vis = fwd["radii"] > 0
    assert vis.sum() > 100
    assert np.array_equal(rgb[vis].view(np.uint32), fwd["rgb"][vis].view(np.uint32))
    assert np.array_equal(cl[vis], fwd["clamped"][vis])


def test_leaf_jacobians_match_oracle_chain(oracle, harness):
    """One Gaussian, one-hot pixel, power 1: the oracle's per-pair chain (backward.cu:276-475,532-583) gives
    u = (dL_dmean2D, dL_dconic) and the leaves; the kernels' Jacobian matrices applied to that u must agree."""
    W, H = 96, 64
    rng = np.random.default_rng(11)
    w2c = np.eye(4, dtype=np.float32)
    w2c[:3, :3] = np.array([[0.8, 0, 0.6], [0, 1, 0], [-0.6, 0, 0.8]], np.float32)
    w2c[:3, 3] = [0.1, 0.05, 0.3]
    cam = oracle.setup_camera(W, H, intrinsics(W, H), w2c)
    view = np.ascontiguousarray(cam.viewmatrix, np.float32); proj = np.ascontiguousarray(cam.projmatrix, np.float32)
    checked = 0
    clamp_cases = 0
    errs = []
    for trial in range(300):
        sc = random_scene(1, 100 + trial, zmin=0.4, zmax=5.0, spread=1.5, scale=0.15)
        mod = 1.0
        fwd = oracle.rasterize_forward(cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors"],
                                       scales=sc["scales"], rotations=sc["rotations"])
        if fwd["radii"][0] == 0:
            continue
        ys, xs = np.nonzero(fwd["n_contrib"] > 0)
        if len(ys) == 0:
            continue
        k = rng.integers(len(ys))
        dL = np.zeros((3, H, W), np.float32)
        dL[:, ys[k], xs[k]] = rng.normal(size=3)
        g = oracle.rasterize_backward(cam, fwd, dL, 1)
        assert g["pair_count"] >= 1
        u = np.array([g["dL_dmeans2D"][0, 0], g["dL_dmeans2D"][0, 1], g["dL_dconic"][0, 0, 0], g["dL_dconic"][0, 0, 1],
                      g["dL_dconic"][0, 1, 1]], np.float32)
        out = np.zeros(16, np.float32)
        harness.h_leaf_from_u(P_(sc["means3D"]), P_(fwd["cov3D"]), P_(sc["scales"]), ctypes.c_float(mod), P_(sc["rotations"]),
                              P_(view), P_(proj), ctypes.c_int(W), ctypes.c_int(H), ctypes.c_float(cam.tanfovx),
                              ctypes.c_float(cam.tanfovy), P_(u), P_(out))
        ref = np.concatenate([g["dL_dmeans3D"][0], g["dL_dcov3D"][0], g["dL_dscales"][0], g["dL_drotations"][0]])
        scale = np.abs(ref).max()
        if scale == 0:
            continue
        # fp32 cancellation inside the chain: the two evaluation orders agree to ~1e-5..1e-4 of the largest leaf
        assert np.abs(out - ref).max() <= 2e-4 * scale, (trial, out, ref)
        errs.append(np.abs(out - ref).max() / scale)
        checked += 1
        pv = oracle.transform_points(w2c, sc["means3D"])[0]
        if abs(pv[0] / pv[2]) > 1.3 * cam.tanfovx or abs(pv[1] / pv[2]) > 1.3 * cam.tanfovy:
            clamp_cases += 1
    assert checked > 100
    assert np.median(errs) < 2e-6
