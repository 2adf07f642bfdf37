"""CPU: the radius bound behind the early frustum test of k_preprocess_views (csrc/fisher_rast.hip, phase A), restated in
NumPy and held against the oracle's own radii: whenever the bound puts a splat outside the tile grid, the oracle (the
restatement of forward.cu:155-256) must have given it radius 0 -- on needle-shaped splats with non-unit quaternions placed
around the image edges, and on the benchmark scene family."""
import numpy as np
import pytest

from scenes import intrinsics


def _bound_says_outside(means_cam, cov3D, K, W, H):
    """The kernel's arithmetic (float32): lambda1 <= kc / z^2 * trace(cov3D) + 0.7, radius <= 3 sqrt(.) + 2."""
    f32 = np.float32
    K = np.asarray(K, np.float64)
    fx, fy = f32(K[0, 0]), f32(K[1, 1])
    tanx, tany = f32(W / (2.0 * K[0, 0])), f32(H / (2.0 * K[1, 1]))
    x, y, z = (means_cam[:, k].astype(f32) for k in range(3))
    tr = (cov3D[:, 0] + cov3D[:, 3] + cov3D[:, 5]).astype(f32)
    with np.errstate(all="ignore"):
        iz = f32(1.0) / z
        # the scorer's camera: viewmatrix = I, projection of setup_camera (recon_helpers.py:4-32): ndc = (2 fx x / (W z) ..., pixel = ((ndc + 1) W - 1) / 2
        ndcx, ndcy = f32(2.0) * fx * x * iz / f32(W), f32(2.0) * fy * y * iz / f32(H)
        px, py = ((ndcx + f32(1.0)) * f32(W) - f32(1.0)) * f32(0.5), ((ndcy + f32(1.0)) * f32(H) - f32(1.0)) * f32(0.5)
        jx = np.minimum(np.abs(x * iz) * f32(1.001), f32(1.3) * tanx)
        jy = np.minimum(np.abs(y * iz) * f32(1.001), f32(1.3) * tany)
        kc = f32(1.02) * (fx * fx * (f32(1.0) + jx * jx) + fy * fy * (f32(1.0) + jy * jy))
        rb = f32(3.0) * np.sqrt(kc * tr * iz * iz + f32(0.7)) + f32(2.0)
        gx, gy = (W + 15) // 16, (H + 15) // 16
        out = (px + rb < 0) | (px - rb > gx * 16 + 16) | (py + rb < 0) | (py - rb > gy * 16 + 16)
    return out & (z > 0.001)


@pytest.mark.parametrize("which", ["border", "room"])
def test_bound_never_drops_a_visible_splat(oracle, which):
    if which == "border":
        from test_gpu_scorer_adversarial import border_scene
        W, H, sc, _ = border_scene()
        K = intrinsics(W, H)
        means, scales, rots, opac, col = sc["means3D"], sc["scales"], sc["rotations"], sc["opacities"], sc["colors"]
    else:
        from fisher_rast import synthetic
        W, H = 160, 112
        K = np.asarray(synthetic.intrinsics(W, H), np.float64)
        act = {k: v.numpy() for k, v in synthetic.activate(synthetic.room_shell(20000, seed=5)).items()}
        w2c = synthetic.invert_rigid(synthetic.candidate_poses(1, seed=5))[0].numpy()
        means = oracle.transform_points(w2c, act["means3D"])
        scales, rots, opac, col = act["scales"], act["rotations"], act["opacities"], act["rgb_colors"]
    cam = oracle.setup_camera(W, H, K, np.eye(4))
    fwd = oracle.rasterize_forward(cam, means, opac, colors_precomp=col, scales=scales, rotations=rots)
    outside = _bound_says_outside(np.asarray(means, np.float32), fwd["cov3D"], K, W, H)
    visible = fwd["radii"] > 0
    assert not (outside & visible).any(), int((outside & visible).sum())
    # ... and on an ordinary scene it is worth having: it removes most of the splats that are in front of the camera but not
    # visible (`border` is built to sit where it cannot)
    front = np.asarray(means)[:, 2] > 0.001
    if which == "room":
        assert outside.sum() > 0.5 * (front & ~visible).sum(), (int(outside.sum()), int((front & ~visible).sum()))
