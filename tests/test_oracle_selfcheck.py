"""The oracle has no reference golden vectors to lean on, so it is pinned by:
 (a) torch autograd of an INDEPENDENT dense differentiable splat renderer (power = 1),
 (b) the exact identity  Fisher(power=2, g) == sum_pixels grad(power=1, g * onehot_pixel)^2,
 (c) hand-computed single-Gaussian answers."""
import numpy as np
import pytest
import torch

from scenes import random_scene, intrinsics, rel_err


def test_expf_accuracy_and_monotone(oracle):
    x = np.concatenate([np.linspace(-30, 1, 4001), np.linspace(-100, 88, 500)]).astype(np.float32)
    e = oracle.expf(x).astype(np.float64)
    t = np.exp(x.astype(np.float64))
    ulp = np.abs(e - t) / np.spacing(t.astype(np.float32)).astype(np.float64)
    assert ulp[np.isfinite(ulp)].max() <= 1.0
    xs = np.sort(x[:4001])
    es = oracle.expf(xs)
    assert np.all(np.diff(es.astype(np.float64)) >= 0)
    assert oracle.expf(np.float32([0.0]))[0] == 1.0


def test_get_higher_msb(oracle):
    L = oracle.lib()
    # rasterizer_impl.cu:35-50 : 256 tiles -> 9, 1024 -> 11 (41 / 43 sorted bits as quoted in SURVEY 2a)
    assert L.orc_get_higher_msb(256) == 9
    assert L.orc_get_higher_msb(1024) == 11
    assert L.orc_get_higher_msb(1) == 1


def test_single_gaussian_known_answer(oracle):
    """Isotropic Gaussian on the optical axis: everything can be computed by hand."""
    W = H = 64
    cam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4))
    s, z, o = 0.1, 2.0, 0.8
    fwd = oracle.rasterize_forward(cam, [[0, 0, z]], [o], colors_precomp=[[1.0, 0.5, 0.25]],
                                   scales=[[s, s, s]], rotations=[[1, 0, 0, 0]])
    f = W / 2.0
    var = (f * s / z) ** 2 + 0.3
    assert fwd["radii"][0] == int(np.ceil(3 * np.sqrt(var)))
    # pixel centre convention: ndc 0 -> (W-1)/2
    assert np.allclose(fwd["means2D"][0], [(W - 1) / 2.0, (H - 1) / 2.0])
    assert np.allclose(fwd["conic_opacity"][0], [1 / var, 0, 1 / var, o], rtol=1e-5)
    assert fwd["depths"][0] == np.float32(z)
    # pixel (32,32) is at offset (+0.5,+0.5) from the mean
    d2 = 0.5
    alpha = o * np.exp(-0.5 * d2 / var)
    assert np.allclose(fwd["color"][:, 32, 32], np.array([1.0, 0.5, 0.25]) * alpha, rtol=1e-5)
    assert np.allclose(fwd["final_T"][32, 32], 1 - alpha, rtol=1e-5)
    assert fwd["n_contrib"][32, 32] == 1
    assert fwd["depth"][0, 32, 32] == np.float32(15.0) or alpha > 0.5  # median default unless T crosses 0.5
    # far corner: no contribution, background 0, depth default 15
    assert fwd["color"][:, 0, 0].max() == 0 and fwd["depth"][0, 0, 0] == np.float32(15.0)
    # one-hot upstream on that pixel, power 1: dL/dcolor = alpha * T(=1)
    dL = np.zeros((3, H, W), np.float32)
    dL[0, 32, 32] = 1.0
    g = oracle.rasterize_backward(cam, fwd, dL, 1)
    assert np.allclose(g["dL_dcolors"][0], [alpha, 0, 0], rtol=1e-5)
    # dL/dopacity = G * dL_dalpha = G * c_r
    assert np.allclose(g["dL_dopacity"][0, 0], np.exp(-0.5 * d2 / var) * 1.0, rtol=1e-5)
    assert g["pair_count"] == int((fwd["n_contrib"] > 0).sum())


def test_cull_and_empty(oracle):
    W = H = 32
    cam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4))
    # behind the camera / at the near plane: culled (auxiliary.h:154)
    fwd = oracle.rasterize_forward(cam, [[0, 0, -1.0], [0, 0, 0.001], [50.0, 0, 1.0]], [0.5, 0.5, 0.5],
                                   colors_precomp=np.ones((3, 3)), scales=np.full((3, 3), 0.01),
                                   rotations=[[1, 0, 0, 0]] * 3)
    assert fwd["radii"].tolist() == [0, 0, 0] and fwd["num_rendered"] == 0
    assert np.all(fwd["ranges"] == 0) and np.all(fwd["final_T"] == 1.0) and np.all(fwd["depth"] == 15.0)
    # P == 0 (rasterize_points.cu:81): zero image
    e = oracle.rasterize_forward(cam, np.zeros((0, 3)), np.zeros((0,)), colors_precomp=np.zeros((0, 3)),
                                 scales=np.zeros((0, 3)), rotations=np.zeros((0, 4)))
    assert e["num_rendered"] == 0 and np.all(e["color"] == 0) and np.all(e["depth"] == 0)
    with pytest.raises(Exception):
        oracle.rasterize_forward(cam, np.zeros((1, 3)), np.zeros(1))


def test_stable_tie_order(oracle):
    """Duplicate Gaussians (equal tile and depth bits) keep ascending index order."""
    W = H = 32
    cam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4))
    m = np.tile(np.array([[0.1, 0.1, 2.0]], np.float32), (5, 1))
    fwd = oracle.rasterize_forward(cam, m, np.full(5, 0.3), colors_precomp=np.random.rand(5, 3),
                                   scales=np.full((5, 3), 0.05), rotations=[[1, 0, 0, 0]] * 5)
    for t in range(fwd["ranges"].shape[0]):
        a, b = fwd["ranges"][t]
        assert fwd["point_list"][a:b].tolist() == sorted(fwd["point_list"][a:b].tolist())
    assert fwd["num_rendered"] == 5 * int(fwd["tiles_touched"][0])


# ------------------------------------------------------------------------------------------------------
def dense_render(means, scales, rot, op, col, m2d, cam, visible, dtype=torch.float64):
    """Independent differentiable splat renderer: every pixel against every Gaussian, sorted by depth,
    same thresholds as forward.cu:331-380 applied as (non-differentiable) masks."""
    W, H = cam.image_width, cam.image_height
    view = torch.tensor(cam.viewmatrix, dtype=dtype).reshape(4, 4).T   # back to row-major math matrix
    proj = torch.tensor(cam.projmatrix, dtype=dtype).reshape(4, 4).T
    P = means.shape[0]
    hom = torch.cat([means, torch.ones(P, 1, dtype=dtype)], 1)
    pv = (view @ hom.T).T[:, :3]
    ph = (proj @ hom.T).T
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :2] * pw[:, None] + m2d[:, :2]
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    r, x, y, z = rot[:, 0], rot[:, 1], rot[:, 2], rot[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
    S = torch.diag_embed(scales)
    Sigma = R @ S @ S @ R.transpose(1, 2)
    fx, fy = W / (2 * cam.tanfovx), H / (2 * cam.tanfovy)
    tz = pv[:, 2]
    tx = torch.clamp(pv[:, 0] / tz, -1.3 * cam.tanfovx, 1.3 * cam.tanfovx) * tz
    ty = torch.clamp(pv[:, 1] / tz, -1.3 * cam.tanfovy, 1.3 * cam.tanfovy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * tx / tz ** 2, zero, fy / tz, -fy * ty / tz ** 2], 1).reshape(P, 2, 3)
    A = J @ view[:3, :3]
    cov = A @ Sigma @ A.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + 0.3
    det = a * c - b * b
    cx, cy, cz = c / det, -b / det, a / det
    order = torch.argsort(tz.detach(), stable=True)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dtype), torch.arange(W, dtype=dtype), indexing="ij")
    T = torch.ones(H, W, dtype=dtype)
    C = torch.zeros(3, H, W, dtype=dtype)
    done = torch.zeros(H, W, dtype=torch.bool)
    for i in order.tolist():
        if not visible[i]:   # culled by the reference's preprocess (near plane / tile rectangle of the 3-sigma radius)
            continue
        dx = px[i] - xs
        dy = py[i] - ys
        power = -0.5 * (cx[i] * dx * dx + cz[i] * dy * dy) - cy[i] * dx * dy
        alpha = op[i] * torch.exp(power)
        alpha_c = torch.clamp(alpha, max=0.99)
        ok = (power.detach() <= 0) & (alpha_c.detach() >= 1.0 / 255.0) & (~done)
        test_T = T * (1 - alpha_c)
        newly_done = ok & (test_T.detach() < 1e-4)
        done = done | newly_done
        ok = ok & (~newly_done)
        w = torch.where(ok, alpha_c * T, torch.zeros_like(T))
        C = C + col[i][:, None, None] * w[None]
        T = torch.where(ok, test_T, T)
    return C, T


@pytest.mark.parametrize("seed", [0, 1])
def test_power1_matches_autograd_of_dense_renderer(oracle, seed):
    W = H = 16   # a single tile: the tile's list is exactly the set of visible Gaussians
    P = 40
    sc = random_scene(P, seed, zmin=1.0, zmax=4.0, spread=0.6, scale=0.08, opacity_mean=0.0)
    sc["opacities"] = np.clip(sc["opacities"], 0.05, 0.9)  # keep alpha below the 0.99 clamp (appendix B.3)
    cam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4))
    fwd = oracle.rasterize_forward(cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors"],
                                   scales=sc["scales"], rotations=sc["rotations"])
    rng = np.random.default_rng(seed + 10)
    dL = rng.normal(size=(3, H, W)).astype(np.float32)
    g = oracle.rasterize_backward(cam, fwd, dL, 1)

    t = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in sc.items()}
    m2d = torch.zeros(P, 3, dtype=torch.float64, requires_grad=True)
    C, T = dense_render(t["means3D"], t["scales"], t["rotations"], t["opacities"], t["colors"], m2d, cam, fwd["radii"] > 0)
    assert rel_err(fwd["color"], C.detach().numpy()) < 2e-5
    assert rel_err(fwd["final_T"], T.detach().numpy()) < 2e-5
    (C * torch.tensor(dL, dtype=torch.float64)).sum().backward()
    for name, ref in (("dL_dmeans3D", t["means3D"].grad), ("dL_dcolors", t["colors"].grad),
                      ("dL_dopacity", t["opacities"].grad.reshape(-1, 1)), ("dL_dscales", t["scales"].grad),
                      ("dL_drotations", t["rotations"].grad), ("dL_dmeans2D", m2d.grad)):
        assert rel_err(g[name], ref.numpy()) < 2e-4, name


def test_power2_identity(oracle):
    """Sum over pixels of squared one-hot-pixel gradients == the power-2 pass (backward.cu:1095-1137)."""
    W = H = 16
    P = 60
    sc = random_scene(P, 3, zmin=0.8, zmax=4.0, spread=0.7, scale=0.08)
    cam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4))
    fwd = oracle.rasterize_forward(cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors"],
                                   scales=sc["scales"], rotations=sc["rotations"])
    gs = 1e-3
    g2 = oracle.rasterize_backward(cam, fwd, np.full((3, H, W), gs, np.float32), 2)
    names = ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dcolors", "dL_dmeans2D", "dL_dcov3D")
    acc = {n: 0.0 for n in names}
    for y in range(H):
        for x in range(W):
            d = np.zeros((3, H, W), np.float32)
            d[:, y, x] = gs
            g1 = oracle.rasterize_backward(cam, fwd, d, 1)
            for n in names:
                acc[n] = acc[n] + g1[n].astype(np.float64) ** 2
    for n in names:
        assert rel_err(acc[n], g2[n]) < 1e-6, n


def test_compute_hessian_layout(oracle):
    W = H = 32
    sc = random_scene(200, 5)
    cam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4))
    w2c = np.eye(4, dtype=np.float32)
    w2c[:3, 3] = [0.1, -0.05, 0.2]
    H4, vis = oracle.compute_hessian(cam, w2c, sc["means3D"], sc["colors"], sc["rotations"], sc["opacities"], sc["scales"], 4)
    H11, vis2 = oracle.compute_hessian(cam, w2c, sc["means3D"], sc["colors"], sc["rotations"], sc["opacities"], sc["scales"], 11)
    assert H4.shape == (200, 4) and H11.shape == (200, 11) and vis == vis2
    assert np.array_equal(H4, H11[:, :4]) and np.all(H11 >= 0) and H4.sum() > 0
