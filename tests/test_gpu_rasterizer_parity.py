"""-m gpu: the HIP rasteriser (through the C ABI) against the oracle on identical inputs.
Bar (north star): tile / sort indices bit-exact; RGB and gradients within 1e-4 relative.  The forward arithmetic
is written to round exactly like the oracle, so the forward outputs are additionally required to be BIT-EXACT."""
import numpy as np
import pytest
import torch

from gpu_util import hip_forward, hip_backward, assert_close, to_dev
from scenes import random_scene, intrinsics

pytestmark = pytest.mark.gpu


def _pose(yaw=0.3, t=(0.2, -0.1, 0.3)):
    w2c = np.eye(4, dtype=np.float32)
    c, s = np.cos(yaw), np.sin(yaw)
    w2c[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
    w2c[:3, 3] = t
    return w2c


def _scene(case):
    if case == "general":
        W, H, P = 256, 256, 20000
        sc = random_scene(P, 0, zmin=-0.5, zmax=8.0, spread=1.5, scale=0.04)
        sc["means3D"][:40, 2] = np.random.default_rng(0).uniform(0.0011, 0.2, 40)   # appendix B.1: huge near splats
        w2c = _pose()
    elif case == "ragged":          # image not a multiple of 16 (B.5), non-square
        W, H, P = 200, 120, 6000
        sc = random_scene(P, 1, scale=0.06)
        w2c = _pose(-0.2)
    elif case == "crowded_tile":    # one tile holds > 4096 splats: multi-round LDS batches + the global-memory sort path
        W, H, P = 64, 64, 9000
        rng = np.random.default_rng(2)
        sc = random_scene(P, 2, scale=0.01)
        z = rng.uniform(1.0, 6.0, P).astype(np.float32)
        sc["means3D"] = np.stack([rng.uniform(-0.02, 0.02, P) * z, rng.uniform(-0.02, 0.02, P) * z, z], 1).astype(np.float32)
        sc["opacities"] = rng.uniform(0.002, 0.05, P).astype(np.float32)
        w2c = np.eye(4, dtype=np.float32)
    elif case == "ties":            # duplicated Gaussians: equal (tile, depth) keys keep index order (B.6)
        W, H, P = 96, 96, 3000
        sc = random_scene(1000, 3, scale=0.08)
        sc = {k: np.concatenate([v, v, v]) for k, v in sc.items()}
        w2c = np.eye(4, dtype=np.float32)
    elif case == "opaque":          # alpha saturating at 0.99 and early termination (B.3, B.4)
        W, H, P = 128, 128, 8000
        sc = random_scene(P, 4, scale=0.15, opacity_mean=6.0)
        w2c = np.eye(4, dtype=np.float32)
    return W, H, sc, w2c


CASES = ["general", "ragged", "crowded_tile", "ties", "opaque"]


@pytest.fixture(scope="module")
def runs(gpu, oracle):
    cache = {}

    def get(case):
        if case not in cache:
            W, H, sc, w2c = _scene(case)
            cam = oracle.setup_camera(W, H, intrinsics(W, H), w2c)
            args = dict(colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
            want = oracle.rasterize_forward(cam, sc["means3D"], sc["opacities"], **args)
            got = hip_forward(gpu, cam, sc["means3D"], sc["opacities"], **args)
            cache[case] = (cam, sc, want, got)
        return cache[case]
    return get


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("case", CASES)
def test_forward_bit_exact(runs, case):
    cam, sc, want, got = runs(case)
    vis = want["radii"] > 0
    assert vis.sum() > 0
    assert np.array_equal(got["radii"], want["radii"])
    assert np.array_equal(bits(got["depths"][vis]), bits(want["depths"][vis]))
    assert np.array_equal(bits(got["means2D"][vis]), bits(want["means2D"][vis]))
    assert np.array_equal(bits(got["conic_opacity"][vis]), bits(want["conic_opacity"][vis]))
    assert np.array_equal(bits(got["cov3D"][vis]), bits(want["cov3D"][vis]))
    # binning: per-tile ranges and the depth-sorted id list, bit for bit
    assert got["num_rendered"] == want["num_rendered"]
    assert np.array_equal(got["ranges"], want["ranges"])
    assert np.array_equal(got["point_list"], want["point_list"])
    if case == "crowded_tile":
        assert got["tile_count"].max() > 4096
    # render
    assert np.array_equal(got["n_contrib"], want["n_contrib"])
    assert np.array_equal(bits(got["final_T"]), bits(want["final_T"]))
    assert np.array_equal(bits(got["color"]), bits(want["color"]))
    assert np.array_equal(bits(got["depth"]), bits(want["depth"]))


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("power", [1, 2])
def test_backward_parity(runs, oracle, gpu, case, power):
    cam, sc, want, got = runs(case)
    H, W = cam.image_height, cam.image_width
    rng = np.random.default_rng(5)
    dL = rng.normal(size=(3, H, W)).astype(np.float32) if power == 1 else np.full((3, H, W), 1e-3, np.float32)
    gw = oracle.rasterize_backward(cam, want, dL, power)
    gg = hip_backward(gpu, cam, got, dL, power)
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations")
    if power == 2:
        # the arbiter (the oracle's statements in binary64 on the same contributor sets): a Gaussian whose reference chain is
        # itself off by r > 0 in binary32 (near-plane giants of `general`: up to 1.2e-4) gets (1e-4 + min(1.25 r, 0.3)) instead of 1e-4
        # (tools/arbiter_diag.py measured a need of 0.95 r at most)
        w64 = oracle.rasterize_forward(cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors"], scales=sc["scales"],
                                       rotations=sc["rotations"], decisions=want)
        ga = oracle.rasterize_backward(cam, w64, dL, power)
    for n in names:
        # power 1 sums signed terms (cancellation): compare against the tensor's scale; power 2 sums squares: 1e-4 relative
        if power == 2:
            o, a = gw[n].astype(np.float64).reshape(gw[n].shape[0], -1), ga[n].reshape(gw[n].shape[0], -1)
            big = np.abs(a) > 1e-7 * np.abs(a).max()
            r = np.where(big, np.abs(o - a) / np.maximum(np.abs(a), 1e-300), 0.0).max(axis=1, keepdims=True)
            tol = (1e-4 + np.minimum(1.25 * r, 0.3)) * np.abs(o) + 1e-7 * np.abs(o).max()
            bad = np.abs(gg[n].astype(np.float64).reshape(o.shape) - o) > tol
            assert not bad.any(), (f"{case}/{n}", int(bad.sum()), float((np.abs(gg[n].reshape(o.shape) - o) / np.maximum(tol, 1e-300)).max()))
            assert (r > 2e-5).mean() < 0.02
            print(f"[{case}/{n}] widened Gaussians: {int((r > 2e-5).sum())} of {r.shape[0]}, largest r {float(r.max()):.2e}")
        else:
            assert_close(gg[n], gw[n], 1e-4, f"{case}/{n}", atol_frac=2e-5)
    assert gg["dL_dsh"].shape == (sc["means3D"].shape[0], 0, 3)


def test_cov3d_precomp_path(gpu, oracle):
    W, H, sc, w2c = _scene("ragged")
    cam = oracle.setup_camera(W, H, intrinsics(W, H), w2c)
    base = oracle.rasterize_forward(cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
    cov = base["cov3D"].copy()
    for i in np.nonzero(base["radii"] == 0)[0]:   # culled splats never had their covariance computed
        cov[i] = [1e-3, 0, 0, 1e-3, 0, 1e-3]
    want = oracle.rasterize_forward(cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors"], cov3D_precomp=cov)
    got = hip_forward(gpu, cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors"], cov3D_precomp=cov)
    assert np.array_equal(got["radii"], want["radii"]) and np.array_equal(got["point_list"], want["point_list"])
    assert np.array_equal(bits(got["color"]), bits(want["color"]))
    dL = np.random.default_rng(1).normal(size=(3, H, W)).astype(np.float32)
    gw = oracle.rasterize_backward(cam, want, dL, 1)
    gg = hip_backward(gpu, cam, got, dL, 1)
    for n in ("dL_dmeans3D", "dL_dcov3D", "dL_dopacity", "dL_dcolors"):
        assert_close(gg[n], gw[n], 1e-4, n, atol_frac=2e-5)
    assert np.all(gg["dL_dscales"] == 0) and np.all(gg["dL_drotations"] == 0)


@pytest.mark.parametrize("deg", [0, 3])
def test_sh_forward(gpu, oracle, deg):
    W = H = 96
    P = 3000
    sc = random_scene(P, 9)
    shs = np.random.default_rng(deg).normal(scale=0.4, size=(P, 16, 3)).astype(np.float32)
    cam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4))._replace(sh_degree=deg, campos=np.array([0.1, -0.2, 0.05], np.float32))
    want = oracle.rasterize_forward(cam, sc["means3D"], sc["opacities"], shs=shs, scales=sc["scales"], rotations=sc["rotations"])
    got = hip_forward(gpu, cam, sc["means3D"], sc["opacities"], shs=shs, scales=sc["scales"], rotations=sc["rotations"])
    vis = want["radii"] > 0
    assert np.array_equal(bits(got["rgb"][vis]), bits(want["rgb"][vis]))
    assert np.array_equal(got["clamped"][vis], want["clamped"][vis])
    assert np.array_equal(bits(got["color"]), bits(want["color"]))


def test_mark_visible_and_empty(gpu, oracle):
    from fisher_rast import ops
    W = H = 64
    sc = random_scene(5000, 12, zmin=-2.0, zmax=3.0)
    cam = oracle.setup_camera(W, H, intrinsics(W, H), _pose())
    want = oracle.mark_visible(cam, sc["means3D"])
    got = ops.mark_visible(to_dev(sc["means3D"], gpu), to_dev(cam.viewmatrix, gpu), to_dev(cam.projmatrix, gpu))
    assert got.dtype == torch.bool and np.array_equal(got.cpu().numpy(), want)
    e = hip_forward(gpu, cam, np.zeros((0, 3), np.float32), np.zeros((0,), np.float32), colors_precomp=np.zeros((0, 3), np.float32),
                    scales=np.zeros((0, 3), np.float32), rotations=np.zeros((0, 4), np.float32))
    assert e["num_rendered"] == 0 and np.all(e["color"] == 0) and np.all(e["depth"] == 0) and e["radii"].shape == (0,)
    # every splat culled: background image, default depth
    behind = sc["means3D"].copy()
    behind[:, 2] = -np.abs(behind[:, 2]) - 1
    cam0 = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4))
    o = hip_forward(gpu, cam0, behind, sc["opacities"], colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
    assert o["num_rendered"] == 0 and np.all(o["radii"] == 0) and np.all(o["color"] == 0) and np.all(o["depth"] == 15.0)
    assert np.all(o["final_T"] == 1.0) and np.all(o["ranges"] == 0)


def test_argument_errors(gpu):
    from fisher_rast import ops, FisherRastError
    m = torch.zeros((4, 3), device=gpu)
    one = torch.ones((4, 3), device=gpu)
    eye = torch.eye(4, device=gpu)
    e = torch.Tensor([])
    with pytest.raises(RuntimeError):
        ops.rasterize_forward(torch.zeros(3, device=gpu), torch.zeros((4, 2), device=gpu), one, one[:, :1], one, torch.ones((4, 4), device=gpu),
                              1.0, e, eye, eye, 1.0, 1.0, 32, 32, e, 0, torch.zeros(3, device=gpu), False)
    with pytest.raises(FisherRastError):   # neither colours nor SHs
        ops.rasterize_forward(torch.zeros(3, device=gpu), m, e, one[:, :1], one, torch.ones((4, 4), device=gpu),
                              1.0, e, eye, eye, 1.0, 1.0, 32, 32, e, 0, torch.zeros(3, device=gpu), False)
    with pytest.raises(FisherRastError):   # CPU tensors: no CPU path
        ops.mark_visible(torch.zeros((4, 3)), torch.eye(4), torch.eye(4))


@pytest.mark.parametrize("deg", [0, 1, 3])
@pytest.mark.parametrize("power", [1, 2])
def test_sh_backward(gpu, oracle, deg, power):
    """SH colours: dL_dsh, and the extra mean gradient through the view direction (backward.cu:20-139), including the
    reference's quirks (no dL_dsh when the degree is 0, backward.cu:1117; float-offset SH pointer, 1067)."""
    W, H = 96, 64
    P = 2500
    sc = random_scene(P, 31 + deg, scale=0.07)
    M = 16
    shs = np.random.default_rng(deg).normal(scale=2.0, size=(P, M, 3)).astype(np.float32)   # large enough to clamp some colours
    cam = oracle.setup_camera(W, H, intrinsics(W, H), _pose(0.1))._replace(sh_degree=deg, campos=np.array([0.3, -0.2, -0.4], np.float32))
    want = oracle.rasterize_forward(cam, sc["means3D"], sc["opacities"], shs=shs, scales=sc["scales"], rotations=sc["rotations"])
    got = hip_forward(gpu, cam, sc["means3D"], sc["opacities"], shs=shs, scales=sc["scales"], rotations=sc["rotations"])
    assert want["clamped"].any()
    dL = np.random.default_rng(7).normal(size=(3, H, W)).astype(np.float32) if power == 1 else np.full((3, H, W), 1e-3, np.float32)
    gw = oracle.rasterize_backward(cam, want, dL, power)
    gg = hip_backward(gpu, cam, got, dL, power)
    fl = 2e-5 if power == 1 else 1e-7
    for n in ("dL_dmeans3D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans2D", "dL_dsh"):
        assert_close(gg[n], gw[n], 1e-4, f"deg{deg}/p{power}/{n}", atol_frac=fl)
    assert gg["dL_dsh"].shape == (P, M, 3)
    if deg == 0:
        assert not gg["dL_dsh"].any()
    else:
        assert np.abs(gg["dL_dsh"][:, : (deg + 1) ** 2]).max() > 0 and not gg["dL_dsh"][:, (deg + 1) ** 2:].any()


def test_autograd_front_end_with_shs(gpu):
    from diff_gaussian_rasterization import GaussianRasterizer, GaussianRasterizationSettings
    P = 800
    sc = random_scene(P, 40)
    t = {k: torch.tensor(v, device=gpu, requires_grad=True) for k, v in sc.items() if k != "colors"}
    shs = torch.randn((P, 4, 3), device=gpu, requires_grad=True)
    eye = torch.eye(4, device=gpu)
    proj = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1.0001, 1], [0, 0, -0.01, 0]], dtype=torch.float32, device=gpu)
    rs = GaussianRasterizationSettings(64, 64, 1.0, 1.0, torch.zeros(3, device=gpu), 1.0, eye, proj, 1, torch.zeros(3, device=gpu), False)
    m2d = torch.zeros((P, 3), device=gpu, requires_grad=True)
    im, radii, depth = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"].reshape(-1, 1), shs=shs,
                                              scales=t["scales"], rotations=t["rotations"])
    assert im.shape == (3, 64, 64) and radii.dtype == torch.int32 and depth.shape == (1, 64, 64)
    im.sum().backward()
    assert shs.grad.shape == (P, 4, 3) and float(shs.grad.abs().sum()) > 0
    assert t["means3D"].grad.shape == (P, 3) and m2d.grad.shape == (P, 3)
    assert torch.isfinite(t["scales"].grad).all() and torch.isfinite(t["rotations"].grad).all()


def test_backward_through_depth_only_gives_zero_gradients(gpu):
    """The reference's backward ignores the gradients of `radii` and `depth` (__init__.py:100-136 there); a loss on the depth image
    alone therefore reaches the Gaussians as zeros, not as an error (autograd is told not to fill in the unused output gradients)."""
    from diff_gaussian_rasterization import GaussianRasterizer, GaussianRasterizationSettings
    P = 500
    sc = random_scene(P, 43)
    t = {k: torch.tensor(v, device=gpu, requires_grad=True) for k, v in sc.items()}
    eye = torch.eye(4, device=gpu)
    proj = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1.0001, 1], [0, 0, -0.01, 0]], dtype=torch.float32, device=gpu)
    rs = GaussianRasterizationSettings(64, 64, 1.0, 1.0, torch.zeros(3, device=gpu), 1.0, eye, proj, 0, torch.zeros(3, device=gpu), False)
    for power in (1, 2):
        for v in t.values():
            v.grad = None
        im, radii, depth = GaussianRasterizer(rs, backward_power=power)(
            means3D=t["means3D"], means2D=torch.zeros((P, 3), device=gpu, requires_grad=True), opacities=t["opacities"].reshape(-1, 1),
            colors_precomp=t["colors"], scales=t["scales"], rotations=t["rotations"])
        depth.sum().backward()
        assert t["means3D"].grad is not None and not t["means3D"].grad.any() and not t["colors"].grad.any()


def test_prefiltered_with_a_culled_point_is_an_error(gpu):
    """auxiliary.h:156-160: with `prefiltered` set, a point behind the near plane makes the reference print "Point is filtered
    although prefiltered is set" and trap the device.  Here the forward raises with that message and the device stays usable;
    a scene with every point in front of the camera renders as without the flag."""
    from diff_gaussian_rasterization import GaussianRasterizer, GaussianRasterizationSettings
    P = 300
    sc = random_scene(P, 41)
    eye = torch.eye(4, device=gpu)
    proj = torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1.0001, 1], [0, 0, -0.01, 0]], dtype=torch.float32, device=gpu)

    def render(means, prefiltered):
        rs = GaussianRasterizationSettings(64, 64, 1.0, 1.0, torch.zeros(3, device=gpu), 1.0, eye, proj, 0, torch.zeros(3, device=gpu), prefiltered)
        return GaussianRasterizer(rs)(means3D=means, means2D=torch.zeros((P, 3), device=gpu), opacities=torch.tensor(sc["opacities"], device=gpu).reshape(-1, 1),
                                      colors_precomp=torch.tensor(sc["colors"], device=gpu), scales=torch.tensor(sc["scales"], device=gpu),
                                      rotations=torch.tensor(sc["rotations"], device=gpu))
    means = torch.tensor(sc["means3D"], device=gpu)
    assert float(means[:, 2].min()) > 0.001
    a, b = render(means, False), render(means, True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    behind = means.clone()
    behind[7, 2] = -1.0
    with pytest.raises(RuntimeError, match="filtered although prefiltered is set"):
        render(behind, True)
    c = render(behind, False)                                  # without the flag the point is simply culled
    assert int(c[1][7]) == 0 and torch.isfinite(c[0]).all()


def test_fused_rgb_depth_silhouette_pair(gpu, oracle):
    """forward_pair / fr_forward_features / fr_backward_pair: the double render of the reference's get_loss
    (models/SLAM/gaussian.py:199-211) on one projection, binning and sort.  Forward images bit-identical to two separate
    calls; every leaf gradient equal to the sum of the two separate backwards; `means2D.grad` = the colour render's only."""
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from fisher_rast import synthetic
    from models.SLAM.utils.recon_helpers import setup_camera
    from models.SLAM.utils.slam_helpers import (transformed_params2rendervar, transformed_params2depthplussilhouette,
                                                 render_rgb_depth_sil)
    P, W, H = 6000, 160, 112
    raw = synthetic.room_shell(P, 21)
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(1, 22))[0].to(gpu)
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=gpu)
    g = torch.Generator().manual_seed(23)
    w_im, w_ds = torch.randn((3, H, W), generator=g).to(gpu), torch.randn((3, H, W), generator=g).to(gpu)
    names = ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales")

    def leaves():
        return {k: raw[k].clone().to(gpu).requires_grad_(True) for k in names}

    def frame_pts(params):
        pts = params["means3D"]
        return (w2c @ torch.cat((pts, torch.ones_like(pts[:, :1])), 1).T).T[:, :3]

    # reference flow: two renders
    pa = leaves()
    tp = frame_pts(pa)
    rv = transformed_params2rendervar(pa, tp)
    dv = transformed_params2depthplussilhouette(pa, w2c, tp)
    rv["means2D"].retain_grad()
    im_a, rad_a, _ = Renderer(raster_settings=cam)(**rv)
    ds_a, _, _ = Renderer(raster_settings=cam)(**dv)
    ((im_a * w_im).sum() + (ds_a * w_ds).sum()).backward()
    # fused flow
    pb = leaves()
    im_b, rad_b, ds_b, rvb = render_rgb_depth_sil(pb, cam, w2c, frame_pts(pb))
    ((im_b * w_im).sum() + (ds_b * w_ds).sum()).backward()
    torch.cuda.synchronize()
    assert torch.equal(im_a, im_b) and torch.equal(ds_a, ds_b) and torch.equal(rad_a, rad_b)
    assert int((rad_a > 0).sum()) > 500
    for k in names:
        assert_close(pb[k].grad.cpu().numpy(), pa[k].grad.cpu().numpy(), 1e-4, f"fused pair d/d{k}", atol_frac=1e-6)
    assert_close(rvb["means2D"].grad.cpu().numpy(), rv["means2D"].grad.cpu().numpy(), 1e-5, "means2D.grad (colour render only)", atol_frac=1e-7)
    # and the depth / silhouette image against the oracle
    n = {k: v.detach().cpu().numpy() for k, v in dv.items() if k != "means2D"}
    ocam = oracle.setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4))
    fw = oracle.rasterize_forward(ocam, n["means3D"], n["opacities"], colors_precomp=n["colors_precomp"], scales=n["scales"], rotations=n["rotations"])
    assert np.array_equal(ds_b.detach().cpu().numpy(), fw["color"])
    with pytest.raises(Exception):
        Renderer(raster_settings=cam, backward_power=2).forward_pair(rv["means3D"], rv["means2D"], rv["opacities"], rv["colors_precomp"],
                                                                    dv["colors_precomp"], scales=rv["scales"], rotations=rv["rotations"])


def test_fused_pair_on_a_crowded_tile(gpu):
    """fr_backward_pair where one tile holds ~9000 splats: the per-strip lists overflow the LDS index, the tile is flagged and
    redone by the scan kernel once per image -- same gradients as two separate backwards."""
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from models.SLAM.utils.recon_helpers import setup_camera
    W, H, sc, w2c = _scene("crowded_tile")
    cam = setup_camera(W, H, intrinsics(W, H), w2c, device=gpu)
    P = sc["means3D"].shape[0]
    g = torch.Generator().manual_seed(31)
    feats0 = torch.rand((P, 3), generator=g)
    w_a, w_b = torch.randn((3, H, W), generator=g).to(gpu), torch.randn((3, H, W), generator=g).to(gpu)

    def leaves():
        d = dict(means3D=to_dev(sc["means3D"], gpu), opacities=to_dev(sc["opacities"].reshape(-1, 1), gpu), scales=to_dev(sc["scales"], gpu),
                 rotations=to_dev(sc["rotations"], gpu), colors=to_dev(sc["colors"], gpu), feats=feats0.clone().to(gpu))
        return {k: v.requires_grad_(True) for k, v in d.items()}
    a = leaves()
    m2a = torch.zeros_like(a["means3D"], requires_grad=True)
    im_a, rad_a, _ = Renderer(raster_settings=cam)(a["means3D"], m2a, a["opacities"], colors_precomp=a["colors"], scales=a["scales"], rotations=a["rotations"])
    ft_a, _, _ = Renderer(raster_settings=cam)(a["means3D"], torch.zeros_like(a["means3D"], requires_grad=True), a["opacities"],
                                                colors_precomp=a["feats"], scales=a["scales"], rotations=a["rotations"])
    ((im_a * w_a).sum() + (ft_a * w_b).sum()).backward()
    b = leaves()
    m2b = torch.zeros_like(b["means3D"], requires_grad=True)
    im_b, rad_b, _, ft_b = Renderer(raster_settings=cam).forward_pair(b["means3D"], m2b, b["opacities"], b["colors"], b["feats"],
                                                                      scales=b["scales"], rotations=b["rotations"])
    ((im_b * w_a).sum() + (ft_b * w_b).sum()).backward()
    assert torch.equal(im_a, im_b) and torch.equal(ft_a, ft_b)
    for k in a:
        assert_close(b[k].grad.cpu().numpy(), a[k].grad.cpu().numpy(), 2e-4, f"crowded pair d/d{k}", atol_frac=1e-6)
    assert_close(m2b.grad.cpu().numpy(), m2a.grad.cpu().numpy(), 2e-4, "crowded pair means2D.grad", atol_frac=1e-6)
