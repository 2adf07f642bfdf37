"""-m gpu: the densification / pruning statistics kernels (fr_densify_stats / fr_densify_masks / fr_prune_mask, mirrored in
models/SLAM/utils/slam_external.py of the package) against oracle/densify_stats.py (slam_external.py:196-200, 345-465;
gaussian.py:289-291 of the reference): accumulators and masks bit for bit, including the values that sit on the thresholds,
0 / 0 gradients, isotropic (1-column) scales, and a whole mapping iteration driven by the drop-in rasteriser."""
import numpy as np
import pytest
import torch

from gpu_util import assert_close

pytestmark = pytest.mark.gpu


def _state(P, seed, cols=3):
    rng = np.random.default_rng(seed)
    radius = np.where(rng.uniform(size=P) < 0.6, rng.integers(1, 60, P), 0).astype(np.int32)
    grad = rng.normal(0, 3e-4, (P, 3)).astype(np.float32)
    grad[:, 2] = 0
    mr = rng.uniform(0, 40, P).astype(np.float32)
    acc = rng.uniform(0, 2e-3, P).astype(np.float32)
    den = rng.integers(0, 6, P).astype(np.float32)
    acc[den == 0] = 0
    ls = rng.normal(np.log(0.04), 0.6, (P, cols)).astype(np.float32)
    lo = rng.normal(0.0, 3.0, (P, 1)).astype(np.float32)
    # sit on the thresholds: exp(log_scale) == 0.05 / 0.1 as nearly as float32 allows, sigmoid == 0.005
    ls[:8] = np.float32(np.log(0.05)); ls[8:16] = np.nextafter(np.float32(np.log(0.05)), np.float32(1)); ls[16:24] = np.float32(np.log(0.1))
    lo[:8, 0] = np.float32(np.log(0.005 / 0.995)); lo[8:16, 0] = np.nextafter(lo[0, 0], np.float32(-10))
    return radius, grad, mr, acc, den, ls, lo


@pytest.mark.parametrize("P,cols", [(1, 3), (1000, 3), (70001, 1)])
def test_stats_and_masks_match_the_restatement(gpu, oracle, P, cols):
    from models.SLAM.utils import slam_external as se
    from oracle import densify_stats as ods
    radius, grad, mr, acc, den, ls, lo = _state(P, 3 + P, cols)
    t = lambda a: torch.from_numpy(a).to(gpu)
    means2D = torch.zeros((P, 3), device=gpu, requires_grad=True)
    means2D.grad = t(grad)
    variables = dict(means2D=means2D, max_2D_radius=t(mr), means2D_gradient_accum=t(acc), denom=t(den))
    # forward-only call (tracking iterations): seen + max radius
    se.update_seen_and_radius(variables, t(radius))
    seen_o, mr_o = ods.seen_and_radius(radius, mr)
    assert np.array_equal(variables['seen'].cpu().numpy(), seen_o) and np.array_equal(variables['max_2D_radius'].cpu().numpy(), mr_o)
    # mapping iteration: accumulate with the reference's `seen` route, then the fused route on a fresh copy
    se.accumulate_mean2d_gradient(variables)
    acc_o, den_o = ods.accumulate_mean2d_gradient(grad, seen_o, acc, den)
    assert np.array_equal(variables['means2D_gradient_accum'].cpu().numpy(), acc_o) and np.array_equal(variables['denom'].cpu().numpy(), den_o)
    v2 = dict(means2D=means2D, max_2D_radius=t(mr), means2D_gradient_accum=t(acc), denom=t(den))
    se.accumulate_mean2d_gradient(v2, radius=t(radius))
    assert torch.equal(v2['seen'], variables['seen']) and torch.equal(v2['max_2D_radius'], variables['max_2D_radius'])
    assert torch.equal(v2['means2D_gradient_accum'], variables['means2D_gradient_accum']) and torch.equal(v2['denom'], variables['denom'])
    # masks
    params = dict(log_scales=t(ls), logit_opacities=t(lo))
    for thr in (0.0002, 0.0):
        c, s = se.densify_masks(params, variables, thr)
        c_o, s_o = ods.densify_masks(acc_o, den_o, ls, thr)
        assert np.array_equal(c.cpu().numpy(), c_o) and np.array_equal(s.cpu().numpy(), s_o)
    assert 0 < int(c_o.sum()) < P or P == 1
    for op_thr, big in ((0.005, None), (0.005, 0.1), (0.05, 0.1 * 3.7)):
        rm = se.prune_mask(params, op_thr, big)
        assert np.array_equal(rm.cpu().numpy(), ods.prune_mask(lo, ls, op_thr, big))


def test_mapping_iteration_with_the_drop_in_rasteriser(gpu, oracle):
    """render (RGB + depth/silhouette pair) -> loss -> backward -> statistics, as get_loss + densify drive them."""
    from fisher_rast import synthetic
    from models.SLAM.utils import slam_external as se
    from models.SLAM.utils.recon_helpers import setup_camera
    from models.SLAM.utils.slam_helpers import render_rgb_depth_sil
    from oracle import densify_stats as ods
    P, W, H = 4000, 96, 96
    params = {k: v.to(gpu).requires_grad_(k != "none") for k, v in synthetic.room_shell(P, seed=9).items()}
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=gpu)
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(1, seed=10))[0].to(gpu)
    pts = params['means3D']
    tp = (w2c @ torch.cat((pts, torch.ones_like(pts[:, :1])), 1).T).T[:, :3]
    im, radius, depth_sil, rv = render_rgb_depth_sil(params, cam, w2c, tp)
    loss = (im - 0.3).abs().sum() + 0.5 * (depth_sil[0] - 2.0).abs().sum()
    loss.backward()
    variables = dict(means2D=rv['means2D'], max_2D_radius=torch.zeros(P, device=gpu), means2D_gradient_accum=torch.zeros(P, device=gpu),
                     denom=torch.zeros(P, device=gpu))
    se.accumulate_mean2d_gradient(variables, radius=radius)
    g = rv['means2D'].grad.cpu().numpy()
    r = radius.cpu().numpy()
    seen_o, mr_o = ods.seen_and_radius(r, np.zeros(P, np.float32))
    acc_o, den_o = ods.accumulate_mean2d_gradient(g, seen_o, np.zeros(P, np.float32), np.zeros(P, np.float32))
    assert seen_o.sum() > 100 and float(acc_o.max()) > 0
    assert np.array_equal(variables['seen'].cpu().numpy(), seen_o) and np.array_equal(variables['max_2D_radius'].cpu().numpy(), mr_o)
    assert np.array_equal(variables['means2D_gradient_accum'].cpu().numpy(), acc_o) and np.array_equal(variables['denom'].cpu().numpy(), den_o)
    # torch's own chain (what the reference runs) agrees to rounding
    want = torch.norm(rv['means2D'].grad[variables['seen'], :2], dim=-1)
    assert torch.allclose(variables['means2D_gradient_accum'][variables['seen']], want, rtol=1e-6, atol=0)
    to_clone, to_split = se.densify_masks(params, variables, float(np.median(acc_o[seen_o])))
    c_o, s_o = ods.densify_masks(acc_o, den_o, params['log_scales'].detach().cpu().numpy(), float(np.median(acc_o[seen_o])))
    assert np.array_equal(to_clone.cpu().numpy(), c_o) and np.array_equal(to_split.cpu().numpy(), s_o) and c_o.sum() > 10


def test_drop_in_get_loss_against_the_two_render_route(gpu):
    """`make_get_loss` (the graft for the reference's module-level get_loss, gaussian.py:184-297) against the reference's own
    structure -- two GaussianRasterizer calls, torch ops for `seen` / `max_2D_radius` -- on the same parameters, with the test's
    stand-ins for the reference's transform_to_frame / calc_loss (mapping mode: L1 depth + 0.8 L1 + 0.2 (1 - ssim) colour,
    the ssim term left out: it is the reference's own torch code either way)."""
    import torch.nn.functional as F
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from fisher_rast import synthetic
    from models.SLAM.gaussian import make_get_loss
    from models.SLAM.utils.recon_helpers import setup_camera
    from models.SLAM.utils.slam_helpers import transformed_params2rendervar, transformed_params2depthplussilhouette
    P, W, H = 5000, 96, 80
    base = synthetic.room_shell(P, seed=12)
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(1, seed=13))[0].to(gpu)
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=gpu)
    g = torch.Generator().manual_seed(3)
    curr_data = dict(cam=cam, w2c=torch.eye(4, device=gpu), im=torch.rand((3, H, W), generator=g).to(gpu),
                     depth=(torch.rand((1, H, W), generator=g) * 4 + 0.5).to(gpu))
    curr_data['depth'][0, :4, :] = 0.0                      # invalid depth rows (mask)

    def transform_to_frame(params, time_idx, gaussians_grad, camera_grad):    # stand-in: the pose is w2c, Gaussians get the gradient
        pts = params['means3D'] if gaussians_grad else params['means3D'].detach()
        return (w2c @ torch.cat((pts, torch.ones_like(pts[:, :1])), 1).T).T[:, :3]

    def calc_loss(curr_data, im, depth, mask, color_mask, use_l1, use_sil_for_loss, ignore_outlier_depth_loss, tracking):
        return dict(depth=torch.abs(curr_data['depth'] - depth)[mask.detach()].mean(), im=0.8 * torch.abs(im - curr_data['im']).mean())

    weights = dict(im=0.5, depth=1.0)

    def fresh():
        params = {k: v.clone().to(gpu).requires_grad_(True) for k, v in base.items()}
        variables = dict(max_2D_radius=torch.full((P,), 3.0, device=gpu), means2D_gradient_accum=torch.zeros(P, device=gpu), denom=torch.zeros(P, device=gpu))
        return params, variables

    # the graft
    params, variables = fresh()
    loss, variables, wl = make_get_loss(transform_to_frame, calc_loss)(params, curr_data, variables, 0, weights, True, 0.5, True, False, mapping=True)
    loss.backward()
    # the reference's structure
    params2, variables2 = fresh()
    tp = transform_to_frame(params2, 0, True, False)
    rv = transformed_params2rendervar(params2, tp)
    ds = transformed_params2depthplussilhouette(params2, curr_data['w2c'], tp)
    rv['means2D'].retain_grad()
    im, radius, _ = Renderer(raster_settings=cam)(**rv)
    depth_sil, _, _ = Renderer(raster_settings=cam)(**ds)
    depth = depth_sil[0].unsqueeze(0)
    unc = (depth_sil[2].unsqueeze(0) - depth ** 2).detach()
    mask = (curr_data['depth'] > 0) & (~torch.isnan(depth)) & (~torch.isnan(unc))
    l2 = calc_loss(curr_data, im, depth, mask, torch.tile(mask, (3, 1, 1)), True, True, False, False)
    loss2 = sum(v * weights[k] for k, v in l2.items())
    loss2.backward()
    seen = radius > 0
    variables2['max_2D_radius'][seen] = torch.max(radius[seen].float(), variables2['max_2D_radius'][seen])
    assert float(loss.detach()) == float(loss2.detach()) and float(wl["loss"].detach()) == float(loss.detach())          # the fused pair's images are bit-identical
    assert torch.equal(variables['seen'], seen) and torch.equal(variables['max_2D_radius'], variables2['max_2D_radius'])
    assert float((variables['max_2D_radius'] > 3.0).sum()) > 10
    for k in ('means3D', 'rgb_colors', 'unnorm_rotations', 'logit_opacities', 'log_scales'):
        assert_close(params[k].grad.cpu().numpy(), params2[k].grad.cpu().numpy(), 2e-4, f"get_loss d/d{k}", atol_frac=2e-6)
    assert_close(variables['means2D'].grad.cpu().numpy(), rv['means2D'].grad.cpu().numpy(), 2e-4, "get_loss means2D.grad", atol_frac=2e-6)
