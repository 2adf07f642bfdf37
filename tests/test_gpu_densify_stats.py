"""-m gpu: the densification / pruning statistics kernels (fr_densify_stats / fr_densify_masks / fr_prune_mask, mirrored in
models/SLAM/utils/slam_external.py of the package) against oracle/densify_stats.py (slam_external.py:196-200, 345-465;
gaussian.py:289-291 of the reference): accumulators and masks bit for bit, including the values that sit on the thresholds,
0 / 0 gradients, isotropic (1-column) scales, and a whole mapping iteration driven by the drop-in rasteriser."""
import numpy as np
import pytest
import torch

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
