"""Render-variable builders on the hot path (models/SLAM/utils/slam_helpers.py:178-188, 235-252, 268-279)."""
import torch
import torch.nn.functional as F


def _scales3(params):
    ls = params['log_scales']
    return ls if ls.shape[-1] == 3 else torch.tile(ls, (1, 3))


def transformed_params2rendervar(params, transformed_pts):
    return {
        'means3D': transformed_pts,
        'colors_precomp': params['rgb_colors'],
        'rotations': F.normalize(params['unnorm_rotations']),
        'opacities': torch.sigmoid(params['logit_opacities']),
        'scales': torch.exp(_scales3(params)),
        'means2D': torch.zeros_like(params['means3D'], requires_grad=True) + 0,
    }


def get_depth_and_silhouette(pts_3D, w2c):
    """Per-Gaussian "colour" (z, 1, z^2) used for the depth + silhouette render."""
    pts4 = torch.cat((pts_3D, torch.ones_like(pts_3D[:, :1])), dim=-1)
    z = (w2c @ pts4.transpose(0, 1)).transpose(0, 1)[:, 2:3]
    return torch.cat((z, torch.ones_like(z), torch.square(z)), dim=1).float()


def transformed_params2depthplussilhouette(params, w2c, transformed_pts):
    return {
        'means3D': transformed_pts,
        'colors_precomp': get_depth_and_silhouette(transformed_pts, w2c),
        'rotations': F.normalize(params['unnorm_rotations']),
        'opacities': torch.sigmoid(params['logit_opacities']),
        'scales': torch.exp(_scales3(params)),
        'means2D': torch.zeros_like(params['means3D'], requires_grad=True) + 0,
    }


def render_rgb_depth_sil(params, cam, w2c, transformed_pts, renderer_cls=None):
    """The two renders of the reference's get_loss (models/SLAM/gaussian.py:199-211) -- RGB, then (depth, silhouette, depth^2)
    on the same Gaussians -- as one call on one projection / binning / sort (`GaussianRasterizer.forward_pair`).
    Returns (im [3,H,W], radius [P], depth_sil [3,H,W], rendervar); `rendervar['means2D']` keeps the colour render's
    screen-space gradient, the statistic the densifier accumulates (gaussian.py:207)."""
    if renderer_cls is None:
        from diff_gaussian_rasterization import GaussianRasterizer as renderer_cls
    rendervar = transformed_params2rendervar(params, transformed_pts)
    rendervar['means2D'].retain_grad()
    feats = get_depth_and_silhouette(transformed_pts, w2c)
    im, radius, _, depth_sil = renderer_cls(raster_settings=cam).forward_pair(
        rendervar['means3D'], rendervar['means2D'], rendervar['opacities'], rendervar['colors_precomp'], feats,
        scales=rendervar['scales'], rotations=rendervar['rotations'])
    return im, radius, depth_sil, rendervar
