"""Drop-in `diff_gaussian_rasterization` for MI355X.

Public surface identical to the reference package
(thirdparty/diff-gaussian-rasterization-modified/diff_gaussian_rasterization/__init__.py:140-204):
`GaussianRasterizationSettings` (NamedTuple, same field order), `GaussianRasterizer(raster_settings,
backward_power=1)` with `forward(means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
cov3D_precomp) -> (color[3,H,W], radii[P] int32, depth[1,H,W])` and `markVisible(positions)`.
`backward_power` is FisherRF's modification: 1 gives gradients, 2 gives the per-Gaussian sums of squared
per-pixel gradients (the diagonal Fisher proxy).  The compute lives in hand-written HIP kernels behind the
C ABI of libfisher_rast.so; `_C` mirrors the reference's pybind module.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool


class _RasterizeGaussians(torch.autograd.Function):
    """Inputs in the reference order (means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
    cov3Ds_precomp, raster_settings, backward_power); gradients come back in that same order
    (__init__.py:125-136 of the reference)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, backward_power):
        rs = raster_settings
        num_rendered, color, radii, geom, binning, img, depth = _C.rasterize_gaussians(
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
            rs.sh_degree, rs.campos, rs.prefiltered)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.backward_power = backward_power
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img)
        # the gradients of radii and depth are ignored (as in the reference's backward): do not have autograd fill zeros for them
        ctx.set_materialize_grads(False)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, _grad_depth):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
        if grad_out_color is None:      # only depth / radii were used downstream
            grad_out_color = torch.zeros((3, rs.image_height, rs.image_width), dtype=torch.float32, device=means3D.device)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = _C.rasterize_gaussians_backward(
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh, rs.sh_degree, rs.campos,
            geom, ctx.num_rendered, binning, img, ctx.backward_power)

        def fit(g, like):
            # absent inputs were passed as empty CPU tensors: autograd wants a gradient of that shape (or None)
            if like is None or like.numel() == 0:
                return None
            return g
        return (grad_means3D, grad_means2D, fit(grad_sh, sh), fit(grad_colors_precomp, colors_precomp), grad_opacities,
                fit(grad_scales, scales), fit(grad_rotations, rotations), fit(grad_cov3Ds_precomp, cov3Ds_precomp),
                None, None)


class _RasterizeGaussiansPair(torch.autograd.Function):
    """Colour image + a second feature image (the reference's depth / silhouette render, gaussian.py:203-211) on ONE
    projection, binning and sort.  Inputs (means3D, means2D, means2D_features, colors_precomp, features, opacities, scales,
    rotations, cov3Ds_precomp, raster_settings); `means2D` receives the colour image's screen-space gradient only, as
    in the reference where each render has its own `means2D` and the densifier reads the colour render's."""

    @staticmethod
    def forward(ctx, means3D, means2D, means2D_features, colors_precomp, features, opacities, scales, rotations,
                cov3Ds_precomp, raster_settings):
        rs = raster_settings
        num_rendered, color, radii, geom, binning, img, depth, feat_img = _C.rasterize_gaussians_pair(
            rs.bg, means3D, colors_precomp, features, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width,
            rs.sh_degree, rs.campos, rs.prefiltered)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, features, means3D, scales, rotations, cov3Ds_precomp, radii, geom, binning, img)
        ctx.mark_non_differentiable(radii, depth)
        return color, radii, depth, feat_img

    @staticmethod
    def backward(ctx, grad_color, _grad_radii, _grad_depth, grad_features):
        rs = ctx.raster_settings
        colors_precomp, features, means3D, scales, rotations, cov3Ds_precomp, radii, geom, binning, img = ctx.saved_tensors
        if grad_color is None:
            grad_color = torch.zeros((3, rs.image_height, rs.image_width), device=means3D.device)
        if grad_features is None:
            grad_features = torch.zeros((3, rs.image_height, rs.image_width), device=means3D.device)
        (g_m2, g_m2f, g_col, g_feat, g_op, g_m3, g_cov, g_sc, g_rot) = _C.rasterize_gaussians_backward_pair(
            rs.bg, means3D, radii, colors_precomp, features, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_color, grad_features, rs.campos, geom, binning, img,
            num_rendered=ctx.num_rendered)

        def fit(g, like):
            return None if (like is None or like.numel() == 0) else g
        return (g_m3, g_m2, g_m2f, g_col, g_feat, g_op, fit(g_sc, scales), fit(g_rot, rotations), fit(g_cov, cov3Ds_precomp), None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, backward_power):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, backward_power)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings, backward_power: int = 1):
        super().__init__()
        self.raster_settings = raster_settings
        self.backward_power = backward_power

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        have_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (have_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings, self.backward_power)

    def forward_pair(self, means3D, means2D, opacities, colors_precomp, features, scales=None, rotations=None,
                     cov3D_precomp=None, means2D_features=None):
        """Not in the reference: `(color, radii, depth, feature_image)` -- what two calls with `colors_precomp` and then
        `features` (same geometry) return, from one projection / binning / sort, with one fused backward (power 1)."""
        if self.backward_power != 1:
            raise Exception('forward_pair is the training-step path: backward_power must be 1')
        have_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (have_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        if means2D_features is None:
            means2D_features = torch.zeros_like(means2D)
        return _RasterizeGaussiansPair.apply(means3D, means2D, means2D_features, colors_precomp, features, opacities, scales,
                                             rotations, cov3D_precomp, self.raster_settings)
