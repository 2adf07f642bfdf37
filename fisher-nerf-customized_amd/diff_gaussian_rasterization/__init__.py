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
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, _grad_depth):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
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
