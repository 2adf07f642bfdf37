"""Stand-in for the reference's pybind module `diff_gaussian_rasterization._C`
(thirdparty/diff-gaussian-rasterization-modified/ext.cpp:14-18): same three functions, same positional
argument lists and return tuples (rasterize_points.h:18-66), implemented on the C ABI of libfisher_rast.so."""
from fisher_rast import ops as _ops


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered):
    return _ops.rasterize_forward(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                                  cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                                  image_width, sh, degree, campos, prefiltered)


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh,
                                 degree, campos, geomBuffer, R, binningBuffer, imageBuffer, power):
    return _ops.rasterize_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                   cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh,
                                   degree, campos, geomBuffer, R, binningBuffer, imageBuffer, power)


def mark_visible(means3D, viewmatrix, projmatrix):
    return _ops.mark_visible(means3D, viewmatrix, projmatrix)


# ---- not in the reference's pybind module: the second feature image on shared geometry (include/fisher_rast.h) ----
def rasterize_gaussians_pair(background, means3D, colors, features, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                             viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, degree, campos, prefiltered):
    """(num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer, depth, feature_image)"""
    import torch
    return _ops.rasterize_forward(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                                  viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, torch.Tensor([]),
                                  degree, campos, prefiltered, features=features)


def rasterize_features(features, raster_cfg_args, geomBuffer, binningBuffer, imageBuffer):
    return _ops.rasterize_forward_features(features, raster_cfg_args, geomBuffer, binningBuffer, imageBuffer)


def rasterize_gaussians_backward_pair(background, means3D, radii, colors, features, scales, rotations, scale_modifier,
                                      cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                      dL_dout_features, campos, geomBuffer, binningBuffer, imageBuffer, num_rendered=0):
    return _ops.rasterize_backward_pair(background, means3D, radii, colors, features, scales, rotations, scale_modifier,
                                        cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                        dL_dout_features, campos, geomBuffer, binningBuffer, imageBuffer, num_rendered=num_rendered)
