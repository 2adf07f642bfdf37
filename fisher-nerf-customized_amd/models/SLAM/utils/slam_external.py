"""Densification / pruning statistics of the training step (models/SLAM/utils/slam_external.py:196-200, 345-465 and the tail of
get_loss, models/SLAM/gaussian.py:289-291) on MI355X: each torch-op chain of the reference is one kernel pass over the
Gaussians (fr_densify_stats / fr_densify_masks / fr_prune_mask, include/fisher_rast.h).

Only the statistics and the masks are accelerated.  What the reference then DOES with a mask -- cloning / splitting the
parameter tensors, rebuilding the Adam state (cat_params_to_optimizer, remove_points) -- is optimiser bookkeeping that stays
reference Python; `densify_masks` / `prune_mask` hand it the same boolean tensors it computes itself.
"""
import ctypes

import torch

from fisher_rast import _lib


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _f32(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def update_seen_and_radius(variables, radius):
    """The tail of get_loss (gaussian.py:289-291): variables['seen'] = radius > 0 and, where seen,
    max_2D_radius = max(radius, max_2D_radius) -- in place."""
    dev = radius.device
    P = int(radius.shape[0])
    radii = radius if (radius.dtype == torch.int32 and radius.is_contiguous()) else radius.to(torch.int32).contiguous()
    seen = torch.empty((P,), dtype=torch.bool, device=dev)
    mr = variables['max_2D_radius']
    assert mr.dtype == torch.float32 and mr.is_contiguous()
    with torch.cuda.device(dev):
        _lib.check(_lib.load().fr_densify_stats(P, radii.data_ptr(), None, mr.data_ptr(), None, None, seen.data_ptr(), _stream(dev)),
                   "fr_densify_stats")
    variables['seen'] = seen
    return variables


def accumulate_mean2d_gradient(variables, radius=None):
    """slam_external.py:196-200: means2D_gradient_accum[seen] += |means2D.grad[seen, :2]|, denom[seen] += 1 -- in place.
    With `radius` (the colour render's radii) the visibility comes from it and max_2D_radius / seen are updated in the same
    pass (the tail of get_loss); without, variables['seen'] is used as the reference does."""
    grad = _f32(variables['means2D'].grad)
    dev = grad.device
    P = int(grad.shape[0])
    acc, den = variables['means2D_gradient_accum'], variables['denom']
    assert acc.dtype == den.dtype == torch.float32 and acc.is_contiguous() and den.is_contiguous()
    lib = _lib.load()
    with torch.cuda.device(dev):
        if radius is not None:
            radii = radius if (radius.dtype == torch.int32 and radius.is_contiguous()) else radius.to(torch.int32).contiguous()
            seen = torch.empty((P,), dtype=torch.bool, device=dev)
            _lib.check(lib.fr_densify_stats(P, radii.data_ptr(), grad.data_ptr(), variables['max_2D_radius'].data_ptr(), acc.data_ptr(),
                                            den.data_ptr(), seen.data_ptr(), _stream(dev)), "fr_densify_stats")
            variables['seen'] = seen
        else:
            radii = variables['seen'].to(torch.int32)             # 1 where seen: max_2D_radius is not touched on this route
            _lib.check(lib.fr_densify_stats(P, radii.data_ptr(), grad.data_ptr(), None, acc.data_ptr(), den.data_ptr(), None, _stream(dev)),
                       "fr_densify_stats")
    return variables


def densify_masks(params, variables, grad_thresh, clone_max_scale=0.05, split_min_scale=0.05):
    """(to_clone, to_split) of densify() (slam_external.py:419-433) as bool tensors.  NOTE the reference evaluates to_split AFTER
    the clones were appended; a clone has max scale <= clone_max_scale, so with the default thresholds the mask over the
    grown array is this mask followed by False for every clone."""
    ls = _f32(params['log_scales'])
    dev = ls.device
    P, cols = int(ls.shape[0]), int(ls.shape[1]) if ls.dim() == 2 else 1
    to_clone = torch.empty((P,), dtype=torch.bool, device=dev)
    to_split = torch.empty((P,), dtype=torch.bool, device=dev)
    acc, den = _f32(variables['means2D_gradient_accum']), _f32(variables['denom'])
    with torch.cuda.device(dev):
        _lib.check(_lib.load().fr_densify_masks(P, acc.data_ptr(), den.data_ptr(), ls.data_ptr(), cols, float(grad_thresh),
                                                float(clone_max_scale), float(split_min_scale), to_clone.data_ptr(), to_split.data_ptr(),
                                                _stream(dev)), "fr_densify_masks")
    return to_clone, to_split


def prune_mask(params, opacity_thresh, big_thresh=None):
    """to_remove of prune_gaussians() / of the removal pass of densify() (slam_external.py:354, 394-396, 452-457):
    sigmoid(logit_opacities) < opacity_thresh, OR-ed with max scale > big_thresh when big_thresh is given
    (0.1 in prune_gaussians, 0.1 * scene_radius in densify)."""
    lo, ls = _f32(params['logit_opacities']).reshape(-1), _f32(params['log_scales'])
    dev = lo.device
    P, cols = int(lo.shape[0]), int(ls.shape[1]) if ls.dim() == 2 else 1
    out = torch.empty((P,), dtype=torch.bool, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().fr_prune_mask(P, lo.data_ptr(), ls.data_ptr(), cols, float(opacity_thresh),
                                             -1.0 if big_thresh is None else float(big_thresh), out.data_ptr(), _stream(dev)), "fr_prune_mask")
    return out
