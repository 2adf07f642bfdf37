"""oracle/densify_stats.py -- TEST INFRASTRUCTURE ONLY.  NOT PRODUCT CODE.

NumPy restatement of the densification / pruning statistics of the reference's training step, float32 throughout:
  * tail of get_loss                 models/SLAM/gaussian.py:289-291       (seen, max_2D_radius)
  * accumulate_mean2d_gradient       models/SLAM/utils/slam_external.py:196-200
  * densify(): grads, to_clone, to_split   slam_external.py:419-433
  * prune_gaussians() / densify(): to_remove   slam_external.py:354, 394-396, 452-457
exp() is the oracle's orc_expf (oracle/fisher_oracle.c), the fixed sequence of IEEE operations the kernels use as well, so
thresholded masks are comparable bit for bit.  torch.norm(g[:, :2], dim=-1) is restated as sqrt(gx*gx + gy*gy) in float32
(torch's 2-norm over two elements reduces the same way; no reference vectors exist for this: parity unpinned).
"""
import numpy as np

from . import ref


def seen_and_radius(radius, max_2D_radius):
    seen = np.asarray(radius) > 0
    out = np.asarray(max_2D_radius, np.float32).copy()
    out[seen] = np.maximum(np.asarray(radius)[seen].astype(np.float32), out[seen])
    return seen, out


def accumulate_mean2d_gradient(grad_means2D, seen, accum, denom):
    g = np.asarray(grad_means2D, np.float32)
    accum, denom = np.asarray(accum, np.float32).copy(), np.asarray(denom, np.float32).copy()
    n = np.sqrt(g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1], dtype=np.float32)
    accum[seen] += n[seen]
    denom[seen] += np.float32(1)
    return accum, denom


def max_scale(log_scales):
    ls = np.asarray(log_scales, np.float32).reshape(len(log_scales), -1)
    return ref.expf(ls).max(axis=1)


def densify_masks(accum, denom, log_scales, grad_thresh, clone_max_scale=0.05, split_min_scale=0.05):
    with np.errstate(invalid="ignore", divide="ignore"):
        grads = (np.asarray(accum, np.float32) / np.asarray(denom, np.float32)).astype(np.float32)
    grads[np.isnan(grads)] = 0.0
    ms = max_scale(log_scales)
    return (grads >= np.float32(grad_thresh)) & (ms <= np.float32(clone_max_scale)), ms > np.float32(split_min_scale)


def prune_mask(logit_opacities, log_scales, opacity_thresh, big_thresh=None):
    x = np.asarray(logit_opacities, np.float32).reshape(-1)
    op = (np.float32(1) / (np.float32(1) + ref.expf(-x))).astype(np.float32)
    rm = op < np.float32(opacity_thresh)
    if big_thresh is not None:
        rm = rm | (max_scale(log_scales) > np.float32(big_thresh))
    return rm
