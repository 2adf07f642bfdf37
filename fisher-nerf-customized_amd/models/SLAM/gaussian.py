"""`GaussianSLAM`: the Fisher-information / render operator surface of the reference class
(models/SLAM/gaussian.py) on MI355X.

In scope (same names, arguments and return conventions as the reference):
    compute_Hessian (1503-1570)   compute_H_train (1338-1348)   pose_eval (1354-1375)
    render_at_pose (555-579)      gs_pts_cnt (1350-1352)        gaussian_points / cur_frame_idx (1590-1598)
    pause / resume / color_refinement / stop (1600-1614)
The SLAM loop itself (init, track_rgbd, densify, keyframe selection ...) is NOT rebuilt: it is reference
Python that stays as it is.  `FisherOps.install(cls)` grafts the accelerated methods onto the reference class
so that tester_gaussians_navigation.py keeps calling `slam.pose_eval(...)` unchanged; `GaussianSLAM` below is
the same operator surface as a standalone object built from a parameter dict (a `params{t}.npz` checkpoint
or synthetic data) for tests and benchmarks.

What changes underneath: instead of one rasteriser forward + backward(power=2) + cat + sum per view, with
per-view allocations and two host syncs, every call batches its views through FisherScorer
(fisher_rast/ops.py -> fr_fisher_views) and synchronises once when the scores are brought to the host.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from diff_gaussian_rasterization import GaussianRasterizer as Renderer
from fisher_rast.ops import FisherScorer
from models.SLAM.utils.common_utils import checkpoint_time_idx, load_params_ckpt, save_params, save_params_ckpt
from models.SLAM.utils.recon_helpers import setup_camera
from models.SLAM.utils.slam_helpers import (transformed_params2rendervar, transformed_params2depthplussilhouette,
                                            render_rgb_depth_sil)
from models.SLAM.utils.slam_external import update_seen_and_radius


def _loss_pixel_mask(depth_sil, gt_depth, sil_thres, reject_outliers, need_presence):
    """Pixels that enter the tracking / mapping loss, from the depth / silhouette / depth^2 render `depth_sil` [3,H,W] -- the rule of
    the reference's get_loss (models/SLAM/gaussian.py:212-233) as one conjunction: a measured depth; with `reject_outliers` an
    absolute depth error below ten times its median (taken over the whole frame: the zeros of unmeasured pixels count, as they do
    there); no NaN in the rendered depth nor in its variance proxy E[z^2] - E[z]^2; with `need_presence` a silhouette above the
    threshold.  Returns (rendered depth [1,H,W], boolean mask [1,H,W])."""
    rendered = depth_sil[0:1]
    keep = gt_depth > 0
    if reject_outliers:
        err = (gt_depth - rendered).abs() * keep
        keep = keep & (err < 10 * err.median())
    keep = keep & ~(torch.isnan(rendered) | torch.isnan(depth_sil[2:3] - rendered ** 2))
    if need_presence:
        keep = keep & (depth_sil[1] > sil_thres)
    return rendered, keep.detach()


def make_get_loss(transform_to_frame, calc_loss):
    """Drop-in for the module-level `get_loss` of the reference (models/SLAM/gaussian.py:184-297): same signature, same return
    `(loss, variables, weighted_losses)`.  What changes: the two rasteriser calls of 205-211 (RGB, then depth / silhouette /
    depth^2 on the same Gaussians) are ONE projection / binning / sort with two compositing passes and one fused backward
    (`render_rgb_depth_sil` -> fr_forward_pair / fr_backward_pair), and the `seen` / `max_2D_radius` update of 289-291 is one
    kernel pass (fr_densify_stats).  The loss terms are the reference's own `calc_loss` and the pose / point transform its own
    `transform_to_frame` (models/SLAM/utils/slam_helpers.py:23-44, 282-317), handed in by the caller; the pixel mask is
    `_loss_pixel_mask`.  The matplotlib dump of 240-284 is not reproduced (visualize_tracking_loss is accepted and ignored).
    Opt-in: `FisherOps.install(cls, patch_get_loss=True)` puts it into the reference module."""
    @torch.enable_grad()
    def get_loss(params, curr_data, variables, iter_time_idx, loss_weights, use_sil_for_loss,
                 sil_thres, use_l1, ignore_outlier_depth_loss, tracking=False,
                 mapping=False, do_ba=False, plot_dir=None, visualize_tracking_loss=False, tracking_iteration=None):
        if not (tracking or mapping):
            raise ValueError("get_loss: one of tracking / mapping must be set")     # (the reference fails with a NameError here)
        # tracking: only the camera pose takes a gradient; mapping: only the Gaussians (gaussian.py:189-199)
        pts = transform_to_frame(params, iter_time_idx, gaussians_grad=not tracking, camera_grad=bool(tracking))
        im, radius, depth_sil, rendervar = render_rgb_depth_sil(params, curr_data['cam'], curr_data['w2c'], pts)
        variables['means2D'] = rendervar['means2D']      # densification reads the colour render's screen-space gradient (gaussian.py:207)
        depth, mask = _loss_pixel_mask(depth_sil, curr_data['depth'], sil_thres, ignore_outlier_depth_loss, tracking and use_sil_for_loss)
        terms = calc_loss(curr_data, im, depth, mask, mask.repeat(3, 1, 1), use_l1, use_sil_for_loss, ignore_outlier_depth_loss, tracking)
        weighted = {name: value * loss_weights[name] for name, value in terms.items()}
        total = sum(weighted.values())
        update_seen_and_radius(variables, radius)        # variables['seen'], variables['max_2D_radius'] (gaussian.py:289-291)
        weighted['loss'] = total
        return total, variables, weighted
    return get_loss


class FisherOps:
    """Mixin with the accelerated Fisher methods.  Needs: self.params (dict of tensors), self.cam
    (GaussianRasterizationSettings), self.keyframe_list (dicts with 'est_w2c')."""

    FISHER_COLUMNS = 4       # [camera-frame mean xyz | opacity]
    H_TRAIN_REG = 0.1        # gaussian.py:1357

    # -- internals -----------------------------------------------------------------------------------
    def _device(self):
        return self.params['means3D'].device

    def _as_w2c(self, m):
        if isinstance(m, np.ndarray):
            m = torch.from_numpy(m)
        return m.to(self._device()).float()

    _PARAM_KEYS = ('means3D', 'rgb_colors', 'unnorm_rotations', 'logit_opacities', 'log_scales')

    def _scorer_key(self, extra):
        """Identity + version of every tensor the scorer is built from: an in-place optimiser step bumps `_version`, a
        densification / pruning pass replaces the tensors -- either way the key changes and the scorer is rebuilt."""
        def sig(t):
            return None if t is None else (id(t), t._version, t.data_ptr(), tuple(t.shape))
        key = [sig(self.params.get(k, None)) for k in self._PARAM_KEYS] + [id(self.cam), self.FISHER_COLUMNS]
        if extra is not None and extra is not False:
            key += [sig(extra[k]) for k in ('means3D', 'rotations', 'opacity', 'scales')]
        return tuple(key)

    def _scorer(self, extra=None):
        """Activated render variables (gaussian.py:1529-1533) wrapped in a FisherScorer, with its packed inputs and workspace.
        Kept between calls while the map is unchanged (the tester scores ~630 path steps per planning round on one map);
        `extra` appends random Gaussians (gaussian_object.py:1971-1992)."""
        key = self._scorer_key(extra)
        cached = getattr(self, "_scorer_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        p = self.params
        with torch.no_grad():
            means = p['means3D'].detach()
            colors = p['rgb_colors'].detach() if p.get('rgb_colors', None) is not None else \
                torch.full_like(means, 0.5)
            rot = F.normalize(p['unnorm_rotations'].detach())
            op = torch.sigmoid(p['logit_opacities'].detach())
            sc = torch.exp(p['log_scales'].detach())
            if sc.shape[-1] == 1:
                sc = torch.tile(sc, (1, 3))
            if extra is not None and extra is not False:
                dev = means.device
                means = torch.cat([means, extra['means3D'].to(dev).float()], dim=0)
                rot = torch.cat([rot, extra['rotations'].to(dev).float()], dim=0)
                op = torch.cat([op, extra['opacity'].to(dev).float().reshape(-1, 1)], dim=0)
                sc = torch.cat([sc, extra['scales'].to(dev).float()], dim=0)
                colors = torch.cat([colors, torch.full((extra['means3D'].shape[0], 3), 0.5, device=dev)], dim=0)
        scorer = FisherScorer(self.cam, means, colors, rot, op, sc, columns=self.FISHER_COLUMNS, dL_dpix=1e-3)
        self._scorer_cache = (key, scorer)
        return scorer

    # -- reference surface ---------------------------------------------------------------------------
    def compute_Hessian(self, rel_w2c, return_points=False, random_gaussian_params=False, return_pose=False):
        """One view's diagonal Fisher proxy (gaussian.py:1503-1570): (N, C) if return_points else flat
        [means(3N) | opacity(N)] -- the reference flattens per block, not per row (1559-1560)."""
        scorer = self._scorer(random_gaussian_params if self.FISHER_COLUMNS == 11 else None)
        w2c = self._as_w2c(rel_w2c).reshape(1, 4, 4)
        N, C = scorer.P, self.FISHER_COLUMNS
        cur_H = torch.zeros((N, C), dtype=torch.float32, device=self._device())
        res = scorer.run(w2c, out_H=cur_H)
        self._last_vis_count = res["vis_count"]
        if not return_points:
            blocks = [cur_H[:, 0:3].reshape(-1), cur_H[:, 3:4].reshape(-1)]
            if C == 11:
                blocks += [cur_H[:, 4:7].reshape(-1), cur_H[:, 7:11].reshape(-1)]
            cur_H = torch.cat(blocks)
        if not return_pose:
            return cur_H
        pose_H = torch.eye(6, device=self._device())
        if C == 11:
            return cur_H, pose_H, int(self._last_vis_count[0].item())
        return cur_H, pose_H

    def compute_H_train(self, random_gaussians=None):
        """Sum of cur_H over the keyframes (gaussian.py:1338-1348), all keyframes in one batched call."""
        if len(self.keyframe_list) == 0:
            return None
        scorer = self._scorer(random_gaussians if self.FISHER_COLUMNS == 11 else None)   # shared with pose_eval's scoring launch
        w2cs = self._stack_poses([kf['est_w2c'] for kf in self.keyframe_list])
        H_train = torch.zeros((scorer.P, self.FISHER_COLUMNS), dtype=torch.float32, device=self._device())
        scorer.run(w2cs, out_H=H_train)
        return H_train

    def _keyframe_key(self):
        """The keyframe poses as (tensor objects, their versions), or None when they cannot be followed (not device tensors).  The
        cache holds on to the tensor OBJECTS, so a later tensor cannot take a freed one's identity."""
        if not getattr(self, "CACHE_H_TRAIN", True):
            return None
        ts = [kf['est_w2c'] for kf in self.keyframe_list]
        if not all(isinstance(t, torch.Tensor) and t.is_cuda for t in ts):
            return None
        return ts, tuple(t._version for t in ts)

    @staticmethod
    def _same_keyframes(a, b):
        return a is not None and b is not None and len(a[0]) == len(b[0]) and a[1] == b[1] and all(x is y for x, y in zip(a[0], b[0]))

    def gs_pts_cnt(self, random_gaussian_params=None):
        """ API Setting """
        return 1

    def _stack_poses(self, poses):
        """[V,4,4] fp32 on the device from a tensor, an array, or a list of either: one stack and one transfer, not V of them"""
        dev = self._device()
        if isinstance(poses, torch.Tensor):
            return poses.reshape(-1, 4, 4).to(dev).float()
        if isinstance(poses, np.ndarray):
            return torch.from_numpy(np.ascontiguousarray(poses.reshape(-1, 4, 4))).to(dev).float()
        if len(poses) and all(isinstance(p, torch.Tensor) and p.device == dev for p in poses):
            return torch.stack(list(poses)).float()
        return torch.from_numpy(np.stack([np.asarray(p.detach().cpu() if isinstance(p, torch.Tensor) else p) for p in poses])).to(dev).float()

    def pose_eval(self, poses, random_gaussian_params=None, criterion=None):
        """Scores of candidate poses (gaussian.py:1354-1375): returns (scores cpu fp32 [V], stack(c2w) [V,4,4]).
        H_train over the keyframes and the candidate scores are two launches on one stream with ONE host synchronisation:
        the two 16-byte status words travel to the host together with the scores."""
        extra = random_gaussian_params if self.FISHER_COLUMNS == 11 else None
        c2w = self._stack_poses(poses)
        scorer = self._scorer(extra)
        V, K = int(c2w.shape[0]), len(self.keyframe_list)
        if 0 < K and max(V, K) <= scorer.max_views_per_launch():
            # H_train is a function of (map, keyframe poses): a planner calls pose_eval for batch after batch of candidates between two
            # mapping steps, so 1 / (H_train + reg) is kept while neither has changed (the scorer is per map version already; the
            # keyframe poses are followed by tensor identity + version -- poses that are not device tensors are not followed, no reuse)
            kkey = self._keyframe_key()
            cached = getattr(self, "_h_inv_cache", None)
            if cached is not None and cached[0] is scorer and self._same_keyframes(cached[1], kkey):
                r2 = scorer.launch(c2w, H_inv=cached[2], poses_are_c2w=True)
                host = torch.cat([r2["status"], r2["scores"].view(torch.int32)]).cpu()
                if int(host[1]) == 0:
                    return host[4:].view(torch.float32).clone(), c2w
            else:
                kf = self._stack_poses([kf['est_w2c'] for kf in self.keyframe_list])
                H_train = torch.zeros((scorer.P, self.FISHER_COLUMNS), dtype=torch.float32, device=self._device())
                r1 = scorer.launch(kf, out_H=H_train)
                # (the poses go in as they are: the library inverts them -- one kernel in place of torch.linalg.inv's dozen launches)
                H_inv = torch.reciprocal(H_train + self.H_TRAIN_REG)
                r2 = scorer.launch(c2w, H_inv=H_inv, poses_are_c2w=True)
                host = torch.cat([r1["status"], r2["status"], r2["scores"].view(torch.int32)]).cpu()      # the one sync
                if int(host[1]) == 0 and int(host[5]) == 0:
                    self._h_inv_cache = (scorer, kkey, H_inv) if kkey is not None else None
                    return host[8:].view(torch.float32).clone(), c2w
            self._h_inv_cache = None
            # the tile-instance buffer was too small (nothing was accumulated or scored): the growing path below repeats both
        H_train = self.compute_H_train(extra)
        H_train_inv = torch.reciprocal(H_train + self.H_TRAIN_REG)
        res = scorer.run(c2w, H_inv=H_train_inv, poses_are_c2w=True)
        scores = res["scores"].cpu()
        return scores, c2w

    def path_scores(self, w2cs, H_inv_per_view):
        """Batched form of the planner's per-step `sum(cur_H * H_train_inv_path)` (tester 1688-1695): every view
        gets its own weight block.  Returns the V sums on the device (the caller takes the log)."""
        scorer = self._scorer()
        return scorer.run(self._as_w2c(w2cs), H_inv=H_inv_per_view, H_inv_per_view=True)["scores"]

    @classmethod
    def install(cls, target_cls, patch_get_loss=False):
        """Graft the accelerated methods onto the reference's class.  `patch_get_loss=True` also replaces the
        module-level `get_loss` of the module `target_cls` lives in by the fused-render form (`make_get_loss`); off by default --
        a caller that only wants the Fisher scorer keeps the reference's training step untouched."""
        for name in ("_device", "_as_w2c", "_stack_poses", "_scorer", "_scorer_key", "_keyframe_key", "_same_keyframes", "_PARAM_KEYS", "compute_Hessian", "compute_H_train",
                     "pose_eval", "path_scores"):
            setattr(target_cls, name, getattr(cls, name))
        # the module-level get_loss of the reference (gaussian.py:184-297), rebuilt around the reference module's own
        # transform_to_frame / calc_loss: one fused render pair instead of two rasteriser calls
        import sys
        mod = sys.modules.get(target_cls.__module__)
        if patch_get_loss and mod is not None and all(hasattr(mod, n) for n in ("get_loss", "transform_to_frame", "calc_loss")):
            mod.get_loss = make_get_loss(mod.transform_to_frame, mod.calc_loss)
        if not hasattr(target_cls, "FISHER_COLUMNS"):
            target_cls.FISHER_COLUMNS = cls.FISHER_COLUMNS
        target_cls.H_TRAIN_REG = cls.H_TRAIN_REG
        return target_cls


class GaussianSLAM(FisherOps):
    """Standalone carrier of the operator surface: a Gaussian map (param dict), a camera and keyframes."""

    def __init__(self, config=None, params=None, intrinsics=None, width=None, height=None, device="cuda"):
        self.config = config
        self.cfg = config
        if config is not None and intrinsics is None:
            cal = config["SLAM"]["Dataset"]["Calibration"] if "SLAM" in config else config["Dataset"]["Calibration"]
            intrinsics = np.array([[cal["fx"], 0.0, cal["cx"]], [0.0, cal["fy"], cal["cy"]], [0.0, 0.0, 1.0]])
            width = width or cal.get("width", None)
            height = height or cal.get("height", None)
        self.device = torch.device(device)
        self.intrinsics = None if intrinsics is None else torch.as_tensor(np.asarray(intrinsics)).float().to(self.device)
        self.params = {}
        self.variables = {}
        self.checkpoint_extras = {}
        self.cam = None
        self.frame_idx = 0
        self.keyframe_list = []
        self.keyframe_time_indices = []
        self.first_frame_w2c = torch.eye(4, device=self.device)
        self.save_dir = self.eval_dir = None
        if params is not None:
            self.load_params(params)
        if intrinsics is not None and width is not None and height is not None:
            self.set_camera(width, height, intrinsics)

    # -- construction helpers (checkpoint format: common_utils.py:45-59, gaussian.py:156-168) ----------
    def load_params(self, params):
        """`params`: a dict of arrays / tensors, or the path of a `params{t}.npz` / `params.npz` checkpoint written by the
        reference (common_utils.py:35-59).  A path is read the way tester_gaussians_navigation.py:2745-2760 does: the extras
        "Uncertainty" / "occ_map" are kept aside, the densification statistics are reset, and
        `keyframe_time_indices{t}.npy` next to `eval_dir` is picked up when it exists."""
        if isinstance(params, (str, os.PathLike)):
            weight_file = os.fspath(params)
            self.params, self.checkpoint_extras = load_params_ckpt(weight_file, device=self.device)
            n = self.params['means3D'].shape[0]
            for k in ('max_2D_radius', 'means2D_gradient_accum', 'denom', 'timestep'):
                self.variables[k] = torch.zeros(n, device=self.device, dtype=torch.float32)
            t = checkpoint_time_idx(weight_file)
            if t is not None:
                self.frame_idx = t
                kf_file = os.path.join(self.eval_dir or os.path.dirname(weight_file), f"keyframe_time_indices{t}.npy")
                if os.path.exists(kf_file):
                    self.keyframe_time_indices = np.load(kf_file).tolist()
            return self
        self.params = {k: torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v).float().to(self.device)
                       for k, v in params.items()}
        return self

    def save_params_ckpt(self, output_dir, time_idx=None, **extra_args):
        """Writes `params{time_idx}.npz` (or `params.npz`) in the reference's format."""
        if time_idx is None:
            return save_params(self.params, output_dir)
        return save_params_ckpt(self.params, output_dir, time_idx, **extra_args)

    def set_camera(self, width, height, intrinsics):
        k = np.asarray(intrinsics.cpu() if isinstance(intrinsics, torch.Tensor) else intrinsics)
        self.intrinsics = torch.as_tensor(k).float().to(self.device)
        # the view matrix is identity: Gaussians are moved into the candidate frame instead (gaussian.py:493-495)
        self.cam = setup_camera(width, height, k, np.eye(4), device=self.device)
        return self

    def add_keyframe(self, est_w2c, **extra):
        kf = dict(est_w2c=self._as_w2c(est_w2c), id=len(self.keyframe_list))
        kf.update(extra)
        self.keyframe_list.append(kf)
        return kf

    # -- rendering (gaussian.py:555-579) ---------------------------------------------------------------
    def render_at_pose(self, c2w, white_bg=True, mask=None):
        rel_w2c = torch.linalg.inv(self._as_w2c(c2w))
        pts = self.params['means3D']
        pts4 = torch.cat((pts, torch.ones_like(pts[:, :1])), dim=1)
        transformed_pts = (rel_w2c @ pts4.T).T[:, :3]
        rendervar = transformed_params2rendervar(self.params, transformed_pts)
        depth_sil_rendervar = transformed_params2depthplussilhouette(self.params, self.first_frame_w2c, transformed_pts)
        im, radius, _, = Renderer(raster_settings=self.cam)(**rendervar)
        self.variables['means2D'] = rendervar['means2D']
        depth_sil, _, _, = Renderer(raster_settings=self.cam)(**depth_sil_rendervar)
        depth = depth_sil[0, :, :].unsqueeze(0)
        return {"render": im, "depth": depth}

    # -- small surface ---------------------------------------------------------------------------------
    @property
    def cur_frame_idx(self):
        return self.frame_idx

    def get_gaussian_xyz(self):
        return self.params['means3D']

    @property
    def gaussian_points(self):
        return self.get_gaussian_xyz()

    def pause(self):
        """ API to be compatible with Mono GS """
        return

    def resume(self):
        """ API to be compatible with Mono GS """
        return

    def color_refinement(self):
        """ API to be compatible with Mono GS """
        return

    def stop(self):
        """ API to be compatible with Mono GS """
        return
