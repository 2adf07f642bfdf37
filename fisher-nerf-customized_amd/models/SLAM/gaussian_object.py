"""`GaussianObjectSLAM`: the object-aware variant's Fisher surface (models/SLAM/gaussian_object.py of the reference).

  compute_Hessian / compute_H_train / pose_eval (1541-1551, 1591-1617, 1940-2045): 11 Fisher columns
      [mean xyz | opacity | scale xyz | rot rxyz], optional random Gaussians appended with colour 0.5,
      `compute_Hessian(..., return_pose=True)` -> (cur_H, eye(6), vis_count) -- inherited from FisherOps, batched through
      fr_fisher_views.
  POp-GS estimators (2049-2176, 1552-1585, 1619-1732): the reference draws K random upstream-gradient images z_k,
      back-propagates each through the power-2 rasteriser and averages squares (diag form) or per-splat outer products
      (block form) of the gradient rows.  Here the K probes of ALL poses of a call are K x V "views" of ONE fr_fisher_views
      launch with a per-view upstream-gradient image (fr_fisher_cfg.dL_dpix_image) and per-view out_H: no autograd graph,
      no per-probe backward pass.  What a probe returns is the reference's quantity -- the power-2 gradient row of every
      Gaussian under z_k -- so diag / block estimates agree with the reference's route to the scorer's 1e-4 bar
      (tests/test_gpu_fisher_parity.py checks both against the oracle and against the drop-in autograd rasteriser).
  Only the accelerated estimators live here.  `install()` grafts them onto the reference class and leaves everything else
  of that class in place.
"""
import torch

from diff_gaussian_rasterization import GaussianRasterizer as Renderer
from models.SLAM.gaussian import FisherOps, GaussianSLAM

# scorer rows are [mean 0:3 | opacity 3 | scale 4:7 | rot 7:11]; the reference's POp-GS vectors are [mean | opacity | rot | scale]
_ROW_BLOCKS = {"mean": (0, 3), "opacity": (3, 4), "rot": (7, 11), "scale": (4, 7)}
_POPGS_BLOCK_ORDER = ("mean", "opacity", "rot", "scale")


def _criterion(name):
    name = str(name).lower()
    if name not in ("topt", "dopt"):
        raise ValueError("criterion must be 'topt' or 'dopt'")
    return name


class ObjectFisherOps(FisherOps):
    FISHER_COLUMNS = 11

    # ---- probes ---------------------------------------------------------------------------------------------------
    def _draw_probes(self, n, zs=None):
        """[n,3,H,W] upstream gradients; `zs` (a sequence of n [3,H,W] tensors) replaces the random draws."""
        dev = self._device()
        H, W = int(self.cam.image_height), int(self.cam.image_width)
        if zs is not None:
            return torch.stack([z.to(dev).float() for z in zs]).reshape(n, 3, H, W)
        return torch.randn((n, 3, H, W), device=dev)

    def _probe_rows(self, w2cs, zs):
        """Power-2 gradient rows under upstream images zs [V,3,H,W] at poses w2cs [V,4,4]: rows [V,N,11], vis_count [V]."""
        scorer = self._scorer()
        V = int(w2cs.shape[0])
        rows = torch.zeros((V, scorer.P, 11), dtype=torch.float32, device=self._device())
        res = scorer.run(w2cs, out_H=rows, out_H_per_view=True, dL_image=zs)
        return rows, res["vis_count"]

    def _pose_probe_rows(self, w2cs, K, zs=None):
        """rows [V,K,N,11] and vis_count [V] for V poses x K probes, one launch."""
        V = int(w2cs.shape[0])
        rows, vis = self._probe_rows(w2cs.repeat_interleave(K, dim=0), self._draw_probes(V * K, zs))
        return rows.reshape(V, K, rows.shape[1], 11), vis.reshape(V, K)[:, 0]

    @staticmethod
    def _flat_diag(rows):
        """mean over the K probes of the squared rows, flattened block-wise [means(3N) | opacity(N) | rot(4N) | scale(3N)]
        (the layout of gaussian_object.py:2100-2107).  rows [V,K,N,11] -> [V,11N]."""
        sq = (rows * rows).mean(dim=1)
        return torch.cat([sq[:, :, a:b].reshape(sq.shape[0], -1) for a, b in (_ROW_BLOCKS[n] for n in _POPGS_BLOCK_ORDER)], dim=1)

    def _diag_batch(self, w2cs, K, chunk_bytes=4 << 30):
        """diag(J^T J) estimates [V,11N] for V poses, K probes each, in as few launches as memory allows."""
        N = int(self.params['means3D'].shape[0])
        per = max(1, int(chunk_bytes // (K * N * 44)))
        return torch.cat([self._flat_diag(self._pose_probe_rows(w2cs[v0:v0 + per], K)[0]) for v0 in range(0, int(w2cs.shape[0]), per)])

    # ---- diagonal criteria (gaussian_object.py:2049-2109, 1552-1569, 1619-1662, 1706-1719) -------------------------------
    def estimate_diag_JtJ_simple(self, w2c, K: int = 4, zs=None):
        """(diag / K, vis_count) for one pose."""
        rows, vis = self._pose_probe_rows(self._as_w2c(w2c).reshape(1, 4, 4), int(K), zs)
        return self._flat_diag(rows)[0], int(vis[0].item())

    def compute_H_train_popgs(self, K: int = 4):
        if len(self.keyframe_list) == 0:
            raise RuntimeError("No keyframes available for POP-GS prior.")
        w2cs = torch.stack([self._as_w2c(kf['est_w2c']) for kf in self.keyframe_list])
        return self._diag_batch(w2cs, int(K)).sum(dim=0)

    @staticmethod
    def _diag_scores(H_train_diag, diags, lam, criterion):
        """T-opt: -sum 1/(H + J + lam); D-opt: sum log(H + lam + J) - sum log(H + lam); both clamped at 1e-12 as the
        reference clamps (1711, 1716-1718).  diags [V,D] -> [V] on the device."""
        prior = H_train_diag.unsqueeze(0) + lam
        post = (prior + diags).clamp_min(1e-12)
        if criterion == "topt":
            return -post.reciprocal().sum(dim=1)
        return post.log().sum(dim=1) - prior.clamp_min(1e-12).log().sum(dim=1)

    @classmethod
    def topt_score_from_diags(cls, H_train_diag, JtJ_diag_pi, lam: float = 1e-6):
        return cls._diag_scores(H_train_diag, JtJ_diag_pi.reshape(1, -1), lam, "topt")[0]

    @classmethod
    def dopt_score_from_diags(cls, H_train_diag, JtJ_diag_pi, lam: float = 1e-6):
        return cls._diag_scores(H_train_diag, JtJ_diag_pi.reshape(1, -1), lam, "dopt")[0]

    def pose_eval_popgs(self, poses, random_gaussian_params=None, criterion: str = "topt", K: int = 4, lam: float = 1e-6):
        """Scores [V] (cpu) and stack(c2w): every pose's K probes and every keyframe's K probes in two launches."""
        crit = _criterion(criterion)
        H_train_diag = self.compute_H_train_popgs(K=K)
        c2w_all = torch.stack([self._as_w2c(c2w) for c2w in poses])
        diags = self._diag_batch(torch.linalg.inv(c2w_all), int(K))
        return self._diag_scores(H_train_diag, diags, lam, crit).cpu(), c2w_all

    # ---- block criteria (gaussian_object.py:2111-2176, 1572-1585, 1660-1732) -------------------------------------------------
    @staticmethod
    def _block_columns(use_rot, use_scale, use_opacity):
        want = {"mean": True, "opacity": use_opacity, "rot": use_rot, "scale": use_scale}
        return [c for n in _POPGS_BLOCK_ORDER if want[n] for c in range(*_ROW_BLOCKS[n])]

    def _visible_indices(self, w2c):
        """Indices with radius > 0 at this pose (what the reference reads off its forward pass, 2141)."""
        p = self.params
        with torch.no_grad():
            pts = p['means3D']
            cam_pts = (w2c @ torch.cat([pts, torch.ones_like(pts[:, :1])], dim=1).T).T[:, :3].contiguous()
            sc = torch.exp(p['log_scales'])
            sc = sc.expand(-1, 3) if sc.shape[-1] == 1 else sc
            _, radius, _ = Renderer(raster_settings=self.cam)(
                means3D=cam_pts, means2D=torch.zeros_like(cam_pts), opacities=torch.sigmoid(p['logit_opacities']),
                colors_precomp=p['rgb_colors'], scales=sc.contiguous(), rotations=torch.nn.functional.normalize(p['unnorm_rotations']))
        return torch.nonzero(radius > 0).reshape(-1)

    def estimate_block_JtJ(self, w2c, K: int = 2, use_rot: bool = True, use_scale: bool = True, use_opacity: bool = True, zs=None):
        """(H_blocks [Nv,d,d] / K, vis_idx [Nv]): per visible splat the outer product of its power-2 gradient row
        [mean3 | opacity | rot4 | scale3] (blocks present as the flags say), averaged over the K probes."""
        w2c = self._as_w2c(w2c)
        vis_idx = self._visible_indices(w2c)
        rows, _ = self._pose_probe_rows(w2c.reshape(1, 4, 4), int(K), zs)
        G = rows[0][:, vis_idx][:, :, self._block_columns(use_rot, use_scale, use_opacity)]      # [K, Nv, d]
        return torch.einsum('kvi,kvj->vij', G, G) / float(K), vis_idx

    def compute_H_train_blocks(self, K: int = 2, **flags):
        """Sum over the keyframes.  The reference does not align the visible sets of different keyframes: it truncates both
        operands to the smaller count and keeps the first keyframe's index list (1577-1584); kept, so that scores match."""
        if len(self.keyframe_list) == 0:
            raise RuntimeError("No keyframes available for POP-GS prior (blocks).")
        per_kf = [self.estimate_block_JtJ(kf['est_w2c'], K=K, **flags) for kf in self.keyframe_list]
        n = min(int(b.shape[0]) for b, _ in per_kf)
        Hm = per_kf[0][0][:n].clone()
        for b, _ in per_kf[1:]:
            # the reference truncates progressively; truncating everything to the overall minimum gives the same rows
            Hm += b[:n]
        return Hm, per_kf[0][1][:n]

    @staticmethod
    def _block_scores(Hm, J, lam, criterion):
        """T-opt: -sum_b trace((H_b + J_b + lam I)^-1); D-opt: sum_b [logdet(H_b + lam I + J_b) - logdet(H_b + lam I)]."""
        eye = torch.eye(Hm.shape[-1], device=Hm.device, dtype=Hm.dtype)
        prior = Hm + lam * eye
        if criterion == "topt":
            return -torch.linalg.inv(prior + J).diagonal(dim1=-2, dim2=-1).sum()
        return (torch.linalg.slogdet(prior + J)[1] - torch.linalg.slogdet(prior)[1]).sum()

    @classmethod
    def t_opt_blocks(cls, Hm_blocks, J_blocks, lam=1e-6):
        return cls._block_scores(Hm_blocks, J_blocks, lam, "topt")

    @classmethod
    def d_opt_blocks(cls, Hm_blocks, J_blocks, lam=1e-6):
        return cls._block_scores(Hm_blocks, J_blocks, lam, "dopt")

    def pose_eval_popgs_blocks(self, poses, random_gaussian_params=None, criterion: str = "topt", K: int = 6, lam: float = 1e-6,
                               use_rot=True, use_scale=True, use_opacity=True):
        crit = _criterion(criterion)
        flags = dict(use_rot=use_rot, use_scale=use_scale, use_opacity=use_opacity)
        Hm, train_idx = self.compute_H_train_blocks(K=K, **flags)
        scores, c2ws = [], []
        for c2w in poses:
            c2w = self._as_w2c(c2w)
            c2ws.append(c2w)
            Jb, cur_idx = self.estimate_block_JtJ(torch.linalg.inv(c2w), K=K, **flags)
            # splats seen both by the prior and from this pose (both index lists ascend, so the pairing is by value)
            in_train = torch.isin(train_idx, cur_idx)
            if not bool(in_train.any()):
                scores.append(float('-inf'))
                continue
            in_cur = torch.isin(cur_idx, train_idx)
            scores.append(float(self._block_scores(Hm[in_train], Jb[in_cur], lam, crit)))
        return torch.tensor(scores), torch.stack(c2ws)

    @classmethod
    def install(cls, target_cls):
        """Graft the 11-column Fisher methods AND the POp-GS estimators onto the reference's GaussianObjectSLAM."""
        target_cls.FISHER_COLUMNS = cls.FISHER_COLUMNS
        super().install(target_cls)
        for name in ("_draw_probes", "_probe_rows", "_pose_probe_rows", "_flat_diag", "_diag_batch", "_diag_scores",
                     "_block_columns", "_visible_indices", "_block_scores", "estimate_diag_JtJ_simple", "compute_H_train_popgs",
                     "pose_eval_popgs", "estimate_block_JtJ", "compute_H_train_blocks", "pose_eval_popgs_blocks"):
            setattr(target_cls, name, cls.__dict__[name])
        return target_cls


class GaussianObjectSLAM(ObjectFisherOps, GaussianSLAM):
    pass
