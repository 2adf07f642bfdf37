"""`GaussianObjectSLAM`: the object-aware variant's Fisher surface (models/SLAM/gaussian_object.py of the reference):
  compute_Hessian / compute_H_train / pose_eval (1541-1551, 1591-1617, 1940-2045): 11 Fisher columns
      [mean xyz | opacity | scale xyz | rot rxyz], optional random Gaussians appended with colour 0.5,
      `compute_Hessian(..., return_pose=True)` -> (cur_H, eye(6), vis_count)  -- batched through fr_fisher_views;
  estimate_diag_JtJ_simple (2049-2109), compute_H_train_popgs (1552-1569), pose_eval_popgs (1619-1662),
  topt_score_from_diags / dopt_score_from_diags (1706-1719): the POp-GS "simple diag" criteria.  These are NOT linear in
      cur_H and need K backward passes with random upstream gradients on one forward, so they run through the drop-in
      autograd rasteriser exactly as the reference does (forward once, `backward(gradient=z, retain_graph=...)` K times on
      the power-2 rasteriser, squares of those gradients averaged -- the reference's own quirk).
  estimate_block_JtJ (2111-2176), compute_H_train_blocks (1572-1585), pose_eval_popgs_blocks (1660-1704),
  t_opt_blocks / d_opt_blocks (1721-1732): the per-splat d x d block form of the same criteria (d <= 11), same route.
  Fused route (default, `fused=True`): the K probes `im.backward(gradient=z_k)` on the power-2 rasteriser are K "views" of
      the batched Fisher kernel with a per-view upstream-gradient IMAGE (fr_fisher_cfg.dL_dpix_image) and per-view
      `out_H` -- one launch for all probes (and, in pose_eval_popgs, for all poses), no autograd graph, no K generic
      backward passes.  Same numbers as the autograd route within the scorer's 1e-4 bar.
"""
import numpy as np
import torch
import torch.nn.functional as F

from diff_gaussian_rasterization import GaussianRasterizer as Renderer
from models.SLAM.gaussian import FisherOps, GaussianSLAM


class ObjectFisherOps(FisherOps):
    FISHER_COLUMNS = 11

    # ---- fused probes -------------------------------------------------------------------------------------------
    def _draw_probes(self, n, zs=None):
        """[n,3,H,W] upstream gradients: the reference draws `torch.randn_like(im)` per probe (2088, 2158)."""
        dev = self._device()
        H, W = int(self.cam.image_height), int(self.cam.image_width)
        if zs is not None:
            return torch.stack([z.to(dev).float() for z in zs]).reshape(n, 3, H, W)
        return torch.randn((n, 3, H, W), device=dev)

    def _probe_rows(self, w2cs, zs):
        """Power-2 gradient rows of the rasteriser under upstream images zs [V,3,H,W] at poses w2cs [V,4,4]:
        rows [V,N,11] in the Fisher column order [mean3 | opacity | scale3 | rot4], and vis_count [V]."""
        scorer = self._scorer()
        V = int(w2cs.shape[0])
        rows = torch.zeros((V, scorer.P, 11), dtype=torch.float32, device=self._device())
        res = scorer.run(w2cs, out_H=rows, out_H_per_view=True, dL_image=zs)
        return rows, res["vis_count"]

    _DIAG_ORDER = ((0, 3), (3, 4), (7, 11), (4, 7))          # [means | opacity | rot | scale] blocks of the 11 columns

    @torch.enable_grad()
    def estimate_diag_JtJ_simple(self, w2c, K: int = 4, zs=None, fused: bool = True):
        """Returns (diag / K, vis_count), diag flat as [means(3N) | opacity(N) | rot(4N) | scale(3N)].
        `zs` (optional list of K [3,H,W] tensors) replaces the reference's `torch.randn_like(im)` draws."""
        dev = self._device()
        w2c = self._as_w2c(w2c)
        if fused:
            rows, vis = self._probe_rows(w2c.reshape(1, 4, 4).expand(K, 4, 4).contiguous(), self._draw_probes(K, zs))
            g = torch.cat([rows[:, :, a:b].reshape(K, -1) for a, b in self._DIAG_ORDER], dim=1)
            return (g * g).sum(dim=0) / float(K), int(vis[0].item())
        p = self.params
        with torch.no_grad():
            pts = p['means3D']
            pts4 = torch.cat([pts, torch.ones(pts.shape[0], 1, device=dev, dtype=torch.float32)], dim=1)
            transformed_pts = (w2c @ pts4.T).T[:, :3].contiguous()
            rgb_colors = p['rgb_colors']
            rotations = F.normalize(p['unnorm_rotations'])
            opacities = torch.sigmoid(p['logit_opacities'])
            scales = torch.exp(p['log_scales'])
            if scales.shape[-1] == 1:
                scales = torch.tile(scales, (1, 3))
        rendervar = {
            'means3D': transformed_pts.requires_grad_(True),
            'opacities': opacities.detach().clone().requires_grad_(True),
            'rotations': rotations.detach().clone().requires_grad_(True),
            'scales': scales.detach().clone().requires_grad_(True),
            'colors_precomp': rgb_colors.detach(),
            'means2D': torch.zeros_like(transformed_pts, requires_grad=True, device=dev),
        }
        im, radius, _ = Renderer(raster_settings=self.cam, backward_power=2)(**rendervar)
        vis_count = int((radius > 0).sum().item())
        diag_accum = None
        for k in range(K):
            z = torch.randn_like(im) if zs is None else zs[k].to(dev)
            for v in rendervar.values():
                if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None:
                    v.grad.zero_()
            im.backward(gradient=z, retain_graph=(k < K - 1))
            g = torch.cat([rendervar['means3D'].grad.detach().reshape(-1), rendervar['opacities'].grad.detach().reshape(-1),
                           rendervar['rotations'].grad.detach().reshape(-1), rendervar['scales'].grad.detach().reshape(-1)], dim=0)
            contrib = g * g
            diag_accum = contrib if diag_accum is None else diag_accum + contrib
        return diag_accum / float(K), vis_count

    def _diag_batch(self, w2cs, K, chunk_bytes=4 << 30):
        """diag(J^T J) estimates [V, 11N] for V poses, K probes each, in as few launches as memory allows."""
        V = int(w2cs.shape[0])
        N = int(self.params['means3D'].shape[0])
        per = max(1, int(chunk_bytes // (K * N * 44)))
        out = []
        for v0 in range(0, V, per):
            w = w2cs[v0:v0 + per]
            n = int(w.shape[0])
            rows, _ = self._probe_rows(w.repeat_interleave(K, dim=0), self._draw_probes(n * K))
            g = torch.cat([rows[:, :, a:b].reshape(n, K, -1) for a, b in self._DIAG_ORDER], dim=2)
            out.append((g * g).sum(dim=1) / float(K))
        return torch.cat(out)

    def compute_H_train_popgs(self, K: int = 4, fused: bool = True):
        if len(self.keyframe_list) == 0:
            raise RuntimeError("No keyframes available for POP-GS prior.")
        if fused:
            w2cs = torch.stack([self._as_w2c(kf['est_w2c']) for kf in self.keyframe_list])
            return self._diag_batch(w2cs, K).sum(dim=0)
        H = None
        for kf in self.keyframe_list:
            cur, _ = self.estimate_diag_JtJ_simple(kf['est_w2c'], K=K, fused=False)
            H = cur if H is None else H + cur
        return H

    @staticmethod
    def topt_score_from_diags(H_train_diag, JtJ_diag_pi, lam: float = 1e-6):
        """T-opt (to maximise): - sum_j 1 / (H_train_j + JtJ_j + lam)."""
        Hpi = H_train_diag + JtJ_diag_pi + lam
        return -torch.sum(1.0 / torch.clamp(Hpi, min=1e-12))

    @staticmethod
    def dopt_score_from_diags(H_train_diag, JtJ_diag_pi, lam: float = 1e-6):
        """D-opt (to maximise): sum_j log(H_train_j + JtJ_j + lam) - sum_j log(H_train_j + lam)."""
        Hm = H_train_diag + lam
        Hpi = Hm + JtJ_diag_pi
        return torch.sum(torch.log(torch.clamp(Hpi, min=1e-12))) - torch.sum(torch.log(torch.clamp(Hm, min=1e-12)))

    def pose_eval_popgs(self, poses, random_gaussian_params=None, criterion: str = "topt", K: int = 4, lam: float = 1e-6,
                        fused: bool = True):
        if criterion.lower() not in ("topt", "dopt"):
            raise ValueError("criterion must be 'topt' or 'dopt'")
        H_train_diag = self.compute_H_train_popgs(K=K, fused=fused)
        if fused:
            c2w_all = torch.stack([self._as_w2c(c2w) for c2w in poses])
            diags = self._diag_batch(torch.linalg.inv(c2w_all), K)
            fn = self.topt_score_from_diags if criterion.lower() == "topt" else self.dopt_score_from_diags
            return torch.tensor([float(fn(H_train_diag, d, lam=lam)) for d in diags]), c2w_all
        scores, c2ws = [], []
        for c2w in poses:
            c2w = self._as_w2c(c2w)
            cur_diag, _ = self.estimate_diag_JtJ_simple(torch.linalg.inv(c2w), K=K, fused=False)
            if criterion.lower() == "topt":
                s = self.topt_score_from_diags(H_train_diag, cur_diag, lam=lam)
            elif criterion.lower() == "dopt":
                s = self.dopt_score_from_diags(H_train_diag, cur_diag, lam=lam)
            else:
                raise ValueError("criterion must be 'topt' or 'dopt'")
            scores.append(s)
            c2ws.append(c2w)
        return torch.tensor(scores), torch.stack(c2ws)

    # ---- block form (gaussian_object.py:2111-2176, 1572-1585, 1660-1732) ------------------------------------------
    @torch.enable_grad()
    def estimate_block_JtJ(self, w2c, K: int = 2, use_rot: bool = True, use_scale: bool = True, use_opacity: bool = True,
                           zs=None, fused: bool = True):
        """Returns (H_blocks [Nv, d, d] / K, vis_idx [Nv]): per visible splat the outer product of its power-2 gradient
        row [mean3 | opacity | rot4 | scale3] (columns present as the flags say), averaged over K random upstream draws.
        `zs` (optional list of K [3,H,W] tensors) replaces the reference's `torch.randn_like(im)` draws."""
        dev = self._device()
        w2c = self._as_w2c(w2c)
        p = self.params
        with torch.no_grad():
            pts = p['means3D']
            pts4 = torch.cat([pts, torch.ones(pts.shape[0], 1, device=dev, dtype=torch.float32)], dim=1)
            transformed_pts = (w2c @ pts4.T).T[:, :3].contiguous()
            rotations = F.normalize(p['unnorm_rotations'])
            opacities = torch.sigmoid(p['logit_opacities'])
            scales = torch.exp(p['log_scales'])
            if scales.shape[-1] == 1:
                scales = scales.repeat(1, 3)
            colors = p['rgb_colors']
        if fused:
            with torch.no_grad():            # one forward for the visible set (radius > 0), then all probes in one launch
                _, radius, _ = Renderer(raster_settings=self.cam)(means3D=transformed_pts, means2D=torch.zeros_like(transformed_pts),
                                                                  opacities=opacities, colors_precomp=colors, scales=scales, rotations=rotations)
            vis_idx = torch.where(radius > 0)[0]
            rows, _ = self._probe_rows(w2c.reshape(1, 4, 4).expand(int(K), 4, 4).contiguous(), self._draw_probes(int(K), zs))
            cols = [0, 1, 2] + ([3] if use_opacity else []) + ([7, 8, 9, 10] if use_rot else []) + ([4, 5, 6] if use_scale else [])
            Gv = rows[:, vis_idx][:, :, cols]                                  # [K, Nv, d] in the order [mean | opacity | rot | scale]
            return torch.einsum('kvi,kvj->vij', Gv, Gv) / float(K), vis_idx
        rvars = {
            'means3D': transformed_pts.requires_grad_(True),
            'rotations': rotations.detach().clone().requires_grad_(use_rot),
            'scales': scales.detach().clone().requires_grad_(use_scale),
            'opacities': opacities.detach().clone().requires_grad_(use_opacity),
            'colors_precomp': colors.detach(),
            'means2D': torch.zeros_like(transformed_pts, requires_grad=True, device=dev),
        }
        im, radius, _ = Renderer(raster_settings=self.cam, backward_power=2)(**rvars)
        vis_idx = torch.where(radius > 0)[0]
        Nv = vis_idx.numel()

        def rows():
            cols = [rvars['means3D'].grad]
            if use_opacity: cols.append(rvars['opacities'].grad)
            if use_rot: cols.append(rvars['rotations'].grad)
            if use_scale: cols.append(rvars['scales'].grad)
            return torch.cat([c.reshape(c.shape[0], -1) for c in cols], dim=1)

        def zero():
            for v in rvars.values():
                if isinstance(v, torch.Tensor) and v.grad is not None:
                    v.grad.zero_()

        d = 3 + (1 if use_opacity else 0) + (4 if use_rot else 0) + (3 if use_scale else 0)   # the reference finds d with a dummy backward
        H_blocks = torch.zeros((Nv, d, d), device=im.device, dtype=im.dtype)
        for k in range(int(K)):
            z = torch.randn_like(im) if zs is None else zs[k].to(dev)
            zero()
            im.backward(gradient=z, retain_graph=(k < K - 1))
            Gv = rows()[vis_idx, :]
            H_blocks += Gv.unsqueeze(2) * Gv.unsqueeze(1)
        zero()
        return H_blocks / float(K), vis_idx

    def compute_H_train_blocks(self, K: int = 2, **kw):
        """Sum over keyframes, aligned the reference's way (truncate to the smaller visible count, keep the first index set)."""
        Hm, vis_ref = None, None
        for kf in self.keyframe_list:
            Hb, vis_idx = self.estimate_block_JtJ(kf['est_w2c'], K=K, **kw)          # (kw may carry fused=False)
            if Hm is None:
                Hm, vis_ref = Hb, vis_idx
            else:
                Nv = min(Hm.shape[0], Hb.shape[0])
                Hm = Hm[:Nv] + Hb[:Nv]
                vis_ref = vis_ref[:Nv]
        if Hm is None:
            raise RuntimeError("No keyframes available for POP-GS prior (blocks).")
        return Hm, vis_ref

    @staticmethod
    def t_opt_blocks(Hm_blocks, J_blocks, lam=1e-6):
        I = torch.eye(Hm_blocks.shape[-1], device=Hm_blocks.device, dtype=Hm_blocks.dtype)
        invH = torch.linalg.inv(Hm_blocks + J_blocks + lam * I)
        return -torch.einsum('bii->', invH)

    @staticmethod
    def d_opt_blocks(Hm_blocks, J_blocks, lam=1e-6):
        I = torch.eye(Hm_blocks.shape[-1], device=Hm_blocks.device, dtype=Hm_blocks.dtype)
        Hm = Hm_blocks + lam * I
        _, log1 = torch.linalg.slogdet(Hm + J_blocks)
        _, log0 = torch.linalg.slogdet(Hm)
        return (log1 - log0).sum()

    def pose_eval_popgs_blocks(self, poses, random_gaussian_params=None, criterion: str = "topt", K: int = 6, lam: float = 1e-6,
                               use_rot=True, use_scale=True, use_opacity=True):
        kw = dict(use_rot=use_rot, use_scale=use_scale, use_opacity=use_opacity)
        Hm_blocks, train_vis_idx = self.compute_H_train_blocks(K=K, **kw)
        train_np = train_vis_idx.detach().cpu().numpy()
        scores, c2ws = [], []
        for c2w in poses:
            c2w = self._as_w2c(c2w)
            Jb, cur_vis_idx = self.estimate_block_JtJ(torch.linalg.inv(c2w), K=K, **kw)
            _, idx_train, idx_cur = np.intersect1d(train_np, cur_vis_idx.detach().cpu().numpy(), return_indices=True)
            if idx_train.size == 0:
                scores.append(float('-inf')); c2ws.append(c2w)
                continue
            Hb = Hm_blocks[torch.from_numpy(idx_train).to(Hm_blocks.device)]
            J = Jb[torch.from_numpy(idx_cur).to(Jb.device)]
            if criterion.lower() == "topt":
                score = self.t_opt_blocks(Hb, J, lam)
            elif criterion.lower() == "dopt":
                score = self.d_opt_blocks(Hb, J, lam)
            else:
                raise ValueError("criterion must be 'topt' or 'dopt'")
            scores.append(score.item())
            c2ws.append(c2w)
        return torch.tensor(scores), torch.stack(c2ws)

    @classmethod
    def install(cls, target_cls):
        """Graft the 11-column Fisher methods AND the POp-GS estimators onto the reference's GaussianObjectSLAM."""
        target_cls.FISHER_COLUMNS = cls.FISHER_COLUMNS
        super().install(target_cls)
        for name in ("_draw_probes", "_probe_rows", "_diag_batch", "_DIAG_ORDER", "estimate_diag_JtJ_simple",
                     "compute_H_train_popgs", "pose_eval_popgs", "estimate_block_JtJ", "compute_H_train_blocks",
                     "pose_eval_popgs_blocks"):
            setattr(target_cls, name, cls.__dict__[name])
        return target_cls


class GaussianObjectSLAM(ObjectFisherOps, GaussianSLAM):
    pass
