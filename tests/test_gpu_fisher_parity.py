"""-m gpu: the fused multi-view Fisher scorer (fr_fisher_views) and the GaussianSLAM operator surface against the
oracle's restatement of gaussian.py:1338-1375,1503-1570 / gaussian_object.py:1940-2045.
Tolerance (north star): 1e-4 relative on Fisher scores; visibility / tile-instance counts exact."""
import numpy as np
import pytest
import torch

from gpu_util import assert_close, to_dev
from scenes import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def config1(gpu, oracle):
    """BASELINE.json configs[0]: 10k random Gaussians, 8 candidate 256x256 views (+4 keyframes for H_train)."""
    from fisher_rast import synthetic
    from models.SLAM.utils.recon_helpers import setup_camera
    P, V, W, H = 10000, 8, 256, 256
    params = synthetic.room_shell(P, seed=1)
    act = synthetic.activate(params)
    K = synthetic.intrinsics(W, H)
    c2w = synthetic.candidate_poses(V, seed=1)
    kf_c2w = synthetic.candidate_poses(4, seed=101)
    w2c = synthetic.invert_rigid(c2w)
    kf_w2c = synthetic.invert_rigid(kf_c2w)
    cam = setup_camera(W, H, K, np.eye(4), device=gpu)
    ocam = oracle.setup_camera(W, H, K, np.eye(4))
    a = {k: v.numpy() for k, v in act.items()}
    args = (a["means3D"], a["rgb_colors"], a["rotations"], a["opacities"], a["scales"])
    return dict(P=P, V=V, W=W, H=H, params=params, act=act, K=K, c2w=c2w, w2c=w2c, kf_w2c=kf_w2c, cam=cam, ocam=ocam, args=args)


def _scorer(cfg, gpu, columns):
    from fisher_rast.ops import FisherScorer
    act = cfg["act"]
    return FisherScorer(cfg["cam"], *(act[k].to(gpu) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")),
                        columns=columns)


@pytest.mark.parametrize("columns", [4, 11])
def test_per_view_hessian_and_scores(config1, gpu, oracle, columns):
    c = config1
    P, V = c["P"], c["V"]
    sc = _scorer(c, gpu, columns)
    # per-view cur_H, materialised
    cur = torch.zeros((V, P, columns), device=gpu)
    r = sc.run(c["w2c"].to(gpu), out_H=cur, out_H_per_view=True)
    H_train_o = oracle.compute_h_train(c["ocam"], c["kf_w2c"].numpy(), *c["args"], columns=columns)
    Ht = torch.zeros((P, columns), device=gpu)
    sc.run(c["kf_w2c"].to(gpu), out_H=Ht)
    assert_close(Ht.cpu().numpy(), H_train_o, 1e-4, "H_train", atol_frac=1e-7)
    H_inv = torch.reciprocal(Ht + 0.1)
    s = sc.run(c["w2c"].to(gpu), H_inv=H_inv)
    torch.cuda.synchronize()
    want_scores, want_vis = oracle.pose_eval(c["ocam"], c["w2c"].numpy(), H_train_o, *c["args"], columns=columns)
    assert np.array_equal(s["vis_count"].cpu().numpy(), want_vis)
    assert np.array_equal(r["vis_count"].cpu().numpy(), want_vis)
    assert rel_err(s["scores"].cpu().numpy(), want_scores) < 1e-4
    for v in range(V):
        H_o, vis, fwd, _ = oracle.compute_hessian(c["ocam"], c["w2c"][v].numpy(), *c["args"], columns=columns, return_all=True)
        assert_close(cur[v].cpu().numpy(), H_o, 1e-4, f"cur_H[{v}]", atol_frac=1e-7)
        assert int(r["num_rendered"][v]) == fwd["num_rendered"]
    # score == sum(cur_H * H_inv) of the materialised tensors (gaussian.py:1367)
    s2 = (cur.double() * H_inv.double()[None]).sum(dim=(1, 2)).cpu().numpy()
    assert rel_err(s["scores"].cpu().numpy(), s2) < 2e-5


def test_reference_style_loop_equals_fused(config1, gpu):
    """The reference's calling pattern (gaussian.py:1536-1556: autograd GaussianRasterizer with backward_power=2, one view
    at a time) and the fused scorer give the same cur_H."""
    from diff_gaussian_rasterization import GaussianRasterizer
    c = config1
    act = {k: v.to(gpu) for k, v in c["act"].items()}
    sc = _scorer(c, gpu, 4)
    for v in (0, 3):
        w2c = c["w2c"][v].to(gpu)
        pts4 = torch.cat((act["means3D"], torch.ones_like(act["means3D"][:, :1])), dim=1)
        tp = (w2c @ pts4.T).T[:, :3].contiguous()
        rv = {'means3D': tp.requires_grad_(True), 'colors_precomp': act["rgb_colors"].clone().requires_grad_(True),
              'rotations': act["rotations"].clone().requires_grad_(True), 'opacities': act["opacities"].clone().requires_grad_(True),
              'scales': act["scales"].clone().requires_grad_(True),
              'means2D': torch.zeros_like(tp, requires_grad=True, device=gpu) + 0}
        rv['means2D'].retain_grad()
        im, radius, _ = GaussianRasterizer(raster_settings=c["cam"], backward_power=2)(**rv)
        im.backward(gradient=torch.ones_like(im) * 1e-3)
        ref_H = torch.cat([tp.grad.reshape(c["P"], -1), rv['opacities'].grad.reshape(c["P"], -1)], dim=1)
        cur = torch.zeros((c["P"], 4), device=gpu)
        r = sc.run(w2c.reshape(1, 4, 4), out_H=cur)
        assert int((radius > 0).sum()) == int(r["vis_count"][0])
        # the torch matmul above and the in-kernel transform may round the camera-frame means differently (ulps)
        assert_close(cur.cpu().numpy(), ref_H.cpu().numpy(), 2e-3, f"view{v}", atol_frac=1e-5)
        assert rel_err(cur.sum().item(), ref_H.sum().item()) < 1e-4
        assert rv['means2D'].grad.shape == (c["P"], 3) and float(rv['means2D'].grad[:, 2].abs().max()) == 0.0


@pytest.mark.parametrize("cls_name,columns", [("GaussianSLAM", 4), ("GaussianObjectSLAM", 11)])
def test_slam_operator_surface(config1, gpu, oracle, cls_name, columns):
    import models.gaussian_slam as mgs
    c = config1
    cls = getattr(mgs, cls_name)
    slam = cls(params={k: v.clone() for k, v in c["params"].items()}, intrinsics=c["K"], width=c["W"], height=c["H"], device=gpu)
    for w in c["kf_w2c"]:
        slam.add_keyframe(w.numpy())          # est_w2c may be numpy (gaussian.py:1518-1519)
    H_train = slam.compute_H_train()
    H_train_o = oracle.compute_h_train(c["ocam"], c["kf_w2c"].numpy(), *c["args"], columns=columns)
    assert H_train.shape == (c["P"], columns)
    assert_close(H_train.cpu().numpy(), H_train_o, 1e-4, "H_train", atol_frac=1e-7)
    poses = [p.to(gpu) for p in c["c2w"]]
    scores, c2ws = slam.pose_eval(poses, random_gaussian_params=None)
    assert scores.device.type == "cpu" and scores.dtype == torch.float32 and scores.shape == (c["V"],)
    assert c2ws.shape == (c["V"], 4, 4)
    # the surface inverts c2w on the device like the reference; feed the oracle the same w2c
    w2c_dev = torch.linalg.inv(torch.stack(poses)).cpu().numpy()
    want, _ = oracle.pose_eval(c["ocam"], w2c_dev, H_train_o, *c["args"], columns=columns)
    assert rel_err(scores.numpy(), want) < 1e-4
    # compute_Hessian return conventions (gaussian.py:1555-1570 / gaussian_object.py:2029-2045)
    H_pts = slam.compute_Hessian(c["w2c"][0].numpy(), return_points=True)
    H_flat = slam.compute_Hessian(c["w2c"][0].to(gpu), return_points=False)
    assert H_pts.shape == (c["P"], columns) and H_flat.shape == (c["P"] * columns,)
    assert torch.equal(H_flat[:3 * c["P"]].reshape(c["P"], 3), H_pts[:, :3]) or rel_err(H_flat[:3 * c["P"]].reshape(c["P"], 3).cpu().numpy(), H_pts[:, :3].cpu().numpy()) < 1e-5
    ret = slam.compute_Hessian(c["w2c"][0].numpy(), return_points=True, return_pose=True)
    assert torch.equal(ret[1], torch.eye(6, device=gpu)) and len(ret) == (3 if columns == 11 else 2)
    assert slam.gs_pts_cnt() == 1 and slam.gaussian_points is slam.params['means3D'] and slam.cur_frame_idx == 0
    assert slam.pause() is None and slam.stop() is None


def test_render_at_pose_matches_oracle(config1, gpu, oracle):
    import models.gaussian_slam as mgs
    c = config1
    slam = mgs.GaussianSLAM(params=c["params"], intrinsics=c["K"], width=c["W"], height=c["H"], device=gpu)
    out = slam.render_at_pose(c["c2w"][2].to(gpu))
    assert out["render"].shape == (3, c["H"], c["W"]) and out["depth"].shape == (1, c["H"], c["W"])
    # feed the oracle exactly what the surface fed the rasteriser: device-side transform and activations
    from models.SLAM.utils.slam_helpers import transformed_params2rendervar, transformed_params2depthplussilhouette
    rel_w2c = torch.linalg.inv(c["c2w"][2].to(gpu))
    pts = slam.params["means3D"]
    tp = (rel_w2c @ torch.cat((pts, torch.ones_like(pts[:, :1])), 1).T).T[:, :3]
    rv = transformed_params2rendervar(slam.params, tp)
    n = {k: v.detach().cpu().numpy() for k, v in rv.items()}
    fw = oracle.rasterize_forward(c["ocam"], n["means3D"], n["opacities"], colors_precomp=n["colors_precomp"],
                                  scales=n["scales"], rotations=n["rotations"])
    assert np.array_equal(out["render"].detach().cpu().numpy().view(np.uint32), fw["color"].view(np.uint32))
    dv = transformed_params2depthplussilhouette(slam.params, slam.first_frame_w2c, tp)
    fd = oracle.rasterize_forward(c["ocam"], n["means3D"], n["opacities"], colors_precomp=dv["colors_precomp"].cpu().numpy(),
                                  scales=n["scales"], rotations=n["rotations"])
    assert np.array_equal(out["depth"].detach().cpu().numpy()[0].view(np.uint32), fd["color"][0].view(np.uint32))


def test_overflow_regrow_and_chunking(config1, gpu):
    c = config1
    sc = _scorer(c, gpu, 4)
    Ht = torch.zeros((c["P"], 4), device=gpu)
    sc.run(c["kf_w2c"].to(gpu), out_H=Ht)
    H_inv = torch.reciprocal(Ht + 0.1)
    base = sc.run(c["w2c"].to(gpu), H_inv=H_inv)["scores"].cpu()
    small = _scorer(c, gpu, 4)
    small.per_view_capacity = 64            # forces the device-side overflow flag and a re-run
    small.WORKSPACE_BUDGET = 3 * (c["P"] * 36 + 64 * 8)   # and view chunking
    again = small.run(c["w2c"].to(gpu), H_inv=H_inv)
    assert torch.equal(again["scores"].cpu(), base)
    Ht2 = torch.zeros((c["P"], 4), device=gpu)
    small2 = _scorer(c, gpu, 4)
    small2.per_view_capacity = 64
    small2.run(c["kf_w2c"].to(gpu), out_H=Ht2)
    assert rel_err(Ht2.cpu().numpy(), Ht.cpu().numpy()) < 1e-5   # atomics: order differs, values agree


def test_scorer_is_reused_until_the_map_changes(config1, gpu):
    """compute_Hessian is called once per path step (tester 1684-1705) on an unchanged map: the packed inputs and the
    workspace are built once; any in-place update or replacement of a parameter tensor rebuilds them."""
    import models.gaussian_slam as mgs
    c = config1
    slam = mgs.GaussianSLAM(params={k: v.clone() for k, v in c["params"].items()}, intrinsics=c["K"], width=c["W"], height=c["H"], device=gpu)
    for w in c["kf_w2c"][:2]:
        slam.add_keyframe(w)
    a = slam._scorer()
    h0 = slam.compute_Hessian(c["w2c"][0], return_points=True)
    slam.pose_eval([p.to(gpu) for p in c["c2w"][:2]])
    assert slam._scorer() is a
    slam.params["logit_opacities"].mul_(0.5)                       # what an optimiser step does
    b = slam._scorer()
    assert b is not a
    h1 = slam.compute_Hessian(c["w2c"][0], return_points=True)
    assert not torch.equal(h0, h1)
    slam.params["means3D"] = slam.params["means3D"].clone()        # what densification / pruning does
    assert slam._scorer() is not b


def test_pose_eval_keeps_h_train_until_map_or_keyframes_change(config1, gpu):
    """pose_eval recomputes H_train over the keyframes in the reference on every call (gaussian.py:1354-1375); here 1 / (H_train + reg)
    is kept while the map and the keyframe poses are what they were: the repeated call must give the same scores, and a new keyframe,
    an in-place change of a keyframe pose or of the map must be seen."""
    import models.gaussian_slam as mgs
    c = config1
    slam = mgs.GaussianSLAM(params={k: v.clone() for k, v in c["params"].items()}, intrinsics=c["K"], width=c["W"], height=c["H"], device=gpu)
    for w in c["kf_w2c"][:2]:
        slam.add_keyframe(w.clone())
    poses = [p.to(gpu) for p in c["c2w"][:4]]
    s0, _ = slam.pose_eval(poses)
    assert slam._h_inv_cache is not None
    held = slam._h_inv_cache[2]
    s1, _ = slam.pose_eval(poses)
    assert slam._h_inv_cache[2] is held and torch.equal(s0, s1)                 # reused, same scores
    # the uncached route gives the same numbers (H_train sums float atomics: equal to rounding)
    slam.CACHE_H_TRAIN = False
    s2, _ = slam.pose_eval(poses)
    slam.CACHE_H_TRAIN = True
    assert torch.allclose(s2, s0, rtol=1e-5)
    slam.add_keyframe(c["kf_w2c"][2].clone())                                     # a new keyframe
    s3, _ = slam.pose_eval(poses)
    assert slam._h_inv_cache[2] is not held and not torch.allclose(s3, s0, rtol=1e-3)
    held = slam._h_inv_cache[2]
    slam.keyframe_list[0]['est_w2c'][0, 3] += 0.25                                # a keyframe pose refined in place
    s4, _ = slam.pose_eval(poses)
    assert slam._h_inv_cache[2] is not held and not torch.equal(s4, s3)
    held = slam._h_inv_cache[2]
    slam.params["logit_opacities"].mul_(0.9)                                      # an optimiser step on the map
    s5, _ = slam.pose_eval(poses)
    assert slam._h_inv_cache[2] is not held and not torch.equal(s5, s4)
    # and against the two-step route of the reference's structure
    Ht = slam.compute_H_train()
    want = slam._scorer().run(torch.stack(poses), H_inv=torch.reciprocal(Ht + slam.H_TRAIN_REG), poses_are_c2w=True)["scores"].cpu()
    assert torch.allclose(s5, want, rtol=1e-5)


def test_per_view_weights_path_eval(config1, gpu):
    """H_inv_view_stride != 0: each view has its own weights (the planner's path evaluation, tester 1688-1705)."""
    c = config1
    sc = _scorer(c, gpu, 4)
    V, P = 4, c["P"]
    g = torch.Generator().manual_seed(0)
    Hi = torch.rand((V, P, 4), generator=g).to(gpu)
    got = sc.run(c["w2c"][:V].to(gpu), H_inv=Hi, H_inv_per_view=True)["scores"].cpu().numpy()
    for v in range(V):
        one = sc.run(c["w2c"][v:v + 1].to(gpu), H_inv=Hi[v])["scores"].cpu().numpy()
        assert one[0] == got[v]


def _autograd_probe_rows(slam, w2c, zs):
    """The reference's route for one pose (gaussian_object.py:2066-2098): one forward of the drop-in autograd rasteriser with
    backward_power=2, then one backward per upstream draw.  Returns ([K] rows [N,11] ordered [mean|opacity|rot|scale], radius)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from models.SLAM.utils.slam_helpers import transformed_params2rendervar
    pts = slam.params["means3D"]
    tp = (w2c @ torch.cat((pts, torch.ones_like(pts[:, :1])), 1).T).T[:, :3].contiguous()
    rv = {k: (v.detach().clone().requires_grad_(True) if k != "means2D" else v) for k, v in transformed_params2rendervar(slam.params, tp).items()}
    im, radius, _ = GaussianRasterizer(raster_settings=slam.cam, backward_power=2)(**rv)
    out = []
    for k, z in enumerate(zs):
        for t in rv.values():
            if t.is_leaf:
                t.grad = None
        im.backward(gradient=z.to(im.device), retain_graph=k + 1 < len(zs))
        out.append(torch.cat([rv["means3D"].grad, rv["opacities"].grad.reshape(-1, 1), rv["rotations"].grad, rv["scales"].grad], 1).clone())
    return out, radius


def test_popgs_diag_estimator(config1, gpu, oracle):
    """estimate_diag_JtJ_simple (gaussian_object.py:2049-2109) with the random upstream gradients supplied:
    diag = mean_k (power-2 gradient under z_k)^2, layout [means | opacity | rot | scale]."""
    import models.gaussian_slam as mgs
    c = config1
    slam = mgs.GaussianObjectSLAM(params=c["params"], intrinsics=c["K"], width=c["W"], height=c["H"], device=gpu)
    K = 2
    g = torch.Generator().manual_seed(5)
    zs = [torch.randn((3, c["H"], c["W"]), generator=g) for _ in range(K)]
    w2c = torch.linalg.inv(c["c2w"][1].to(gpu))
    diag, vis = slam.estimate_diag_JtJ_simple(w2c, K=K, zs=zs)                  # fused: all probes in one fr_fisher_views launch
    rows_ag, radius = _autograd_probe_rows(slam, w2c, zs)                        # the reference's autograd route
    sq = sum(r * r for r in rows_ag) / K
    diag_ag = torch.cat([sq[:, 0:3].reshape(-1), sq[:, 3:4].reshape(-1), sq[:, 4:8].reshape(-1), sq[:, 8:11].reshape(-1)])
    vis_ag = int((radius > 0).sum())
    P = c["P"]
    assert diag.shape == (P * 11,) and vis == vis_ag
    assert_close(diag.cpu().numpy(), diag_ag.double().cpu().numpy(), 2e-4, "diag_JtJ fused vs autograd", atol_frac=1e-7)
    # oracle on the very same device-side render variables
    from models.SLAM.utils.slam_helpers import transformed_params2rendervar
    pts = slam.params["means3D"]
    tp = (w2c @ torch.cat((pts, torch.ones_like(pts[:, :1])), 1).T).T[:, :3]
    n = {k: v.detach().cpu().numpy() for k, v in transformed_params2rendervar(slam.params, tp).items()}
    fw = oracle.rasterize_forward(c["ocam"], n["means3D"], n["opacities"], colors_precomp=n["colors_precomp"], scales=n["scales"], rotations=n["rotations"])
    acc = 0.0
    for z in zs:
        gz = oracle.rasterize_backward(c["ocam"], fw, z.numpy(), 2)
        gcat = np.concatenate([gz["dL_dmeans3D"].reshape(-1), gz["dL_dopacity"].reshape(-1), gz["dL_drotations"].reshape(-1), gz["dL_dscales"].reshape(-1)]).astype(np.float64)
        acc = acc + gcat ** 2
    want = acc / K
    assert vis == int((fw["radii"] > 0).sum())
    assert_close(diag.cpu().numpy(), want, 2e-4, "diag_JtJ", atol_frac=1e-7)
    for kf in c["kf_w2c"][:2]:
        slam.add_keyframe(kf)
    Ht = slam.compute_H_train_popgs(K=1)
    t = slam.topt_score_from_diags(Ht, diag, lam=1e-6)
    d = slam.dopt_score_from_diags(Ht, diag, lam=1e-6)
    assert torch.isfinite(t) and torch.isfinite(d) and float(d) >= 0.0 and float(t) < 0.0
    scores, c2ws = slam.pose_eval_popgs([p.to(gpu) for p in c["c2w"][:2]], criterion="dopt", K=1)
    assert scores.shape == (2,) and c2ws.shape == (2, 4, 4)
    # same draws through both routes: seed the generator the probes come from
    torch.manual_seed(11); s_f, _ = slam.pose_eval_popgs([p.to(gpu) for p in c["c2w"][:2]], criterion="topt", K=2, lam=1e-3)
    assert bool(torch.isfinite(s_f).all()) and bool((s_f < 0).all())


def test_popgs_block_estimator(config1, gpu, oracle):
    """estimate_block_JtJ (gaussian_object.py:2111-2176): per visible splat the outer product of its power-2 gradient row
    [mean3 | opacity | rot4 | scale3] under supplied upstream draws; then the block T-/D-opt scores (1660-1732)."""
    import models.gaussian_slam as mgs
    c = config1
    slam = mgs.GaussianObjectSLAM(params=c["params"], intrinsics=c["K"], width=c["W"], height=c["H"], device=gpu)
    K = 2
    g = torch.Generator().manual_seed(6)
    zs = [torch.randn((3, c["H"], c["W"]), generator=g) for _ in range(K)]
    w2c = torch.linalg.inv(c["c2w"][2].to(gpu))
    Hb, vis_idx = slam.estimate_block_JtJ(w2c, K=K, zs=zs)                       # fused route
    rows_ag, radius = _autograd_probe_rows(slam, w2c, zs)                         # autograd route
    vis_ag = torch.nonzero(radius > 0).reshape(-1)
    Hb_ag = sum(r[vis_ag].unsqueeze(2) * r[vis_ag].unsqueeze(1) for r in rows_ag) / K
    assert torch.equal(vis_idx, vis_ag)
    assert_close(Hb.cpu().numpy(), Hb_ag.double().cpu().numpy(), 4e-4, "block_JtJ fused vs autograd", atol_frac=1e-7)
    from models.SLAM.utils.slam_helpers import transformed_params2rendervar
    pts = slam.params["means3D"]
    tp = (w2c @ torch.cat((pts, torch.ones_like(pts[:, :1])), 1).T).T[:, :3]
    n = {k: v.detach().cpu().numpy() for k, v in transformed_params2rendervar(slam.params, tp).items()}
    fw = oracle.rasterize_forward(c["ocam"], n["means3D"], n["opacities"], colors_precomp=n["colors_precomp"], scales=n["scales"], rotations=n["rotations"])
    vis = np.where(fw["radii"] > 0)[0]
    assert np.array_equal(vis_idx.cpu().numpy(), vis)
    want = np.zeros((vis.size, 11, 11))
    for z in zs:
        gz = oracle.rasterize_backward(c["ocam"], fw, z.numpy(), 2)
        G = np.concatenate([gz["dL_dmeans3D"], gz["dL_dopacity"].reshape(-1, 1), gz["dL_drotations"], gz["dL_dscales"]], 1).astype(np.float64)[vis]
        want += G[:, :, None] * G[:, None, :]
    want /= K
    assert Hb.shape == (vis.size, 11, 11)
    assert_close(Hb.cpu().numpy(), want, 4e-4, "block_JtJ", atol_frac=1e-7)
    # column subsets keep the reference's order [mean | opacity | rot | scale]
    Hs, _ = slam.estimate_block_JtJ(w2c, K=K, zs=zs, use_rot=False)
    keep = [0, 1, 2, 3, 8, 9, 10]
    assert_close(Hs.cpu().numpy(), want[:, keep][:, :, keep], 4e-4, "block_JtJ(no rot)", atol_frac=1e-7)
    # block criteria against float64 numpy on the same blocks
    for kf in c["kf_w2c"][:2]:
        slam.add_keyframe(kf)
    Hm, tv = slam.compute_H_train_blocks(K=1)
    assert Hm.shape[1:] == (11, 11) and tv.shape[0] == Hm.shape[0]
    lam = 1e-3
    Hm64, J64 = Hm[:64].double().cpu().numpy(), Hb[:64].double().cpu().numpy()
    I = np.eye(11)
    t_want = -np.trace(np.linalg.inv(Hm64 + J64 + lam * I), axis1=1, axis2=2).sum()
    d_want = (np.linalg.slogdet(Hm64 + lam * I + J64)[1] - np.linalg.slogdet(Hm64 + lam * I)[1]).sum()
    t_got = float(slam.t_opt_blocks(Hm[:64].double(), Hb[:64].double(), lam))
    d_got = float(slam.d_opt_blocks(Hm[:64].double(), Hb[:64].double(), lam))
    assert abs(t_got - t_want) <= 1e-5 * abs(t_want) and abs(d_got - d_want) <= 1e-6 * max(1.0, abs(d_want))   # ill-conditioned blocks, two LAPACKs
    scores, c2ws = slam.pose_eval_popgs_blocks([p.to(gpu) for p in c["c2w"][:2]], criterion="dopt", K=1, lam=1e-3)
    assert scores.shape == (2,) and c2ws.shape == (2, 4, 4) and bool(torch.isfinite(scores).all())


def _crowded_scene(P, seed):
    """P tiny, faint splats that all project into the 16x16 tile at the image centre of a 48x48 view."""
    rng = np.random.default_rng(seed)
    z = rng.uniform(2.0, 6.0, P).astype(np.float32)
    # pixel = 24 * x / z + 23.5 ; tile (1,1) spans pixels 16..31
    u = rng.uniform(16.5, 30.5, P); v = rng.uniform(16.5, 30.5, P)
    means = np.stack([(u - 23.5) / 24.0 * z, (v - 23.5) / 24.0 * z, z], 1).astype(np.float32)
    scales = np.full((P, 3), 0.004, np.float32) * z[:, None]
    rot = np.tile(np.array([[1, 0, 0, 0]], np.float32), (P, 1))
    op = rng.uniform(0.005, 0.012, P).astype(np.float32)
    col = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    return means, col, rot, op, scales


@pytest.mark.parametrize("P,expect", [(70_000, "tile_over_65535"), (30_000, "strip_list_over_3840")])
@pytest.mark.parametrize("columns", [4, 11])
def test_fallback_paths_for_crowded_tiles(gpu, oracle, P, expect, columns):
    """Tiles whose lists do not fit k_fisher_tile_v2's LDS index (more than 65535 splats in a tile, or more than 3840
    contributing splats per 16x4 strip) are flagged and redone by the scan kernel: same results."""
    from fisher_rast.ops import FisherScorer
    from models.SLAM.utils.recon_helpers import setup_camera
    from fisher_rast import synthetic
    W = H = 48
    K = synthetic.intrinsics(W, H)
    means, col, rot, op, sc = _crowded_scene(P, 3)
    cam = setup_camera(W, H, K, np.eye(4), device=gpu)
    ocam = oracle.setup_camera(W, H, K, np.eye(4))
    w2c = np.eye(4, dtype=np.float32)[None]
    H_o, vis, fwd, g = oracle.compute_hessian(ocam, w2c[0], means, col, rot, op, sc, columns=columns, return_all=True)
    counts = fwd["ranges"][:, 1] - fwd["ranges"][:, 0]
    if expect == "tile_over_65535":
        assert counts.max() > 65535
    else:
        assert 3840 * 4 < counts.max() <= 65535 and fwd["n_contrib"].max() > 300
    t = [torch.from_numpy(a).to(gpu) for a in (means, col, rot, op, sc)]
    scorer = FisherScorer(cam, *t, columns=columns)
    cur = torch.zeros((P, columns), device=gpu)
    r = scorer.run(torch.from_numpy(w2c).to(gpu), out_H=cur)
    assert int(r["vis_count"][0]) == vis and int(r["num_rendered"][0]) == fwd["num_rendered"]
    assert_close(cur.cpu().numpy(), H_o, 1e-4, "cur_H", atol_frac=1e-7)
    Hi = torch.rand((P, columns), generator=torch.Generator().manual_seed(1)).to(gpu)
    s = scorer.run(torch.from_numpy(w2c).to(gpu), H_inv=Hi)["scores"].cpu().numpy()
    want = float((H_o.astype(np.float64) * Hi.cpu().double().numpy()).sum())
    assert abs(s[0] - want) <= 1e-4 * abs(want)
    # the single-view gradient path has the same fallback (k_backward_lin_tile -> k_backward_tile, u only)
    from gpu_util import hip_forward, hip_backward
    got = hip_forward(gpu, ocam, means, op, colors_precomp=col, scales=sc, rotations=rot)
    assert np.array_equal(got["point_list"], fwd["point_list"])
    dL = np.random.default_rng(2).normal(size=(3, H, W)).astype(np.float32)
    gw = oracle.rasterize_backward(ocam, fwd, dL, 1)
    gg = hip_backward(gpu, ocam, got, dL, 1)
    for n in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dcolors"):
        assert_close(gg[n], gw[n], 1e-4, n, atol_frac=2e-5)


def test_more_tiles_than_the_lds_histogram_holds(gpu, oracle):
    """T = 65 x 65 = 4225 tiles > FR_MAX_LDS_TILES: binning falls back to global counters."""
    from fisher_rast import synthetic
    from gpu_util import hip_forward
    W = H = 1040
    P = 3000
    act = {k: v.numpy() for k, v in synthetic.activate(synthetic.room_shell(P, seed=8)).items()}
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(1, seed=8))[0].numpy()
    tp = oracle.transform_points(w2c, act["means3D"])
    cam = oracle.setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4))
    kw = dict(colors_precomp=act["rgb_colors"], scales=act["scales"], rotations=act["rotations"])
    want = oracle.rasterize_forward(cam, tp, act["opacities"], **kw)
    got = hip_forward(gpu, cam, tp, act["opacities"], **kw)
    assert got["num_rendered"] == want["num_rendered"] and np.array_equal(got["ranges"], want["ranges"])
    assert np.array_equal(got["point_list"], want["point_list"])
    assert np.array_equal(np.ascontiguousarray(got["color"]).view(np.uint32), want["color"].view(np.uint32))
    # and the batched scorer on the same image size
    from fisher_rast.ops import FisherScorer
    from models.SLAM.utils.recon_helpers import setup_camera
    gcam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=gpu)
    t = {k: torch.from_numpy(v).to(gpu) for k, v in act.items()}
    sc = FisherScorer(gcam, t["means3D"], t["rgb_colors"], t["rotations"], t["opacities"], t["scales"])
    cur = torch.zeros((P, 4), device=gpu)
    sc.run(torch.from_numpy(w2c[None]).to(gpu), out_H=cur)
    H_o, _ = oracle.compute_hessian(cam, w2c, act["means3D"], act["rgb_colors"], act["rotations"], act["opacities"], act["scales"])
    assert_close(cur.cpu().numpy(), H_o, 1e-4, "cur_H", atol_frac=1e-7)
    # score-only mode on this image size: the scorer's records come from k_fisher_records (visibility from radii, on the side
    # stream beside the sorts, after the scatter has read the depths it rewrites) instead of phase C of the multi-view front end
    Hi = torch.rand((P, 4), generator=torch.Generator().manual_seed(1)) * 2.0 + 0.1
    got = sc.run(torch.from_numpy(np.stack([w2c, w2c])).to(gpu), H_inv=Hi.to(gpu))["scores"].cpu().numpy()
    want_s = float((H_o.astype(np.float64) * Hi.numpy().astype(np.float64)).sum())
    assert got[0] == got[1] and abs(got[0] - want_s) <= 1e-4 * abs(want_s)


@pytest.mark.parametrize("P,V,W,H,columns,seed", [
    (2, 1, 32, 32, 4, 40),            # two Gaussians, one view
    (63, 3, 50, 70, 4, 41),           # fewer Gaussians than a wave, ragged image, V < views per preprocess workgroup
    (300, 9, 72, 40, 11, 42),         # V = 8 + 1: a second, partial view chunk
    (777, 17, 48, 48, 4, 43),         # V odd: plain tile mapping (V % 8 != 0)
    (2500, 8, 96, 64, 11, 44),
    (1800, 5, 33, 17, 4, 45),         # image smaller than two tiles in one direction
])
def test_odd_shapes_against_oracle(gpu, oracle, P, V, W, H, columns, seed):
    """Shapes around the decomposition boundaries of the multi-view front end (256 Gaussians x 8 views per workgroup),
    the XCD-aware tile mapping (V % 8) and the 16 x 16 tiling."""
    from fisher_rast import synthetic
    from fisher_rast.ops import FisherScorer
    from models.SLAM.utils.recon_helpers import setup_camera
    act = synthetic.activate(synthetic.room_shell(P, seed))
    K = synthetic.intrinsics(W, H)
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, seed))
    kf = synthetic.invert_rigid(synthetic.candidate_poses(2, seed + 100))
    cam = setup_camera(W, H, K, np.eye(4), device=gpu)
    ocam = oracle.setup_camera(W, H, K, np.eye(4))
    a = {k: v.numpy() for k, v in act.items()}
    args = (a["means3D"], a["rgb_colors"], a["rotations"], a["opacities"], a["scales"])
    sc = FisherScorer(cam, *(act[k].to(gpu) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")), columns=columns)
    Ht = torch.zeros((P, columns), device=gpu)
    sc.run(kf.to(gpu), out_H=Ht)
    H_train_o = oracle.compute_h_train(ocam, kf.numpy(), *args, columns=columns)
    assert_close(Ht.cpu().numpy(), H_train_o, 1e-4, "H_train", atol_frac=1e-7)
    r = sc.run(w2c.to(gpu), H_inv=torch.reciprocal(Ht + 0.1))
    want, vis = oracle.pose_eval(ocam, w2c.numpy(), H_train_o, *args, columns=columns)
    assert np.array_equal(r["vis_count"].cpu().numpy(), vis)
    got = r["scores"].cpu().numpy()
    assert np.all(np.abs(got - want) <= 1e-4 * np.maximum(np.abs(want), 1e-30) + 1e-12 * np.abs(want).max()), (got, want)
    for v in range(min(V, 3)):
        _, _, fwd, _ = oracle.compute_hessian(ocam, w2c[v].numpy(), *args, columns=columns, return_all=True)
        assert int(r["num_rendered"][v]) == fwd["num_rendered"]


def test_object_variant_with_appended_random_gaussians(config1, gpu, oracle):
    """GaussianObjectSLAM.compute_Hessian / pose_eval with `random_gaussian_params` (gaussian_object.py:1971-1992): the extra
    Gaussians are appended with colour 0.5, their rotations un-normalised and their opacity taken as given."""
    import models.gaussian_slam as mgs
    c = config1
    slam = mgs.GaussianObjectSLAM(params=c["params"], intrinsics=c["K"], width=c["W"], height=c["H"], device=gpu)
    g = torch.Generator().manual_seed(17)
    Nr = 1500
    rg = dict(means3D=(torch.rand((Nr, 3), generator=g) - 0.5) * torch.tensor([8.0, 2.0, 8.0]),
              rotations=torch.randn((Nr, 4), generator=g) * 1.3,                 # deliberately not unit length
              opacity=torch.rand((Nr, 1), generator=g) * 0.8 + 0.1,
              scales=torch.rand((Nr, 3), generator=g) * 0.05 + 0.01)
    w2c = torch.linalg.inv(c["c2w"][3])
    cur_H, pose_H, vis = slam.compute_Hessian(w2c.to(gpu), return_points=True, random_gaussian_params=rg, return_pose=True)
    P = c["P"]
    assert cur_H.shape == (P + Nr, 11) and pose_H.shape == (6, 6)
    a = {k: v.numpy() for k, v in c["act"].items()}
    means = np.concatenate([a["means3D"], rg["means3D"].numpy()]).astype(np.float32)
    cols = np.concatenate([a["rgb_colors"], np.full((Nr, 3), 0.5, np.float32)]).astype(np.float32)
    rots = np.concatenate([a["rotations"], rg["rotations"].numpy()]).astype(np.float32)
    ops = np.concatenate([a["opacities"].reshape(-1, 1), rg["opacity"].numpy()]).astype(np.float32)
    scs = np.concatenate([a["scales"], rg["scales"].numpy()]).astype(np.float32)
    H_o, vis_o, _, _ = oracle.compute_hessian(c["ocam"], w2c.numpy(), means, cols, rots, ops, scs, columns=11, return_all=True)
    assert vis == vis_o
    # un-normalised quaternions make the rotation rows a difference of large terms: the floor is 1e-5 of the largest entry here
    assert_close(cur_H.cpu().numpy(), H_o, 1e-4, "cur_H with random Gaussians", atol_frac=1e-5)
    assert float(cur_H[P:].abs().sum()) > 0                                          # the appended ones do receive information
    for kf in c["kf_w2c"][:2]:
        slam.add_keyframe(kf)
    scores, c2ws = slam.pose_eval([p.to(gpu) for p in c["c2w"][:3]], random_gaussian_params=rg)
    H_train_o = oracle.compute_h_train(c["ocam"], c["kf_w2c"][:2].numpy(), means, cols, rots, ops, scs, columns=11)
    want, _ = oracle.pose_eval(c["ocam"], c["w2c"][:3].numpy(), H_train_o, means, cols, rots, ops, scs, columns=11)
    assert rel_err(scores.numpy(), want) < 1e-4 and c2ws.shape == (3, 4, 4)
