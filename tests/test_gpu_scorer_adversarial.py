"""-m gpu: the headline scorer kernels (fr_fisher_views: score-only single pass, out_H two-pass, per-view H_inv; 4 and 11
Fisher columns) on the adversarial scene families of test_gpu_rasterizer_parity.py -- near-plane giants, image sizes that
are not multiples of 16, one tile with thousands of splats, depth ties, alpha saturating at 0.99 with T < 1e-4 kills --
and on a scene built to SIT ON the thresholds of forward.cu:347-363 (alpha within 1e-6 of 1/255, T within rounding of
1e-4, power == 0), against the oracle's compute_Hessian / pose_eval (gaussian.py:1338-1375, 1503-1570;
gaussian_object.py:1940-2045).  Tolerance (north star): 1e-4 relative on the scores -- every family, no exception -- and on the
entries of cur_H; an entry's tolerance is widened only by a multiple of what the reference's OWN binary32 chain loses on that
Gaussian, measured against the arbiter (the oracle's statements in binary64 on the same contributor sets, oracle/ref.py;
tests/test_arbiter_cpu.py): needle-shaped and near-plane splats, where the reference itself is off by up to several per cent."""
import numpy as np
import pytest
import torch

from scenes import random_scene, intrinsics, rel_err
from test_gpu_rasterizer_parity import _scene, CASES

pytestmark = pytest.mark.gpu


def _views(base_w2c, n):
    """n candidate poses around the family's pose: the first is the pose itself, the others yaw / shift it a little."""
    out = []
    for k in range(n):
        yaw, t = 0.07 * k, np.array([0.05 * k, -0.02 * k, 0.03 * k], np.float32)
        c, s = np.cos(yaw), np.sin(yaw)
        d = np.eye(4, dtype=np.float32)
        d[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
        d[:3, 3] = t
        out.append((d @ base_w2c).astype(np.float32))
    return np.stack(out)


def threshold_scene(oracle, W=96, H=96, seed=7):
    """A normal population plus splats tuned to the reference's thresholds.
    (a) 150 small splats whose opacity is set so that alpha at one pixel next to their centre is (1/255)(1 + d),
        |d| <= 1e-6 -- the `alpha < 1/255` test of forward.cu:351 decides on the last bits of exp();
    (b) a stack of 4 wide splats of opacity 0.9 on one spot: T after them is 1e-4 up to rounding, and falls through
        1e-4 across the neighbouring pixels (forward.cu:358-363);
    (c) 30 splats centred exactly on pixel centres (power == 0 there)."""
    rng = np.random.default_rng(seed)
    sc = random_scene(1500, seed, scale=0.06)
    cam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4))
    # (c): means projecting onto pixel centres: ndc = ((px + 0.5) * 2 / W - 1), x = ndc * z (tanfov = 1)
    n_c = 30
    zc = rng.uniform(1.0, 4.0, n_c).astype(np.float32)
    pxc = rng.integers(8, W - 8, n_c); pyc = rng.integers(8, H - 8, n_c)
    mc = np.stack([((pxc + 0.5) * 2.0 / W - 1.0) * zc, ((pyc + 0.5) * 2.0 / H - 1.0) * zc, zc], 1).astype(np.float32)
    # (b): four wide, nearly flat splats in front of everything at one spot
    mb = np.array([[0.1, 0.05, 0.30 + 0.01 * k] for k in range(4)], np.float32)
    # (a): small splats anywhere
    n_a = 150
    za = rng.uniform(0.8, 5.0, n_a).astype(np.float32)
    ma = np.stack([rng.uniform(-0.8, 0.8, n_a) * za, rng.uniform(-0.8, 0.8, n_a) * za, za], 1).astype(np.float32)

    def grow(key, extra):
        sc[key] = np.concatenate([sc[key], extra.astype(np.float32)])
    n0 = sc["means3D"].shape[0]
    grow("means3D", np.concatenate([mc, mb, ma]))
    n_new = n_c + 4 + n_a
    grow("scales", np.concatenate([np.full((n_c, 3), 0.05), np.full((4, 3), 0.25), np.exp(rng.normal(np.log(0.04), 0.3, (n_a, 3)))]))
    rot = rng.normal(size=(n_new, 4)); rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    grow("rotations", rot)
    grow("opacities", np.concatenate([np.full(n_c, 0.6), np.full(4, 0.9), np.full(n_a, 0.5)]))
    grow("colors", rng.uniform(0, 1, (n_new, 3)))
    # tune (a) from the projected conics (they do not depend on opacity)
    fwd = oracle.rasterize_forward(cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors"], scales=sc["scales"],
                                   rotations=sc["rotations"])
    ia = np.arange(n0 + n_c + 4, n0 + n_new)
    xy = fwd["means2D"][ia].astype(np.float32); con = fwd["conic_opacity"][ia].astype(np.float32)
    pix = np.floor(xy) + np.array([2.0, 1.0], np.float32)                 # a pixel about two columns / one row off the centre
    d = xy - pix
    power = np.float32(-0.5) * (con[:, 0] * d[:, 0] * d[:, 0] + con[:, 2] * d[:, 1] * d[:, 1]) - con[:, 1] * d[:, 0] * d[:, 1]
    delta = rng.uniform(-1e-6, 1e-6, n_a)
    op = (1.0 / 255.0) * (1.0 + delta) / np.exp(power.astype(np.float64))
    ok = (fwd["radii"][ia] > 0) & (power < 0) & (op < 0.98)
    sc["opacities"][ia[ok]] = op[ok].astype(np.float32)
    assert ok.sum() > 60
    return W, H, sc, np.eye(4, dtype=np.float32)


def border_scene(W=112, H=80, seed=11):
    """Splats in a band around the frustum's edges, with strongly anisotropic scales and quaternions that are NOT unit
    (forward.cu:120-151 does not normalise them, so cov3D is scaled by |q|^4): whether such a splat reaches a tile is
    decided by its radius alone.  This is what the early frustum test of k_preprocess_views has to get right -- it may only
    drop a splat the reference drops (empty tile rectangle, forward.cu:233-236); vis_count and the scores would show it."""
    rng = np.random.default_rng(seed)
    P = 3000
    z = rng.uniform(0.6, 5.0, P)
    band = rng.uniform(0.7, 1.9, P) * rng.choice([-1.0, 1.0], P)          # tan of the angle off the axis: the image edge is at 1
    free = rng.uniform(-1.6, 1.6, P)
    side = rng.random(P) < 0.5
    xz, yz = np.where(side, band, free), np.where(side, free, band)
    means = np.stack([xz * z, yz * z, z], 1).astype(np.float32)
    scales = np.exp(rng.normal(np.log(0.08), 1.0, (P, 3))).clip(0.004, 0.6).astype(np.float32)
    rot = rng.normal(size=(P, 4))
    rot *= (rng.uniform(0.5, 1.6, P) / np.linalg.norm(rot, axis=1))[:, None]
    sc = dict(means3D=means, scales=scales, rotations=rot.astype(np.float32),
              opacities=rng.uniform(0.05, 0.9, P).astype(np.float32), colors=rng.uniform(0, 1, (P, 3)).astype(np.float32))
    return W, H, sc, np.eye(4, dtype=np.float32)


def _family(case, oracle):
    if case == "thresholds":
        return threshold_scene(oracle)
    if case == "border":
        return border_scene()
    return _scene(case)


@pytest.fixture(scope="module")
def family(gpu, oracle):
    cache = {}

    def get(case):
        if case not in cache:
            from fisher_rast.ops import FisherScorer
            from models.SLAM.utils.recon_helpers import setup_camera
            W, H, sc, w2c = _family(case, oracle)
            K = intrinsics(W, H)
            cam = setup_camera(W, H, K, np.eye(4), device=gpu)
            ocam = oracle.setup_camera(W, H, K, np.eye(4))
            w2cs = _views(w2c, 3)
            args = (sc["means3D"], sc["colors"], sc["rotations"], sc["opacities"], sc["scales"])
            t = [torch.from_numpy(np.ascontiguousarray(a)).to(gpu) for a in args]
            scorers = {C: FisherScorer(cam, *t, columns=C) for C in (4, 11)}
            # (cur_H of the oracle, cur_H of the arbiter, vis_count) per view
            cur = {C: [oracle.compute_hessian(ocam, w, *args, columns=C, arbiter=True) for w in w2cs] for C in (4, 11)}
            cache[case] = dict(P=sc["means3D"].shape[0], w2cs=w2cs, args=args, scorers=scorers, cur=cur, ocam=ocam)
        return cache[case]
    return get


# multiple of the reference chain's own binary32 deviation (per Gaussian) by which an entry's tolerance is widened.  Measured need
# (tools/arbiter_diag.py): 0.95 at most, 4 and 11 columns alike -- every out_H mode runs on records in the u = -conic d basis now
# (k_fisher_tile_v3h, k_fisher_tile_v3g); round 2's two-pass kernel for 11 columns (the reference's own (dx, dy) chain with a
# different rounding sequence) needed 5.1.
K_DEV = {4: 1.25, 11: 1.25}
# ... and the widening is capped: the worst Gaussian the arbiter has found is off by 0.22 (11 columns); a chain that loses more than
# that gets no further allowance, so a kernel regression on ill-conditioned splats still fails
WIDEN_CAP = 0.3


def _entry_tolerance(o, a, C):
    """[P, C] tolerance: (1e-4 + min(K r_G, cap)) |o| + 1e-7 max|o|, r_G = the largest relative deviation of the binary32 oracle `o` from the
    arbiter `a` over the Gaussian's columns (entries below 1e-7 of the largest are not rated)."""
    o64, big = o.astype(np.float64), np.abs(a) > 1e-7 * np.abs(a).max()
    r = np.where(big, np.abs(o64 - a) / np.maximum(np.abs(a), 1e-300), 0.0).max(axis=1, keepdims=True)
    return (1e-4 + np.minimum(K_DEV[C] * r, WIDEN_CAP)) * np.abs(o64) + 1e-7 * np.abs(o64).max(), r


def _k_needed(got, o, a, C):
    """the multiple of r_G the worst entry of `got` actually needed beyond the flat 1e-4 (reported, so that K_DEV can be judged)"""
    r = _entry_tolerance(o, a, C)[1]
    o64 = o.astype(np.float64)
    excess = (np.abs(got - o64) - 1e-7 * np.abs(o64).max()) / np.maximum(np.abs(o64), 1e-300) - 1e-4
    return float(np.where((excess > 0) & (r > 0), excess / np.maximum(r, 1e-300), 0.0).max())


@pytest.mark.parametrize("columns", [4, 11])
@pytest.mark.parametrize("case", CASES + ["thresholds", "border"])
def test_scorer_modes_on_adversarial_families(family, gpu, case, columns):
    f = family(case)
    P, V, C = f["P"], len(f["w2cs"]), columns
    sc = f["scorers"][C]
    w2c = torch.from_numpy(f["w2cs"]).to(gpu)
    cur_o = np.stack([h for h, _, _ in f["cur"][C]])                    # [V, P, C] binary32 oracle
    cur_a = np.stack([h for _, h, _ in f["cur"][C]])                    # ... and the arbiter
    vis_o = np.array([v for _, _, v in f["cur"][C]])
    # keyframes = views 1.. (so that H_train differs from the view being scored), reg 0.1 as gaussian.py:1367
    H_train_o = cur_o[1:].sum(0, dtype=np.float32)
    H_inv_o = (np.float32(1.0) / (H_train_o + np.float32(0.1))).astype(np.float32)
    want = (cur_o.astype(np.float64) * H_inv_o.astype(np.float64)[None]).sum(axis=(1, 2))

    # out_H, per view (compute_Hessian) and accumulated (compute_H_train)
    cur = torch.zeros((V, P, C), device=gpu)
    r = sc.run(w2c, out_H=cur, out_H_per_view=True)
    assert np.array_equal(r["vis_count"].cpu().numpy(), vis_o)
    tols = []
    n_wide = 0
    for v in range(V):
        tol, r_G = _entry_tolerance(cur_o[v], cur_a[v], C)
        tols.append(tol)
        n_wide += int((r_G > 2e-5).sum())
        got = cur[v].cpu().numpy().astype(np.float64)
        bad = np.abs(got - cur_o[v]) > tol
        assert not bad.any(), (case, v, int(bad.sum()), float((np.abs(got - cur_o[v]) / np.maximum(tol, 1e-300)).max()))
    # the widening is the exception: Gaussians whose reference chain is off by more than 2e-5 are at most 1.5 % of the (view,
    # Gaussian) pairs (measured on the CPU: crowded_tile 1.2-1.5 % -- deep contributors behind thousands of splats --, border
    # 0.5-1.0 %, ragged 0.5-0.7 %, thresholds 0.08 %, general 0.007 %, ties / opaque none)
    assert n_wide <= 0.02 * V * P, (case, n_wide)
    r_max = max(float(_entry_tolerance(cur_o[v], cur_a[v], C)[1].max()) for v in range(V))
    need = max(_k_needed(cur[v].cpu().numpy().astype(np.float64), cur_o[v], cur_a[v], C) for v in range(V))
    print(f"[{case}-{C}] widened (view, Gaussian) pairs: {n_wide} of {V * P} ({100.0 * n_wide / (V * P):.3f} %), largest r_G {r_max:.2e}, "
          f"K needed {max(need, 0.0):.2f} of K_DEV {K_DEV[C]}")
    Ht = torch.zeros((P, C), device=gpu)
    sc.run(w2c[1:], out_H=Ht)
    bad = np.abs(Ht.cpu().numpy().astype(np.float64) - H_train_o) > np.sum(tols[1:], axis=0)
    assert not bad.any(), (case, "H_train", int(bad.sum()))

    # score-only (the single-pass kernel), H_inv shared by the views: 1e-4 on every family
    H_inv = torch.from_numpy(H_inv_o).to(gpu)
    s = sc.run(w2c, H_inv=H_inv)
    assert np.array_equal(s["vis_count"].cpu().numpy(), vis_o)
    assert rel_err(s["scores"].cpu().numpy(), want) < 1e-4, (case, s["scores"].cpu().numpy(), want)

    # per-view H_inv (the path evaluator's mode)
    g = torch.Generator().manual_seed(5)
    Hv = (torch.rand((V, P, C), generator=g) * 3.0 + 0.05)
    want_pv = (cur_o.astype(np.float64) * Hv.numpy().astype(np.float64)).sum(axis=(1, 2))
    s_pv = sc.run(w2c, H_inv=Hv.to(gpu), H_inv_per_view=True)
    assert rel_err(s_pv["scores"].cpu().numpy(), want_pv) < 1e-4, (case, s_pv["scores"].cpu().numpy(), want_pv)

    # scores and materialised cur_H of the SAME launch sequence agree (gaussian.py:1367)
    s2 = (cur.double() * H_inv.double()[None]).sum(dim=(1, 2)).cpu().numpy()
    assert rel_err(s["scores"].cpu().numpy(), s2) < 1e-4
