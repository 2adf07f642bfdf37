"""CPU: the ARBITER (oracle/fisher_oracle.c built with -DORC_DOUBLE: the reference's statements in binary64 on the same
binary32 inputs, every decision -- culling, radii, tile lists, sort order, each pixel's contributor set -- taken from the
binary32 run) against the oracle proper.  |oracle - arbiter| is the rounding error of the reference's own binary32 chain
(backward.cu:276-475, 850-1140), i.e. what no second binary32 implementation can be asked to reproduce.

What it settles (round-2 verdict, `border` family held to 1e-3):
  * on the SCORES (gaussian.py:1367) the binary32 chain is within 1e-5 of exact on every family, needle-shaped splats
    included -- so a kernel that misses 1e-4 there is wrong, not unlucky.  Round 2's scorer record (the H_inv-weighted
    quadratic form expanded into a polynomial in dx, dy) was: it squares the condition number of conic * d; the record in
    g = conic * d (fr_math.h: fr_mean_rows_g / fr_scorer_poly_g) is as accurate as the reference chain itself;
  * on single ENTRIES of cur_H the binary32 chain itself is off by up to several per cent for a handful of needle-shaped
    or near-plane splats: tests/test_gpu_scorer_adversarial.py widens the tolerance of exactly those Gaussians by a
    multiple of this measured deviation and of nothing else."""
import ctypes

import numpy as np
import pytest

from scenes import intrinsics
from test_gpu_rasterizer_parity import _scene
from test_gpu_scorer_adversarial import border_scene, _views


def _family(case):
    return border_scene() if case == "border" else _scene(case)


@pytest.mark.parametrize("case,columns", [("border", 4), ("border", 11), ("general", 4)])
def test_binary32_chain_against_the_arbiter(oracle, case, columns):
    W, H, sc, w2c = _family(case)
    ocam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4))
    w2cs = _views(w2c, 3)
    args = (sc["means3D"], sc["colors"], sc["rotations"], sc["opacities"], sc["scales"])
    res = [oracle.compute_hessian(ocam, w, *args, columns=columns, arbiter=True) for w in w2cs]   # asserts equal pair counts / n_contrib
    o = np.stack([r[0] for r in res]).astype(np.float64)
    a = np.stack([r[1] for r in res])
    H_train = o[1:].astype(np.float32).sum(0, dtype=np.float32)
    H_inv = (np.float32(1.0) / (H_train + np.float32(0.1))).astype(np.float64)
    so, sa = (o * H_inv[None]).sum((1, 2)), (a * H_inv[None]).sum((1, 2))
    assert np.all(np.abs(so - sa) <= 1e-5 * np.abs(sa)), (case, so, sa)          # measured: <= 3.6e-6 (border), 3.5e-6 (general)
    # single entries: the deviation is concentrated in a few Gaussians
    worst = 0.0
    for v in range(len(w2cs)):
        big = np.abs(a[v]) > 1e-7 * np.abs(a[v]).max()
        rel = np.abs(o[v] - a[v])[big] / np.abs(a[v])[big]
        worst = max(worst, float(rel.max()))
        assert (rel > 1e-4).mean() < 0.05
        assert np.median(rel) < 2e-6
    if case == "border":
        assert worst > 1e-3          # measured: 3.6e-2 (4 columns), 2.2e-1 (11): why a flat 1e-4 per entry cannot hold there


def test_scorer_record_is_as_well_conditioned_as_the_reference_chain(oracle, harness):
    """The pair factor F = sum_c H_inv[c] leaf_c^2 / w^2 that k_fisher_tile_v3's record encodes (fr_scorer_poly_g, evaluated as the
    walk does) against the exact value (arbiter build of the reference's per-pair chain, orc_pair_leaves), on the needle-shaped
    splats of `border`: weighted by G^2 over three rings of each splat's footprint, as the score weighs them."""
    cf, cd = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)
    pf = lambda x: x.ctypes.data_as(cf)
    pd = lambda x: x.ctypes.data_as(cd)
    L32, L64 = oracle.lib(), oracle.lib64()
    W, H, sc, w2c = border_scene()
    cam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4))
    w2cs = _views(w2c, 3)
    pts = oracle.transform_points(w2cs[2], sc["means3D"])
    kw = dict(colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
    fwd = oracle.rasterize_forward(cam, pts, sc["opacities"], **kw)
    fwd64 = oracle.rasterize_forward(cam, oracle.transform_points(w2cs[2], sc["means3D"], arbiter=True), sc["opacities"], decisions=fwd, **kw)
    rng = np.random.default_rng(0)
    hv = rng.uniform(0.5, 10, 11).astype(np.float32)
    view = np.ascontiguousarray(cam.viewmatrix, np.float32); proj = np.ascontiguousarray(cam.projmatrix, np.float32)
    view64, proj64 = view.astype(np.float64), proj.astype(np.float64)
    powers = np.repeat([0.5, 2.0, 4.0], 16)
    G2 = np.exp(-2 * powers)
    rows = []
    for i in np.nonzero(fwd["radii"] > 0)[0][::3]:
        con = np.ascontiguousarray(fwd64["conic_opacity"][i])
        ev, evec = np.linalg.eigh(np.array([[con[0], con[1]], [con[1], con[2]]]))
        if ev.min() <= 0:
            continue
        th = np.tile(np.linspace(0, 2 * np.pi, 16, endpoint=False), 3)
        d = (evec @ (np.sqrt(2 * powers) * np.stack([np.cos(th), np.sin(th)]) / np.sqrt(ev)[:, None])).T
        n = len(d)
        dx, dy = np.ascontiguousarray(d[:, 0], np.float32), np.ascontiguousarray(d[:, 1], np.float32)
        og, od, c3 = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(3, np.float32)
        mean, cov = np.ascontiguousarray(pts[i], np.float32), np.ascontiguousarray(fwd["cov3D"][i], np.float32)
        s, r = np.ascontiguousarray(sc["scales"][i], np.float32), np.ascontiguousarray(sc["rotations"][i], np.float32)
        harness.h_scorer_pair_factor(ctypes.c_int(11), pf(mean), pf(cov), pf(s), ctypes.c_float(1.0), pf(r), pf(view), pf(proj),
                                     ctypes.c_int(W), ctypes.c_int(H), ctypes.c_float(cam.tanfovx), ctypes.c_float(cam.tanfovy),
                                     pf(hv), ctypes.c_float(sc["opacities"][i]), ctypes.c_int(n), pf(dx), pf(dy), pf(og), pf(od), pf(c3))
        F64, F32 = np.zeros(n), np.zeros(n)
        m64, cov64 = np.ascontiguousarray(fwd64["inputs"]["means3D"][i]), np.ascontiguousarray(fwd64["cov3D"][i])
        s64, r64 = s.astype(np.float64), r.astype(np.float64)
        co32 = np.ascontiguousarray(fwd["conic_opacity"][i], np.float32)
        out, outf = np.zeros(11), np.zeros(11, np.float32)
        for k in range(n):
            L64.orc_pair_leaves(pd(m64), pd(cov64), pd(s64), ctypes.c_double(1.0), pd(r64), pd(view64), pd(proj64), ctypes.c_int(W), ctypes.c_int(H),
                                ctypes.c_double(cam.tanfovx), ctypes.c_double(cam.tanfovy), pd(con), ctypes.c_double(float(dx[k])),
                                ctypes.c_double(float(dy[k])), ctypes.c_double(1.0), pd(out))
            F64[k] = (hv.astype(np.float64) * out ** 2).sum()
            L32.orc_pair_leaves(pf(mean), pf(cov), pf(s), ctypes.c_float(1.0), pf(r), pf(view), pf(proj), ctypes.c_int(W), ctypes.c_int(H),
                                ctypes.c_float(cam.tanfovx), ctypes.c_float(cam.tanfovy), pf(co32), ctypes.c_float(float(dx[k])),
                                ctypes.c_float(float(dy[k])), ctypes.c_float(1.0), pf(outf))
            F32[k] = (hv.astype(np.float64) * outf.astype(np.float64) ** 2).sum()
        den = (G2 * F64).sum()
        rows.append((den, abs((G2 * (og - F64)).sum()), abs((G2 * (od - F64)).sum()), abs((G2 * (F32 - F64)).sum())))
    rows = np.array(rows)
    assert len(rows) > 400
    err_g, err_d, err_ref = (rows[:, k].sum() / rows[:, 0].sum() for k in (1, 2, 3))
    # measured (all visible splats): the record 7.7e-6, the reference's binary32 chain 7.0e-6, round 2's record 2.2e-4
    assert err_ref < 3e-5
    assert err_g < 3e-5 and err_g < 3 * err_ref + 1e-6
    assert err_d > 1e-4


def test_rows_over_gamma_u_give_the_reference_leaves(oracle, harness):
    """fr_mean_rows_g / fr_scale_rot_jacobian (what every record kernel stores: k_fisher_tile_v3h, k_backward_sq_rows) applied to
    gamma(u) = (ux, uy, ux^2, ux uy, uy^2), u = -conic d, against the reference's per-pair chain evaluated exactly (arbiter build of
    orc_pair_leaves, backward.cu:1016-1090 -> 276-475, 532-583): mean, scale and rotation leaves of random splats at random
    offsets inside their footprint."""
    cf, cd = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)
    pf = lambda x: x.ctypes.data_as(cf)
    pd = lambda x: x.ctypes.data_as(cd)
    L64 = oracle.lib64()
    from scenes import random_scene
    W, H = 96, 64
    w2c = np.eye(4, dtype=np.float32)
    w2c[:3, :3] = np.array([[0.8, 0, 0.6], [0, 1, 0], [-0.6, 0, 0.8]], np.float32)
    w2c[:3, 3] = [0.1, 0.05, 0.3]
    cam = oracle.setup_camera(W, H, intrinsics(W, H), w2c)
    view = np.ascontiguousarray(cam.viewmatrix, np.float32); proj = np.ascontiguousarray(cam.projmatrix, np.float32)
    rng = np.random.default_rng(4)
    errs = []
    for trial in range(300):
        sc = random_scene(1, 300 + trial, zmin=0.4, zmax=5.0, spread=1.5, scale=0.15)
        if trial % 3 == 0:
            sc["scales"] = (sc["scales"] * np.array([[8.0, 1.0, 0.15]], np.float32)).astype(np.float32)     # needles
        fwd = oracle.rasterize_forward(cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
        if fwd["radii"][0] == 0:
            continue
        con = fwd["conic_opacity"][0].astype(np.float64)
        ev, evec = np.linalg.eigh(np.array([[con[0], con[1]], [con[1], con[2]]]))
        if ev.min() <= 0:
            continue
        th, pw = rng.uniform(0, 2 * np.pi), rng.uniform(0.2, 4.0)
        d = evec @ (np.sqrt(2 * pw) * np.array([np.cos(th), np.sin(th)]) / np.sqrt(ev))
        dx, dy = np.float32(d[0]), np.float32(d[1])
        out = np.zeros(10, np.float32)
        c32 = np.ascontiguousarray(fwd["conic_opacity"][0, :3], np.float32)
        harness.h_leaves_from_rows_g(pf(sc["means3D"]), pf(fwd["cov3D"]), pf(sc["scales"]), ctypes.c_float(1.0), pf(sc["rotations"]), pf(view), pf(proj),
                                     ctypes.c_int(W), ctypes.c_int(H), ctypes.c_float(cam.tanfovx), ctypes.c_float(cam.tanfovy), pf(c32),
                                     ctypes.c_float(float(dx)), ctypes.c_float(float(dy)), pf(out))
        want = np.zeros(11)
        f64 = lambda a: np.ascontiguousarray(np.asarray(a, np.float32).astype(np.float64))
        L64.orc_pair_leaves(pd(f64(sc["means3D"][0])), pd(f64(fwd["cov3D"][0])), pd(f64(sc["scales"][0])), ctypes.c_double(1.0), pd(f64(sc["rotations"][0])),
                            pd(f64(view)), pd(f64(proj)), ctypes.c_int(W), ctypes.c_int(H), ctypes.c_double(cam.tanfovx), ctypes.c_double(cam.tanfovy),
                            pd(f64(fwd["conic_opacity"][0])), ctypes.c_double(float(dx)), ctypes.c_double(float(dy)), ctypes.c_double(1.0), pd(want))
        ref = np.concatenate([want[0:3], want[4:11]])
        for lo, hi in ((0, 3), (3, 6), (6, 10)):                # per leaf group: against the group's largest entry
            scale = np.abs(ref[lo:hi]).max()
            if scale > 0:
                errs.append(np.abs(out[lo:hi] - ref[lo:hi]).max() / scale)
    errs = np.asarray(errs)
    assert len(errs) > 400
    assert np.median(errs) < 3e-6 and errs.max() < 5e-4, (float(np.median(errs)), float(errs.max()))
