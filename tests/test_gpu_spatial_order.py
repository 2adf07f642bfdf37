"""-m gpu: fr_spatial_order (the Gaussians along a Z-curve) and fr_fisher_cfg.order (FisherScorer(spatial_order=True), off by default)
-- a layout hint: visible counts and tile-instance counts with and without it are identical, the scores agree to 1e-4 (the
contributor sets are the same; two DISTINCT splats of bit-equal depth in one tile -- about one pair per view -- composite in Z-curve
order instead of index order, which moves a score by ~1e-5), out_H likewise, per-view weights are read at the caller's index,
duplicated Gaussians keep the caller's order (the sort is stable), and the group test of the projection kernel (whole rounds of 256
neighbours skipped per view) never drops a Gaussian the per-Gaussian test keeps."""
import numpy as np
import pytest
import torch

from scenes import rel_err

pytestmark = pytest.mark.gpu


def _morton_np(m):
    """k_knn_morton restated: 10 bits per axis over the bounding box, x lowest"""
    m = m.astype(np.float32)
    lo, hi = m.min(0), m.max(0)
    ext = (hi - lo).astype(np.float32)
    u = np.where(ext > 0, (m - lo) / np.where(ext > 0, ext, 1), 0).astype(np.float32)
    q = (np.clip(u, 0, 1) * np.float32(1023.0)).astype(np.uint32).astype(np.uint64)

    def spread(x):
        x = (x | (x << 16)) & 0x030000FF
        x = (x | (x << 8)) & 0x0300F00F
        x = (x | (x << 4)) & 0x030C30C3
        x = (x | (x << 2)) & 0x09249249
        return x
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


@pytest.mark.parametrize("P", [1, 2, 255, 256, 257, 5000, 70001])
def test_spatial_order_is_a_stable_sort_along_the_z_curve(gpu, P):
    from fisher_rast.ops import spatial_order_of
    g = torch.Generator().manual_seed(P)
    m = torch.rand((P, 3), generator=g) * torch.tensor([10.0, 2.5, 10.0]) - torch.tensor([5.0, 1.25, 5.0])
    if P > 300:
        m[100:200] = m[0:100]                     # duplicates: equal codes
        m[P - 50:] = m[200:250]
    order = spatial_order_of(m.to(gpu)).cpu().numpy().astype(np.int64)
    assert order.shape == (P,) and np.array_equal(np.sort(order), np.arange(P))
    code = _morton_np(m.numpy())
    want = np.argsort(code, kind="stable")
    c = code[order]
    assert np.all(c[1:] >= c[:-1])
    # stable: inside a run of equal codes the indices ascend
    same = c[1:] == c[:-1]
    assert np.all(order[1:][same] > order[:-1][same])
    assert np.array_equal(order, want)


@pytest.fixture(scope="module")
def scene(gpu):
    from fisher_rast import synthetic
    from models.SLAM.utils.recon_helpers import setup_camera
    P, V, W, H = 80_000, 16, 256, 256
    act = {k: v.to(gpu) for k, v in synthetic.activate(synthetic.room_shell(P, seed=21)).items()}
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, seed=21)).to(gpu)
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=gpu)
    return dict(P=P, V=V, act=act, w2c=w2c, cam=cam)


def _scorer(s, columns, spatial):
    from fisher_rast.ops import FisherScorer
    return FisherScorer(s["cam"], *(s["act"][k] for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")), columns=columns,
                        spatial_order=spatial)


@pytest.mark.parametrize("columns", [4, 11])
def test_the_order_changes_no_result(scene, gpu, columns):
    s = scene
    P, V = s["P"], s["V"]
    a, b = _scorer(s, columns, False), _scorer(s, columns, True)
    assert a.order is None and b.order is not None and b.order.dtype == torch.int32
    Hi = (torch.rand((P, columns), generator=torch.Generator().manual_seed(2)) * 2 + 0.05).to(gpu)
    ra, rb = a.run(s["w2c"], H_inv=Hi), b.run(s["w2c"], H_inv=Hi)
    assert torch.equal(ra["vis_count"], rb["vis_count"]) and torch.equal(ra["num_rendered"], rb["num_rendered"])
    assert int(ra["vis_count"].min()) > 0
    assert rel_err(rb["scores"].cpu().numpy(), ra["scores"].cpu().numpy()) < 1e-4       # (equal but for the order of bit-equal depths)
    # per-view weights are rows of the CALLER's indexing
    Hv = (torch.rand((V, P, columns), generator=torch.Generator().manual_seed(3)) * 2 + 0.05).to(gpu)
    pa, pb = a.run(s["w2c"], H_inv=Hv, H_inv_per_view=True), b.run(s["w2c"], H_inv=Hv, H_inv_per_view=True)
    assert rel_err(pb["scores"].cpu().numpy(), pa["scores"].cpu().numpy()) < 1e-4
    # the diagonal, per view and accumulated: rows of the caller's indexing, equal up to the order of the float atomics
    Ha = torch.zeros((V, P, columns), device=gpu)
    Hb = torch.zeros((V, P, columns), device=gpu)
    a.run(s["w2c"], out_H=Ha, out_H_per_view=True)
    b.run(s["w2c"], out_H=Hb, out_H_per_view=True)
    assert float(Ha.max()) > 0 and rel_err(Hb.sum(dim=0).cpu().numpy(), Ha.sum(dim=0).cpu().numpy()) < 1e-3
    nz = (Ha.abs().sum(dim=(0, 2)) > 0)
    assert torch.equal(nz, (Hb.abs().sum(dim=(0, 2)) > 0))                               # the same Gaussians are touched
    # H_inv and out_H in one launch: the two-pass fall-back ignores the order
    Hc = torch.zeros((P, columns), device=gpu)
    rc = b.run(s["w2c"][:3], H_inv=Hi, out_H=Hc)
    assert rel_err(rc["scores"].cpu().numpy(), ra["scores"][:3].cpu().numpy()) < 1e-4


def test_group_test_never_drops_a_survivor(gpu, oracle):
    """Rounds of 256 Z-curve neighbours are skipped per view by a bounding-sphere test: stress it with poses INSIDE the cloud, at its
    edges, looking away from it, with huge and tiny splats mixed, a non-finite mean, and a camera with an off-centre principal point --
    the visible counts must equal the oracle's (which tests every Gaussian)."""
    from fisher_rast import synthetic
    from fisher_rast.ops import FisherScorer
    from models.SLAM.utils.recon_helpers import setup_camera
    P, W, H = 6000, 160, 96
    raw = synthetic.room_shell(P, seed=33, kind="uniform_box")
    raw["log_scales"][::7] += 2.0                         # some splats 7x larger than their neighbours
    raw["means3D"][123] = float("nan")
    act = synthetic.activate(raw)
    K = np.array([[70.0, 0, 95.0], [0, 85.0, 30.0], [0, 0, 1]])          # principal point far from the centre
    poses = synthetic.candidate_poses(12, seed=34)
    poses[0, :3, 3] = torch.tensor([4.9, 0.0, 4.9])     # in a corner
    poses[1, :3, 3] = torch.tensor([9.0, 0.0, 0.0])     # outside the room
    w2c = synthetic.invert_rigid(poses)
    cam = setup_camera(W, H, K, np.eye(4), device=gpu)
    ocam = oracle.setup_camera(W, H, K, np.eye(4))
    a = {k: v.numpy() for k, v in act.items()}
    sc = FisherScorer(cam, *(act[k].to(gpu) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")), spatial_order=True)
    Hi = torch.ones((P, 4), device=gpu)
    r = sc.run(w2c.to(gpu), H_inv=Hi)
    for v in range(w2c.shape[0]):
        want = oracle.compute_hessian(ocam, w2c[v].numpy(), a["means3D"], a["rgb_colors"], a["rotations"], a["opacities"], a["scales"], return_all=True)
        assert int(r["vis_count"][v]) == want[1], (v, int(r["vis_count"][v]), want[1])
        assert int(r["num_rendered"][v]) == want[2]["num_rendered"]
