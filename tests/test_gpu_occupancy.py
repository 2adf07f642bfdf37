"""Planner-side kernels (include/fisher_occ.h, planning/astar.py of the package) against the CPU restatement of the
reference planner (oracle/occupancy_frontier.py): occupancy map bit-exact after every update, free space / frontier /
target masks and the selected cells identical for the three selection rules, erosion and cell binning identical."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(gpu, W, H, grid, n_frames, seed, method="combined"):
    from fisher_rast import synthetic
    from oracle.occupancy_frontier import OccupancyMap, room_depth
    from planning import AstarPlanner
    K = synthetic.intrinsics(W, H)
    poses = synthetic.candidate_poses(n_frames, seed).numpy().astype(np.float32)
    pl = AstarPlanner(device=gpu, cell_size=0.05, frontier_select_method=method)
    start = np.eye(4, dtype=np.float32)
    pl.init(torch.from_numpy(start), torch.from_numpy(np.asarray(K, dtype=np.float32)))
    assert tuple(pl.grid_dim) == (768, 768)
    if grid != 768:                                    # smaller maps for the quick cases
        pl.grid_dim = np.array([grid, grid])
        pl.occ_map = torch.zeros((3, grid, grid), device=gpu)
        pl.occ_map[0] = 1.
        pl.occ_map[2, grid // 2 - 1:grid // 2 + 2, grid // 2 - 1:grid // 2 + 2] = 2.
        pl.cam_pos = np.array([grid // 2, grid // 2])
    om = OccupancyMap(K, grid_dim=(grid, grid), cell_size=0.05, map_center=(0.0, 0.0), height_range=(pl.height_lower, pl.height_upper),
                      pcd_far_distance=pl.pcd_far_distance)
    om.occ_map[2, grid // 2 - 1:grid // 2 + 2, grid // 2 - 1:grid // 2 + 2] = 2.0      # AstarPlanner.init, astar.py:98
    depths = [room_depth(p, W, H, K) for p in poses]
    return pl, om, poses, depths


@pytest.mark.parametrize("W,H,grid,downsample", [(64, 48, 256, 1), (128, 128, 768, 1), (96, 64, 320, 2)])
def test_update_occ_map_bit_exact(gpu, W, H, grid, downsample):
    pl, om, poses, depths = _setup(gpu, W, H, grid, 5, 300)
    assert np.array_equal(pl.occ_map.cpu().numpy(), om.occ_map)
    for t, (p, d) in enumerate(zip(poses, depths)):
        pl.update_occ_map(d, torch.from_numpy(p).to(gpu), t, downsample=downsample)
        om.update_occ_map(d, p, downsample=downsample)
        got = pl.occ_map.cpu().numpy()
        assert np.array_equal(pl.cam_pos, om.cam_pos)
        assert np.array_equal(got, om.occ_map), f"frame {t}: {np.abs(got - om.occ_map).max()} max abs diff, {(got != om.occ_map).sum()} cells"
    idx = om.occ_map.argmax(axis=0)
    assert (idx == 2).sum() > 200 and (idx == 1).sum() > 10


@pytest.mark.parametrize("method", ["combined", "largest", "closest"])
@pytest.mark.parametrize("with_points", [False, True])
def test_freespace_and_frontiers_match(gpu, method, with_points):
    from fisher_rast import synthetic
    pl, om, poses, depths = _setup(gpu, 128, 128, 768, 6, 301, method)
    for t, (p, d) in enumerate(zip(poses, depths)):
        pl.update_occ_map(d, torch.from_numpy(p).to(gpu), t)
        om.update_occ_map(d, p)
    pts = synthetic.room_shell(150_000, 2)["means3D"] if with_points else None
    pts_np = None if pts is None else pts.numpy()
    free = pl.build_connected_freespace(None if pts is None else pts.to(gpu))
    want_free = om.build_connected_freespace(pts_np)
    assert free.dtype == np.uint8 and np.array_equal(free, want_free) and want_free.sum() > 500
    det = {}
    want_pts, want_free2 = om.build_frontiers(pts_np, method=method, details=det)
    got_pts, got_free = pl.build_frontiers(None if pts is None else pts.to(gpu))
    assert np.array_equal(got_free, want_free2)
    assert np.array_equal(pl.frontier, det["frontier"])
    assert want_pts is not None and "target" in det
    assert np.array_equal(pl.target_frontier, det["target"])
    if with_points:
        assert got_pts.shape == want_pts.shape and np.array_equal(got_pts, want_pts)      # np.where order, float64 affine map
    else:
        assert got_pts.shape == (1, 2)                                                     # FBE rule (astar.py:655-679)
        d = np.linalg.norm(want_pts - om.cam_pos[None, :], axis=1)
        ok = np.where(d >= 0.5)[0]
        assert np.array_equal(got_pts[0], want_pts[ok[np.argmin(d[ok])]])


def test_empty_map_has_no_frontier(gpu):
    pl, om, _, _ = _setup(gpu, 64, 48, 256, 1, 302)
    pts, free = pl.build_frontiers(None)
    wpts, wfree = om.build_frontiers(None)
    assert np.array_equal(free, wfree)
    assert (pts is None) == (wpts is None)


def test_erode_cells_and_candidate_filter(gpu):
    from scipy import ndimage
    from oracle.occupancy_frontier import discretize_coords
    pl, om, poses, depths = _setup(gpu, 128, 128, 768, 6, 303)
    for t, (p, d) in enumerate(zip(poses, depths)):
        pl.update_occ_map(d, torch.from_numpy(p).to(gpu), t)
    free = pl.build_connected_freespace(None)
    from fisher_rast import _lib
    import ctypes
    lib = _lib.load()
    cfg = pl._occ_cfg()
    src = torch.from_numpy(free).to(gpu)
    for k in (3, 10, 11):
        dst = torch.empty_like(src)
        _lib.check(lib.fr_occ_erode(ctypes.byref(cfg), src.data_ptr(), dst.data_ptr(), k, None), "fr_occ_erode")
        want = ndimage.binary_erosion(free.astype(bool), structure=np.ones((k, k), bool), border_value=1).astype(np.uint8)
        assert np.array_equal(dst.cpu().numpy(), want), k
    g = torch.Generator().manual_seed(9)
    xyz = (torch.rand((5000, 3), generator=g) - 0.5) * 50.0            # beyond the map on both sides: clamps
    cells = pl.cells_of(xyz.to(gpu)).cpu().numpy()
    want = discretize_coords(xyz[:, 0].numpy(), xyz[:, 2].numpy(), (768, 768), 0.05, (0.0, 0.0))
    assert np.array_equal(cells, want)


def test_candidate_samplers_match_the_restatement(gpu):
    """fr_occ_ring_candidates (generate_candidate[_object] + the free-space filter, astar.py:1383-1430) and
    fr_occ_free_candidates (sample_random_candidate, astar.py:782-837) against oracle/occupancy_frontier.py on the same seed:
    poses within float32 rounding of sin / cos (1e-6), the keep flags and the pose count exact."""
    from scipy import ndimage
    from oracle.occupancy_frontier import ring_candidates, free_candidates
    pl, om, poses, depths = _setup(gpu, 128, 128, 768, 6, 303)
    for t, (p, d) in enumerate(zip(poses, depths)):
        pl.update_occ_map(d, torch.from_numpy(p).to(gpu), t)
    free = pl.build_connected_freespace(None)
    pl.cam_height = 0.25
    centers = torch.from_numpy(poses[:, [0, 2], 3]).to(gpu)                 # around the observed camera positions
    for K, seed, expansion in ((64, 1234, 1.0), (257, 99, 1.5), (1500, 7, 2.25)):
        pl.K = K
        cand = pl.generate_candidate(centers, expansion=expansion, seed=seed)
        want, _ = ring_candidates(centers.cpu().numpy(), K, pl.min_range, pl.radius * expansion, pl.cam_height, seed)
        assert cand.shape == (K, 4, 4)
        assert np.abs(cand.cpu().numpy() - want).max() < 2e-6
        R = cand[:, :3, :3]
        assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3, device=gpu).expand(K, 3, 3), atol=1e-5)
        # fused filter: flags from the restatement evaluated on the GPU's own positions (so a last-bit difference in sin / cos cannot flip a cell)
        er = ndimage.binary_erosion(free.astype(bool), structure=np.ones((10, 10), bool), border_value=1)
        kept = pl.generate_candidate_in_freespace(centers, free, expansion=expansion, seed=seed)
        xy = cand[:, [0, 2], 3].cpu().numpy()
        col = ((xy[:, 0] - np.float32(0.0)) / np.float32(0.05) + np.float32(384)).astype(np.int64)
        row = ((xy[:, 1] - np.float32(0.0)) / np.float32(0.05) + np.float32(384)).astype(np.int64)
        keep = er[row, col]
        assert er.sum() > 40 and 0 < keep.sum() < K
        assert torch.equal(kept, cand[torch.from_numpy(keep).to(gpu)])
        assert torch.equal(pl.filter_candidates_in_freespace(cand, free), kept)
        _, keep_o = ring_candidates(centers.cpu().numpy(), K, pl.min_range, pl.radius * expansion, pl.cam_height, seed, eroded=er)
        assert (keep_o != keep).sum() <= 1                                   # the restatement's own positions: at most a boundary case
    # the object ring uses its own parameters
    pl.K_object, pl.radius_object, pl.min_range_object = 33, 0.8, 0.3
    got = pl.generate_candidate_object(centers, expansion=2, seed=5)
    want, _ = ring_candidates(centers.cpu().numpy(), 33, 0.3, 1.6, pl.cam_height, 5)
    assert np.abs(got.cpu().numpy() - want).max() < 2e-6
    # a filter on an almost empty free space keeps everything (astar.py:1389)
    few = np.zeros_like(free); few[380:386, 380:386] = 1
    assert pl.generate_candidate_in_freespace(centers, few, seed=3).shape[0] == pl.K
    # torch.manual_seed makes the default seeding reproducible
    torch.manual_seed(3); a = pl.generate_candidate(centers)
    torch.manual_seed(3); b = pl.generate_candidate(centers)
    assert torch.equal(a, b) and not torch.equal(a, pl.generate_candidate(centers))
    # uniformly placed poses in the free space
    rp = pl.sample_random_candidate(np.array([0.0, 0.4, 0.0]), free, seed=77)
    er11 = ndimage.binary_erosion(free.astype(bool), structure=np.ones((11, 11), bool), border_value=1).astype(np.uint8)
    want = free_candidates(er11, 0.4, 77)
    assert rp.shape == want.shape and rp.shape[0] == int(er11.sum()) // 4 > 100
    assert np.abs(rp.cpu().numpy() - want).max() < 2e-6
    xz = rp[:, [0, 2], 3].cpu().numpy()                                       # the planner's own cell rule (astar.py:93-94)
    assert er11[(xz[:, 1] / 0.05 + 384).astype(np.int64), (xz[:, 0] / 0.05 + 384).astype(np.int64)].all()   # every pose sits in the eroded free space


def test_bad_arguments_are_reported(gpu):
    import ctypes
    from fisher_rast import _lib
    lib = _lib.load()
    cfg = _lib.OccCfg(0, 768, 0.05, 0.0, 0.0, -0.6, 0.6, 10.0)
    assert lib.fr_occ_workspace_bytes(ctypes.byref(cfg)) == 0
    assert lib.fr_occ_erode(ctypes.byref(cfg), 1, 2, 3, None) == _lib.FR_EINVAL
    assert b"fr_occ_cfg" in lib.fr_last_error()


def test_scene_bounds_give_a_non_square_map(gpu):
    """AstarPlanner.init with scene_bounds (astar.py:76-86): grid from the bounds, map centre off the origin; update, free space
    and frontiers on a 281 x 201 map against the CPU restatement."""
    from fisher_rast import synthetic
    from oracle.occupancy_frontier import OccupancyMap, room_depth
    from planning import AstarPlanner
    W, H = 96, 80
    K = synthetic.intrinsics(W, H)
    pl = AstarPlanner(device=gpu, cell_size=0.05, frontier_select_method="closest")
    lower, upper = np.array([-6.0, -1.5, -4.0]), np.array([8.0, 1.5, 6.0])
    pl.init(torch.eye(4), torch.from_numpy(np.asarray(K, dtype=np.float32)), scene_bounds=(lower, upper))
    gw, gh = int(pl.grid_dim[0]), int(pl.grid_dim[1])
    assert (gw, gh) == (281, 201) and tuple(pl.occ_map.shape) == (3, gh, gw)
    mc = pl.map_center.cpu().numpy()
    assert np.allclose(mc, [1.0, 1.0])
    om = OccupancyMap(K, grid_dim=(gw, gh), cell_size=0.05, map_center=(float(np.float32(mc[0])), float(np.float32(mc[1]))),
                      height_range=(pl.height_lower, pl.height_upper), pcd_far_distance=pl.pcd_far_distance)
    cz, cx = int(pl.cam_pos[0]), int(pl.cam_pos[1])
    om.occ_map[2, cz - 1:cz + 2, cx - 1:cx + 2] = 2.0
    assert np.array_equal(pl.occ_map.cpu().numpy(), om.occ_map)
    poses = synthetic.candidate_poses(5, 305).numpy().astype(np.float32)
    for t, p in enumerate(poses):
        d = room_depth(p, W, H, K)
        pl.update_occ_map(d, torch.from_numpy(p).to(gpu), t)
        om.update_occ_map(d, p)
        assert np.array_equal(pl.cam_pos, om.cam_pos)
        assert np.array_equal(pl.occ_map.cpu().numpy(), om.occ_map), t
    det = {}
    want_pts, want_free = om.build_frontiers(None, method="closest", details=det)
    got_pts, got_free = pl.build_frontiers(None)
    assert np.array_equal(got_free, want_free) and want_free.sum() > 200
    assert np.array_equal(pl.frontier, det["frontier"]) and np.array_equal(pl.target_frontier, det["target"])
    assert got_pts.shape == (1, 2)
