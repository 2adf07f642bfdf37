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
    # candidate poses: same construction as the reference, then the eroded-free-space filter
    pl.cam_height = 0.0
    torch.manual_seed(3)
    cand = pl.generate_candidate(torch.from_numpy(poses[:, [0, 2], 3]).to(gpu), expansion=1)      # around the observed camera positions
    assert cand.shape == (pl.K, 4, 4) and torch.allclose(cand[:, :3, :3] @ cand[:, :3, :3].transpose(1, 2), torch.eye(3, device=gpu).expand(pl.K, 3, 3), atol=1e-5)
    kept = pl.filter_candidates_in_freespace(cand, free)
    er = ndimage.binary_erosion(free.astype(bool), structure=np.ones((10, 10), bool), border_value=1)
    xy = cand[:, [0, 2], 3].cpu().numpy()
    cx = ((xy[:, 0] - 0.0) / 0.05 + 768 // 2).astype(np.int64); cz = ((xy[:, 1] - 0.0) / 0.05 + 768 // 2).astype(np.int64)
    assert kept.shape[0] == int(er[cz, cx].sum()) and 0 < kept.shape[0] <= pl.K


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
