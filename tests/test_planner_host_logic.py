"""CPU-only: the host-side logic of the planner mirror (planning/astar.py of the package: init, the FBE target rule) and the
NumPy restatement of the candidate samplers (oracle/occupancy_frontier.py) that the GPU kernels are checked against."""
import math

import numpy as np
import torch


def test_init_without_and_with_scene_bounds():
    """astar.py:66-103: 768 x 768 cells centred on the first pose, or the x-z extent of the bounds at the fixed cell size;
    every cell unknown except the 3 x 3 block under the camera."""
    from planning import AstarPlanner
    pl = AstarPlanner(device="cpu", cell_size=0.05)
    pose = torch.eye(4, dtype=torch.float64); pose[0, 3], pose[1, 3], pose[2, 3] = 1.5, 0.3, -2.0
    pl.init(pose, torch.eye(3))
    assert tuple(pl.grid_dim) == (768, 768) and pl.occ_map.shape == (3, 768, 768) and pl.cam_height == 0.3
    assert np.allclose(pl.map_center.numpy(), [1.5, -2.0]) and tuple(pl.cam_pos) == (384, 384)
    assert float(pl.occ_map[0].min()) == 1.0 and float(pl.occ_map[1].abs().max()) == 0.0
    assert float(pl.occ_map[2].sum()) == 18.0 and float(pl.occ_map[2, 383:386, 383:386].min()) == 2.0
    lo, hi = np.array([-3.0, -1.0, -2.0]), np.array([4.0, 2.0, 5.0])
    pose[2, 3] = 1.0
    pl.init(pose.numpy(), torch.eye(3), scene_bounds=(lo, hi))
    assert tuple(pl.grid_dim) == (int(7.0 / 0.05 + 1), int(7.0 / 0.05 + 1)) == (141, 141)
    assert np.allclose(pl.map_center.numpy(), [0.5, 1.5]) and pl.occ_map.shape == (3, 141, 141)
    col = int((1.5 - 0.5) / 0.05 + 141 // 2); row = int((1.0 - 1.5) / 0.05 + 141 // 2)
    assert tuple(pl.cam_pos) == (row, col) and float(pl.occ_map[2, row, col]) == 2.0


def test_fbe_target_rule():
    """astar.py:655-679: nearest selected cell at least 0.5 away from cam_pos, else the fixed fallback direction."""
    from planning import AstarPlanner
    pl = AstarPlanner(device="cpu")
    pl.cam_pos = np.array([10, 20])
    sel = np.array([[10.2, 20.1], [13.0, 24.0], [10.0, 21.0], [9.0, 20.0]])
    got = pl._fbe_point(sel)
    d = np.linalg.norm(sel - pl.cam_pos[None, :], axis=1)
    assert got.shape == (1, 2) and np.array_equal(got[0], sel[2]) and d[2] == d[3] == 1.0      # first of the tied minima
    near = np.array([[10.1, 20.1], [10.0, 20.3]])
    fb = pl._fbe_point(near)
    a = math.pi * 5 / 4
    assert np.allclose(fb, pl.cam_pos[None, :] + np.array([[-math.cos(a), -math.sin(a)]]) * 0.5)


def test_sampler_restatement_properties():
    from oracle.occupancy_frontier import occ_uniform, ring_candidates, free_candidates
    k = np.arange(200000)
    for j in range(3):
        u = occ_uniform(12345, k, j)
        assert u.dtype == np.float32 and u.min() >= 0.0 and u.max() < 1.0
        assert abs(float(u.mean()) - 0.5) < 3e-3 and abs(float(u.var()) - 1 / 12) < 2e-3
        h = np.histogram(u, bins=16, range=(0, 1))[0]
        assert h.min() > 0.9 * len(k) / 16
    assert abs(np.corrcoef(occ_uniform(7, k, 0), occ_uniform(7, k, 1))[0, 1]) < 0.01
    assert not np.array_equal(occ_uniform(1, k[:64], 0), occ_uniform(2, k[:64], 0))
    centers = np.array([[0.0, 0.0], [3.0, -1.0]], np.float32)
    c2w, keep = ring_candidates(centers, 4096, 0.2, 1.0, 0.4, seed=9)
    assert keep.all() and np.allclose(c2w[:, 1, 3], 0.4) and np.allclose(c2w[:, 3], [0, 0, 0, 1])
    off = c2w[:, [0, 2], 3][:, None, :] - centers[None]
    r = np.linalg.norm(off, axis=2).min(axis=1)
    assert r.min() >= 0.2 - 1e-5 and r.max() <= 1.0 + 1e-5
    R = c2w[:, :3, :3]
    assert np.allclose(R @ R.transpose(0, 2, 1), np.eye(3)[None], atol=1e-5) and np.allclose(np.linalg.det(R), 1.0, atol=1e-5)
    # the camera looks along +z of its frame back towards the centre it was sampled around (yaw theta + pi, then x and y flipped)
    ci = np.linalg.norm(off, axis=2).argmin(axis=1)
    to_centre = centers[ci] - c2w[:, [0, 2], 3]
    fwd = R[:, [0, 2], 2]
    cosang = (to_centre * fwd).sum(1) / np.linalg.norm(to_centre, axis=1)
    assert cosang.min() > 0.999 and np.allclose(R[:, 1, 1], -1.0)
    er = np.zeros((768, 768), np.uint8); er[300:420, 350:400] = 1
    rp = free_candidates(er, 0.25, seed=4)
    assert rp.shape == (120 * 50 // 4, 4, 4) and np.allclose(rp[:, 1, 3], 0.25)
    col = np.floor((rp[:, 0, 3] - 0.0) / 0.05).astype(int) + 384; row = np.floor((rp[:, 2, 3]) / 0.05).astype(int) + 384
    assert er[row, col].all()
