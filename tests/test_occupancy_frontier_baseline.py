"""The CPU occupancy/frontier restatement is a reported baseline only (bench.py); this checks that it behaves sanely on
the synthetic room: free space grows around the camera, frontiers appear at the edge of the observed region."""
import numpy as np


def test_occupancy_and_frontiers_on_synthetic_room():
    from fisher_rast import synthetic
    from oracle.occupancy_frontier import OccupancyMap, room_depth
    W = H = 64
    K = synthetic.intrinsics(W, H)
    m = OccupancyMap(K, grid_dim=(256, 256), cell_size=0.05)
    pose = np.eye(4)
    d = room_depth(pose, W, H, K)
    assert d.shape == (1, H, W) and np.isclose(d[0, H // 2, W // 2], 5.0, atol=0.1)   # wall 5 m ahead of the origin
    for _ in range(3):                      # every cell starts at unknown = 1 and an observation adds at most 1: a cell flips
        m.update_occ_map(d, pose)           # once it has been seen more than once (planning/astar.py:94-96, 301)
    idx = m.occ_map.argmax(axis=0)
    assert (idx == 2).sum() > 100 and (idx == 1).sum() > 10          # free wedge and occupied wall cells
    pts, free = m.build_frontiers(None)
    assert free.sum() > 100 and free[m.cam_pos[0], m.cam_pos[1]] == 1
    assert pts is not None and pts.shape[1] == 2
    # looking the other way as well removes part of the frontier behind the camera
    back = np.eye(4); back[0, 0] = -1; back[2, 2] = -1
    for _ in range(3):
        m.update_occ_map(room_depth(back, W, H, K), back)
    _, free2 = m.build_frontiers(None)
    assert free2.sum() > free.sum()
