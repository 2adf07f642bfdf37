"""One synthetic planning round through the drop-in surfaces, in the order tester_gaussians_navigation.py runs them
(BASELINE.json configs[4] needs Habitat + HM3D, which are not available offline): occupancy update from depth frames ->
frontiers -> candidate poses around the frontier -> free-space filter -> GaussianSLAM.pose_eval -> best view."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_planning_round_on_the_synthetic_room(gpu):
    from fisher_rast import synthetic
    from oracle.occupancy_frontier import room_depth            # synthetic depth frames (input generator)
    from planning import AstarPlanner
    import models.gaussian_slam as mgs
    W = H = 128
    K = synthetic.intrinsics(W, H)
    P = 60_000
    params = {k: v.to(gpu) for k, v in synthetic.room_shell(P, 7).items()}
    slam = mgs.GaussianSLAM(params=params, intrinsics=K, width=W, height=H, device=gpu)
    planner = AstarPlanner(device=gpu, cell_size=0.05, frontier_select_method="combined", sample_view_num=48, sample_range=1.0,
                           min_range=0.2)
    start = np.eye(4, dtype=np.float32)
    planner.init(torch.from_numpy(start), torch.from_numpy(np.asarray(K, dtype=np.float32)))
    # the agent looks around from the start pose, twice (a cell leaves "unknown" once it has been observed more than once:
    # every cell starts at 1 and an observation adds at most 1, astar.py:94-96, 301); each heading becomes a keyframe of the map
    for t, yaw in enumerate((0.0, 0.5 * np.pi, np.pi, 1.5 * np.pi) * 2):
        c2w = np.eye(4, dtype=np.float32)
        c, s = np.cos(yaw), np.sin(yaw)
        c2w[:3, :3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
        planner.update_occ_map(room_depth(c2w, W, H, K), torch.from_numpy(c2w).to(gpu), t)
        if t < 4:
            slam.add_keyframe(torch.linalg.inv(torch.from_numpy(c2w)).to(gpu))
    frontier, free_space = planner.build_frontiers(slam.params["means3D"])
    assert free_space.sum() > 500 and free_space[planner.cam_pos[0], planner.cam_pos[1]] == 1
    assert frontier is not None and frontier.shape[1] == 2
    torch.manual_seed(0)
    cands = planner.generate_candidate(torch.from_numpy(frontier).float().to(gpu), expansion=1)
    cands = planner.filter_candidates_in_freespace(cands, free_space)
    assert 0 < cands.shape[0] <= 48
    # every kept candidate stands on a cell of the eroded free space
    cells = planner.cells_of(cands[:, :3, 3]).cpu().numpy()
    assert free_space[cells[:, 1], cells[:, 0]].all()
    scores, c2ws = slam.pose_eval(list(cands))
    assert scores.shape == (cands.shape[0],) and c2ws.shape == (cands.shape[0], 4, 4)
    assert bool(torch.isfinite(scores).all()) and bool((scores >= 0).all()) and float(scores.max()) > 0
    best = int(torch.argmax(scores))
    # the information gain is not flat across the surviving candidates
    if scores.numel() > 1:
        assert float(scores[best]) > 1.05 * float(scores.min())
    # and the same scores come out of the sharded helper when there is one rank
    from fisher_rast import distributed as D
    sc = slam._scorer()
    kf = torch.stack([kf["est_w2c"] for kf in slam.keyframe_list])
    s2, _ = D.pose_eval_sharded(sc, kf, torch.linalg.inv(c2ws))
    assert torch.allclose(s2.cpu(), scores, rtol=1e-5, atol=0)
