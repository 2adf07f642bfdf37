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


def test_install_on_reference_shaped_classes(gpu):
    """INTEGRATION.md 4 / 4b, functionally: classes that only carry the attributes the reference's constructors set get the
    accelerated methods through `install` and produce the same numbers as the standalone carriers."""
    from fisher_rast import synthetic
    from oracle.occupancy_frontier import room_depth
    from models.SLAM.gaussian import FisherOps
    from models.SLAM.utils.recon_helpers import setup_camera
    from planning.astar import OccupancyOps
    import planning
    import models.gaussian_slam as mgs
    W = H = 96
    K = synthetic.intrinsics(W, H)
    params = {k: v.to(gpu) for k, v in synthetic.room_shell(8000, 9).items()}
    kfs = [w.to(gpu) for w in synthetic.invert_rigid(synthetic.candidate_poses(3, 109))]
    poses = [p.to(gpu) for p in synthetic.candidate_poses(5, 9)]

    class RefSLAM:                                   # what models/SLAM/gaussian.py:GaussianSLAM holds when pose_eval is called
        def __init__(self):
            self.params = params
            self.cam = setup_camera(W, H, K, np.eye(4), device=gpu)
            self.keyframe_list = [dict(est_w2c=w, id=i) for i, w in enumerate(kfs)]
    FisherOps.install(RefSLAM)
    got, c2ws = RefSLAM().pose_eval(poses)
    ours = mgs.GaussianSLAM(params=params, intrinsics=K, width=W, height=H, device=gpu)
    for w in kfs:
        ours.add_keyframe(w)
    want, _ = ours.pose_eval(poses)
    assert torch.equal(got, want) and c2ws.shape == (5, 4, 4)

    class RefPlanner:                                # attributes of planning/astar.py:AstarPlanner.__init__ / init (23-103)
        def __init__(self):
            self.device = gpu
            self.cell_size, self.height_upper, self.height_lower = 0.05, 0.6, -0.6
            self.K, self.radius, self.min_range = 16, 1.0, 0.2
            self.frontier_select_method, self.pcd_far_distance = "largest", 10.0
            self.grid_dim = np.array([768, 768])
            self.intrinsics = torch.from_numpy(np.asarray(K, dtype=np.float32))
            pose = torch.eye(4)
            self.cam_height = pose[1, 3]                                     # a 0-dim tensor in the reference
            self.occ_map = torch.zeros((3, 768, 768), device=gpu)
            self.occ_map[0] = 1.
            self.cam_pos = np.array([384, 384])
            self.occ_map[2, 383:386, 383:386] = 2.
            self.map_center = torch.from_numpy(pose[[0, 2], 3].numpy()).to(gpu)
            self.frame_idx = 0
    OccupancyOps.install(RefPlanner)
    ref_pl = RefPlanner()
    our_pl = planning.AstarPlanner(device=gpu, cell_size=0.05, frontier_select_method="largest", sample_view_num=16)
    our_pl.init(torch.eye(4), torch.from_numpy(np.asarray(K, dtype=np.float32)))
    for t, p in enumerate(synthetic.candidate_poses(6, 309).numpy().astype(np.float32)):
        d = room_depth(p, W, H, K)
        ref_pl.update_occ_map(d, torch.from_numpy(p).to(gpu), t)
        our_pl.update_occ_map(d, torch.from_numpy(p).to(gpu), t)
    assert torch.equal(ref_pl.occ_map, our_pl.occ_map)
    f1, s1 = ref_pl.build_frontiers(params["means3D"])
    f2, s2 = our_pl.build_frontiers(params["means3D"])
    assert np.array_equal(s1, s2) and ((f1 is None) == (f2 is None)) and (f1 is None or np.array_equal(f1, f2))
    torch.manual_seed(1)
    cand = ref_pl.generate_candidate(torch.zeros((3, 2), device=gpu))
    assert cand.shape == (16, 4, 4)
