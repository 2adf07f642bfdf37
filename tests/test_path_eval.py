"""Path-EIG evaluator.  CPU: the pose roll-out against vectors produced by running the reference's own
`compute_next_campos` (tests/golden/reference_pose_helpers.npz, made by tests/golden/make_reference_vectors.py).
GPU: the batched evaluator against the reference's serial loop (tester 1664-1727) restated on the oracle."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_rollout_matches_reference_vectors():
    from fisher_rast.path_eval import rollout
    g = np.load(os.path.join(HERE, "golden", "reference_pose_helpers.npz"))
    for si, (fs, ta) in enumerate(g["steps"]):
        for i in range(g["poses"].shape[0]):
            got = rollout(g["poses"][i], g["actions"][i], fs, ta)
            assert np.abs(got - g[f"traj_{si}"][i]).max() < 1e-12


def serial_path_eig(oracle, cam, args, start, actions, final_EIG, H_train, lam, acc, w_point, w_end, fs=0.065, ta=10.0):
    """The reference loop, one compute_Hessian per step (tester 1684-1716)."""
    from fisher_rast.path_eval import compute_next_campos
    H_path = H_train.copy()
    pose = np.array(start, dtype=np.float64)
    total = 0.0
    done = []
    for a in actions:
        pose = compute_next_campos(pose, int(a), fs, ta)
        w2c = np.linalg.inv(pose).astype(np.float32)
        cur_H, _ = oracle.compute_hessian(cam, w2c, *args)
        point_EIG = np.log(np.sum(cur_H.astype(np.float64) / (H_path.astype(np.float64) + lam)))
        done.append(a)
        if (len(done) + 1) % acc == 0:
            total += w_point * point_EIG
            H_path = H_path + cur_H
    if w_end > 0:
        return total / len(done) + w_end * final_EIG
    return (total + final_EIG) / len(done)


@pytest.mark.gpu
def test_batched_paths_match_serial_reference_loop(gpu, oracle):
    import torch
    from fisher_rast import synthetic
    from fisher_rast.ops import FisherScorer
    from fisher_rast.path_eval import evaluate_paths
    from models.SLAM.utils.recon_helpers import setup_camera
    P, W, H = 3000, 96, 96
    act = synthetic.activate(synthetic.room_shell(P, seed=9))
    a = {k: v.numpy() for k, v in act.items()}
    args = (a["means3D"], a["rgb_colors"], a["rotations"], a["opacities"], a["scales"])
    K = synthetic.intrinsics(W, H)
    cam = setup_camera(W, H, K, np.eye(4), device=gpu)
    ocam = oracle.setup_camera(W, H, K, np.eye(4))
    sc = FisherScorer(cam, *(act[k].to(gpu) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")))
    kf = synthetic.invert_rigid(synthetic.candidate_poses(2, seed=10))
    H_train = oracle.compute_h_train(ocam, kf.numpy(), *args)
    start = synthetic.candidate_poses(1, seed=11)[0].numpy().astype(np.float64)
    rng = np.random.default_rng(3)
    paths = [list(rng.integers(1, 4, size=n)) for n in (9, 4, 3, 7)]     # one path has no accumulation step at acc=4
    finals = [0.3, -0.1, 0.7, 0.2]
    for (acc, w_end, lam) in ((4, 0.0, 0.1), (2, 0.5, 1e-6)):
        got = evaluate_paths(sc, start, paths, finals, torch.from_numpy(H_train).to(gpu), H_reg_lambda=lam,
                             acc_H_train_every=acc, path_point_weight=1.0, path_end_weight=w_end)
        want = [serial_path_eig(oracle, ocam, args, start, p, f, H_train, lam, acc, 1.0, w_end) for p, f in zip(paths, finals)]
        assert np.allclose(got, want, rtol=2e-4, atol=1e-6), (got, want)
