"""Path-EIG evaluator: the planner's scoring of candidate action sequences
(tester_gaussians_navigation.py:1664-1727) on top of the batched Fisher scorer.

Reference loop, per path: roll the camera forward one action at a time (`compute_next_campos`,
models/SLAM/utils/slam_external.py:44-65), call `compute_Hessian` at EVERY step, and on every
`acc_H_train_every`-th step add `log sum(cur_H / (H_path + lambda))` to the path value and fold `cur_H` into `H_path`.
Only the accumulation steps influence the result, and step m of a path needs the cur_H of its steps < m, so the batched
form runs in rounds: round m scores the m-th accumulation step of ALL paths in one `fr_fisher_views` call with per-view
weights (`H_inv_view_stride`) and per-view `out_H` blocks, then updates every path's `H_path`.
"""
import numpy as np
import torch


def compute_next_campos(cam_H, action_id, forward_step_size=0.065, turn_angle=10.):
    """One agent action applied to a camera-to-world matrix (float64, like the reference).
    1: forward along +z of the camera; 2 / 3: yaw by -/+ turn_angle degrees about the camera's y axis; else no-op."""
    next_H = np.array(cam_H, dtype=np.float64, copy=True)
    if action_id == 1:
        next_H[:3, 3] = cam_H[:3, 3] + cam_H[:3, :3] @ np.array([0.0, 0.0, forward_step_size])
    elif action_id in (2, 3):
        a = np.deg2rad(turn_angle)
        s = -np.sin(a) if action_id == 2 else np.sin(a)
        R = np.array([[np.cos(a), 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, np.cos(a)]])
        next_H[:3, :3] = cam_H[:3, :3] @ R
    return next_H


def rollout(start_c2w, actions, forward_step_size=0.065, turn_angle=10.):
    H = np.array(start_c2w, dtype=np.float64)
    out = np.zeros((len(actions), 4, 4))
    for t, a in enumerate(actions):
        H = compute_next_campos(H, int(a), forward_step_size, turn_angle)
        out[t] = H
    return out


def evaluate_paths(scorer, start_c2w, path_actions, final_EIGs, H_train, *, forward_step_size=0.065, turn_angle=10.,
                   H_reg_lambda=0.1, acc_H_train_every=5, path_point_weight=1.0, path_pose_weight=0.0,
                   path_end_weight=0.0, vol_weighted_H=False, gs_pts_cnt=1.0, cam_height=None):
    """Returns the list of total_path_EIG values (tester 1713-1718).  `scorer` is a FisherScorer of the current map,
    `H_train` the [P,C] keyframe accumulator.  pose_H is the identity placeholder of the reference, so the pose term
    log(det(pose_H)) is zero whatever `path_pose_weight` is."""
    dev = scorer.dev
    P, C = scorer.P, scorer.columns
    start = np.array(start_c2w, dtype=np.float64, copy=True)
    if cam_height is not None:
        start[1, 3] = cam_height
    n_paths = len(path_actions)
    # accumulation steps: 1-based step s with (s + 1) % acc == 0
    acc_steps = [[s for s in range(1, len(a) + 1) if (s + 1) % acc_H_train_every == 0] for a in path_actions]
    poses = [rollout(start, a, forward_step_size, turn_angle) for a in path_actions]
    totals = [0.0] * n_paths
    H_path = {i: H_train.clone() for i in range(n_paths) if acc_steps[i]}
    rounds = max((len(s) for s in acc_steps), default=0)
    for m in range(rounds):
        active = [i for i in range(n_paths) if len(acc_steps[i]) > m]
        c2w = np.stack([poses[i][acc_steps[i][m] - 1] for i in active])
        w2c = torch.from_numpy(np.linalg.inv(c2w)).float().to(dev)
        H_inv = torch.stack([torch.reciprocal(H_path[i] + H_reg_lambda) for i in active])
        if vol_weighted_H:
            H_inv = H_inv / gs_pts_cnt
        last_round = {i: len(acc_steps[i]) == m + 1 for i in active}
        cur = torch.zeros((len(active), P, C), dtype=torch.float32, device=dev)
        res = scorer.run(w2c, H_inv=H_inv, H_inv_per_view=True, out_H=cur, out_H_per_view=True)
        point_EIG = torch.log(res["scores"]).cpu().numpy()
        for k, i in enumerate(active):
            totals[i] += path_point_weight * float(point_EIG[k])
            if not last_round[i]:
                H_path[i] = H_path[i] + cur[k]
    out = []
    for i in range(n_paths):
        n = max(len(path_actions[i]), 1)
        if path_end_weight > 0:
            out.append(totals[i] / n + path_end_weight * float(final_EIGs[i]))
        else:
            out.append((totals[i] + float(final_EIGs[i])) / n)
    return out
