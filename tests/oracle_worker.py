"""TEST INFRASTRUCTURE: one oracle process of the full-size parity test (tests/test_gpu_fullsize_properties.py starts one per
host core with subprocess).  Scene = fisher_rast.synthetic.room_shell(P, seed), poses = candidate_poses(n_poses, pose_seed).

    python oracle_worker.py hessian P seed W H n_poses pose_seed v0 v1 out.npy            -> cur_H[v1 - v0, P, 4] (float32)
    python oracle_worker.py scores  P seed W H n_poses pose_seed v0 v1 out.npy H_inv.npy  -> scores[v1 - v0] (float64),
                                                                                             vis_count, num_rendered
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np   # noqa: E402


def main():
    mode = sys.argv[1]
    P, seed, W, H, n_poses, pose_seed, v0, v1 = (int(x) for x in sys.argv[2:10])
    out = sys.argv[10]
    import torch
    torch.set_num_threads(1)
    from fisher_rast import synthetic
    from oracle import ref
    act = {k: v.numpy() for k, v in synthetic.activate(synthetic.room_shell(P, seed)).items()}
    args = (act["means3D"], act["rgb_colors"], act["rotations"], act["opacities"], act["scales"])
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(n_poses, pose_seed)).numpy()
    cam = ref.setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4))
    if mode == "hessian":
        np.save(out, np.stack([ref.compute_hessian(cam, w2c[v], *args, columns=4)[0] for v in range(v0, v1)]))
        return
    H_inv = np.load(sys.argv[11]).astype(np.float64)
    rows = []
    for v in range(v0, v1):
        cur_H, vis, fwd, _ = ref.compute_hessian(cam, w2c[v], *args, columns=4, return_all=True)
        rows.append((float(np.sum(cur_H.astype(np.float64) * H_inv)), float(vis), float(fwd["num_rendered"])))
    np.save(out, np.asarray(rows, dtype=np.float64))


if __name__ == "__main__":
    main()
