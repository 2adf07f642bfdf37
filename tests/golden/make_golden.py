#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

PROVENANCE: these vectors are produced by the CPU ORACLE of this repository (oracle/fisher_oracle.c through
oracle/ref.py), NOT by the CUDA reference -- the reference ships no fixtures for this path and its rasteriser cannot be
built or run in this pipeline (no nvcc, no NVIDIA device, GLM submodule empty).  They pin the oracle
against accidental change and give the GPU tests a data-only target that needs neither the oracle build nor
/root/reference.  Inputs are seeded NumPy; run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "fisher-nerf-customized_amd"))

from oracle import ref           # noqa: E402
from scenes import random_scene, intrinsics   # noqa: E402


def raster_case(name, W, H, P, seed, w2c):
    sc = random_scene(P, seed, zmin=-0.3, zmax=6.0, spread=1.4, scale=0.07)
    sc["means3D"][:5, 2] = [0.002, 0.01, 0.05, 0.1, 0.19]     # near-camera splats (appendix B.1)
    sc["means3D"][5:8] = sc["means3D"][8]                      # duplicates: equal-depth ties (B.6)
    sc["scales"][5:8] = sc["scales"][8]; sc["rotations"][5:8] = sc["rotations"][8]
    cam = ref.setup_camera(W, H, intrinsics(W, H), w2c)
    fwd = ref.rasterize_forward(cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
    rng = np.random.default_rng(seed + 1000)
    dL1 = rng.normal(size=(3, H, W)).astype(np.float32)
    g1 = ref.rasterize_backward(cam, fwd, dL1, 1)
    g2 = ref.rasterize_backward(cam, fwd, np.full((3, H, W), 1e-3, np.float32), 2)
    out = dict(W=W, H=H, w2c=w2c, viewmatrix=cam.viewmatrix, projmatrix=cam.projmatrix, campos=cam.campos,
               tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, dL1=dL1, **{f"in_{k}": v for k, v in sc.items()})
    for k in ("radii", "means2D", "depths", "conic_opacity", "cov3D", "point_list", "ranges", "color", "depth", "final_T", "n_contrib"):
        out[f"fwd_{k}"] = fwd[k]
    out["num_rendered"] = fwd["num_rendered"]
    for k in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations", "dL_dconic"):
        out[f"p1_{k}"] = g1[k]
        out[f"p2_{k}"] = g2[k]
    out["pairs"] = g2["pair_count"]
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, "P", P, "R", fwd["num_rendered"], "pairs", g2["pair_count"])


def fisher_case(name, P, V, Kf, W, H, seed):
    from fisher_rast import synthetic
    act = {k: v.numpy() for k, v in synthetic.activate(synthetic.room_shell(P, seed)).items()}
    args = (act["means3D"], act["rgb_colors"], act["rotations"], act["opacities"], act["scales"])
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, seed)).numpy()
    kf = synthetic.invert_rigid(synthetic.candidate_poses(Kf, seed + 100)).numpy()
    cam = ref.setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4))
    out = dict(W=W, H=H, w2c=w2c, kf_w2c=kf, **{f"in_{k}": v for k, v in act.items()})
    for C in (4, 11):
        Ht = ref.compute_h_train(cam, kf, *args, columns=C)
        s, vis = ref.pose_eval(cam, w2c, Ht, *args, columns=C)
        H0, _ = ref.compute_hessian(cam, w2c[0], *args, columns=C)
        out[f"H_train_{C}"] = Ht; out[f"scores_{C}"] = s; out[f"vis_{C}"] = vis; out[f"curH0_{C}"] = H0
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, "scores", out["scores_4"])


if __name__ == "__main__":
    w = np.eye(4, dtype=np.float32)
    w[:3, :3] = np.array([[0.96, 0, 0.28], [0, 1, 0], [-0.28, 0, 0.96]], np.float32)
    w[:3, 3] = [0.2, -0.1, 0.25]
    raster_case("raster_64x48.npz", 64, 48, 500, 21, w)
    raster_case("raster_50x70_identity.npz", 50, 70, 300, 22, np.eye(4, dtype=np.float32))
    fisher_case("fisher_room_2k.npz", 2000, 4, 2, 96, 96, 23)
