"""`params{t}.npz` round trip (common_utils.py:28-59, tester_gaussians_navigation.py:2745-2760).
The fixture tests/golden/reference_params7.npz was written by the reference's own `save_params_ckpt`
(tests/golden/make_reference_vectors.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

FIXTURE = os.path.join(ROOT, "tests", "golden", "reference_params7.npz")
PARAM_KEYS = ['cam_trans', 'cam_unnorm_rots', 'log_scales', 'logit_opacities', 'means3D', 'rgb_colors', 'unnorm_rotations']


def test_reads_a_checkpoint_written_by_the_reference(tmp_path):
    from models.SLAM.gaussian import GaussianSLAM
    ref = dict(np.load(FIXTURE, allow_pickle=True))
    np.save(tmp_path / "keyframe_time_indices7.npy", np.array([0, 3, 5]))
    ck = tmp_path / "params7.npz"
    ck.write_bytes(open(FIXTURE, "rb").read())
    slam = GaussianSLAM(params=str(ck), device="cpu")
    assert sorted(slam.params.keys()) == PARAM_KEYS                      # "Uncertainty" / "occ_map" kept aside (tester 2750)
    for k in PARAM_KEYS:
        assert slam.params[k].dtype == torch.float32
        assert np.array_equal(slam.params[k].numpy(), ref[k].astype(np.float32))
    assert sorted(slam.checkpoint_extras.keys()) == ["Uncertainty", "occ_map"]
    n = ref["means3D"].shape[0]
    for k in ('max_2D_radius', 'means2D_gradient_accum', 'denom', 'timestep'):   # tester 2752-2755
        assert slam.variables[k].shape == (n,) and float(slam.variables[k].abs().sum()) == 0.0
    assert slam.cur_frame_idx == 7
    assert slam.keyframe_time_indices == [0, 3, 5]                      # tester 2759-2760


def test_writes_what_the_reference_writes(tmp_path):
    from models.SLAM.gaussian import GaussianSLAM
    from models.SLAM.utils import common_utils as cu
    ref = dict(np.load(FIXTURE, allow_pickle=True))
    slam = GaussianSLAM(params={k: ref[k] for k in PARAM_KEYS}, device="cpu")
    path = slam.save_params_ckpt(str(tmp_path), 7, Uncertainty=torch.tensor(ref["Uncertainty"]), occ_map=ref["occ_map"])
    assert os.path.basename(path) == "params7.npz"
    got = dict(np.load(path, allow_pickle=True))
    assert sorted(got.keys()) == sorted(ref.keys())
    for k in ref:
        assert got[k].dtype == ref[k].dtype and got[k].shape == ref[k].shape and np.array_equal(got[k], ref[k]), k
    # final-map form and the file-name convention
    final = slam.save_params_ckpt(str(tmp_path))
    assert os.path.basename(final) == "params.npz" and sorted(np.load(final).keys()) == PARAM_KEYS
    assert cu.checkpoint_time_idx("/a/b/params123.npz") == 123 and cu.checkpoint_time_idx("params.npz") is None
