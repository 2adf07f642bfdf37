"""-m gpu: kernel generations and record layouts of fr_fisher_views' out_H / score modes on the same scene, each in a fresh
process (FR_DEBUG_MODE is read once per process): k_fisher_tile_v3h on compact records (default), on dense records
(FR_DEBUG_MODE=19), and the two-pass kernel of round 1 (FR_DEBUG_MODE=9, still used for 11 columns and gradient images), held
to the oracle's compute_hessian; and the scores of the two record layouts against each other, bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

from gpu_util import assert_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, V, W, H, SEED = 2500, 3, 96, 80, 17


@pytest.fixture(scope="module")
def want(oracle):
    from fisher_rast import synthetic
    act = {k: v.numpy() for k, v in synthetic.activate(synthetic.room_shell(P, seed=SEED)).items()}
    cam = oracle.setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4))
    w2cs = synthetic.invert_rigid(synthetic.candidate_poses(V, seed=SEED)).numpy()
    return np.stack([oracle.compute_hessian(cam, w, act["means3D"], act["rgb_colors"], act["rotations"], act["opacities"],
                                            act["scales"])[0] for w in w2cs])


@pytest.mark.parametrize("mode,what", [("0", "k_fisher_tile_v3h"), ("9", "two-pass kernel of round 1"), ("19", "k_fisher_tile_v3h on dense records")])
def test_forced_out_h_kernel_matches_the_oracle(gpu, want, tmp_path, mode, what):
    out = str(tmp_path / f"h_{mode}.npy")
    env = dict(os.environ, FR_DEBUG_MODE=mode)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "outh_variant_probe.py"), out, str(P), str(V), str(W), str(H), str(SEED)],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)
    assert got.shape == want.shape
    for v in range(V):
        assert_close(got[v], want[v], 1e-4, f"{what}: cur_H[{v}]", atol_frac=1e-7)


def test_compact_and_dense_records_give_the_same_scores_bit_for_bit(gpu, tmp_path):
    """Keys that carry the slot (compact records, the default) must sort exactly like keys that carry the Gaussian index
    (FR_DEBUG_MODE=19: dense [V, P] records): the same pairs in the same order through the same arithmetic, summed in a fixed
    order -- identical float32 scores."""
    scores = {}
    for mode in ("0", "19"):
        out = str(tmp_path / f"s_{mode}.npy")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "outh_variant_probe.py"), out, "20000", "5", "200", "136", "23"],
                           cwd=ROOT, env=dict(os.environ, FR_DEBUG_MODE=mode), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        scores[mode] = np.load(out.replace(".npy", "_scores.npy"))
    assert scores["0"].shape == (5,) and np.isfinite(scores["0"]).all() and (scores["0"] > 0).all()
    assert np.array_equal(scores["0"], scores["19"]), (scores["0"], scores["19"])
