"""CPU-only: the threshold scene of tests/test_gpu_scorer_adversarial.py is only a test if it really sits on the thresholds
of forward.cu:347-363 -- checked here with the oracle's forward pass."""
import numpy as np

from scenes import intrinsics
from test_gpu_scorer_adversarial import threshold_scene


def test_threshold_scene_really_sits_on_the_thresholds(oracle):
    """The construction above is only a test if pairs do land within 1e-6 of 1/255 and a pixel's T within 1e-6 of 1e-4."""
    W, H, sc, _ = threshold_scene(oracle)
    cam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4))
    fwd = oracle.rasterize_forward(cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors"], scales=sc["scales"],
                                   rotations=sc["rotations"])
    vis = np.nonzero(fwd["radii"] > 0)[0]
    xy = fwd["means2D"][vis].astype(np.float64); co = fwd["conic_opacity"][vis].astype(np.float64)
    near = 0
    for (x, y), (cx, cy, cz, o) in zip(xy, co):
        px = np.arange(max(0, int(x) - 4), min(W, int(x) + 5)); py = np.arange(max(0, int(y) - 4), min(H, int(y) + 5))
        dx = x - px[None, :]; dy = y - py[:, None]
        a = o * np.exp(-0.5 * (cx * dx * dx + cz * dy * dy) - cy * dx * dy)
        near += int((np.abs(a * 255.0 - 1.0) < 2e-6).sum())
    assert near >= 50
    T = fwd["final_T"]
    assert (np.abs(T / 1e-4 - 1.0) < 1e-3).sum() >= 1 and (T < 1e-4).sum() == 0
