"""-m gpu: the per-tile key sort at the sizes where its tiers change hands (one wave with 1 / 2 / 4 / 8 keys per lane, four
waves with 4 / 8 / 16, sixteen waves with 8 / 16, the global-memory network beyond 16384), with tied depths.  Every Gaussian of
a scene lands in one tile (a second tile gets a third as many), so the segment length is exactly n; the expected order is
(depth bits, index) ascending -- what the reference's stable radix sort over (tile | depth) keys leaves
(rasterizer_impl.cu:70-111, 335-345)."""
import numpy as np
import pytest

from gpu_util import hip_forward

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1000, 1023, 1024, 1025, 2047, 2048, 2049,
         3000, 4095, 4096, 4097, 8191, 8192, 8193, 12000, 16383, 16384, 16385, 20000]


def _scene(n, seed, distinct=None):
    from fisher_rast.synthetic import intrinsics
    W, H = 32, 16
    K = np.asarray(intrinsics(W, H), np.float64)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    rng = np.random.default_rng(seed)
    m = n // 3
    P = n + m
    px = np.concatenate([rng.uniform(4.0, 11.0, n), rng.uniform(20.0, 27.0, m)])      # tile 0, tile 1
    py = rng.uniform(4.0, 11.0, P)
    # ~4 Gaussians per depth value: ties; `distinct`: only that many depth values in all (long runs of equal depth, ordered by index)
    z = rng.choice(np.linspace(1.0, 3.0, distinct if distinct else max(2, P // 4)), P).astype(np.float32)
    means = np.stack([(px - cx) / fx * z, (py - cy) / fy * z, z], 1).astype(np.float32)
    return W, H, dict(means3D=means, opacities=np.full((P, 1), 0.01, np.float32),
                      colors=rng.uniform(0, 1, (P, 3)).astype(np.float32),
                      scales=np.full((P, 3), 1e-4, np.float32),
                      rotations=np.tile(np.array([1, 0, 0, 0], np.float32), (P, 1)))


def _check(gpu, oracle, n, distinct=None):
    from fisher_rast.synthetic import intrinsics
    W, H, sc = _scene(n, 1000 + n, distinct)
    cam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4, dtype=np.float32))
    got = hip_forward(gpu, cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
    m = n // 3
    assert got["num_rendered"] == n + m and (got["radii"] > 0).all()
    rngs = got["ranges"]
    assert rngs[0, 1] - rngs[0, 0] == n and (m == 0 or rngs[1, 1] - rngs[1, 0] == m)
    dbits = got["depths"].view(np.uint32).astype(np.uint64)
    for t, members in ((0, np.arange(n)), (1, np.arange(n, n + m))):
        if len(members) == 0:
            continue
        want_keys = np.sort((dbits[members] << np.uint64(32)) | members.astype(np.uint64))
        seg = got["keys"][rngs[t, 0]:rngs[t, 1]]
        assert np.array_equal(seg, want_keys), (n, t, int((seg != want_keys).sum()))
    assert len(np.unique(got["depths"])) < n + m or n < 3          # the scene does contain ties


@pytest.mark.parametrize("n", SIZES)
def test_segment_of_n_keys_is_sorted_by_depth_then_index(gpu, oracle, n):
    _check(gpu, oracle, n)


@pytest.mark.parametrize("n,distinct", [(2500, 1), (3000, 7), (4096, 40), (4097, 3), (6000, 64), (8192, 100), (8192, 2), (5000, 1200), (9000, 5)])
def test_clustered_depths(gpu, oracle, n, distinct):
    """Depths that do NOT spread: a handful of distinct values (long runs of equal depth, which must come out in index order) -- or,
    with ONE value, keys that differ in their low words only."""
    _check(gpu, oracle, n, distinct)
