"""-m gpu: simple-knn distCUDA2 against the brute-force oracle (parity unpinned against upstream: the submodule is
not vendored in the reference; semantics = mean squared distance to the 3 nearest other points)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,seed", [(1, 0), (3, 0), (5, 1), (1000, 2), (5000, 3), (40000, 4)])
def test_knn_exact(gpu, oracle, P, seed):
    import simple_knn._C as knn
    rng = np.random.default_rng(seed)
    pts = rng.normal(size=(P, 3)).astype(np.float32)
    if P >= 1000:
        pts[: P // 4] *= 0.01           # a dense cluster
        pts[P // 2: P // 2 + 50] = pts[0]   # exact duplicates -> zero distances
    got = knn.distCUDA2(torch.from_numpy(pts).to(gpu)).cpu().numpy()
    if P <= 5000:
        want = oracle.knn_dist2(pts)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    else:
        from scipy.spatial import cKDTree
        d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
        want = (d[:, 1:] ** 2).mean(1)
        assert np.allclose(got, want, rtol=1e-5, atol=1e-12)


def test_knn_empty(gpu):
    import simple_knn._C as knn
    assert knn.distCUDA2(torch.zeros((0, 3), device=gpu)).shape == (0,)
