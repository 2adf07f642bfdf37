"""-m gpu: the power-2 backward through the rasteriser ABI cut into chunks (fr_backward_ws with a scratch buffer: k_backward_sq_slots /
_chunks / _prefix / _leaves -- at most 64 candidates of one 16 x 4 strip per piece of work, the state in front of a chunk from the
composed maps of the chunks behind it) against the single-pass walk (fr_backward: one workgroup per tile, k_backward_sq_walk) on the
same forward -- the same pairs contribute, a chunk's starting state differs by rounding only -- and against the oracle
(backward.cu:850-1140 restated, power 2)."""
import numpy as np
import pytest
import torch

from gpu_util import hip_forward, hip_backward, to_dev

pytestmark = pytest.mark.gpu
NAMES = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations")


def _both(gpu, cam, fwd, dL, power=2):
    from fisher_rast import ops
    t = fwd["tensors"]
    geom, binning, img = fwd["buffers"]
    out = {}
    for seg in (True, False):
        o = ops.rasterize_backward(t["bg"], t["means3D"], fwd["radii_t"], t["colors"], t["scales"], t["rotations"], cam.scale_modifier,
                                   t["cov3D"], t["view"], t["proj"], cam.tanfovx, cam.tanfovy, to_dev(dL, gpu), t["sh"], cam.sh_degree,
                                   t["campos"], geom, fwd["num_rendered"], binning, img, power, segmented=seg)
        torch.cuda.synchronize()
        out[seg] = {n: x.cpu().numpy().astype(np.float64) for n, x in zip(NAMES[:3] + ("dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"), o)}
    return out[True], out[False]


def _room(P, W, H, seed):
    from fisher_rast import synthetic
    act = synthetic.activate(synthetic.room_shell(P, seed))
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(3, seed))[1].numpy()
    pts = act["means3D"].numpy()
    tp = (w2c @ np.concatenate([pts, np.ones((P, 1), np.float32)], 1).T).T[:, :3].astype(np.float32)
    return dict(means3D=np.ascontiguousarray(tp), opacities=act["opacities"].numpy(), colors=act["rgb_colors"].numpy(),
                scales=act["scales"].numpy(), rotations=act["rotations"].numpy())


@pytest.mark.parametrize("power", [2, 1])
@pytest.mark.parametrize("P,W,H,bg", [(200_000, 256, 256, 0.0), (60_000, 128, 96, 0.3), (500_000, 256, 256, 0.0), (120_000, 250, 130, 1.0), (600_000, 800, 560, 0.0)])
def test_chunks_match_the_single_pass(gpu, oracle, P, W, H, bg, power):
    from fisher_rast.synthetic import intrinsics
    sc = _room(P, W, H, 2)
    cam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4, dtype=np.float32))._replace(bg=np.full(3, bg, np.float32))
    fwd = hip_forward(gpu, cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
    T = ((W + 15) // 16) * ((H + 15) // 16)
    longest = int(fwd["tile_count"].max())
    L = max(256, -(-(-(-fwd["num_rendered"] // (3 * T))) // 64) * 64)
    print(f"P={P} {W}x{H}: {fwd['num_rendered']} tile instances, longest list {longest}, segment length {L}: {-(-longest // L)} segments in the longest")
    assert longest > 2 * L, "the scene does not cut any list"
    rng = np.random.default_rng(3)
    dL = (rng.normal(size=(3, H, W)) * 1e-3).astype(np.float32)
    if power == 1:
        dL = rng.normal(size=(3, H, W)).astype(np.float32)
    seg, one = _both(gpu, cam, fwd, dL, power)
    for n in NAMES:
        a, b = seg[n], one[n]
        assert np.abs(b).max() > 0, n
        if power == 2:
            err = np.abs(a - b) / (np.abs(b) + 1e-6 * np.abs(b).max())
            assert err.max() < 2e-5, (n, float(err.max()), int((err > 2e-5).sum()))
            assert ((a != 0) == (b != 0)).all(), n
        else:
            # signed sums (cancellation): against the magnitude of the entry plus a sliver of the tensor's scale, as test_backward_parity[1-*] does
            tol = 1e-4 * np.abs(b) + 2e-5 * np.abs(b).max()
            assert (np.abs(a - b) <= tol).all(), (n, float((np.abs(a - b) / tol).max()))


def test_chunks_against_the_oracle(gpu, oracle):
    """A scene small enough for the oracle's backward whose lists are still cut (96 x 64, 24 tiles; L = 256)."""
    from fisher_rast.synthetic import intrinsics
    W, H, P = 96, 64, 30_000
    sc = _room(P, W, H, 5)
    cam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4, dtype=np.float32))
    args = dict(colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
    want = oracle.rasterize_forward(cam, sc["means3D"], sc["opacities"], **args)
    got = hip_forward(gpu, cam, sc["means3D"], sc["opacities"], **args)
    assert got["num_rendered"] == want["num_rendered"] and int(got["tile_count"].max()) > 3 * 256
    dL = np.full((3, H, W), 1e-3, np.float32)
    gw = oracle.rasterize_backward(cam, want, dL, 2)
    gg = hip_backward(gpu, cam, got, dL, 2)
    for n in NAMES:
        o = gw[n].astype(np.float64).reshape(P, -1)
        g = gg[n].astype(np.float64).reshape(P, -1)
        tol = 1e-4 * np.abs(o) + 1e-6 * np.abs(o).max()
        bad = np.abs(g - o) > tol
        assert not bad.any(), (n, int(bad.sum()), float((np.abs(g - o) / np.maximum(tol, 1e-300)).max()))


@pytest.mark.parametrize("P,W,H,bg", [(200_000, 256, 256, 0.0), (120_000, 250, 130, 0.7)])
def test_pair_chunks_match_the_tile_kernel(gpu, oracle, P, W, H, bg):
    """fr_backward_pair_ws (colour image + a second feature image, six channels in the chunk maps) against fr_backward_pair's two-pass tile
    kernel on the same forward: every output within 1e-4 of the entry plus a sliver of the tensor's scale (signed sums)."""
    from fisher_rast import ops
    from fisher_rast.synthetic import intrinsics
    sc = _room(P, W, H, 7)
    cam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4, dtype=np.float32))._replace(bg=np.full(3, bg, np.float32))
    fwd = hip_forward(gpu, cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
    assert int(fwd["tile_count"].max()) > 2 * 256
    rng = np.random.default_rng(11)
    feats = to_dev(rng.uniform(0, 1, (P, 3)).astype(np.float32), gpu)
    dLa, dLb = to_dev(rng.normal(size=(3, H, W)).astype(np.float32), gpu), to_dev(rng.normal(size=(3, H, W)).astype(np.float32), gpu)
    t = fwd["tensors"]
    geom, binning, img = fwd["buffers"]
    out = {}
    for seg in (True, False):
        o = ops.rasterize_backward_pair(t["bg"], t["means3D"], fwd["radii_t"], t["colors"], feats, t["scales"], t["rotations"], cam.scale_modifier,
                                        t["cov3D"], t["view"], t["proj"], cam.tanfovx, cam.tanfovy, dLa, dLb, t["campos"], geom, binning, img,
                                        num_rendered=fwd["num_rendered"], segmented=seg)
        torch.cuda.synchronize()
        out[seg] = [x.cpu().numpy().astype(np.float64) for x in o]
    names = ("means2D", "means2D_features", "colors", "features", "opacity", "means3D", "cov3D", "scales", "rotations")
    for n, a, b in zip(names, out[True], out[False]):
        assert np.abs(b).max() > 0, n
        tol = 1e-4 * np.abs(b) + 2e-5 * np.abs(b).max()
        assert (np.abs(a - b) <= tol).all(), (n, float((np.abs(a - b) / tol).max()))


def test_a_scratch_laid_out_for_too_few_instances_gives_zero_gradients(gpu, oracle):
    """fr_backward_ws with num_rendered below what the forward reported: the chunk kernels notice (status[0] on the device) and leave the
    gradients zero instead of writing past the slots the scratch has."""
    import ctypes
    from fisher_rast import ops, _lib
    from fisher_rast.synthetic import intrinsics
    P, W, H = 60_000, 128, 96
    sc = _room(P, W, H, 2)
    cam = oracle.setup_camera(W, H, intrinsics(W, H), np.eye(4, dtype=np.float32))
    fwd = hip_forward(gpu, cam, sc["means3D"], sc["opacities"], colors_precomp=sc["colors"], scales=sc["scales"], rotations=sc["rotations"])
    t = fwd["tensors"]
    geom, binning, img = fwd["buffers"]
    dL = to_dev(np.full((3, H, W), 1e-3, np.float32), gpu)
    for claimed, expect_zero in ((fwd["num_rendered"] // 4, True), (fwd["num_rendered"], False)):
        o = ops.rasterize_backward(t["bg"], t["means3D"], fwd["radii_t"], t["colors"], t["scales"], t["rotations"], cam.scale_modifier, t["cov3D"], t["view"],
                                   t["proj"], cam.tanfovx, cam.tanfovy, dL, t["sh"], cam.sh_degree, t["campos"], geom, claimed, binning, img, 2)
        torch.cuda.synchronize()
        assert bool((o[3] == 0).all()) == expect_zero
