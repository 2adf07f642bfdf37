"""-m gpu: every kernel generation the PRODUCT library still holds behind fr_fisher_views, each forced by the inputs that select it
(the product build reads no environment variable; the FR_DEBUG_MODE switches of earlier rounds live in the -DFR_AB rig only):

  default                           out_H, 4 columns             k_fisher_tile_v3h on compact records
  H_inv AND out_H in one launch     (path evaluation)            k_preprocess_views<0> + k_fisher_tile_v2<C, true, true>
  image beyond 4096 tiles           out_H 4 / 11 columns, scores k_preprocess + k_fisher_records (dense [V, P] records) + k_fisher_tile_v3h /
                                                                 k_fisher_tile_v2<11, false, true> / k_fisher_tile_v3<16, 4, false>
  (packed key lists, tile_capacity 0: tests/test_gpu_tile_segments.py)

all held to the oracle's compute_hessian (entries 1e-4 + 1e-7 max, scores 1e-4)."""
import numpy as np
import pytest
import torch

from gpu_util import assert_close

pytestmark = pytest.mark.gpu
P, V, W, H, SEED = 2500, 3, 96, 80, 17


def _scene(gpu, oracle, P, V, W, H, seed, columns):
    from fisher_rast import synthetic
    from fisher_rast.ops import FisherScorer
    from models.SLAM.utils.recon_helpers import setup_camera
    act = synthetic.activate(synthetic.room_shell(P, seed=seed))
    a = {k: v.numpy() for k, v in act.items()}
    ocam = oracle.setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4))
    w2cs = synthetic.invert_rigid(synthetic.candidate_poses(V, seed=seed))
    want = np.stack([oracle.compute_hessian(ocam, w, a["means3D"], a["rgb_colors"], a["rotations"], a["opacities"], a["scales"],
                                            columns=columns)[0] for w in w2cs.numpy()])
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=gpu)
    sc = FisherScorer(cam, *(act[k].to(gpu) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")), columns=columns)
    return sc, w2cs.to(gpu), want


@pytest.mark.parametrize("columns", [4, 11])
def test_out_h_and_scores_in_one_launch(gpu, oracle, columns):
    """H_inv and out_H together (fisher_rast/path_eval.py: per-view weights, per-view diagonals): the two-pass kernel
    k_fisher_tile_v2<C, true, true> behind the record-less multi-view front end."""
    sc, w2c, want = _scene(gpu, oracle, P, V, W, H, SEED, columns)
    Hi = (torch.rand((P, columns), generator=torch.Generator().manual_seed(3)) * 2.0 + 0.05)
    out = torch.zeros((V, P, columns), device=gpu)
    r = sc.run(w2c, H_inv=Hi.to(gpu), out_H=out, out_H_per_view=True)
    got = out.cpu().numpy()
    for v in range(V):
        assert_close(got[v], want[v], 1e-4, f"two-pass kernel, {columns} columns: cur_H[{v}]", atol_frac=1e-7)
    ws = (want.astype(np.float64) * Hi.double().numpy()[None]).sum(axis=(1, 2))
    assert np.all(np.abs(r["scores"].cpu().double().numpy() - ws) <= 1e-4 * np.abs(ws))
    # ... and against the default kernels (k_fisher_tile_v3h / v3g for the diagonal, k_fisher_tile_v3 for the scores)
    out2 = torch.zeros((V, P, columns), device=gpu)
    sc.run(w2c, out_H=out2, out_H_per_view=True)
    s2 = sc.run(w2c, H_inv=Hi.to(gpu))["scores"].cpu().double().numpy()
    for v in range(V):
        assert_close(out2[v].cpu().numpy(), want[v], 1e-4, f"default kernel, {columns} columns: cur_H[{v}]", atol_frac=1e-7)
    assert np.all(np.abs(s2 - ws) <= 1e-4 * np.abs(ws))


@pytest.mark.parametrize("columns", [4, 11])
def test_images_beyond_4096_tiles_use_the_single_view_front_end(gpu, oracle, columns):
    """65 x 65 tiles: no LDS tile histogram -> k_preprocess (radii, dense [V, P] records rewritten by k_fisher_records beside the
    sorts), k_scan_tiles + k_scatter_keys, then k_fisher_tile_v3h (4 columns) / k_fisher_tile_v2<11, false, true> for the diagonal
    and k_fisher_tile_v3<16, 4, false> on the dense records for the scores."""
    sc, w2c, want = _scene(gpu, oracle, 3000, 2, 1040, 1040, 8, columns)
    assert sc.tiles == 65 * 65
    out = torch.zeros((2, 3000, columns), device=gpu)
    sc.run(w2c, out_H=out, out_H_per_view=True)
    for v in range(2):
        assert_close(out[v].cpu().numpy(), want[v], 1e-4, f"{columns} columns, 4225 tiles: cur_H[{v}]", atol_frac=1e-7)
    Hi = (torch.rand((3000, columns), generator=torch.Generator().manual_seed(5)) * 2.0 + 0.05)
    s = sc.run(w2c, H_inv=Hi.to(gpu))["scores"].cpu().double().numpy()
    ws = (want.astype(np.float64) * Hi.double().numpy()[None]).sum(axis=(1, 2))
    assert np.all(np.abs(s - ws) <= 1e-4 * np.abs(ws)), (s, ws)
