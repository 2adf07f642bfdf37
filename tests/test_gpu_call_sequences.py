"""-m gpu: a FisherScorer is a long-lived object that a planner calls hundreds of times in every mode; it keeps state between the
calls (workspace, fr_fisher_cfg.reuse_static, the tile-segment sizes).  A seeded random sequence of calls on ONE scorer must give, call
by call, what a FRESH scorer gives for the same call: scores bit for bit (they are deterministic), diagonals to the order of their
float atomics."""
import numpy as np
import pytest
import torch

from scenes import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("columns,seed", [(4, 0), (4, 1), (11, 2)])
def test_random_call_sequences_equal_fresh_scorers(gpu, columns, seed):
    from fisher_rast import synthetic
    from fisher_rast.ops import FisherScorer
    from models.SLAM.utils.recon_helpers import setup_camera
    P, W, H, NV = 12000, 112, 80, 24
    act = {k: v.to(gpu) for k, v in synthetic.activate(synthetic.room_shell(P, seed=50 + seed)).items()}
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=gpu)
    w2c_all = synthetic.invert_rigid(synthetic.candidate_poses(NV, seed=60 + seed)).to(gpu)
    args = [act[k] for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")]
    sc = FisherScorer(cam, *args, columns=columns)
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    shared = [(torch.rand((P, columns), generator=g) * 2 + 0.05).to(gpu) for _ in range(2)]
    modes = ["score", "score", "score_same", "score_per_view", "outh", "outh_per_view", "both", "image"]
    last_shared = shared[0]
    for step in range(28):
        mode = modes[int(rng.integers(len(modes)))]
        V = int(rng.choice([1, 2, 3, 8, 9, 16]))
        v0 = int(rng.integers(0, NV - V + 1))
        w = w2c_all[v0:v0 + V]
        fresh = FisherScorer(cam, *args, columns=columns)
        kw = {}
        if mode == "score":
            last_shared = shared[int(rng.integers(2))]
            if rng.random() < 0.3:
                last_shared.mul_(1.5)                                   # in place: same tensor, new version
            kw = dict(H_inv=last_shared)
        elif mode == "score_same":
            kw = dict(H_inv=last_shared)
        elif mode == "score_per_view":
            kw = dict(H_inv=(torch.rand((V, P, columns), generator=g) + 0.05).to(gpu), H_inv_per_view=True)
        elif mode == "image":
            kw = dict(dL_image=(torch.randn((V, 3, H, W), generator=g) * 1e-3).to(gpu))
        elif mode == "both":
            kw = dict(H_inv=(torch.rand((V, P, columns), generator=g) + 0.05).to(gpu), H_inv_per_view=True)
        outs = []
        for s in (sc, fresh):
            k2 = dict(kw)
            if mode in ("outh", "image"):
                k2["out_H"] = torch.zeros((P, columns), device=gpu)
            elif mode in ("outh_per_view", "both"):
                k2["out_H"] = torch.zeros((V, P, columns), device=gpu)
                k2["out_H_per_view"] = True
            r = s.run(w, **k2)
            outs.append((r, k2.get("out_H")))
        (ra, Ha), (rb, Hb) = outs
        assert torch.equal(ra["vis_count"], rb["vis_count"]) and torch.equal(ra["num_rendered"], rb["num_rendered"]), (step, mode)
        if ra["scores"] is not None:
            if mode == "both":            # (the two-pass kernel sums its scores with atomics)
                assert rel_err(ra["scores"].cpu().numpy(), rb["scores"].cpu().numpy()) < 1e-5, (step, mode, V)
            else:
                assert torch.equal(ra["scores"], rb["scores"]), (step, mode, V, ra["scores"], rb["scores"])
            assert float(ra["scores"].min()) > 0
        if Ha is not None:
            assert float(Hb.abs().max()) > 0 and rel_err(Ha.cpu().numpy(), Hb.cpu().numpy()) < 1e-5, (step, mode, V)
