"""CPU, world_size = 2, gloo: the view-sharding / all-gather / all-reduce layer (fisher_rast/distributed.py) with the
oracle standing in for the per-rank scorer.  Covers uneven shards and the empty-shard case."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, V, K, out_dir):
    for p in (ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fisher_rast import distributed as D, synthetic
        from oracle import ref
        P, W, H = 1500, 64, 64
        act = {k: v.numpy() for k, v in synthetic.activate(synthetic.room_shell(P, seed=5)).items()}
        args = (act["means3D"], act["rgb_colors"], act["rotations"], act["opacities"], act["scales"])
        cam = ref.setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4))
        w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, seed=6))
        kf = synthetic.invert_rigid(synthetic.candidate_poses(K, seed=7))
        calls = []

        def accumulate(w, Hacc):
            calls.append(("kf", int(w.shape[0])))
            for m in w:
                Hacc += torch.from_numpy(ref.compute_hessian(cam, m.numpy(), *args)[0])

        H_train = D.sharded_h_train(accumulate, kf, torch.zeros((P, 4)))
        H_inv = torch.reciprocal(H_train + 0.1)

        def score(w):
            calls.append(("views", int(w.shape[0])))
            s, _ = ref.pose_eval(cam, w.numpy(), H_train.numpy(), *args)
            return torch.from_numpy(s).float()

        n_points = P

        class OracleScorer:                     # the slice of FisherScorer's surface that sharded_point_score_max uses
            P, columns = n_points, 4

            def run(self, w, out_H=None, out_H_per_view=False):
                for i, m in enumerate(w):
                    out_H[i] += torch.from_numpy(ref.compute_hessian(cam, m.numpy(), *args)[0])

        best = D.sharded_point_score_max(OracleScorer(), w2c, H_inv, chunk=2)
        scores = D.sharded_scores(score, w2c)
        lo, hi = D.shard_bounds(V, rank, world)
        assert [c for c in calls if c[0] == "views"] == ([("views", hi - lo)] if hi > lo else [])
        torch.save(dict(scores=scores, H_train=H_train, lo=lo, hi=hi, best=best), os.path.join(out_dir, f"r{rank}.pt"))
        if rank == 0:
            Hs = ref.compute_h_train(cam, kf.numpy(), *args)
            s, _ = ref.pose_eval(cam, w2c.numpy(), Hs, *args)
            # gaussian.py:1284-1303: running max over the views of the per-point score, starting from zeros
            mx = torch.zeros((P,))
            for m in w2c:
                cur = torch.from_numpy(ref.compute_hessian(cam, m.numpy(), *args)[0])
                mx = torch.maximum(mx, (cur * H_inv).sum(dim=1))
            torch.save(dict(scores=torch.from_numpy(s).float(), H_train=torch.from_numpy(Hs), best=mx), os.path.join(out_dir, "serial.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("V,K", [(5, 3), (1, 1)])
def test_sharded_pose_eval_world2(tmp_path, V, K):
    port = 29500 + (os.getpid() % 2000) + V
    mp.spawn(_worker, args=(2, port, V, K, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    serial = torch.load(tmp_path / "serial.pt")
    assert torch.equal(r0["scores"], r1["scores"]) and r0["scores"].shape == (V,)
    assert torch.equal(r0["H_train"], r1["H_train"])
    assert (r0["lo"], r0["hi"], r1["lo"], r1["hi"]) == ((0, 3, 3, 5) if V == 5 else (0, 1, 1, 1))
    assert torch.allclose(r0["H_train"], serial["H_train"], rtol=1e-5, atol=1e-12)
    assert torch.allclose(r0["scores"], serial["scores"], rtol=1e-5)
    assert torch.equal(r0["best"], r1["best"]) and torch.equal(r0["best"], serial["best"])       # max is exact in any order


def test_shard_bounds_partition():
    from fisher_rast.distributed import shard_bounds
    for n in (0, 1, 7, 64, 512, 513):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _replicate_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fisher_rast import distributed as D, synthetic
        P = 700
        full = synthetic.room_shell(P, seed=9)                       # what only the planner process holds (slam.params)
        H_full = torch.rand((P, 4), generator=torch.Generator().manual_seed(3))
        if rank == 0:
            params, H_inv = full, H_full
        elif rank == 1:
            params, H_inv = None, None                               # a worker rank with no map at all
        else:
            params = {k: torch.zeros_like(v) for k, v in full.items()}   # a stale replica: zeros of the right shape are reused in place
            H_inv = torch.zeros((P, 4))
        stale_ptr = None if params is None or rank == 0 else params["means3D"].data_ptr()
        # the replicas differ before the broadcast: the fingerprint check must say so on every rank
        mine = [v for v in (params or {"x": torch.zeros(3)}).values()]
        differs = False
        try:
            D.assert_replicated(mine[:1])
        except RuntimeError as ex:
            differs = "different replicas" in str(ex)
        got, H_got = D.replicate_map(params, H_inv, src=0, device=torch.device("cpu"))
        D.assert_replicated([got[k] for k in sorted(got)] + [H_got])
        ok = all(torch.equal(got[k], full[k]) for k in full) and torch.equal(H_got, H_full) and set(got) == set(full)
        reused = stale_ptr is None or got["means3D"].data_ptr() == stale_ptr
        torch.save(dict(ok=ok, differs=differs, reused=reused), os.path.join(out_dir, f"rep{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_replicate_map_world3(tmp_path):
    """the map and H_inv are broadcast from the one rank that holds them; a rank that starts from nothing (None) or
    from zeros ends with rank 0's tensors bit for bit, and `assert_replicated` tells differing replicas from equal ones."""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_replicate_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    for r in range(3):
        d = torch.load(tmp_path / f"rep{r}.pt")
        assert d["ok"] and d["differs"] and d["reused"], (r, d)


def test_replicate_map_is_identity_without_a_process_group():
    from fisher_rast import distributed as D
    p = {"a": torch.ones(3)}
    got, h = D.replicate_map(p, None)
    assert got is p and h is None
    D.assert_replicated([p["a"]])
