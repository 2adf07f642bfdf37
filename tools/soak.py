#!/usr/bin/env python3
"""Randomised self-consistency soak of fr_fisher_views (no oracle: sizes the oracle cannot reach in reasonable time).
For random scenes / image sizes / view counts the score-only launch (k_fisher_tile_v3) must agree with
sum(cur_H * H_inv) of the out_H launch (k_fisher_tile_v3h or k_fisher_tile_v2) of the same views, the visible counts and the
tile-instance counts of the two launches must be equal, a second score-only launch must reproduce the first bit for bit, and
so must a scorer on packed key lists (the default keeps fixed key segments).
usage: tools/soak.py [rounds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
from fisher_rast import synthetic
from fisher_rast.ops import FisherScorer
from models.SLAM.utils.recon_helpers import setup_camera

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
worst = 0.0
for r in range(rounds):
    P = int(10 ** rng.uniform(2.0, 5.6))
    V = int(rng.integers(1, 24))
    W, H = int(rng.integers(33, 400)), int(rng.integers(33, 400))
    if rng.random() < 0.15:                                  # beyond 4096 tiles: the single-view front end + k_fisher_records
        W, H, V, P = int(rng.integers(1030, 1100)), int(rng.integers(1030, 1100)), int(rng.integers(1, 4)), min(P, 60_000)
    C = int(rng.choice([4, 11]))
    seed = int(rng.integers(0, 10_000))
    raw = synthetic.room_shell(P, seed)
    if rng.random() < 0.3:                                   # some scenes with much larger splats (crowded tiles, long lists)
        raw["log_scales"] = raw["log_scales"] + float(rng.uniform(0.5, 1.5))
    act = {k: v.to(dev) for k, v in synthetic.activate(raw).items()}
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
    sc = FisherScorer(cam, act["means3D"], act["rgb_colors"], act["rotations"], act["opacities"], act["scales"], columns=C)
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, seed + 1)).to(dev)
    Hinv = (torch.rand((P, C), device=dev) * 2.0 + 0.05)
    cur = torch.zeros((V, P, C), device=dev)
    a = sc.run(w2c, out_H=cur, out_H_per_view=True)
    b = sc.run(w2c, H_inv=Hinv)
    b2 = sc.run(w2c, H_inv=Hinv)
    pk = FisherScorer(cam, act["means3D"], act["rgb_colors"], act["rotations"], act["opacities"], act["scales"], columns=C)
    pk.tile_capacity = 0                                     # packed key lists (scan + scatter kernel): bit-identical scores
    bp = pk.run(w2c, H_inv=Hinv)
    want = (cur.double() * Hinv.double()[None]).sum(dim=(1, 2))
    got = b["scores"].double()
    denom = want.abs().clamp_min(1e-30)
    rel = float(((got - want).abs() / denom)[want.abs() > 1e-12].max()) if bool((want.abs() > 1e-12).any()) else 0.0
    ok = (torch.equal(a["vis_count"], b["vis_count"]) and torch.equal(a["num_rendered"], b["num_rendered"])
          and torch.equal(b["scores"], b2["scores"]) and torch.equal(b["scores"], bp["scores"]) and rel < 3e-4
          and bool(torch.isfinite(got).all()))
    worst = max(worst, rel)
    print(f"round {r:2d}: P={P:7d} V={V:2d} {W:3d}x{H:3d} C={C:2d}  rel err {rel:.2e}  segments {sc.tile_capacity:6d}  {'ok' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        sys.exit(1)
    del sc, pk, cur
print(f"soak ok: {rounds} rounds, worst relative difference {worst:.2e}")
