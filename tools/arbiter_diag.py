"""GPU box: where do the scorer's kernels sit between the binary32 oracle and the binary64 arbiter (oracle/ref.py, arbiter=True)
on the ill-conditioned test families?  python tools/arbiter_diag.py [case ...]   (diagnostic for tests/test_gpu_scorer_adversarial.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "fisher-nerf-customized_amd")]
import numpy as np
import torch
from oracle import ref
from scenes import intrinsics
import test_gpu_scorer_adversarial as A
from fisher_rast.ops import FisherScorer
from models.SLAM.utils.recon_helpers import setup_camera

ref.build()
dev = torch.device("cuda:0")
for case in (sys.argv[1:] or ["border", "general"]):
    W, H, sc, w2c = A._family(case, ref)
    K = intrinsics(W, H)
    cam = setup_camera(W, H, K, np.eye(4), device=dev)
    ocam = ref.setup_camera(W, H, K, np.eye(4))
    w2cs = A._views(w2c, 3)
    args = (sc["means3D"], sc["colors"], sc["rotations"], sc["opacities"], sc["scales"])
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in args]
    P = sc["means3D"].shape[0]
    for C in (4, 11):
        res = [ref.compute_hessian(ocam, w, *args, columns=C, arbiter=True) for w in w2cs]
        o = np.stack([r[0] for r in res]).astype(np.float64)
        a = np.stack([r[1] for r in res])
        Htr = o[1:].astype(np.float32).sum(0, dtype=np.float32)
        Hinv = (np.float32(1) / (Htr + np.float32(0.1))).astype(np.float32)
        so, sa = (o * Hinv[None]).sum((1, 2)), (a * Hinv[None]).sum((1, 2))
        scorer = FisherScorer(cam, *t, columns=C)
        wd = torch.from_numpy(w2cs).to(dev)
        s = scorer.run(wd, H_inv=torch.from_numpy(Hinv).to(dev))["scores"].cpu().numpy().astype(np.float64)
        cur = torch.zeros((3, P, C), device=dev)
        scorer.run(wd, out_H=cur, out_H_per_view=True)
        g = cur.cpu().numpy().astype(np.float64)
        sg = (g * Hinv[None]).sum((1, 2))
        print(f"{case}-{C}: scores  |gpu-o32|/o32 {np.abs(s - so) / so}  |gpu-a64| {np.abs(s - sa) / sa}  |o32-a64| {np.abs(so - sa) / sa}  outH-sum vs a64 {np.abs(sg - sa) / sa}")
        for v in range(3):
            scale = np.abs(a[v]).max()
            big = np.abs(a[v]) > 1e-7 * scale
            eo = np.abs(o[v] - a[v])[big] / np.abs(a[v])[big]
            eg = np.abs(g[v] - a[v])[big] / np.abs(a[v])[big]
            ego = np.abs(g[v] - o[v])[big] / np.abs(o[v])[big]
            # entries the GPU misses at 1e-4 against the oracle, and how far the oracle itself is from the arbiter there
            miss = ego > 1e-4
            ratio = (np.abs(g[v] - o[v])[big][miss] / np.maximum(np.abs(o[v] - a[v])[big][miss], 1e-300)) if miss.any() else np.zeros(0)
            # per-Gaussian conditioning estimate: the largest relative deviation of the oracle from the arbiter over the Gaussian's columns
            ra = np.where(big, np.abs(o[v] - a[v]) / np.maximum(np.abs(a[v]), 1e-300), 0.0).max(axis=1, keepdims=True) * np.ones_like(a[v])
            exc = np.maximum(np.abs(g[v] - o[v]) - 1e-4 * np.abs(o[v]) - 1e-7 * np.abs(o[v]).max(), 0.0)
            need = exc / np.maximum(ra * np.abs(o[v]), 1e-300)
            print(f"   view {v}: K needed with the per-Gaussian estimate: {need[exc > 0].max() if (exc > 0).any() else 0:.2f}  (entries over 1e-4 + floor: {(exc > 0).sum()}, of them with r_G < 2e-5: {((exc > 0) & (ra < 2e-5)).sum()})")
            print(f"   view {v}: entries {big.sum()}  o32-a64: max {eo.max():.2e} n>1e-4 {(eo > 1e-4).sum()} | gpu-a64: max {eg.max():.2e} n>1e-4 {(eg > 1e-4).sum()}"
                  f" | gpu-o32: max {ego.max():.2e} n>1e-4 {miss.sum()}  (there |gpu-o32|/|o32-a64|: max {ratio.max() if ratio.size else 0:.2f}, "
                  f"n with o32-a64 < 2e-5: {(eo[miss] < 2e-5).sum() if miss.any() else 0})")
