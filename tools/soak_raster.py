#!/usr/bin/env python3
"""Cross-check of the walking forward / backward kernels against the round-1 kernels on random scenes the oracle cannot reach
in reasonable time.  Run once per kernel generation (FR_DEBUG_MODE is read once per process); the parent compares the dumps:
   tools/soak_raster.py dump <out.npz> [rounds] [seed]      (child: renders + backward, writes hashes and sums)
   tools/soak_raster.py [rounds] [seed]                      (parent: runs the child with FR_DEBUG_MODE unset and with 16 / 17)
Forward outputs (colour, depth, radii, n_contrib, final_T) must be bit-identical, power-1 gradients equal to 2e-5."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")):
    sys.path.insert(0, p)
import numpy as np


def child(out, rounds, seed):
    import torch
    import __graft_entry__ as entry
    entry.build()
    from fisher_rast import synthetic, ops
    from models.SLAM.utils.recon_helpers import setup_camera
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    res = {}
    e = torch.Tensor([])
    for r in range(rounds):
        P = int(10 ** rng.uniform(2.5, 5.5)); W, H = int(rng.integers(40, 640)), int(rng.integers(40, 640)); s = int(rng.integers(0, 10_000))
        raw = synthetic.room_shell(P, s)
        if rng.random() < 0.4:
            raw["log_scales"] = raw["log_scales"] + float(rng.uniform(0.5, 1.8))
        act = {k: v.to(dev) for k, v in synthetic.activate(raw).items()}
        cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
        w2c = synthetic.invert_rigid(synthetic.candidate_poses(1, s))[0].to(dev)
        pts = act["means3D"]
        tp = (w2c @ torch.cat((pts, torch.ones_like(pts[:, :1])), 1).T).T[:, :3].contiguous()
        R, color, radii, geom, binning, img, depth = ops.rasterize_forward(cam.bg, tp, act["rgb_colors"], act["opacities"], act["scales"], act["rotations"], 1.0, e,
                                                                       cam.viewmatrix, cam.projmatrix, cam.tanfovx, cam.tanfovy, H, W, e, 0, cam.campos, False)
        dL = torch.randn((3, H, W), generator=torch.Generator().manual_seed(s)).to(dev)
        grads = ops.rasterize_backward(cam.bg, tp, radii, act["rgb_colors"], act["scales"], act["rotations"], 1.0, e, cam.viewmatrix, cam.projmatrix,
                                       cam.tanfovx, cam.tanfovy, dL, e, 0, cam.campos, geom, R, binning, img, 1)
        torch.cuda.synchronize()
        L = ops.workspace_layout(P, W, H, 1)
        im = img.cpu().numpy()
        fwd = b"".join([color.cpu().numpy().tobytes(), depth.cpu().numpy().tobytes(), radii.cpu().numpy().tobytes(),
                        im[L["final_T"]:L["final_T"] + 4 * W * H].tobytes(), im[L["n_contrib"]:L["n_contrib"] + 4 * W * H].tobytes()])
        res[f"fwd{r}"] = np.frombuffer(hashlib.sha256(fwd).digest(), np.uint8)
        res[f"cfg{r}"] = np.array([P, W, H, int(R)])
        for k, gt in enumerate(grads):
            if isinstance(gt, torch.Tensor) and gt.numel() > 0:
                res[f"g{r}_{k}"] = gt.double().cpu().numpy().reshape(-1)[:: max(1, gt.numel() // 20000)].copy()
    np.savez(out, **res)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "dump":
        child(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8, int(sys.argv[4]) if len(sys.argv) > 4 else 1)
        return
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    outs = {}
    for mode in ("0", "16", "17"):
        out = f"/tmp/soak_raster_{mode}.npz"
        # (modes 16 / 17 exist in the -DFR_AB rig only: tools/build_variant.sh rig)
        rig = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "ab_rig.so")
        if not os.path.exists(rig):
            subprocess.check_call(["bash", os.path.join(os.path.dirname(os.path.abspath(__file__)), "build_variant.sh"), "rig"])
        env = dict(os.environ, FR_DEBUG_MODE=mode, FISHER_RAST_SO=rig)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "dump", out, str(rounds), str(seed)], env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = dict(np.load(out))
    a = outs["0"]
    worst = 0.0
    for r in range(rounds):
        P, W, H, R = a[f"cfg{r}"]
        same = all(np.array_equal(a[f"fwd{r}"], outs[m][f"fwd{r}"]) for m in ("16", "17"))
        rel = 0.0
        for k in [k for k in a if k.startswith(f"g{r}_")]:
            for m in ("16", "17"):
                d = np.abs(a[k] - outs[m][k]).max(); s = np.abs(a[k]).max()
                rel = max(rel, d / s if s > 0 else d)
        worst = max(worst, rel)
        print(f"round {r}: P={P} {W}x{H} instances={R}  forward identical: {same}  gradients max rel diff {rel:.2e}")
        assert same and rel < 2e-5
    print(f"soak_raster ok: {rounds} scenes, worst gradient difference {worst:.2e}")


if __name__ == "__main__":
    main()
