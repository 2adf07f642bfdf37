"""tools/fe_ablate.py (GPU box): what the direct key scatter inside k_preprocess_views_c costs, by leaving parts of it out.
A -DFR_ABLATE build of the library (tools/_build/, never the package directory) with FR_DEBUG_MODE
    30 = everything, 31 = no key sweep, 32 = the sweep without the key stores, 33 = synthetic list entries (no list loads, no stores),
    34 = no range claims (no global atomics), 35 = phase A only (frustum test + survivor lists), 36 = phases A + B (no records, no list
    entries), 37 = all phases, every record written to the same few cache lines (the arithmetic without its HBM traffic);
every mode raises the overflow flag so that the kernels behind the scan return at once -- the time of one launch is the front end's.
One child process per mode (FR_DEBUG_MODE is read once per process)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_build", "libfisher_rast_ablate.so")
CSRC = os.path.join(ROOT, "fisher-nerf-customized_amd", "csrc")


def build():
    srcs = [os.path.join(CSRC, f) for f in ("fisher_rast.hip", "fisher_occ.hip")]
    deps = srcs + [os.path.join(CSRC, "fr_math.h")]
    if os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in deps):
        return
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-DFR_ABLATE", "-o", SO] + srcs)


def child(mode):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")]
    import numpy as np
    import torch
    from fisher_rast import synthetic
    from fisher_rast.ops import FisherScorer
    from models.SLAM.utils.recon_helpers import setup_camera
    dev = torch.device("cuda:0")
    P, V, W, H = 500_000, 64, 256, 256
    act = synthetic.activate(synthetic.room_shell(P, 2))
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
    sc = FisherScorer(cam, *(act[k].to(dev) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")))
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(V, 2)).to(dev)
    Hi = torch.rand((P, 4), generator=torch.Generator().manual_seed(1)).to(dev)
    if "--outh" in sys.argv:
        # the out_H launch of 16 keyframes (k_fisher_tile_v3h): 0 = everything, 28 = no global atomics, 29 = no LDS atomics either
        kf = synthetic.invert_rigid(synthetic.candidate_poses(16, 102)).to(dev)
        Ht = torch.zeros((P, 4), device=dev)
        for _ in range(3):
            sc.launch(kf, out_H=Ht)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            sc.launch(kf, out_H=Ht)
        e1.record()
        torch.cuda.synchronize()
        print("mode", mode, "H_train of 16 keyframes %.3f ms per launch" % (e0.elapsed_time(e1) / 10), flush=True)
        return
    for _ in range(3):
        r = sc.launch(w2c, H_inv=Hi)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        r = sc.launch(w2c, H_inv=Hi)
    e1.record()
    torch.cuda.synchronize()
    print("mode", mode, "front end %.3f ms per launch" % (e0.elapsed_time(e1) / 10), "status", r["status"].cpu().tolist(), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        build()
        outh = ["--outh"] if "--outh" in sys.argv else []
        modes = [int(a) for a in sys.argv[1:] if a != "--outh"] or ([0, 28, 29] if outh else [30, 31, 32, 33, 34, 35, 36, 37])
        for m in modes:
            env = dict(os.environ, FR_DEBUG_MODE=str(m), FISHER_RAST_SO=SO)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", str(m)] + outh, env=env)
