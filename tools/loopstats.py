"""tools/loopstats.py (GPU box): loop-trip counters / s_memtime shares of k_fisher_tile_v3 on the bench workload, from a
-DFR_LOOPSTATS build of the library that is compiled ON DEMAND into tools/_build/ (never into the package directory) and loaded
through FISHER_RAST_SO.  One child process per mode (FR_DEBUG_MODE is read once per process):
    python tools/loopstats.py            -> modes 2..7 and 10..12 (fisher_rast.hip: k_fisher_tile_v3, FR_LOOPSTATS)
    python tools/loopstats.py 4 5        -> only these
Mode 4 = wave-level walk iterations, 5 = contributing (pixel, splat) pairs, 6 = lane-level walk steps, 2 = candidates, 3 = chunks,
7 = steps of the busiest lane, 10 / 11 / 12 = s_memtime ticks (/64) of the key stream / chunk set-up / walk."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_build", "libfisher_rast_loopstats.so")
CSRC = os.path.join(ROOT, "fisher-nerf-customized_amd", "csrc")


def build():
    srcs = [os.path.join(CSRC, f) for f in ("fisher_rast.hip", "fisher_occ.hip")]
    deps = srcs + [os.path.join(CSRC, "fr_math.h")]
    if os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in deps):
        return
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-DFR_LOOPSTATS", "-o", SO] + srcs)


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
    r = sc.run(w2c, H_inv=Hi)
    print("mode", mode, "sum over 64 views", float(r["scores"].double().sum()), "tile instances", int(r["num_rendered"].sum()), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        build()
        modes = [int(a) for a in sys.argv[1:]] or [2, 3, 4, 5, 6, 7, 10, 11, 12]
        for m in modes:
            env = dict(os.environ, FR_DEBUG_MODE=str(m), FISHER_RAST_SO=SO)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", str(m)], env=env)
