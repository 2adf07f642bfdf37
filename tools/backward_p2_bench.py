"""tools/backward_p2_bench.py (GPU box): the power-2 backward through the rasteriser ABI, chunked (fr_backward_ws with scratch) against the
single pass (one workgroup per tile), ms per call at several image sizes of the benchmark room."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")]
import numpy as np, torch
from fisher_rast import synthetic, ops
from models.SLAM.utils.recon_helpers import setup_camera
dev = torch.device("cuda:0")
e = torch.Tensor([])
SIZES = ((500_000, 256, 256), (2_000_000, 512, 512), (2_000_000, 320, 240), (500_000, 128, 128))
if len(sys.argv) > 1:
    SIZES = (tuple(int(x) for x in sys.argv[1:4]),)
for P, W, H in SIZES:
    act = {k: v.to(dev) for k, v in synthetic.activate(synthetic.room_shell(P, 2)).items()}
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(2, 2))[1].to(dev)
    pts = act["means3D"]
    tp = (w2c @ torch.cat((pts, torch.ones_like(pts[:, :1])), 1).T).T[:, :3].contiguous()
    R, color, radii, geom, binning, img, depth = ops.rasterize_forward(cam.bg, tp, act["rgb_colors"], act["opacities"], act["scales"], act["rotations"], 1.0, e,
                                                                        cam.viewmatrix, cam.projmatrix, cam.tanfovx, cam.tanfovy, H, W, e, 0, cam.campos, False)
    dL = torch.full((3, H, W), 1e-3, device=dev)
    out = {}
    for seg in (True, False):
        def run():
            return ops.rasterize_backward(cam.bg, tp, radii, act["rgb_colors"], act["scales"], act["rotations"], 1.0, e, cam.viewmatrix, cam.projmatrix,
                                          cam.tanfovx, cam.tanfovy, dL, e, 0, cam.campos, geom, R, binning, img, 2, segmented=seg)
        for _ in range(3):
            g = run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g = run()
        torch.cuda.synchronize()
        out[seg] = ((time.perf_counter() - t0) / 10 * 1e3, g[3].clone())
    rel = float(((out[True][1] - out[False][1]).abs() / (out[False][1].abs() + 1e-6 * out[False][1].abs().max())).max())
    T = ((W + 15) // 16) * ((H + 15) // 16)
    print(f"P={P} {W}x{H} ({T} tiles, {R} tile instances): chunked {out[True][0]:.3f} ms, single pass {out[False][0]:.3f} ms per backward; dL_dmeans3D max rel diff {rel:.1e}")
