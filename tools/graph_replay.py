"""tools/graph_replay.py (GPU box): one 64-view scorer launch captured into a HIP graph (torch.cuda.CUDAGraph: the library's
side-stream fork / joins are captured with it) and replayed, against plain launches -- what the launch gaps of a step are worth."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "fisher-nerf-customized_amd")]
import numpy as np, torch
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
want = sc.run(w2c, H_inv=Hi)["scores"].clone()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


plain = timed(lambda: sc.launch(w2c, H_inv=Hi))
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for _ in range(3):
        sc.launch(w2c, H_inv=Hi)
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    r = sc.launch(w2c, H_inv=Hi)
g.replay(); torch.cuda.synchronize()
assert torch.equal(r["scores"], want), "replayed scores differ"
graph = timed(g.replay)
print("plain launches %.3f ms per step, graph replay %.3f ms per step" % (plain, graph))
