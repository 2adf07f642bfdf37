#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric): candidate views scored per second
(+ Fisher scores per second) at 256x256 over 500k Gaussians.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic input: the fused scorer (fr_fisher_views) ranks
`--views` (64) candidate poses per GPU against a resident 500k-Gaussian map and a resident H_inv = 1/(H_train+0.1):
project / cull / per-(view, Gaussian) scorer records -> tile binning -> per-tile depth sort -> one front-to-back
scoring pass (sum(cur_H * H_inv) without materialising cur_H).  Inputs are in HBM before the timed region; each step ends
with the asynchronous copy of the scores to the host.
N = 1 : BASELINE.json configs[1] (500k Gaussians, 64 candidate 256x256 views, seed 2).
N > 1 : configs[2]'s shape.  Default = weak scaling: every rank scores its own 64-view slice of a 64*N-view candidate
        set (seed 3); `--total-views T` = strong scaling: T views (configs[2]: 512) split over the ranks.  Either way the
        per-view scores are exchanged with ONE all-gather per step (RCCL over xGMI).
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline` objects, plus the
reported side numbers: serial_loop, compute_hessian_v1_ms, config4, columns_11, occupancy / frontier.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "fisher-nerf-customized_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np   # noqa: E402
import torch         # noqa: E402
import torch.distributed as dist   # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)
BYTES_PER_TILE_INSTANCE = 8 + 32 + 52   # k_fisher_tile_v3 on packed key lists: sorted key + {recA, recB} + the 13 floats of recQ it uses, each moved once
BYTES_PER_TILE_INSTANCE_FIXED = 8 + 80  # ... with fixed key segments (the scorer's default): sorted key + the 80-byte record (20 floats, all used)


# ---------------------------------------------------------------------------------------------------------------------
# CPU legs: run BEFORE anything initialises the GPU, so that worker processes can simply be forked
# ---------------------------------------------------------------------------------------------------------------------
_CPU = {}


def _cpu_worker(v):
    from oracle import ref
    c = _CPU
    cur_H, _ = ref.compute_hessian(c["cam"], c["w2c"][v], *c["args"], columns=c["columns"])
    return float(np.sum(cur_H.astype(np.float64) * c["H_inv"].astype(np.float64)))


def host_cores():
    """The box's CPU share (a one-GPU box grants 16 cores; os.cpu_count() reports the whole host)."""
    n = int(os.environ.get("FR_BENCH_CORES", "0"))
    if n > 0:
        return n
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(P, W, H, seed, n_views, columns, V_total):
    """The oracle (CPU restatement of the reference path: forward + fused backward(power=2) + weighted sum) on a bounded
    sample of the same workload: the first n_views of the step's V_total candidate views on ONE host core, then 2 views per core
    on ALL host cores (one forked process per core; the oracle itself is scalar C).  H_inv from 1 keyframe instead of 16.
    The oracle's scores and its H_inv are kept (`_parity`): the GPU scores the same views with the same H_inv after the
    timed region and the line reports the largest relative difference (`parity`)."""
    import multiprocessing as mp
    from oracle import ref
    from fisher_rast import synthetic
    act = {k: v.numpy() for k, v in synthetic.activate(synthetic.room_shell(P, seed)).items()}
    args = (act["means3D"], act["rgb_colors"], act["rotations"], act["opacities"], act["scales"])
    cam = ref.setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4))
    cores = host_cores()
    n_all = min(2 * cores, V_total)
    n_views = min(n_views, V_total)
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(V_total, seed)).numpy()       # the poses of the timed step
    kf = synthetic.invert_rigid(synthetic.candidate_poses(1, seed + 100)).numpy()
    H_train = ref.compute_h_train(cam, kf, *args, columns=columns)
    H_inv = (np.float32(1.0) / (H_train + np.float32(0.1))).astype(np.float32)
    _CPU.update(cam=cam, w2c=w2c, args=args, columns=columns, H_inv=H_inv)
    t0 = time.perf_counter()
    one = [_cpu_worker(v) for v in range(n_views)]
    dt1 = time.perf_counter() - t0
    out = dict(value=n_views / dt1, unit="candidate-views/s", cores=1, kind="port",
               sample=f"{n_views} of the {P}-Gaussian {W}x{H} candidate views through oracle/fisher_oracle.c "
                      f"(forward + fused backward power=2 + weighted sum), {dt1:.1f} s on 1 host core",
               seconds=dt1)
    out["_parity"] = dict(scores=np.asarray(one, np.float64), H_inv=H_inv)
    if cores > 1:
        with mp.get_context("fork").Pool(cores) as pool:
            pool.map(_cpu_worker, range(cores))                      # page the scene into every worker
            t0 = time.perf_counter()
            allc = pool.map(_cpu_worker, range(n_all), chunksize=1)
            dtn = time.perf_counter() - t0
        assert np.allclose(allc[:min(n_views, n_all)], one[:min(n_views, n_all)], rtol=1e-12)
        out["all_cores"] = dict(value=n_all / dtn, unit="candidate-views/s", cores=cores, seconds=dtn,
                                sample=f"{n_all} views, one oracle process per core on {cores} host cores, {dtn:.1f} s")
        if n_all > n_views:
            out["_parity"]["scores"] = np.asarray(allc, np.float64)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# GPU side numbers (outside the headline timed region)
# ---------------------------------------------------------------------------------------------------------------------
def gpu_occupancy_frontier(dev, W, H, seed, n_frames=4, n_gaussians=200_000):
    """The planner-side kernels (fr_occ_update / fr_occ_freespace / fr_occ_frontiers) on the same synthetic frames as
    the CPU occupancy baseline: ms per map update and per frontier build, map resident, one host sync per build."""
    from fisher_rast import synthetic
    from oracle.occupancy_frontier import room_depth           # input generator of the baseline (not the thing measured)
    from planning import AstarPlanner
    K = synthetic.intrinsics(W, H)
    poses = synthetic.candidate_poses(n_frames, seed + 200).numpy().astype(np.float32)
    pts = synthetic.room_shell(n_gaussians, seed)["means3D"].to(dev)
    pl = AstarPlanner(device=dev, cell_size=0.05, frontier_select_method="combined")
    pl.init(torch.eye(4), torch.from_numpy(np.asarray(K, dtype=np.float32)))
    depths = [torch.from_numpy(room_depth(p, W, H, K)).to(dev) for p in poses]
    c2ws = [torch.from_numpy(p).to(dev) for p in poses]
    pl.update_occ_map(depths[0], c2ws[0], 0)                    # warm-up (workspace allocation)
    pl.init(torch.eye(4), torch.from_numpy(np.asarray(K, dtype=np.float32)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t, (d, c) in enumerate(zip(depths, c2ws)):
        pl.update_occ_map(d, c, t)
    torch.cuda.synchronize()
    t_up = (time.perf_counter() - t0) / n_frames
    pl.build_frontiers(pts)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        fr, free = pl.build_frontiers(pts)
    torch.cuda.synchronize()
    t_fr = (time.perf_counter() - t0) / reps
    centers = torch.from_numpy(np.asarray(fr, dtype=np.float32)).to(dev) if fr is not None else torch.zeros((1, 2), device=dev)
    pl.cam_height = 0.0
    pl.generate_candidate_in_freespace(centers, free, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        cand = pl.generate_candidate_in_freespace(centers, free, seed=r)
    torch.cuda.synchronize()
    t_c = (time.perf_counter() - t0) / reps
    return dict(ms_per_update=1e3 * t_up, ms_per_frontier_build=1e3 * t_fr, ms_per_candidate_batch=1e3 * t_c, frames=n_frames,
                free_cells=int(free.sum()), frontier_cells=0 if fr is None else int(len(fr)), candidates_kept=int(cand.shape[0]),
                kernels="fr_occ_update / fr_occ_freespace / fr_occ_frontiers / fr_occ_erode + fr_occ_ring_candidates "
                        "(csrc/fisher_occ.hip), wall time incl. host glue")


def measured_copy_bandwidth(dev, nbytes=1 << 30, reps=5):
    """Device-to-device copy rate of this box, read + write bytes per second (recorded beside the nominal HBM peak)."""
    a = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def api_latencies(dev, raw_params, W, H, seed, V):
    """Drop-in calls as the planner makes them (scores on the host when the call returns):
      pose_eval(poses)      gaussian.py:1354-1375 -- H_train over 16 keyframes + V candidate scores;
      compute_Hessian(w2c)  gaussian.py:1503-1570 -- ONE view, what the tester calls once per path step, ~630 times per
                            planning round (tester_gaussians_navigation.py:1684-1705)."""
    import models.gaussian_slam as mgs
    from fisher_rast import synthetic
    slam = mgs.GaussianSLAM(params={k: v.to(dev) for k, v in raw_params.items()}, intrinsics=synthetic.intrinsics(W, H), width=W, height=H, device=dev)
    for kf in synthetic.invert_rigid(synthetic.candidate_poses(16, seed + 100)):
        slam.add_keyframe(kf.to(dev))
    poses = [p.to(dev) for p in synthetic.candidate_poses(V, seed)]
    slam.pose_eval(poses)
    torch.cuda.synchronize()
    reps = 5
    # every call recomputes H_train, as the reference does (the product keeps 1 / (H_train + reg) while map and keyframes are unchanged)
    slam.CACHE_H_TRAIN = False
    slam.pose_eval(poses)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        scores, _ = slam.pose_eval(poses)
    torch.cuda.synchronize()
    t_pe = (time.perf_counter() - t0) / reps
    slam.CACHE_H_TRAIN = True
    slam.pose_eval(poses)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        scores, _ = slam.pose_eval(poses)
    torch.cuda.synchronize()
    t_pe_kept = (time.perf_counter() - t0) / reps
    w2cs = synthetic.invert_rigid(torch.stack(poses[:16]).cpu()).to(dev)
    slam.compute_Hessian(w2cs[0], return_points=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for w in w2cs:
        h = slam.compute_Hessian(w, return_points=True)
        float(h[0, 0])                                          # the caller reads the result (tester 1690-1695)
    t_ch = (time.perf_counter() - t0) / len(w2cs)
    return (dict(ms_per_call=1e3 * t_pe, ms_per_call_h_train_kept=1e3 * t_pe_kept, views=V, keyframes=16,
                 what="GaussianSLAM.pose_eval(poses): H_train over the keyframes + all candidate scores, result on the host; "
                      "`h_train_kept`: repeated calls on an unchanged map and keyframe set (1 / (H_train + reg) is kept)"),
            1e3 * t_ch)


def serial_loop(dev, raw_params, W, H, seed, n_views, H_inv):
    """the reference's calling pattern on this repository's own kernels -- one view at a time through the
    drop-in autograd `GaussianRasterizer(backward_power=2)` (gaussian.py:1523-1567, 1362-1370: torch transform, fresh
    rendervar, forward, backward, cat, sum(...).item()): per-view allocations and host syncs."""
    from diff_gaussian_rasterization import GaussianRasterizer as Renderer
    from fisher_rast import synthetic
    from models.SLAM.utils.recon_helpers import setup_camera
    params = {k: v.to(dev) for k, v in raw_params.items()}
    P = params["means3D"].shape[0]
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
    w2cs = synthetic.invert_rigid(synthetic.candidate_poses(n_views, seed)).to(dev)

    def one_view(rel_w2c):
        with torch.no_grad():
            pts = params['means3D']
            tp = (rel_w2c @ torch.cat((pts, torch.ones(P, 1, device=dev)), dim=1).T).T[:, :3]
            rot = torch.nn.functional.normalize(params['unnorm_rotations'])
            op = torch.sigmoid(params['logit_opacities'])
            sc = torch.exp(params['log_scales'])
            sc = torch.tile(sc, (1, 3)) if sc.shape[-1] == 1 else sc
        rv = {'means3D': tp.requires_grad_(True), 'colors_precomp': params['rgb_colors'].detach().clone().requires_grad_(True),
              'rotations': rot.requires_grad_(True), 'opacities': op.requires_grad_(True), 'scales': sc.requires_grad_(True),
              'means2D': torch.zeros_like(tp, requires_grad=True, device=dev) + 0}
        im, radius, _ = Renderer(raster_settings=cam, backward_power=2)(**rv)
        im.backward(gradient=torch.ones_like(im) * 1e-3)
        int((radius > 0).sum().item())
        cur_H = torch.cat([tp.grad.detach().reshape(P, -1), op.grad.detach().reshape(P, -1)], dim=1)
        return torch.sum(cur_H * H_inv).item()

    one_view(w2cs[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for v in range(n_views):
        one_view(w2cs[v])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(value=n_views / dt, unit="candidate-views/s", views=n_views, ms_per_view=1e3 * dt / n_views,
                what="BASELINE.md B4: reference-style serial loop (autograd GaussianRasterizer(backward_power=2), one view at a "
                     "time, per-view allocations and .item() syncs) on this repository's kernels -- the 1x the batched scorer is "
                     "quoted against; no CUDA number exists")


def config4_train_step(dev, P=2_000_000, size=512, reps=5):
    """BASELINE.json configs[3]: 2M Gaussians, 512x512, one forward + power=1 backward with dL_dpix = N(0,1) (seed 44) through
    the single-view rasteriser ABI (train-step proxy): ms per step, tile instances R, and the sort's key traffic rate
    (16 B per tile instance -- key read + written once -- over the summed sort-kernel time is not separable here, so the rate
    is quoted over the whole forward: a lower bound)."""
    from fisher_rast import synthetic, ops
    from models.SLAM.utils.recon_helpers import setup_camera
    W = H = size
    act = {k: v.to(dev) for k, v in synthetic.activate(synthetic.room_shell(P, 4)).items()}
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(1, 4))[0].to(dev)
    pts = act["means3D"]
    tp = (w2c @ torch.cat((pts, torch.ones_like(pts[:, :1])), 1).T).T[:, :3].contiguous()
    dL = torch.randn((3, H, W), generator=torch.Generator().manual_seed(44)).to(dev)
    e = torch.Tensor([])

    def fwd():
        return ops.rasterize_forward(cam.bg, tp, act["rgb_colors"], act["opacities"], act["scales"], act["rotations"], 1.0, e,
                                     cam.viewmatrix, cam.projmatrix, cam.tanfovx, cam.tanfovy, H, W, e, 0, cam.campos, False)

    def step():
        R, color, radii, geom, binning, img, depth = fwd()
        ops.rasterize_backward(cam.bg, tp, radii, act["rgb_colors"], act["scales"], act["rotations"], 1.0, e, cam.viewmatrix,
                               cam.projmatrix, cam.tanfovx, cam.tanfovy, dL, e, 0, cam.campos, geom, R, binning, img, 1)
        return R

    R = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        R = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        fwd()
    torch.cuda.synchronize()
    dtf = (time.perf_counter() - t0) / reps
    return dict(ms_per_step=1e3 * dt, ms_forward=1e3 * dtf, gaussians=P, image=[H, W], tile_instances=int(R),
                sort_GBps_lower_bound=16.0 * R / dtf / 1e9,
                what="BASELINE.json configs[3]: forward + backward(power=1), one view, through the drop-in rasteriser ABI; "
                     "sort rate = 16 B per tile instance over the WHOLE forward time (preprocess + scan + scatter + sort + render)")


def valu_ceiling():
    """tools/valu_ceiling.hip --quick, 5 waves per SIMD: wave64 VALU instructions per second the chip retires with independent
    FMAs (the ceiling the scorer is priced against), with one dependent chain of scalar-operand FMAs per wave, and with
    FMAs that read three vector registers (the form real code mostly has)."""
    exe = os.path.join(ROOT, "tools", "_build", "valu_ceiling")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "--quick"], capture_output=True, text=True, timeout=120).stdout
        rows = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
        return {"independent_fma_wave_insts_per_s": rows[0]["chip_wave_insts_per_s"],
                "dependent_chain_scalar_operands_wave_insts_per_s": rows[1]["chip_wave_insts_per_s"],
                "three_vgpr_fma_wave_insts_per_s": rows[2]["chip_wave_insts_per_s"] if len(rows) > 2 else None,
                "source": "tools/valu_ceiling.hip --quick, run inside this bench (5 waves per SIMD, 1024 SIMDs)"}
    except Exception as ex:          # measurement aid only: never fail the bench on it, but say what happened
        return {"error": repr(ex)}


def pmc_record(P, V, W, C):
    """Hardware counters of the k_fisher_tile_v4 dispatches of ONE step, collected by tools/pmc_collect.sh (rocprofv3 --pmc, separate
    passes) and folded by tools/pmc_summary.py into profiles/pmc_k_fisher_tile_v4.json.  The record carries the sha256 of the kernel's
    gfx950 machine code as it sat in the library the counters were taken on (tools/codeobj.py); it is used only when the library
    loaded NOW holds the same code -- recomputed here, so there is no stamp to maintain by hand."""
    from fisher_rast import _lib
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import codeobj
    f = os.path.join(ROOT, "profiles", "pmc_k_fisher_tile_v4.json")
    if not os.path.exists(f):
        return None, {"file": None, "fresh": False, "why": "no PMC record committed"}
    pm = json.load(open(f))
    now = codeobj.kernel_code_id(_lib.SO_PATH, "k_fisher_tile_v4")
    meta = {"file": "profiles/pmc_k_fisher_tile_v4.json", "commit": pm.get("commit"), "kernel_code_id": pm.get("kernel_code_id"),
            "kernel_code_id_loaded": now, "fresh": True,
            "source": "cached: rocprofv3 --pmc passes of tools/pmc_collect.sh, not measured in this run"}
    if pm.get("kernel_code_id") is None or pm.get("kernel_code_id") != now:
        meta.update(fresh=False, why="the loaded library's k_fisher_tile_v4 is not the code the counters were taken on")
    elif not (pm.get("gaussians") == P and pm.get("views") == V and pm.get("size") == W and pm.get("columns") == C):
        meta.update(fresh=False, why="counters were taken on another workload")
    return (pm if meta["fresh"] else None), meta


def self_launch(n_ranks, argv):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    bench.py <same arguments>` as a CHILD process (never a re-exec; nothing here has touched the GPU), pass its output through
    and return its exit code.  The ranks rendezvous on 127.0.0.1 at a port that was free a moment ago."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in child.stdout:                                # rank 0's JSON line (and anything else the ranks print)
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gaussians", type=int, default=500_000)
    ap.add_argument("--views", type=int, default=64, help="candidate views per GPU per step (weak scaling)")
    ap.add_argument("--total-views", type=int, default=0, help="strong scaling: this many views split over the ranks (configs[2]: 512)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--columns", type=int, default=4)
    ap.add_argument("--cpu-views", type=int, default=8, help="views of the 1-core CPU-baseline sample (0 = skip every side measurement)")
    ap.add_argument("--dump-scores", type=str, default="", help="rank 0 writes the gathered scores of the last step to this .npy (tests)")
    ap.add_argument("--seed", type=int, default=0, help="scene / pose seed (default: 2 on one GPU = configs[1], 3 on several = configs[2])")
    ap.add_argument("--tile-capacity", type=int, default=-1, help="A/B: keys per fixed (view, tile) segment (FisherScorer's default when negative; 0 = packed lists)")
    ap.add_argument("--spatial-order", action="store_true", help="A/B: lay the Gaussians out along a Z-curve (fr_spatial_order; FisherScorer's option, off by default)")
    ap.add_argument("--synthetic-hinv", action="store_true",
                    help="tests: H_inv = seeded uniform weights instead of 1/(H_train+0.1) (H_train is accumulated with float atomics, "
                         "so its last bits differ from run to run; with fixed weights the scores of two runs can be compared bit for bit)")
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: this process has made no GPU call; it starts the N ranks as a fresh child
        # (torch.distributed.run, one process per GPU), relays rank 0's JSON line and exits with the child's code
        raise SystemExit(self_launch(a.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} was started inside a process group of {world} rank(s)")

    import __graft_entry__ as entry
    P, W, H, C = a.gaussians, a.size, a.size, a.columns
    seed = a.seed if a.seed else (2 if world == 1 else 3)
    cpu_legs = None
    if rank == 0:
        entry.build()                                   # (imports the package; touches no device)
        if world == 1 and a.cpu_views > 0:
            # host-core baselines first: worker processes are forked before this process has any GPU state
            from oracle import occupancy_frontier
            cpu_legs = dict(cpu_baseline=cpu_baseline(P, W, H, seed, a.cpu_views, C, a.total_views if a.total_views > 0 else a.views),
                            cpu_occupancy_frontier=occupancy_frontier.time_baseline(n_frames=4, W=W, H=H, seed=seed))

    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback on the product path)"
    # FR_BENCH_BACKEND=gloo with FR_BENCH_ONE_DEVICE=1 rehearses the N > 1 control flow on a single GPU (all ranks on cuda:0,
    # collectives through the host); the measured configuration is always nccl (RCCL), one rank per GPU
    backend = os.environ.get("FR_BENCH_BACKEND", "nccl")
    if os.environ.get("FR_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # FR_FORCE_COLLECTIVES=1 under `torch.distributed.run --nproc-per-node 1`: a process group of one rank, so that RCCL is loaded
    # and the step's all-gather / the H_train all-reduce run on the device tensors (rehearsal of the transport on one GPU)
    dist_on = world > 1 or (os.environ.get("FR_FORCE_COLLECTIVES") == "1" and "RANK" in os.environ)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        dist.barrier()                                  # rank 0 has built the library
    from fisher_rast import synthetic, _lib, distributed as D
    from fisher_rast.ops import FisherScorer
    from models.SLAM.utils.recon_helpers import setup_camera

    strong = a.total_views > 0
    V_total = a.total_views if strong else a.views * world
    # the map lives on ONE rank (the planner process of tester_gaussians_navigation.py:1618-1649 holds slam.params); the other
    # ranks receive their replica by broadcast (once per planning round, outside the timed region)
    raw, act, t_rep = None, None, 0.0
    if rank == 0 or not dist_on:
        raw = synthetic.room_shell(P, seed)
        act = {k: v.to(dev) for k, v in synthetic.activate(raw).items()}
    if dist_on:
        torch.cuda.synchronize()
        dist.barrier()
        t_rep = time.perf_counter()
        act, _ = D.replicate_map(act, None, src=0, device=dev)
        torch.cuda.synchronize()
        t_rep = time.perf_counter() - t_rep
        D.assert_replicated([act[k] for k in sorted(act)])
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
    scorer = FisherScorer(cam, *(act[k] for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")),
                          columns=C, dL_dpix=1e-3, spatial_order=a.spatial_order, **({} if a.tile_capacity < 0 else {"tile_capacity": a.tile_capacity}))
    w2c_all = synthetic.invert_rigid(synthetic.candidate_poses(V_total, seed)).to(dev)
    lo, hi = D.shard_bounds(V_total, rank, world)
    w2c = w2c_all[lo:hi].contiguous()
    V = hi - lo
    kf = synthetic.invert_rigid(synthetic.candidate_poses(16, seed + 100)).to(dev)

    # H_train: keyframes sharded over ranks + one all-reduce(SUM)  (outside the timed region: it is an input)
    H_train = torch.zeros((P, C), dtype=torch.float32, device=dev)
    D.sharded_h_train(lambda w, Hacc: scorer.run(w, out_H=Hacc), kf, H_train)
    H_inv = torch.reciprocal(H_train + 0.1)
    if a.synthetic_hinv:
        H_inv = (torch.rand((P, C), generator=torch.Generator().manual_seed(seed + 7)) * 3.0 + 0.05).to(dev)
    first = scorer.run(w2c, H_inv=H_inv)      # sizes the tile-instance buffer (may re-run on overflow)
    num_rendered = first["num_rendered"].cpu().numpy().astype(np.int64)
    vis_count = first["vis_count"].cpu().numpy().astype(np.int64)

    host_scores = torch.empty((V_total,), dtype=torch.float32).pin_memory()
    gather = D.ScoreGather(V_total, dev) if dist_on else None         # the step's all-gather, buffers allocated once

    def step():
        r = scorer.launch(w2c, H_inv=H_inv)
        if dist_on:
            host_scores.copy_(gather(r["scores"]), non_blocking=True)
        else:
            host_scores.copy_(r["scores"], non_blocking=True)
        return r

    for _ in range(a.warmup):
        step()
    lib = _lib.load()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    lib.fr_profile_enable(1)
    t0 = time.perf_counter()
    last = None
    for _ in range(a.steps):
        last = step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    buf = (ctypes.c_float * max(a.steps * 16, 1))()
    n_ev = lib.fr_profile_fetch(buf, a.steps * 16)
    lib.fr_profile_enable(0)
    assert int(last["status"].cpu()[1]) == 0, "tile-instance buffer overflowed inside the timed region"
    # a step launches the dominant kernel once per view group (FisherScorer.n_streams groups on separate streams)
    launches_per_step = max(1, n_ev // max(a.steps, 1))
    kern_ms = float(np.mean([buf[i] for i in range(n_ev)])) if n_ev > 0 else float("nan")

    t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max.item())
    if rank == 0 and a.dump_scores:
        np.save(a.dump_scores, host_scores.numpy().copy())

    if rank == 0:
        views_per_s = V_total * a.steps / dt
        # tile instances one launch processes: the keys actually listed (status[0]); `num_rendered` keeps the reference's count
        # (radius rectangles), the front end lists a splat only in the tiles its alpha footprint reaches
        R = float(int(last["status"].cpu()[0]))
        T = ((W + 15) // 16) * ((H + 15) // 16)
        # algorithmic bytes of ONE k_fisher_tile_v4 launch: per tile instance the sorted key (8 B), the
        # 32-byte {recA, recB} record and the 52 used bytes of the recQ record, each moved once; plus one partial score per (view, tile)
        per_instance = BYTES_PER_TILE_INSTANCE_FIXED if scorer.tile_capacity > 0 else BYTES_PER_TILE_INSTANCE
        kern_bytes = (R * per_instance + 4.0 * V * T) / launches_per_step
        ach = kern_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms == kern_ms else None
        # whole-path algorithmic bytes per view, SURVEY.md 8(d)
        B_view = (12 * P + 44 * vis_count.mean() + 24 * num_rendered.mean() + 40 * num_rendered.mean() +
                  40 * num_rendered.mean() + 24 * W * H + 8 * W * H + 4 * C * P)
        pm, pmc_meta = pmc_record(P, V, W, C)
        if pm and pm.get("dispatches_per_step") != launches_per_step:
            pm, pmc_meta = None, dict(pmc_meta, fresh=False, why="counters were taken with another number of view groups per step")
        # per LAUNCH like `achieved`: the record sums the step's dispatches
        traffic = pm["hbm_bytes_per_step"] / launches_per_step if pm and pm.get("hbm_bytes_per_step") else None
        ceil = valu_ceiling() if world == 1 else None
        valu = None
        if pm and pm.get("SQ_INSTS_VALU") and kern_ms == kern_ms:
            step_ms = kern_ms * launches_per_step                    # the step's dispatches, summed
            rate = pm["SQ_INSTS_VALU"] / (step_ms * 1e-3)
            valu = {"bound": "valu", "insts_per_step": pm["SQ_INSTS_VALU"], "achieved_wave_insts_per_s": rate,
                    "lds_insts_per_step": pm.get("SQ_INSTS_LDS"), "salu_insts_per_step": pm.get("SQ_INSTS_SALU")}
            if ceil and "independent_fma_wave_insts_per_s" in ceil:
                valu["peak_wave_insts_per_s"] = ceil["independent_fma_wave_insts_per_s"]
                valu["frac"] = rate / ceil["independent_fma_wave_insts_per_s"]
            if pm.get("contributing_pairs_per_step"):
                # the pair-proportional work next to the GB/s (no atomics on the scoring path)
                valu["pairs_per_s"] = pm["contributing_pairs_per_step"] / (step_ms * 1e-3)
                valu["wave_insts_per_pair"] = pm["SQ_INSTS_VALU"] / pm["contributing_pairs_per_step"]
                valu["walk_iterations_per_step"] = pm.get("walk_iterations_per_step")
        out = {
            "metric": "candidate-views/sec", "value": views_per_s, "unit": "candidate-views/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{P} Gaussians (room_shell seed {seed}), {V_total} candidate {W}x{H} views per step"
                                   f" ({V} on this GPU), Fisher columns {C}, H_inv from 16 keyframes; BASELINE.json configs[{1 if world == 1 else 2}]",
                       "gaussians": P, "views_per_gpu": V, "views_total": V_total, "image": [H, W], "columns": C,
                       "tile_instances_per_view": float(num_rendered.mean()), "listed_tile_instances_per_view": R / max(V, 1),
                       "visible_per_view": float(vis_count.mean()),
                       "parallelism": f"views sharded over {world} GPU(s), scores all-gathered" if world > 1 else "1 GPU"},
            "fisher_scores_per_s": views_per_s * P * C,
            "fisher_scores_per_s_what": "nominal: views/s x P x columns (SURVEY 8d's definition); the scoring path contracts cur_H with "
                                        "H_inv on the fly and never materialises these entries",
            "build": {"build_id": lib.fr_build_id().decode(), "so_path": os.path.relpath(_lib.SO_PATH, ROOT)},
            # `bound`: the contract's HBM roofline (algorithmic bytes over time against 8 TB/s) is what achieved / peak / frac hold;
            # what actually limits the kernel is its per-pair VALU work: `binding` and the `valu` block (frac of the calibrated ceiling)
            "roofline": {"bound": "hbm", "binding": "valu", "kernel": "k_fisher_tile_v4" if scorer.tile_capacity > 0 else "k_fisher_tile_v3", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (ach / HBM_PEAK_GBS) if ach is not None else None, "traffic": traffic,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": kern_bytes,
                         "bytes_per_tile_instance": per_instance, "launches_per_step": launches_per_step,
                         "views_per_launch": V // launches_per_step, "valu": valu, "valu_ceiling": ceil, "pmc": pmc_meta,
                         "note": "the kernel is bound by per-pair VALU work, not by bytes (no dense contraction, no MFMA): "
                                 "`valu.frac` = achieved wave64 VALU instructions per second over the calibrated ceiling"},
            "path": {"bytes_per_view": float(B_view), "achieved_GBps": float(B_view * views_per_s / world / 1e9),
                     "frac_of_hbm_peak": float(B_view * views_per_s / world / 1e9 / HBM_PEAK_GBS)},
        }
        if world == 1:
            out["roofline"]["measured_copy_GBps"] = measured_copy_bandwidth(dev)
        if world == 1 and a.cpu_views > 0:
            if C == 4:
                # the same workload with the 11 Fisher columns of GaussianObjectSLAM, reported next to the headline
                sc11 = FisherScorer(cam, *(act[k].to(dev) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")),
                                    columns=11, dL_dpix=1e-3)
                Ht11 = torch.zeros((P, 11), dtype=torch.float32, device=dev)
                sc11.run(kf, out_H=Ht11)
                Hi11 = torch.reciprocal(Ht11 + 0.1)
                sc11.run(w2c, H_inv=Hi11)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                n11 = max(3, a.steps // 4)
                for _ in range(n11):
                    r11 = sc11.launch(w2c, H_inv=Hi11)
                    host_scores.copy_(r11["scores"], non_blocking=True)
                torch.cuda.synchronize()
                d11 = (time.perf_counter() - t1) / n11
                out["columns_11"] = {"value": V / d11, "unit": "candidate-views/s", "ms_per_step": 1e3 * d11,
                                     "fisher_scores_per_s": V / d11 * P * 11, "steps": n11}
                del sc11, Ht11, Hi11
            par = cpu_legs["cpu_baseline"].pop("_parity")
            out.update(cpu_legs)
            # the cpu_baseline leg's oracle scores against the GPU's on the SAME views with the SAME H_inv (outside the timed region)
            n_par = len(par["scores"])
            g_par = scorer.run(w2c_all[:n_par], H_inv=torch.from_numpy(par["H_inv"]).to(dev))["scores"].cpu().double().numpy()
            rel = np.abs(g_par - par["scores"]) / np.abs(par["scores"])
            out["parity"] = {"views": int(n_par), "max_rel_err": float(rel.max()), "median_rel_err": float(np.median(rel)),
                             "tolerance": 1e-4, "ok": bool(rel.max() < 1e-4),
                             "what": "scores of the first views of this step through oracle/fisher_oracle.c (the cpu_baseline leg) "
                                     "vs fr_fisher_views, same 1-keyframe H_inv; all 64 views with the 16-keyframe H_inv: "
                                     "tests/test_gpu_fullsize_properties.py::test_all_64_scores_of_configs1_against_the_oracle"}
            out["gpu_occupancy_frontier"] = gpu_occupancy_frontier(dev, W, H, seed)
            if C == 4:
                out["serial_loop"] = serial_loop(dev, raw, W, H, seed, 16, H_inv)
                # BASELINE.md holds no published number for this metric, so vs_baseline stays null; the ratio to the
                # reference-style serial loop on this repository's own kernels is context, under its own name
                out["vs_serial_loop"] = views_per_s / out["serial_loop"]["value"]
                out["api_pose_eval"], out["compute_hessian_v1_ms"] = api_latencies(dev, raw, W, H, seed, V)
                del scorer
                torch.cuda.empty_cache()
                out["config4"] = config4_train_step(dev)
        else:
            out["cpu_baseline"] = None
        if dist_on:
            out["collectives"] = {"backend": dist.get_backend(), "world_size": world,
                                  "per_step": "all_gather_into_tensor of the per-view scores (fisher_rast.distributed.ScoreGather)",
                                  "replication": {"what": "fisher_rast.distributed.replicate_map: the activated map broadcast from rank 0, "
                                                          "one dist.broadcast per tensor (outside the timed region)",
                                                  "tensors": len(act), "bytes": int(sum(v.numel() * v.element_size() for v in act.values())),
                                                  "ms": 1e3 * t_rep}}
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
