#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric): candidate views scored per second
(+ Fisher scores per second) at 256x256 over 500k Gaussians.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic input: the fused scorer (fr_fisher_views) ranks
`--views` (64) candidate poses per GPU against a resident 500k-Gaussian map and a resident H_inv = 1/(H_train+0.1):
project/cull -> tile binning -> per-tile depth sort -> transmittance pass -> backward(power=2) -> sum(cur_H*H_inv).
Inputs are in HBM before the timed region; each step ends with the asynchronous copy of the scores to the host.
N = 1 : BASELINE.json configs[1] (500k Gaussians, 64 candidate 256x256 views, seed 2).
N > 1 : configs[2]'s shape, weak scaling: every rank scores its own 64-view slice of a 64*N-view candidate set
        (seed 3) and the per-view scores are exchanged with one RCCL all-gather per step.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes
import json
import os
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


def cpu_baseline(P, W, H, seed, n_views, columns):
    """The oracle (CPU restatement of the reference path, scalar, 1 thread) on a bounded sample of the same workload:
    n_views candidate views of the same 500k-Gaussian scene, same H_inv construction (1 keyframe instead of 16)."""
    from oracle import ref
    from fisher_rast import synthetic
    act = {k: v.numpy() for k, v in synthetic.activate(synthetic.room_shell(P, seed)).items()}
    args = (act["means3D"], act["rgb_colors"], act["rotations"], act["opacities"], act["scales"])
    cam = ref.setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4))
    w2c = synthetic.invert_rigid(synthetic.candidate_poses(n_views, seed)).numpy()
    kf = synthetic.invert_rigid(synthetic.candidate_poses(1, seed + 100)).numpy()
    H_train = ref.compute_h_train(cam, kf, *args, columns=columns)
    t0 = time.perf_counter()
    scores, _ = ref.pose_eval(cam, w2c, H_train, *args, columns=columns)
    dt = time.perf_counter() - t0
    return dict(value=n_views / dt, unit="candidate-views/s", cores=1, kind="port",
                sample=f"{n_views} of the {P}-Gaussian {W}x{H} candidate views through oracle/fisher_oracle.c "
                       f"(forward + fused backward power=2 + weighted sum), {dt:.1f} s on 1 host core",
                seconds=dt)


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
    return dict(ms_per_update=1e3 * t_up, ms_per_frontier_build=1e3 * t_fr, frames=n_frames,
                free_cells=int(free.sum()), frontier_cells=0 if fr is None else int(len(fr)),
                kernels="fr_occ_update / fr_occ_freespace / fr_occ_frontiers (csrc/fisher_occ.hip), wall time incl. host glue")


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


def api_pose_eval_latency(dev, act_raw, W, H, seed, V):
    """End-to-end latency of the drop-in call `GaussianSLAM.pose_eval(poses)` (gaussian.py:1354-1375): activation of the raw
    parameters, H_train over 16 keyframes, V candidate scores, scores on the host -- what the planner waits for."""
    import models.gaussian_slam as mgs
    from fisher_rast import synthetic
    slam = mgs.GaussianSLAM(params={k: v.to(dev) for k, v in act_raw.items()}, intrinsics=synthetic.intrinsics(W, H), width=W, height=H, device=dev)
    for kf in synthetic.invert_rigid(synthetic.candidate_poses(16, seed + 100)):
        slam.add_keyframe(kf.to(dev))
    poses = [p.to(dev) for p in synthetic.candidate_poses(V, seed)]
    slam.pose_eval(poses)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        scores, _ = slam.pose_eval(poses)
    torch.cuda.synchronize()
    return dict(ms_per_call=1e3 * (time.perf_counter() - t0) / reps, views=V, keyframes=16,
                what="GaussianSLAM.pose_eval(poses): H_train over the keyframes + all candidate scores, result on the host")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gaussians", type=int, default=500_000)
    ap.add_argument("--views", type=int, default=64, help="candidate views per GPU per step")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--columns", type=int, default=4)
    ap.add_argument("--cpu-views", type=int, default=12, help="views of the CPU-baseline sample (0 = skip)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback on the product path)"
    # FR_BENCH_BACKEND=gloo with FR_BENCH_ONE_DEVICE=1 rehearses the N > 1 control flow on a single GPU (all ranks on cuda:0,
    # collectives through the host); the measured configuration is always nccl (RCCL), one rank per GPU
    backend = os.environ.get("FR_BENCH_BACKEND", "nccl")
    if os.environ.get("FR_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if world > 1:
        dist.barrier()
    from fisher_rast import synthetic, _lib, distributed as D
    from fisher_rast.ops import FisherScorer
    from models.SLAM.utils.recon_helpers import setup_camera

    P, V, W, H, C = a.gaussians, a.views, a.size, a.size, a.columns
    seed = 2 if world == 1 else 3
    act = synthetic.activate(synthetic.room_shell(P, seed))
    cam = setup_camera(W, H, synthetic.intrinsics(W, H), np.eye(4), device=dev)
    scorer = FisherScorer(cam, *(act[k].to(dev) for k in ("means3D", "rgb_colors", "rotations", "opacities", "scales")),
                          columns=C, dL_dpix=1e-3)
    w2c_all = synthetic.invert_rigid(synthetic.candidate_poses(V * world, seed)).to(dev)
    lo, hi = D.shard_bounds(V * world, rank, world)
    w2c = w2c_all[lo:hi].contiguous()
    kf = synthetic.invert_rigid(synthetic.candidate_poses(16, seed + 100)).to(dev)

    # H_train: keyframes sharded over ranks + one all-reduce(SUM)  (outside the timed region: it is an input)
    H_train = torch.zeros((P, C), dtype=torch.float32, device=dev)
    D.sharded_h_train(lambda w, Hacc: scorer.run(w, out_H=Hacc), kf, H_train)
    H_inv = torch.reciprocal(H_train + 0.1)
    first = scorer.run(w2c, H_inv=H_inv)      # sizes the tile-instance buffer (may re-run on overflow)
    num_rendered = first["num_rendered"].cpu().numpy().astype(np.int64)
    vis_count = first["vis_count"].cpu().numpy().astype(np.int64)

    host_scores = torch.empty((V * world,), dtype=torch.float32).pin_memory()
    gathered = torch.empty((V * world,), dtype=torch.float32, device=dev)

    def step():
        r = scorer.launch(w2c, H_inv=H_inv)
        if world > 1:
            dist.all_gather_into_tensor(gathered, r["scores"])
            host_scores.copy_(gathered, non_blocking=True)
        else:
            host_scores.copy_(r["scores"], non_blocking=True)
        return r

    for _ in range(a.warmup):
        step()
    lib = _lib.load()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    lib.fr_profile_enable(1)
    t0 = time.perf_counter()
    last = None
    for _ in range(a.steps):
        last = step()
    torch.cuda.synchronize()
    if world > 1:
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
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max.item())

    if rank == 0:
        views_per_s = V * world * a.steps / dt
        R = float(num_rendered.sum())          # tile instances of this rank's V views
        T = ((W + 15) // 16) * ((H + 15) // 16)
        # algorithmic bytes of ONE k_fisher_tile_v2 launch: per tile instance pass 1 reads the key (8)
        # and the 32-byte splat record; pass 2 reads those again plus the packed static record (mean, cov3D, rgb, H_inv:
        # 64 B at C = 4, 128 B at C = 11); plus one partial score per (view, tile).
        kern_bytes = (R * ((8 + 32) + (8 + 32) + (64 if C == 4 else 128)) + 4.0 * V * T) / launches_per_step
        ach = kern_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms == kern_ms else None
        # whole-path algorithmic bytes per view, SURVEY.md 8(d)
        B_view = (12 * P + 44 * vis_count.mean() + 24 * num_rendered.mean() + 40 * num_rendered.mean() +
                  40 * num_rendered.mean() + 24 * W * H + 8 * W * H + 4 * C * P)
        traffic = None
        valu = None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_k_fisher_tile_v2.json")
        if os.path.exists(pmc_file):
            try:
                pm = json.load(open(pmc_file))
                if pm.get("gaussians") == P and pm.get("views") == V // launches_per_step and pm.get("size") == W and pm.get("columns") == C:
                    traffic = pm.get("hbm_bytes_per_launch")
                    if pm.get("SQ_INSTS_VALU") and kern_ms == kern_ms:
                        # the kernel's real limiter: wave-level VALU instructions (one per 4 cycles per SIMD) against the
                        # 1024 SIMDs of the chip at the nominal 2.4 GHz peak clock (a lower bound on the busy fraction)
                        valu = {"insts_per_launch": pm["SQ_INSTS_VALU"], "simds": 1024, "clock_ghz": 2.4,
                                "issue_frac": pm["SQ_INSTS_VALU"] * 4 / (1024 * kern_ms * 1e-3 * 2.4e9),
                                "source": "profiles/pmc_k_fisher_tile_v2.json (rocprofv3 --pmc SQ_INSTS_VALU)"}
                        if pm.get("contributing_pairs_per_launch"):
                            # the pair-proportional work next to the GB/s (no atomics on the scoring path)
                            valu["pairs_per_s"] = pm["contributing_pairs_per_launch"] / (kern_ms * 1e-3)
                            valu["wave_insts_per_pair"] = pm["SQ_INSTS_VALU"] / pm["contributing_pairs_per_launch"]
                        if pm.get("SQ_ACTIVE_INST_VALU") and pm.get("SQ_BUSY_CYCLES"):
                            # clock-independent: VALU-active quad-cycles per SIMD over the kernel's busy cycles
                            # (SQ_BUSY_CYCLES is summed over the 32 shader engines; SQ_ACTIVE_INST_* count 4-cycle quanta)
                            valu["busy_frac_profiled"] = (pm["SQ_ACTIVE_INST_VALU"] * 4 / 1024) / (pm["SQ_BUSY_CYCLES"] / 32)
            except Exception:
                traffic = None
        out = {
            "metric": "candidate-views/sec", "value": views_per_s, "unit": "candidate-views/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{P} Gaussians (room_shell seed {seed}), {V} candidate {W}x{H} views per GPU per step, "
                                   f"Fisher columns {C}, H_inv from 16 keyframes; BASELINE.json configs[{1 if world == 1 else 2}]",
                       "gaussians": P, "views_per_gpu": V, "image": [H, W], "columns": C,
                       "tile_instances_per_view": float(num_rendered.mean()), "visible_per_view": float(vis_count.mean()),
                       "parallelism": f"views sharded over {world} GPU(s), scores all-gathered" if world > 1 else "1 GPU"},
            "fisher_scores_per_s": views_per_s * P * C,
            "roofline": {"bound": "hbm", "kernel": "k_fisher_tile_v2", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (ach / HBM_PEAK_GBS) if ach is not None else None, "traffic": traffic,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": kern_bytes, "launches_per_step": launches_per_step,
                         "views_per_launch": V // launches_per_step, "valu": valu},
            "path": {"bytes_per_view": float(B_view), "achieved_GBps": float(B_view * views_per_s / world / 1e9),
                     "frac_of_hbm_peak": float(B_view * views_per_s / world / 1e9 / HBM_PEAK_GBS)},
        }
        if world == 1:
            out["roofline"]["measured_copy_GBps"] = measured_copy_bandwidth(dev)
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
        if world == 1 and a.cpu_views > 0:
            out["cpu_baseline"] = cpu_baseline(P, W, H, seed, a.cpu_views, C)
            # the reference's CPU occupancy / frontier step (planning/astar.py), timed on the same host cores
            from oracle import occupancy_frontier
            out["cpu_occupancy_frontier"] = occupancy_frontier.time_baseline(n_frames=4, W=W, H=H, seed=seed)
            out["gpu_occupancy_frontier"] = gpu_occupancy_frontier(dev, W, H, seed)
            if C == 4:
                out["api_pose_eval"] = api_pose_eval_latency(dev, synthetic.room_shell(P, seed), W, H, seed, V)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
