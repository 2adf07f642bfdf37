"""-m gpu: the N > 1 control flow of bench.py on ONE GPU -- two fresh rank processes (started by torch.distributed.run
before either has touched the GPU), collectives through gloo (FR_BENCH_BACKEND=gloo, FR_BENCH_ONE_DEVICE=1): view sharding,
H_train all-reduce, score all-gather, max-over-ranks timing, the JSON line.  With fixed weights (--synthetic-hinv; H_train
is accumulated with float atomics and differs in its last bits from run to run) the gathered scores must equal the 1-rank
run bit for bit, in weak and in strong scaling.  The measured configuration (nccl = RCCL, one rank per GPU) needs the
multi-GPU node the driver has; this is the rehearsal of everything but the transport."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "2", "--warmup", "1", "--cpu-views", "0", "--gaussians", "30000", "--size", "128", "--seed", "3", "--synthetic-hinv"]


def _run(cmd, env_extra, out_npy):
    env = dict(os.environ, **env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd + ["--dump-scores", out_npy], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line), np.load(out_npy)


def _two_ranks(args, out_npy, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2"] + COMMON + args
    return _run(cmd, {"FR_BENCH_BACKEND": "gloo", "FR_BENCH_ONE_DEVICE": "1"}, out_npy)


def test_two_rank_bench_equals_one_rank(gpu, tmp_path):
    port = 29700 + os.getpid() % 200
    one, s1 = _run([sys.executable, "bench.py", "--gpus", "1", "--views", "16"] + COMMON, {}, str(tmp_path / "one.npy"))
    weak, sw = _two_ranks(["--views", "8"], str(tmp_path / "weak.npy"), port)            # 8 views per rank = the same 16 views
    strong, ss = _two_ranks(["--total-views", "16"], str(tmp_path / "strong.npy"), port + 1)
    assert s1.shape == (16,) and np.isfinite(s1).all() and (s1 > 0).all()
    assert np.array_equal(sw, s1) and np.array_equal(ss, s1)
    for j, scaling in ((weak, "weak"), (strong, "strong")):
        assert j["n_gpus"] == 2 and j["scaling"] == scaling and j["config"]["views_total"] == 16 and j["config"]["views_per_gpu"] == 8
        assert j["metric"] == one["metric"] == "candidate-views/sec" and j["value"] > 0 and j["cpu_baseline"] is None
        assert j["roofline"]["kernel"] == "k_fisher_tile_v4" and j["roofline"]["kernel_ms"] > 0
    assert one["n_gpus"] == 1 and one["build"]["build_id"].startswith("FRSRC:")


def test_plain_command_starts_its_own_ranks(gpu, tmp_path):
    """The driver's command form, no launcher: `python bench.py --gpus 2 ...` must start its two ranks itself (a fresh
    torch.distributed.run child; the parent never touches the GPU), replicate the map from rank 0 by broadcast, shard the 16
    views, and print ONE JSON line whose `collectives` shows that two ranks took part.  Scores = the 1-rank run, bit for bit."""
    one, s1 = _run([sys.executable, "bench.py", "--gpus", "1", "--views", "16"] + COMMON, {}, str(tmp_path / "one.npy"))
    env = {"FR_BENCH_BACKEND": "gloo", "FR_BENCH_ONE_DEVICE": "1"}
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        assert k not in os.environ
    two, s2 = _run([sys.executable, "bench.py", "--gpus", "2", "--total-views", "16"] + COMMON, env, str(tmp_path / "two.npy"))
    assert two["n_gpus"] == 2 and two["collectives"]["world_size"] == 2 and two["collectives"]["backend"] == "gloo"
    rep = two["collectives"]["replication"]
    assert rep["tensors"] == 5 and rep["bytes"] == 30000 * 4 * (3 + 3 + 4 + 1 + 3)
    assert two["config"]["views_total"] == 16 and two["config"]["views_per_gpu"] == 8 and two["scaling"] == "strong"
    assert np.array_equal(s1, s2)
    weak, sw = _run([sys.executable, "bench.py", "--gpus", "2", "--views", "8"] + COMMON, env, str(tmp_path / "weak.npy"))
    assert weak["scaling"] == "weak" and weak["config"]["views_total"] == 16 and np.array_equal(sw, s1)


def test_one_rank_process_group_runs_the_collectives_through_rccl(gpu, tmp_path):
    """No multi-GPU node here, but the transport can still be exercised: bench.py under torch.distributed.run with ONE rank and
    the nccl backend (= RCCL), FR_FORCE_COLLECTIVES=1: the process group is created, the H_train all-reduce and the per-step
    all_gather_into_tensor run on device tensors, and the scores equal the plain one-process run bit for bit."""
    port = 29950 + os.getpid() % 40
    plain, s0 = _run([sys.executable, "bench.py", "--gpus", "1", "--views", "16"] + COMMON, {}, str(tmp_path / "plain.npy"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "1", "--views", "16"] + COMMON
    coll, s1 = _run(cmd, {"FR_FORCE_COLLECTIVES": "1", "FR_BENCH_BACKEND": "nccl"}, str(tmp_path / "rccl.npy"))
    assert coll["collectives"]["backend"] == "nccl" and coll["collectives"]["world_size"] == 1
    assert "collectives" not in plain
    assert np.array_equal(s0, s1)
