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
        assert j["roofline"]["kernel"] == "k_fisher_tile_v3" and j["roofline"]["kernel_ms"] > 0
    assert one["n_gpus"] == 1 and one["build"]["build_id"].startswith("FRSRC:")
