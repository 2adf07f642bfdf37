"""CPU: `python bench.py --gpus N` without a launcher must start its ranks as a CHILD `torch.distributed.run` (never a re-exec, no GPU
call in the parent), pass the arguments through, relay the child's output and return its exit code; inside a launcher it must refuse
a world size that does not match --gpus.  (The GPU form of the same command: tests/test_gpu_bench_multirank.py.)"""
import io
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeChild:
    def __init__(self, lines, rc):
        self.stdout = io.StringIO("".join(lines))
        self._rc = rc

    def wait(self):
        return self._rc


def test_self_launch_builds_the_torchrun_command(monkeypatch, capsys):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_popen(cmd, env=None, stdout=None, text=None, bufsize=None):
        seen.update(cmd=cmd, env=env)
        return _FakeChild(['{"metric": "candidate-views/sec", "n_gpus": 4}\n'], 7)

    monkeypatch.setattr(bench.subprocess, "Popen", fake_popen)
    rc = bench.self_launch(4, ["--gpus", "4", "--steps", "3", "--total-views", "512"])
    assert rc == 7                                                     # the child's code is the parent's
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "3", "--total-views", "512"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") is not None
    assert '"n_gpus": 4' in capsys.readouterr().out                   # rank 0's line is passed through


def test_main_goes_through_self_launch_only_without_a_launcher(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    monkeypatch.setattr(bench, "self_launch", lambda n, argv: calls.append((n, list(argv))) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "2"])
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0 and calls == [(8, ["--gpus", "8", "--steps", "2"])]
    # inside a launcher: a rank, and the world size must be the one asked for
    calls.clear()
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("LOCAL_RANK", "0")
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert calls == [] and "process group of 2" in str(ex.value.code)
